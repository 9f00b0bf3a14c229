"""Every kernel of one eager PPO minibatch update in launch order, with its duration (torch profiler):
    python tools/update_kernels.py [n_envs=512]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=n_envs, device_index=0, auto_reset=True))
model = PPO("MultiInputPolicy", env, n_steps=4096 // n_envs, batch_size=4096, n_epochs=1, seed=0,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
model.graph_update = False
model.collect_rollouts(); model.train(); model.train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model.train()
    torch.cuda.synchronize()
rows = [(e.time_range.start, e.time_range.end - e.time_range.start, e.name) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
rows.sort()
tot = 0.0
for t0, d, kn in rows:
    tot += d
    print(f"{d:8.1f} us  {kn[:150]}")
print(f"total {tot:.0f} us in {len(rows)} kernels")
