"""Which tensor-library operations the kernels of one eager PPO minibatch update belong to (torch profiler, innermost CPU op per kernel):
    python tools/update_ops.py"""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=512, device_index=0, auto_reset=True))
model = PPO("MultiInputPolicy", env, n_steps=8, batch_size=4096, n_epochs=1, seed=0,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
model.graph_update = False
model.collect_rollouts(); model.train(); model.train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model.train()
    torch.cuda.synchronize()
by_op = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
# innermost: an event none of whose children launched these kernels
for e in evs:
    child_k = set()
    for c in e.cpu_children:
        for k in c.kernels: child_k.add(id(k))
    own = [k for k in e.kernels if id(k) not in child_k]
    for k in own:
        r = by_op[e.name]; r[0] += 1; r[1] += k.duration; r[2][k.name[:60]] += 1
tot = sum(r[1] for r in by_op.values())
print(f"total device time {tot:.0f} us")
for name, (c, t, ks) in sorted(by_op.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:9.1f} us x{c:4d}  {name[:50]:50s} {dict(ks.most_common(2))}")
