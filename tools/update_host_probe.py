"""Host cost of launching the captured PPO update: after a device synchronise (empty queue, nothing to wait for) how long do the two hipGraph launches of one minibatch take to
RETURN, and how long until the GPU is done? If the first is close to the second the update is bound by launch overhead, not by its kernels.
    python tools/update_host_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
torch.backends.cudnn.benchmark = True
env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/acorn_env.xml", time_horizon=50), n_envs=4096, device_index=0, auto_reset=True))
model = PPO("MultiInputPolicy", env, n_steps=8, batch_size=4096, n_epochs=2, seed=0, async_slice=96, async_capacity=1024, async_budget_us=2000,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
model.collect_rollouts()
for _ in range(3): model.train()
u = model._upd
assert u is not None and u["fwd"] is not None
host1 = host2 = total = 0.0; n = 50
for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u["fwd"].replay(); t1 = time.perf_counter()
    u["apply"].replay(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    host1 += t1 - t0; host2 += t2 - t1; total += t3 - t0
print(f"one minibatch from an empty queue: forward/backward graph launch returns after {host1 / n * 1e3:.3f} ms, clip+Adam graph launch after {host2 / n * 1e3:.3f} ms more, "
      f"GPU done after {total / n * 1e3:.3f} ms")
# back to back, as train() issues them
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(64):
    u["fwd"].replay(); u["apply"].replay()
th = time.perf_counter() - t0
torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"64 minibatches back to back: host returns after {th / 64 * 1e3:.3f} ms per minibatch, GPU done after {tt / 64 * 1e3:.3f} ms per minibatch")
