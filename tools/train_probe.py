"""Sanity probe: PPO over the time-sliced engine for a minute; prints episode return / length and losses per rollout.
    python tools/train_probe.py <object|mixed> <seconds> [overlap] [--seed S] [--dtype f32|bf16] [--steps N] [no-find] [no-fused-loss]
--steps N: stop after N env transitions instead of after <seconds> (learning curves of variants compared at equal samples)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, MixedBatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
obj = sys.argv[1] if len(sys.argv) > 1 else "sand_ball"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60
overlap = "overlap" in sys.argv[3:]
def _opt(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
seed = _opt("--seed", 0, int); dtype = _opt("--dtype", "f32", str); max_steps = _opt("--steps", 0, int)
import torch
if obj == "mixed":          # four objects x two directions, 512 envs each, one batch set
    cfg = default_config(time_horizon=50)
    env = GpuVecEnv(MixedBatchedRobotEnv(cfg, envs_per_group=512, device_index=0, auto_reset=True))
else:
    cfg = default_config(sim_env=f"/xmls/{obj}_env.xml", time_horizon=50)
    env = GpuVecEnv(BatchedRobotEnv(cfg, n_envs=4096, device_index=0, auto_reset=True))
model = PPO("MultiInputPolicy", env, n_steps=8, batch_size=4096, n_epochs=2, seed=seed, autocast_dtype=torch.bfloat16 if dtype == "bf16" else None, async_slice=96, async_capacity=1024, async_budget_us=2000, ent_coef=0.0, overlap_update=overlap, miopen_find="no-find" not in sys.argv[3:],
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
if "no-fused-loss" in sys.argv[3:]:
    model.fused_loss = False
ar = model._async
t0 = time.time(); it = 0; last = (0.0, 0.0, 0.0)
while (model.num_timesteps < max_steps) if max_steps else (time.time() - t0 < secs):
    model.collect_rollouts(); st = model.train(); it += 1
    if it % 5 == 0:
        c, r, l = float(ar.ep_count.item()), float(ar.ep_ret_sum.item()), float(ar.ep_len_sum.item())
        dc = max(1.0, c - last[0])
        f = env.env.batch.out["fault"]
        faults = [int((f & b).ne(0).sum()) for b in (1, 2, 4)]
        print(f"iter {it:4d} faults(div/ovf/cap) {faults} t {time.time() - t0:6.1f}s timesteps {model.num_timesteps:9d} episodes {int(c):7d} ep_rew_mean(last window) {(r - last[1]) / dc:8.4f} "
              f"ep_len {(l - last[2]) / dc:6.1f} loss {float(st['loss']):9.4f} value_loss {float(st['value_loss']):9.4f}", flush=True)
        last = (c, r, l)
model.finish_updates()
print("fps", model.num_timesteps / (time.time() - t0), "overlap_update", overlap, "seed", seed, "policy dtype", dtype, "options", [x for x in sys.argv[3:] if x.startswith("no-")])
