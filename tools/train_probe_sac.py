"""Sanity probe: SAC (+ hindsight relabelling) over the time-sliced engine -- does the algorithm the reference really trains (train_agent.py:57-92)
learn on this engine? Prints wall time, transitions, updates, mean episode return / length of the last window and the losses.
    python tools/train_probe_sac.py <object> <seconds> [her] [--seed S] [--envs N] [--gsteps G] [--batch B] [--ent auto|auto_0.1|0.05] [--lr 3e-4] [--gamma 0.99]
her: --her_buffer reward term + FlatHerReplayBuffer ('future', n_sampled_goal 4: the reference's HER branch); otherwise plain SAC (:80-92)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, GpuVecEnv, HerReplayBuffer
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
obj = sys.argv[1] if len(sys.argv) > 1 else "sand_ball"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60
her = "her" in sys.argv[3:]
def _opt(name, default, cast):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
seed = _opt("--seed", 0, int); n = _opt("--envs", 1024, int); gsteps = _opt("--gsteps", 2, int); batch = _opt("--batch", 512, int)
ent = _opt("--ent", "auto", str); lr = _opt("--lr", 3e-4, float); gamma = _opt("--gamma", 0.99, float)
cfg = default_config(sim_env=f"/xmls/{obj}_env.xml", time_horizon=50, her_buffer=her)
env = GpuVecEnv(BatchedRobotEnv(cfg, n_envs=n, device_index=0, auto_reset=True))
kw = dict(replay_buffer_class=HerReplayBuffer, replay_buffer_kwargs=dict(n_sampled_goal=4, goal_selection_strategy="future", online_sampling=True, max_episode_length=50)) if her else {}
model = SAC("MultiInputPolicy", env, buffer_size=400_000, learning_starts=4 * n, batch_size=batch, seed=seed, train_freq=1, gradient_steps=gsteps,
            ent_coef=(ent if ent.startswith("auto") else float(ent)), learning_rate=lr, gamma=gamma,
            async_slice=96, async_capacity=n // 4, async_budget_us=2000,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]), **kw)
t0 = time.time(); last = [0.0, 0.0, 0.0]; state = {"next": 5.0}


class Probe:
    def init_callback(self, m): pass
    def on_training_start(self, *a): pass
    def on_training_end(self): pass
    def on_step(self):
        t = time.time() - t0
        if t >= state["next"]:
            state["next"] += 5.0
            c, r, l = (float(model.ep_stats[k].item()) for k in ("count", "ret_sum", "len_sum"))
            dc = max(1.0, c - last[0])
            lg = model.logger or {}
            print(f"t {t:6.1f}s ticks {model._async_ticks:6d} timesteps {model.num_timesteps:9d} updates {model._n_updates:7d} episodes {int(c):7d} "
                  f"ep_rew_mean(last window) {(r - last[1]) / dc:8.4f} ep_len {(l - last[2]) / dc:6.1f} critic_loss {float(lg.get('critic_loss', float('nan'))):9.4f} "
                  f"actor_loss {float(lg.get('actor_loss', float('nan'))):9.4f} ent_coef {float(lg.get('ent_coef', float('nan'))):7.4f}", flush=True)
            last[:] = [c, r, l]
        return t < secs


model.learn(total_timesteps=10**12, callback=Probe())
print("fps", model.num_timesteps / (time.time() - t0), "her", her, "seed", seed, "envs", n, "gradient steps per tick", gsteps, "batch", batch, "ent_coef", ent, "lr", lr, "gamma", gamma,
      "relabelled fraction of a sample", float(model.replay_buffer.sample(4096)["relabelled"].float().mean()) if her else None)
