"""Where does bf16 training go NaN? PPO with the bf16 policy over the time-sliced engine, checking records, losses, parameters and gradients after every
rollout:  python tools/bf16_graph_probe.py graph|eager   (graph = the captured hipGraph update, the default path; eager = uncaptured).
Finding (MI355X, round 2): eager trains cleanly for 12.5 M steps; the captured update turns every parameter NaN within ~10 rollouts while the
rollout records stay finite and the f32 captured update is fine."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
graph = sys.argv[1] == "graph"
cfg = default_config(sim_env="/xmls/sand_ball_env.xml", time_horizon=50)
env = GpuVecEnv(BatchedRobotEnv(cfg, n_envs=4096, device_index=0, auto_reset=True))
model = PPO("MultiInputPolicy", env, n_steps=8, batch_size=4096, n_epochs=2, seed=0, autocast_dtype=torch.bfloat16, async_slice=96, async_capacity=1024, async_budget_us=2000,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
if not graph: model.graph_update = False
ar = model._async
fin = lambda t: bool(torch.isfinite(t.float()).all())
for it in range(400):
    model.collect_rollouts()
    sel = ar.training_indices()
    bad = [n for n, t in (("actions", ar.actions[sel]), ("logp", ar.log_probs[sel]), ("values", ar.values[sel]), ("rewards", ar.rewards[sel]), ("adv", ar.advantages[sel]), ("ret", ar.returns[sel])) if not fin(t)]
    st = model.train()
    pbad = [n for n, p in model.policy.named_parameters() if not fin(p)]
    gbad = [n for n, p in model.policy.named_parameters() if p.grad is not None and not fin(p.grad)]
    if it % 20 == 0 or bad or pbad or gbad or not fin(st["loss"]):
        lp = ar.log_probs[sel]
        print(f"it {it} steps {model.num_timesteps} loss {float(st['loss']):.4f} vl {float(st['value_loss']):.4f} pl {float(st['policy_loss']):.4f} records-bad {bad} params-bad {pbad[:3]} grads-bad {gbad[:3]} "
              f"|logp|max {float(lp.abs().max()):.1f} |adv|max {float(ar.advantages[sel].abs().max()):.1f} log_std {model.policy.log_std.detach().float().cpu().numpy().round(2)} pmax {max(float(p.detach().abs().max()) for p in model.policy.parameters()):.3g}", flush=True)
    if bad or pbad or not fin(st["loss"]):
        break
