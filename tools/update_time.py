"""Wall time of the captured PPO update (32768 records, minibatches of 4096, 2 epochs) on synthetic rollout data, variants side by side:
    python tools/update_time.py [find] [tune] [notune] [autograd] [nofork]
find: torch.backends.cudnn.benchmark = True (MIOpen find mode on the first eager steps); tune: a TunableOp search for the GEMMs (writes /tmp/grip_tunableop.csv:
what assets/tunableop_gfx950.csv is a copy of); notune: without the recorded choices of that file; autograd: the autograd graph instead of the
explicit launch sequence (sb3/fused_update.py); nofork: the explicit sequence with its weight-gradient GEMMs on the main stream"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
if "find" in sys.argv[1:]:
    torch.backends.cudnn.benchmark = True
if "notune" in sys.argv[1:]:
    os.environ["GRIP_TUNABLEOP"] = "0"
if "tune" in sys.argv[1:]:           # TunableOp: every GEMM shape is timed over the library's candidates at its first (eager) call, the winner is used from then on
    torch.cuda.tunable.set_filename("/tmp/grip_tunableop.csv"); torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(30); torch.cuda.tunable.set_max_tuning_iterations(20)
env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/acorn_env.xml", time_horizon=50), n_envs=4096, device_index=0, auto_reset=True))
model = PPO("MultiInputPolicy", env, n_steps=8, batch_size=4096, n_epochs=2, seed=0, async_slice=96, async_capacity=1024, async_budget_us=2000,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
if "autograd" in sys.argv[1:]:
    model.explicit_update = False
if "nofork" in sys.argv[1:]:
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.fused_update import FusedUpdate
    FusedUpdate.fork_weight_grads = False
model.collect_rollouts()
for _ in range(3): model.train()
torch.cuda.synchronize(); t = time.perf_counter(); n = 10
for _ in range(n): st = model.train()
host = (time.perf_counter() - t) / n                    # the host has enqueued everything: if this is the whole time, the update is bound by launch overhead, not by the GPU
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f"update {'find' if torch.backends.cudnn.benchmark else 'default'}{' + TunableOp' if 'tune' in sys.argv[1:] else ''} {'explicit' if model._fused is not None else 'autograd'}{' recorded GEMM choices' if (torch.cuda.tunable.is_enabled() and 'tune' not in sys.argv[1:]) else ''}{' nofork' if 'nofork' in sys.argv[1:] else ''}: {dt * 1e3:.2f} ms per train() = {dt / 16 * 1e3:.3f} ms per minibatch (host-side enqueue {host / 16 * 1e3:.3f} ms); loss {float(st['loss']):.6f}")
