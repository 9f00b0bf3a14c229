#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched rollout engine + PPO on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

N > 1: run as one rank per GPU under torch.distributed.run (the driver does that); when WORLD_SIZE is not set, this process
spawns `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a CHILD
before anything touches a GPU, relays rank 0's JSON line and exits with the child's code.

One "step" = one pass of the hot path over one batch = `--envs` completed env transitions per GPU: the PPO actor-critic forward
(AugmentedNatureCNN + heads) runs on every finished env, the engine runs RobotEnv.step (controller + all physics.step()
sub-steps + reward/done), the observation kernel renders the 5x64x64 uint8 observation, the transition is stored in the HBM
rollout buffer; every `--rollout` steps a full PPO update (GAE, `--epochs` epochs of minibatch forward/backward/Adam, one
flattened-gradient all-reduce per minibatch when N > 1) runs inside the timed region.

Synthetic input: every env starts from the deterministic reset state; actions are a ~ U(-1, 1)^6 from a
counter-based generator keyed by (seed, rank, env, t) (`--actions rng`, the default) -- the policy still runs on every decision
and the update trains on the log-probabilities of those actions, so the arithmetic is that of training, but the physics workload
does not drift as the policy learns. `--actions policy` samples from the (randomly initialised, learning) policy instead.
Before the W warm-up steps an UNTIMED pre-roll of `--preroll` rollout-only steps takes the envs from the common reset state to
desynchronised, mixed episode phases (fresh episodes are free motion: 160 physics.step() calls per macro step; the stationary mix
of pushing / grasping / failing episodes ~235 under uniform actions; all envs start in step, and the synchrony of their episodes takes
a few episode lengths to decay: 800 steps), so that the value does not depend on K and W (20 / 5 and 200 / 20 agree within 2 %).

Default schedule: asynchronous time slices (grip_batch_advance; sb3/async_rollout.py) -- every env runs on its own clock, finished
envs are re-decided every tick, and a step is counted when `--envs` transitions have completed, whichever envs they came from. Only
transitions PPO trains on are counted, as completed (a rollout that ran out of ticks counts what it finished and is reported in
`short_rollouts`). `--lockstep` runs the classic vector-env schedule for comparison.
Workload = BASELINE.json configs[1]: acorn_env (labelled stand-in hull: the reference checkout has no acorn.stl), 4096 envs per GPU,
direction 0, default flags. Weak scaling: per-GPU work is fixed as N grows. Defaults: 20 warm-up + 200 timed steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_PEAK_LANE_OPS = 157.3e12 / 2   # fp32 vector peak 157.3 TFLOP/s (MI355X_MICROARCH.md) = 7.9e13 lane-instructions/s (an FMA is 2 flops): 1024 SIMDs x 32 lanes/clk x 2.4 GHz
# algorithmic bytes per env per macro-step launch of k_macro_step: state in 47 x 4 + action 24,
# state out 40 x 4, reward 4 + done 1 + goals 16 + info 64
MACRO_BYTES_PER_ENV = 47 * 4 + 24 + 40 * 4 + 4 + 1 + 16 + 64
OBS_BYTES_PER_ENV = 5 * 64 * 64 + 14 * 4 + 8


def cpu_policy_leg(threads, seconds=4.0):
    """AugmentedNatureCNN + PPO heads on the host cores (torch CPU, fp32): forward samples/s (no grad) and forward + backward samples/s,
    a bounded sample (about `seconds` each) at batch 512 -- the rate is flat in the batch size well below BASELINE.md's 4096."""
    try:
        from mujoco_rl_manipulate_unknown_objects_amd.sb3.policies import ActorCriticPolicy
        from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
        from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.sensor import RGBDSensor
        from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.actuator import Actuator
        from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import default_config
        cfg = default_config()
        prev = torch.get_num_threads(); torch.set_num_threads(max(1, int(threads)))
        pol = ActorCriticPolicy(RGBDSensor(config=cfg).setup_observation_space(), Actuator(config=cfg).setup_action_space(),
                                features_extractor_class=AugmentedNatureCNN, net_arch=[256, 256])
        B = 512
        obs = {"observation": torch.randint(0, 256, (B, 5, 64, 64), dtype=torch.uint8)}
        act = torch.zeros(B, 6)
        with torch.no_grad():
            pol(obs)
        t0 = time.time(); k = 0
        with torch.no_grad():
            while time.time() - t0 < seconds:
                pol(obs); k += 1
        fwd = B * k / (time.time() - t0)
        t0 = time.time(); k2 = 0
        while time.time() - t0 < seconds:
            v, lp, ent = pol.evaluate_actions(obs, act)
            (lp.mean() + v.mean()).backward(); pol.zero_grad(set_to_none=True); k2 += 1
        fb = B * k2 / (time.time() - t0)
        torch.set_num_threads(prev)
        return {"fwd_samples_per_s": fwd, "fwd_bwd_samples_per_s": fb, "threads": int(threads), "batch": B, "dtype": "f32",
                "sample": f"{k} forward and {k2} forward + backward passes of batch {B}"}
    except Exception as ex:          # noqa: BLE001 -- a reported baseline, never a reason to fail the bench
        return {"error": str(ex)}


def cpu_baseline(obj, seconds_target=15.0):
    """The C oracle (oracle/, `port`: the reference's own dm_control + SB3 stack is not installable here) timed
    on this box's host cores over a bounded sample of the same workload (macro step + observation, no policy)."""
    from oracle import orc
    threads = orc.lib().orc_num_threads()
    m = orc.Model(obj)
    n = 32 * threads
    b = orc.BatchOracle(m, n)
    rng = np.random.default_rng(0)
    obs = np.zeros((n, 5, 64, 64), np.uint8)
    # the GPU leg's input: every env from the deterministic reset state, actions a ~ U(-1, 1)^6. The oracle envs are young (a bounded sample:
    # 1 + <= 64 macro steps per env, i.e. fresh episodes, mostly free motion), so its macro steps are SHORTER than the GPU leg's stationary mix -- the line
    # carries its own mean_substeps_per_env_step and the workload-independent mj_substeps_per_s to compare by
    b.step(rng.uniform(-1, 1, size=(n, 6)), obs=obs)        # warm-up
    t0 = time.time(); steps = 0; sub = 0
    while time.time() - t0 < seconds_target and steps < 64:
        sub += b.step(rng.uniform(-1, 1, size=(n, 6)), obs=obs); steps += 1
    dt = time.time() - t0
    # one env on one thread (the analogue of BASELINE.json configs[0], the reference's own single-env run)
    e = orc.EnvOracle(m); e.reset(); e.step(np.zeros(6, np.float32)); e.observation()
    t1 = time.time(); k1 = 0
    while time.time() - t1 < 3.0 and k1 < 200:
        o = e.step(rng.uniform(-1, 1, size=6).astype(np.float32)); e.observation(); k1 += 1
        if o.done:
            e.reset()
    dt1 = time.time() - t1
    pol = cpu_policy_leg(threads)
    sim_rate = n * steps / dt
    return {"value": sim_rate, "unit": "env-steps/s", "cores": threads, "kind": "port",
            # BASELINE.md section 2, "CPU policy": the actor-critic over AugmentedNatureCNN on the same host cores, and what the SAME work as the
            # GPU line (macro step + observation + one policy forward per step + 2 epochs of forward / backward per sample) then runs at
            "policy": pol,
            "with_policy": (None if not pol.get("fwd_samples_per_s") else
                            {"value": 1.0 / (1.0 / sim_rate + 1.0 / pol["fwd_samples_per_s"] + 2.0 / pol["fwd_bwd_samples_per_s"]), "unit": "env-steps/s",
                             "what": "serial sum on the same cores: simulation + observation, one policy forward, 2 epochs of forward + backward per env step"}),
            "action_source": "U(-1,1)^6 (numpy generator, seed 0): the GPU leg's stream", "mj_substeps_per_s": sub / dt, "mean_substeps_per_env_step": sub / max(1, n * steps),
            "sample": f"{n} envs x {steps} macro steps from the reset state incl. observation render (value: no policy; with_policy: + the policy legs); {sub / dt:.0f} mj-substeps/s",
            "single_env_single_thread": {"value": k1 / dt1, "unit": "env-steps/s", "sample": f"1 env x {k1} macro steps incl. observation render"},
            "note": "reference dm_control+SB3 stack cannot be installed here; its recorded whole-training rate is 3.95-12.97 env-steps/s (BASELINE.md)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (SURVEY.md 8d: >= 200 macro steps after 20 warm-up steps)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--object", default="acorn")
    ap.add_argument("--direction", type=int, default=0)
    ap.add_argument("--rollout", type=int, default=8, help="PPO n_steps (rollout length per env)")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--minibatch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ppo", action="store_true", help="diagnostic: rollout only (INVALID as a headline number)")
    ap.add_argument("--lockstep", action="store_true", help="classic vector-env schedule instead of asynchronous time slices")
    ap.add_argument("--state-dtype", choices=["f32", "f16"], default="f32", help="HBM storage of qpos / qvel / ctrl (BASELINE.json configs[4]: f16; arithmetic is always f32)")
    ap.add_argument("--mixed", action="store_true", help="BASELINE.json configs[3]: {acorn, sand_ball, sugar_cube, bread_crumb} x direction {0, 45}, "
                                                         "--envs / 8 each, sorted by group (not the headline workload)")
    ap.add_argument("--slice", type=int, default=96, help="physics.step() calls per env per tick (async schedule)")
    ap.add_argument("--capacity", type=int, default=0, help="finished envs decided per tick (async schedule); default 3/8 of the envs (1536 of 4096: round 4's kernel finishes ~1400 macro steps per 3 ms tick; 5/16 capped the rate at 287 k)")
    ap.add_argument("--policy-dtype", choices=["f32", "bf16"], default="f32", help="autocast dtype of the policy / PPO update (physics is always f32)")
    ap.add_argument("--overlap-update", action="store_true", help="PPO update of rollout i on a second stream while rollout i+1 is collected (one update of policy lag)")
    ap.add_argument("--pipeline", action="store_true", help="decide for tick t on a side stream while tick t+1 advances (lag 2)")
    ap.add_argument("--budget-us", type=int, default=2000, help="wall-clock cap of a wavefront's slice in microseconds (async schedule; 0 = none)")
    ap.add_argument("--cold-portal", action="store_true", help="diagnostic: the comparison build whose narrow phase starts every portal refinement from scratch (engine.select_library)")
    ap.add_argument("--fixed-slice", action="store_true", help="keep --slice / --budget-us for the whole run (default: they follow the measured length of the macro steps)")
    ap.add_argument("--actions", choices=["rng", "policy"], default=None, help="default: rng with the time-sliced schedule, policy with --lockstep. ""rng: synthetic U(-1,1) action stream keyed by (seed, rank, env, t) (SURVEY.md 8d); "
                                                                                 "policy: samples of the learning policy")
    ap.add_argument("--preroll", type=int, default=800, help="untimed rollout-only steps before the warm-up: desynchronised, mixed episode phases")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if a.capacity <= 0:
        a.capacity = max(8, (3 * a.envs // 8) // 8 * 8)
    if a.actions is None:
        a.actions = "policy" if a.lockstep else "rng"
    if a.mixed:
        a.no_cpu_baseline = True
        if a.envs % 8 or a.capacity % 8:
            raise SystemExit("--mixed needs --envs and --capacity divisible by 8")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one. Nothing in this process has touched a GPU yet (torch is imported, no HIP call made), and
        # the ranks are CHILD processes -- never an exec of a process that initialised the GPU.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if lines:
            print(lines[-1], flush=True)
        else:
            sys.stderr.write(child.stdout[-4000:])
        raise SystemExit(child.returncode if child.returncode or lines else 1)
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run without a launcher: bench.py spawns one)")
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of RCCL (never a headline number)
    rehearsal = os.environ.get("GRIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    if a.cold_portal:
        from mujoco_rl_manipulate_unknown_objects_amd import engine as _engine
        _engine.select_library(cold_portal=True)
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, MixedBatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN

    cfg = default_config(sim_env=f"/xmls/{a.object}_env.xml", direction=a.direction)
    if a.mixed:
        env = GpuVecEnv(MixedBatchedRobotEnv(cfg, envs_per_group=a.envs // 8, device_index=local, auto_reset=True))
    else:
        env = GpuVecEnv(BatchedRobotEnv(cfg, n_envs=a.envs, device_index=local, auto_reset=True))
    model = PPO("MultiInputPolicy", env, n_steps=a.rollout, batch_size=a.minibatch, n_epochs=a.epochs, seed=1234 + rank,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]),
                async_slice=0 if a.lockstep else a.slice, async_capacity=min(a.capacity, a.envs), async_budget_us=a.budget_us,
                autocast_dtype=torch.bfloat16 if a.policy_dtype == "bf16" else None, overlap_update=a.overlap_update and not a.lockstep,
                async_auto_slice=not a.fixed_slice and not a.lockstep and a.slice == 96 and a.budget_us == 2000)
    batch = env.env.batch
    if a.state_dtype == "f16":
        batch.set_state_storage("f16")
    ar = model._async
    if ar is not None and a.pipeline:
        ar.enable_pipeline()
    if a.actions == "rng":
        if ar is None:
            raise SystemExit("--actions rng needs the time-sliced schedule (the fused recorder draws the stream); use --actions policy with --lockstep")
        ar.set_synthetic_actions(a.seed, rank)
    counted = {"done": 0, "short": 0}

    def run_async(nsteps, train=True):
        """nsteps x envs completed transitions, a PPO update after every `rollout` x envs of them (a shorter last rollout is
        trained on as well: every timed step carries its share of the update, whatever K is)."""
        done_steps = 0
        while done_steps < nsteps:
            chunk = min(a.rollout, nsteps - done_steps)
            ar.set_target(chunk * a.envs); model.n_steps = chunk
            before = model.num_timesteps
            model.collect_rollouts()
            got = model.num_timesteps - before                  # transitions that really completed (polled on the device counter)
            counted["done"] += min(got, ar.target); counted["short"] += int(got < ar.target)
            if train and not a.no_ppo:
                model.train()
            done_steps += chunk
        ar.set_target(a.rollout * a.envs); model.n_steps = a.rollout

    def run(nsteps, train=True):
        """nsteps vec-env steps with a PPO update after every `rollout` of them."""
        if ar is not None:
            return run_async(nsteps, train)
        done_steps = 0; subs = 0
        while done_steps < nsteps:
            # collect_rollouts always does n_steps steps; trim the last chunk
            chunk = min(a.rollout, nsteps - done_steps)
            model.n_steps = chunk; model.rollout_buffer.n_steps = chunk
            model.collect_rollouts()
            counted["done"] += chunk * a.envs
            if train and not a.no_ppo:
                model.train()
            done_steps += chunk
        model.n_steps = a.rollout; model.rollout_buffer.n_steps = a.rollout

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # priming, before the W warm-up steps and never timed: one full rollout + update, so that MIOpen's algorithm search and
    # the hipGraph captures (tick, minibatch update) are behind us whatever W and K are. The policy and optimiser are put
    # back afterwards (in place: the graphs keep their addresses), so the timed steps see the same training progress as
    # without priming; only the envs have moved on.
    snap_p = [p.detach().clone() for p in model.policy.parameters()]
    run(a.rollout)
    if hasattr(model, "finish_updates"):
        model.finish_updates()
    torch.cuda.synchronize()
    with torch.no_grad():
        for p, q in zip(model.policy.parameters(), snap_p):
            p.copy_(q)
        for st in model.optimizer.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        if getattr(model, "overlap_update", False):
            for p, q in zip(model.policy_rollout.parameters(), snap_p):
                p.copy_(q)
    del snap_p
    # untimed pre-roll, rollout only: from the common reset state to desynchronised, mixed episode phases (module docstring)
    if a.preroll > 0:
        run(a.preroll, train=False)
    run(a.warmup)
    if hasattr(model, "finish_updates"):
        model.finish_updates()
    sync()
    batch.kernel_time(reset=True)
    if hasattr(batch, "device_time"):
        batch.device_time(reset=True)
    sub_before = None
    # count sub-steps of the timed region on device without host syncs: accumulate n_substeps after each step
    sub_acc = torch.zeros((), dtype=torch.int64, device=env.device)
    orig_step = env.step

    def counted_step(actions):
        r = orig_step(actions)
        sub_acc.add_(r[3]["n_substeps"].sum())
        return r
    env.step = counted_step
    if ar is not None:
        ar.reset_counters(); ticks0 = ar.total_ticks
    counted["done"] = 0; counted["short"] = 0
    sync()
    t0 = time.perf_counter()
    run(a.steps)
    sync()
    dt = time.perf_counter() - t0
    env.step = orig_step
    tmax = torch.tensor([dt], dtype=torch.float64, device=env.device)
    subs = (sub_acc if ar is None else ar.substeps_total).double().reshape(1)
    cnt = torch.tensor([float(counted["done"]), float(counted["short"])], dtype=torch.float64, device=env.device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(subs, op=dist.ReduceOp.SUM); dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt = float(tmax.item()); total_sub = float(subs.item())
    total_env_steps = float(cnt[0].item()); short_rollouts = int(cnt[1].item())
    replicas_identical = None
    if world > 1:           # data parallelism keeps the replicas bit-identical: compare a checksum of every parameter across the ranks
        with torch.no_grad():
            cs = torch.stack([p.detach().double().sum() for p in model.policy.parameters()] + [p.detach().double().abs().sum() for p in model.policy.parameters()])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
    # a number measured while the policy had diverged is not a measurement of training: the parameters must be finite after the timed region (round 4: an update path
    # that left the rollout graph reading freed parameter storage produced NaN losses at full speed, and only a training probe noticed)
    policy_finite = bool(all(torch.isfinite(p).all() for p in model.policy.parameters()))
    if not policy_finite:
        raise RuntimeError("bench.py: the policy's parameters are not finite after the timed region -- the update diverged; no benchmark line for this run")
    ev_ms, ev_n = batch.kernel_time(reset=True)          # host events: the eager launches only (a replayed graph's launches cannot be bracketed)
    k_ms, k_n = ev_ms, ev_n
    d_ms, d_n = None, 0
    if ar is not None and hasattr(batch, "device_time"):
        # every launch of the timed region, replayed graphs' included: start / end stamps the kernel takes with the device's wall clock
        # (grip_batch_device_time, include/grip_sim.h) on the stream it runs on
        d_ms, d_n = batch.device_time(reset=True)
        if d_n > 0:
            k_ms, k_n = d_ms, int(d_n)
    ar_ms, ar_n = (model.allreduce_ms() if (world > 1 and hasattr(model, "allreduce_ms")) else (None, 0))

    if rank == 0:
        value = total_env_steps / dt            # transitions that completed and were trained on, all ranks / max-over-ranks time
        # ALGORITHMIC bytes of one launch, as SURVEY.md 8(d) defines them -- state only: qpos 14 + qvel 13 + ctrl 7 + warm start 13 words in,
        # qpos 14 + qvel 13 + warm start 13 words out per env (348 B f32; qpos / qvel / ctrl are 2-byte words with --state-dtype f16), plus, for
        # the envs whose macro step ends in this launch, the action read (24 B) and the outputs written (reward 4, done 1, goals 16, info 64).
        # What the kernel's OWN bookkeeping moves is listed separately as overhead_bytes_per_launch and is not part of `achieved`: the
        # suspended macro-step context of the time-sliced schedule (32 + 64-byte record each way) and what the narrow phase remembers per
        # pair lane across slices (16 lanes x 16 B each way, + the 48-byte portal each way for the ~0.5 pairs in contact per env).
        state_bytes = (47 + 40) * 4 - ((34 + 27) * 2 if a.state_dtype == "f16" else 0)
        if ar is None:
            fin_per_launch = float(a.envs); overhead_bytes = 0
        else:
            n_ticks = max(1, ar.total_ticks - ticks0)
            fin_per_launch = total_env_steps / max(world, 1) / n_ticks
            overhead_bytes = (2 * (32 + 64) + 2 * 16 * 16 + 2 * 48 // 2) * a.envs
        macro_bytes = state_bytes * a.envs + int(fin_per_launch * (24 + 85))
        sched = ("lock-step vector env" if ar is None else
                 f"asynchronous time slices (<= {ar.S} physics steps and <= {ar.eng.budget_us} us per wavefront and tick"
                 + (f" at the end of the run: slice / budget follow the measured macro-step length, {ar.ladder}" if ar.ladder else "")
                 + f", {min(a.capacity, a.envs)} decisions/tick)" + ("; comparison build: cold-started portal refinement" if a.cold_portal else "") + ("; PPO update overlapped with the next rollout (policy lag 1)" if a.overlap_update else ""))
        achieved = macro_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node), " + ("mixed objects" if a.mixed else f"{a.object}_env") + f" {a.envs} envs/GPU", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("f32" if a.policy_dtype == "f32" else "f32 physics / bf16 policy") + (" (physics state stored as f16)" if a.state_dtype == "f16" else ""),
            "data": "synthetic (deterministic reset state; " + (f"actions U(-1,1)^6 from a counter-based generator keyed by (seed={a.seed}, rank, env, t)" if a.actions == "rng"
                                                                else "actions sampled from the randomly initialised, learning PPO policy") +
                    f"; {a.preroll} untimed rollout-only pre-roll steps to mixed episode phases)",
            "config": {"workload": ("mixed {acorn (stand-in hull), sand_ball, sugar_cube, bread_crumb} x direction {0, 45}, " if a.mixed else
                                    f"{a.object}_env ({'stand-in hull; ' if a.object == 'acorn' else ''}direction {a.direction}), ") +
                                   f"{a.envs} envs/GPU, macro-step + observation + PPO actor-critic fwd each step, "
                                   f"PPO update every {a.rollout} steps ({a.epochs} epochs, minibatch {a.minibatch})",
                       "envs_per_gpu": a.envs, "parallelism": f"dp{world}", "ppo_in_timed_region": not a.no_ppo, "schedule": sched,
                       "step": f"{a.envs} completed env transitions per GPU"},
            # the workload-independent companion of `value`: physics.step() calls per second (env-steps/s depends on how many calls the action
            # source makes a macro step cost: ~232 for the synthetic U(-1,1) stream, ~312 for actions sampled from the learning policy)
            "action_source": a.actions, "mj_substeps_per_s": total_sub / dt, "mean_substeps_per_env_step": total_sub / max(total_env_steps, 1.0),
            "replicas_identical": replicas_identical, "policy_finite_after_run": policy_finite,
            "update_path": "explicit launch sequence (sb3/fused_update.py)" if getattr(model, "_fused", None) is not None else "autograd",
            # N > 1: device time of the one collective per optimiser step (the flat 4 MB fp32 gradient bucket over RCCL), rank 0's view
            "allreduce_ms_per_optimizer_step": ar_ms, "allreduce_steps_timed": ar_n, "env_steps_counted": total_env_steps, "env_steps_nominal": a.envs * world * a.steps, "short_rollouts": short_rollouts,
            # `achieved` / `peak` / `frac` are the HBM view the contract asks for (SURVEY.md 8(d) bytes / launch duration against 8 TB/s); `bound` names what binds
            # the kernel: instruction issue (roofline.valu: `frac` there is the issue-slot figure, the lane figure beside it)
            "roofline": {"bound": "valu-issue", "kernel": "k_macro_step", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms_avg": k_ms, "launches": k_n,
                         "launch_timing": ("device wall-clock stamps of every launch of the timed region (first workgroup's start to the last wave's end, 100 MHz), "
                                           "graph replays included" if k_n != ev_n or k_ms != ev_ms else "HIP events around the launch on its stream"),
                         # both timings always, under fixed names (launch_ms_avg above is the device figure when there is one: `launch_timing` says which)
                         "launch_ms_avg_device": d_ms if d_n else None, "launches_device": int(d_n), "launch_ms_avg_events": ev_ms, "launches_events": ev_n,
                         "algorithmic_bytes_per_launch": macro_bytes, "overhead_bytes_per_launch": overhead_bytes,
                         "bytes_definition": "SURVEY.md 8(d): physics state in + out (348 B f32 per env and launch) + action / outputs of the macro steps that end in the launch; "
                                             "overhead = suspended macro-step context + narrow-phase pair memory of the time-sliced schedule (not counted in achieved)",
                         "note": "the macro-step kernel is bound by instruction issue, not by HBM (7e-5 of the HBM roofline by construction: SURVEY.md 8(d)); round 5 measured what a "
                                 "SIMD issues (tools/hiptests/t_simd_rate.hip): a wave issues at most one instruction per ~4.5-5 cycles (scalar, compare and cross-lane instructions ~8.5), two "
                                 "waves of pure vector streams one per 2.4 (VOP2) ... 3.5 (VOP3) cycles per SIMD; this kernel's two waves issue one VALU instruction per ~4 (DESIGN.md)"},
        }
        if ar is not None:
            out["ticks"] = ar.total_ticks - ticks0
            # HBM traffic and instruction counts cannot be read from inside the process: they come from the committed rocprofv3 --pmc
            # passes of THIS command at THIS configuration (profiles/r03_pmc_summary.json: separate passes, gfx950 FETCH_SIZE
            # correction) and are attached only when the run is that configuration; otherwise traffic stays null.
            try:
                from mujoco_rl_manipulate_unknown_objects_amd import engine as _eng
                pmc_file = next(f for f in ("r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                pmj = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                same_cfg = (pmj.get("config") == {"object": a.object, "envs": a.envs, "state_dtype": a.state_dtype, "mixed": bool(a.mixed)})
                pm = pmj["kernels"]["k_macro_step"]
                # counters of ANOTHER build say nothing about this one: the summary records the fingerprint of the sources it was collected on
                # (engine.source_fingerprint(), tools/pmc_run.sh) and is attached only when the running tree has the same one
                same_src = pmj.get("csrc_sha16") == _eng.source_fingerprint()
                if same_cfg and not a.lockstep and not same_src:
                    out["roofline"]["traffic_note"] = (f"profiles/{pmc_file} was collected on sources {pmj.get('csrc_sha16')}, this run is {_eng.source_fingerprint()}: "
                                                       "counter-derived fields (traffic, valu) left out")
                if same_cfg and not a.lockstep and same_src:
                    rf = out["roofline"]
                    rf["traffic"] = pm["hbm_bytes_per_launch_fetch_doubled"]
                    rf["traffic_source"] = (f"profiles/{pmc_file}, same command and configuration: (2 x FETCH_SIZE + WRITE_SIZE) per launch; raw "
                                            f"{pm['hbm_bytes_per_launch_raw']:.0f} B")
                    # the bound that really binds this kernel: VALU issue. lane-operations per physics.step() of one env from the PMC
                    # pass (SQ_INSTS_VALU x 64 lanes / env-substeps of the launch) x the LIVE physics.step() rate of this run
                    # the bound that really binds this kernel: VALU issue. Lane-operations of one launch from the PMC pass (SQ_INSTS_VALU x 64
                    # lanes, same command and configuration) / the LIVE launch duration of this run -- formed like the HBM figure above
                    lane_ops = pm["SQ_INSTS_VALU"] * 64.0 / (k_ms * 1e-3) if k_ms > 0 else 0.0
                    # `frac`: VALU instructions the SIMD's two waves issue per quad-cycle (2 x SQ_INSTS_VALU / SQ_WAVE_CYCLES; SQ_ACTIVE_INST_VALU counts exactly one
                    # quad-cycle per instruction) -- the issue-slot figure. Calibration on pure streams, two waves per SIMD (profiles/r05_simd_issue_rates.txt): VOP2
                    # 1.7, VOP3 1.15 per quad-cycle, so 1.0 is not a ceiling of the hardware but ~0.9 is where this instruction mix (15 % scalar, compares,
                    # cross-lane, LDS) saturates two waves. The lane view (64 lanes x instructions / time against the fp32 peak) is `lane_frac`.
                    rf["valu"] = {"frac": min(1.0, 2.0 * pm["active_inst_valu_frac"]), "frac_is": "VALU instructions issued per quad-cycle and SIMD (two waves)",
                                  "achieved": lane_ops, "peak": VALU_PEAK_LANE_OPS, "unit": "fp32 lane-ops/s", "lane_frac": lane_ops / VALU_PEAK_LANE_OPS,
                                  "wave_issue_frac": pm.get("active_inst_any_frac"),
                                  "lane_ops_per_launch": pm["SQ_INSTS_VALU"] * 64.0, "lane_ops_per_env_substep": pm.get("valu_lane_ops_per_env_substep"),
                                  "valu_active_frac": pm["active_inst_valu_frac"], "wait_frac": pm["wait_any_frac"],
                                  # how much of an issued wave instruction's 64 lanes can do anything: two envs x 16 lanes own a wave; lanes 32..63
                                  # are enabled clones of lanes 0..31 that take a second data set where one instruction stream can serve two (support
                                  # searches of a pair's two hulls, the cooperative vertex scan, two of a contact's four constraint rows) and
                                  # repeat the lower half's arithmetic everywhere else
                                  "enabled_lane_frac": 1.0, "distinct_lane_frac_outside_shared_phases": 0.5,
                                  # what really bounds the kernel: VALU instruction ISSUE. SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES is per wave; the two waves
                                  # of a SIMD share its one vector ALU, so the ALU is busy 2 x that fraction of the time ("Round 4")
                                  "simd_valu_busy_frac": min(1.0, 2.0 * pm["active_inst_valu_frac"]), "waves_per_simd": 2,
                                  "valu_instructions_per_env_substep": (pm.get("physics_only_probe") or {}).get("valu_wave_instructions_per_env_substep"),
                                  "source": f"profiles/{pmc_file} (SQ_INSTS_VALU per launch of the same command) / live launch duration"}
            except Exception:
                pass
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(a.object)
            except Exception as ex:      # the oracle is optional test infrastructure
                out["cpu_baseline"] = {"value": None, "error": str(ex)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
