"""Which physics.step() calls run the Newton solver into its iteration cap (fault bit 4), and what the oracle makes of exactly those states.
    python tools/build_variant.py capdump -DGRIP_CAPDUMP          (the diagnostic build that records them: csrc/grip_physics.h, GRIP_CAPDUMP)
    python tools/newton_cap_probe.py [object] [ticks] [out.json] [variant]
Runs the physics-only time-sliced workload of tools/physics_rate.py (4096 envs, U(-1,1) actions, 144 / 3000 us, capacity 2048) on the capdump build, reads the
records -- the state each capped step started from, the kernel's last iterate, per Newton iteration the scaled gradient / step length / cost / line-search
evaluations -- and replays every record on the CPU oracle (oracle/grip_physics.c:solve_newton: MuJoCo's 100 iterations at tolerance 1e-10, fp64): its iteration
count, its converged qacc against the kernel's last iterate, the one-step velocity difference h * |dqacc|."""
import sys, os, json, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine
from oracle import orc

obj = sys.argv[1] if len(sys.argv) > 1 else "acorn"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
out_path = sys.argv[3] if len(sys.argv) > 3 else None
variant = sys.argv[4] if len(sys.argv) > 4 else "capdump"
engine.LIB_PATH = os.path.join(engine.CSRC, f"libgrip_sim_{variant}.so")
REC, WORDS, MAXIT = 256, 256, 30
n, cap, S, bud = 4096, 2048, 144, 3000
b = engine.Batch(obj, n, auto_reset=1)
lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(0)
total = torch.zeros(1, dtype=torch.int64, device="cuda"); subs = torch.zeros(1, dtype=torch.int64, device="cuda"); f4 = torch.zeros(1, dtype=torch.int64, device="cuda")
ar = torch.arange(cap, device="cuda")
for _ in range(ticks):
    act = torch.rand(cap, 6, device="cuda", generator=g) * 2 - 1
    out = b.advance(act, S, lst, cnt, bud)
    valid = (ar < cnt) & (lst >= 0)
    idx = lst.clamp(min=0).long()
    total.add_(valid.sum()); subs.add_((out["n_substeps"][idx] * valid).sum()); f4.add_((((out["fault"][idx] & 4) != 0) & valid).sum())
torch.cuda.synchronize()
buf = (C.c_float * (REC * WORDS))(); cnt2 = (C.c_uint * 2)()
rc = engine.lib().grip_debug_capdump(buf, cnt2)
cnt_c = C.c_uint(cnt2[0]); restarts = int(cnt2[1])
assert rc == 0, rc
recs = np.ctypeslib.as_array(buf).reshape(REC, WORDS)[: min(cnt_c.value, REC)].copy()
print(f"{obj}: {int(total.item())} macro steps, {int(subs.item())} physics.step() calls, {int(f4.item())} finished macro steps with fault bit 4, {cnt_c.value} solves recorded (capped or restarted; the first {REC} kept), {restarts} solves restarted from qacc_smooth", flush=True)

m = orc.Model(obj)
h = m.scalar("timestep") if hasattr(m, "scalar") else 2e-3
rows = []
for r in recs:
    s = orc.Sim(m)
    s.d.arr("qpos")[:] = r[8:22]; s.d.arr("qvel")[:] = r[24:37]; s.d.arr("ctrl")[:] = r[40:47]; s.d.arr("qacc_warmstart")[:] = r[48:61]
    s.d.arr("xfrc")[1][2] = 0.438 * 9.81                      # robot_env.py:64-65 (B_EE = 1)
    s.forward()
    qo = np.array(s.d.arr("qacc")); qk = r[64:77].astype(np.float64); qs_k = r[80:93].astype(np.float64); qs_o = np.array(s.d.arr("qacc_smooth"))
    # ... and cold (no warm start): does the oracle need the warm start to get there?
    s2 = orc.Sim(m)
    s2.d.arr("qpos")[:] = r[8:22]; s2.d.arr("qvel")[:] = r[24:37]; s2.d.arr("ctrl")[:] = r[40:47]; s2.d.arr("xfrc")[1][2] = 0.438 * 9.81
    s2.forward()
    qc = np.array(s2.d.arr("qacc"))
    hist = r[96:96 + 4 * MAXIT].reshape(MAXIT, 4)
    pairs = sorted({(int(c.g1), int(c.g2)) for c in s.contacts()})
    rows.append(dict(kind={1: 'capped', 2: 'restarted'}.get(int(r[4]), '?'), kernel_ncon=int(r[0]), kernel_iters=int(r[1]), coupled=int(r[2]) & 1, took_warm=int(r[3]),
                     oracle_ncon=int(s.d.ncon), oracle_iters=int(s.d.solver_iter), oracle_iters_cold=int(s2.d.solver_iter), oracle_pairs=pairs,
                     dqacc_max=float(np.abs(qk - qo).max()), dqacc_vs_cold_oracle=float(np.abs(qk - qc).max()), oracle_warm_vs_cold=float(np.abs(qo - qc).max()), qacc_max=float(np.abs(qo).max()), one_step_dv=float(2e-3 * np.abs(qk - qo).max()),
                     dqacc_smooth_max=float(np.abs(qs_k - qs_o).max()),
                     grad_first=float(hist[0, 0]), grad_last=float(hist[-1, 0]), grad_min=float(hist[:, 0].min()), alpha_last=float(hist[-1, 1]),
                     cost_first=float(hist[0, 2]), cost_last=float(hist[-1, 2]), ls_evals_mean=float(hist[:, 3].mean()),
                     grad_hist=[float(x) for x in hist[:, 0]], alpha_hist=[float(x) for x in hist[:, 1]], cost_hist=[float(x) for x in hist[:, 2]],
                     qpos=[float(x) for x in r[8:22]], qvel=[float(x) for x in r[24:37]], ctrl=[float(x) for x in r[40:47]], warm=[float(x) for x in r[48:61]]))
for i, w in enumerate(rows):
    print(f"[{i}] {w['kind']} after {w['kernel_iters']} iterations, ncon k/o {w['kernel_ncon']}/{w['oracle_ncon']} pairs {w['oracle_pairs']} coupled {w['coupled']} warm {w['took_warm']} | oracle iters {w['oracle_iters']} (cold {w['oracle_iters_cold']}) | "
          f"|qacc_k - qacc_o| max {w['dqacc_max']:.3e} of {w['qacc_max']:.3e} -> dv {w['one_step_dv']:.2e} m/s; vs the oracle's cold solve {w['dqacc_vs_cold_oracle']:.3e} (oracle warm vs cold {w['oracle_warm_vs_cold']:.3e}) | scaled grad first {w['grad_first']:.2e} min {w['grad_min']:.2e} last {w['grad_last']:.2e} "
          f"alpha last {w['alpha_last']:.3g} ls evals {w['ls_evals_mean']:.1f} cost {w['cost_first']:.6g} -> {w['cost_last']:.6g}")
for kind in ("capped", "restarted"):
    sel = [w for w in rows if w["kind"] == kind]
    if sel:
        dq = np.array([w["dqacc_vs_cold_oracle"] for w in sel]); it = np.array([w["kernel_iters"] for w in sel]); qa = np.array([w["qacc_max"] for w in sel])
        print(f"{kind}: {len(sel)} records; kernel iterations median {np.median(it):.0f} max {it.max()}; |qacc_kernel - qacc_oracle(cold, converged)| median {np.median(dq):.2e} p90 {np.quantile(dq, 0.9):.2e} max {dq.max():.2e} "
              f"(relative to max |qacc|: median {np.median(dq / qa):.1e} max {(dq / qa).max():.1e}); one-step velocity difference max {2e-3 * dq.max():.1e} m/s")
summary = dict(object=obj, ticks=ticks, macro_steps=int(total.item()), physics_steps=int(subs.item()), macro_steps_with_fault_bit_4=int(f4.item()), capped_solves=int(cnt_c.value), restarted_solves=restarts, records=rows)
if out_path:
    json.dump(summary, open(out_path, "w"), indent=1)
