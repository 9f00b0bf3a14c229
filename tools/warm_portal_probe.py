"""The narrow phase starts the portal refinement of a touching pair from the portal it converged to one physics.step() earlier;
csrc/libgrip_sim_cold.so (-DGRIP_COLD_PORTAL) is the comparison build that starts from scratch as libccd / MuJoCo do. Reports the
cost of a step in contact-rich states and how far the trajectories of the two builds move apart over a few steps.
usage: python tools/warm_portal_probe.py [object]   (both libraries must have been built; each runs in a child process)"""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, time
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    engine.LIB_PATH = sys.argv[2]; obj = sys.argv[3]
    n = 4096
    b = engine.Batch(obj, n, auto_reset=1)
    state_file = os.path.join(ROOT, "gpurun_out", f"warm_probe_state_{obj}.npz")
    if not os.path.exists(state_file):         # the common start state comes from the first library run
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        for t in range(150):
            b.step(torch.rand(n, 6, device="cuda", generator=g) * 2 - 1)
        torch.cuda.synchronize()
        q0 = b.get_state(); os.makedirs(os.path.dirname(state_file), exist_ok=True)
        np.savez(state_file, *q0)
    z = np.load(state_file); q0 = tuple(z[f"arr_{i}"] for i in range(4))
    out = {}
    for k in (1, 5, 25):                       # k physics.step() calls from the same state: one launch, memo alive inside it
        b.set_state(*q0); b.substep(k); torch.cuda.synchronize()
        q = b.get_state(); out[f"qpos_after_{k}"] = q[0][:512].tolist(); out[f"qvel_after_{k}"] = q[1][:512].tolist()
    b.set_state(*q0)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); b.substep(100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100 * 1e6)
    out["us_per_step"] = min(ts)
    d = b.debug_forward(); out["ncon_hist"] = np.bincount(d["ncon"]).tolist()
    print("JSON" + json.dumps(out)); sys.exit(0)
import numpy as np
obj = sys.argv[1] if len(sys.argv) > 1 else "acorn"
sf = os.path.join(ROOT, "gpurun_out", f"warm_probe_state_{obj}.npz")
if os.path.exists(sf):
    os.remove(sf)
res = {}
for name in ("libgrip_sim_cold.so", "libgrip_sim.so"):
    lib = os.path.join(ROOT, "mujoco_rl_manipulate_unknown_objects_amd", "csrc", name)
    r = subprocess.run([sys.executable, __file__, "--child", lib, obj], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("JSON")]
    if not line:
        print(name, "failed:", r.stderr[-600:]); sys.exit(1)
    res[name] = json.loads(line[0][4:])
c, w = res["libgrip_sim_cold.so"], res["libgrip_sim.so"]
print(f"{obj}: physics.step() of 4096 envs in states of 150 random macro steps: cold {c['us_per_step']:.1f} us, warm {w['us_per_step']:.1f} us per launch; contacts {c['ncon_hist']}")
for k in (1, 5, 25):
    dq = np.abs(np.array(c[f"qpos_after_{k}"]) - np.array(w[f"qpos_after_{k}"])); dv = np.abs(np.array(c[f"qvel_after_{k}"]) - np.array(w[f"qvel_after_{k}"]))
    print(f"  after {k:2d} steps: max |dqpos| {dq.max():.2e} (99th pct {np.percentile(dq.max(1), 99):.2e}), max |dqvel| {dv.max():.2e} (99th pct {np.percentile(dv.max(1), 99):.2e}), envs identical {(dq.max(1) == 0).mean() * 100:.0f} %")
