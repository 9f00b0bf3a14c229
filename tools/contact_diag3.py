"""Diagnostic (GPU box): worst one-step errors along the oracle's contact trajectories -- contacts and accelerations side by side."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc
from mujoco_rl_manipulate_unknown_objects_amd import engine
from test_gpu_contact import oracle_trajectory
z = np.load(os.path.join(ROOT, "tests", "golden", "contact_states.npz"))
obj = sys.argv[1]
m = orc.Model(obj)
cat = z[f"{obj}/category"]
pick = []
for c, k in (("push_reward", 3), ("close_code3_break", 3), ("close_code1", 1), ("close_code2", 1), ("hull_contact_move", 2), ("pad_grasp_nonzero", 2)):
    pick += list(np.where(cat == c)[0][:k])
pre, post, cons = [], [], []
for i in pick:
    a, b_, c_, mg = oracle_trajectory(orc, m, z, obj, i)
    pre += a; post += b_; cons += c_
n = len(pre)
f32 = lambda k: np.array([s[k] for s in pre], np.float32)
b = engine.Batch(obj, n)
b.set_state(f32(0), f32(1), f32(2), f32(3))
dbg = b.debug_forward()
b.substep(1); torch.cuda.synchronize()
gq, gv, _, _ = b.get_state()
nv = np.array([s[1] for s in post]); nq = np.array([s[0] for s in post])
ev = np.abs(gv - nv).max(1)
same = np.array([sorted((int(dbg["con"][k, c, 7]), int(dbg["con"][k, c, 8])) for c in range(dbg["ncon"][k])) == cons[k] for k in range(n)])
worst = list(np.argsort(-ev)[:6]) + list(np.where(~same)[0][:4])
for k in worst:
    s = orc.Sim(m)
    s.qpos[:] = f32(0)[k].astype(np.float64); s.qvel[:] = f32(1)[k].astype(np.float64); s.ctrl[:] = f32(2)[k].astype(np.float64); s.qacc_warmstart[:] = f32(3)[k].astype(np.float64)
    s.d.xfrc[1][2] = 0.438 * 9.81
    s.fwd_position(); s.forward()
    print(f"--- state {k}: qvel err {ev[k]:.3e}, contact sets same {same[k]}, oracle solver iters {s.d.solver_iter}, gpu newton iters {dbg['con'][k, 0, 9]:.0f}")
    for c in range(s.d.ncon):
        cc = s.d.con[c]
        print("   orc", cc.g1, cc.g2, "dist %.6f" % cc.dist, "pos", np.round(list(cc.pos), 5), "n", np.round(list(cc.frame)[:3], 5))
    for c in range(dbg["ncon"][k]):
        g = dbg["con"][k, c]
        print("   gpu", int(g[7]), int(g[8]), "dist %.6f" % g[6], "pos", np.round(g[:3], 5), "n", np.round(g[3:6], 5))
    print("   qacc orc", np.round(np.array(s.qacc), 3)); print("   qacc gpu", np.round(dbg["qacc"][k], 3))
    print("   qacc_smooth orc", np.round(np.array(s.qacc_smooth), 3)); print("   warm", np.round(f32(3)[k], 3))
