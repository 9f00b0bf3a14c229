"""Diagnostic: per-phase cycle shares of one physics.step() (uses the -DGRIP_STAMPS build, never the shipped library)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, "libgrip_sim_stamps.so")
obj = sys.argv[1] if len(sys.argv) > 1 else "acorn"
mode = sys.argv[2] if len(sys.argv) > 2 else "rest"
n, k = 64, 100
b = engine.Batch(obj, n); b.reset(); b.substep(300); torch.cuda.synchronize()
if mode == "push":
    q, v, c, w = b.get_state(); q[:, 0] = 0.2; q[:, 2] = 0.05; b.set_state(qpos=q)
    c[:, 0] = 1.0; b.set_state(ctrl=c); b.substep(60); torch.cuda.synchronize()
b.substep(k); torch.cuda.synchronize()
out = (C.c_ulonglong * 20)()
assert engine.lib().grip_debug_stamps(out) == 0
names = ["kinematics", "collide", "mass+bias+qs", "make_constraints", "solve: other (LS, bookkeeping)", "integrate", "solve: constraint pass", "solve: tri-solves+gather", "solve: assemble rows", "solve: cholesky", "solve: line search", "-", "-", "-", "solve: prologue", "solve: stage logic", "solve: hessian_vectors", "solve: p readback, Mp, Jp", "solve: loop exit", "solve: final gathers"]
tot = sum(out[:11]) + sum(out[14:20])
for nm, v in zip(names, out):
    if v: print(f"{nm:34s} {v / k:10.0f} cycles/substep  {100 * v / tot:5.1f} %")
print("total", tot / k, "cycles/substep;  ncon", np.bincount(b.debug_forward()["ncon"]))
