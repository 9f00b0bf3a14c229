"""Diagnostic: whole-GPU per-phase cycle shares of the physics kernel in the time-sliced workload (random actions).
Uses the -DGRIP_STAMPS build (libgrip_sim_stamps.so), never the shipped library."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, "libgrip_sim_stamps.so")
obj = sys.argv[1] if len(sys.argv) > 1 else "acorn"
n, cap = 4096, 1024
b = engine.Batch(obj, n, auto_reset=1)
lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(0)
out = (C.c_ulonglong * 32)()
def ticks(k):
    for _ in range(k):
        b.advance(torch.randn(cap, 6, device="cuda", generator=g).clamp(-1, 1), 96, lst, cnt, 2000)
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 150        # 150 ticks: fresh episodes (free motion); ~3000: envs at mixed episode phases
ticks(warm); engine.lib().grip_debug_stamps(out)
ticks(100); engine.lib().grip_debug_stamps(out)
names = ["kinematics", "collide", "mass+bias+qs", "make_constraints", "solve: other (bookkeeping)", "integrate", "solve: constraint pass", "solve: tri-solves+gather", "solve: assemble rows", "solve: cholesky", "solve: line search"]
names2 = {14: "solve: prologue (mrow, jar of both starts)", 15: "solve: stage logic after pricing", 16: "solve: hessian_vectors + sync",
          17: "solve: p readback, M p, J p, g0/g1", 18: "solve: loop exit", 19: "solve: final gathers"}
tot = sum(out[:11]) + sum(out[14:20])
names3 = {20: "collide: prologue (frames, sphere tests, portal rebuild)", 21: "collide: votes + cooperative support", 22: "collide: per-lane support of geom 2",
          23: "collide: per-lane support of geom 1", 24: "collide: floor branch (neighbours inside the margin)", 25: "collide: portal phase logic", 26: "collide: loop exit"}
for i, nm in names3.items():
    print(f"   {nm:58s} {100 * out[i] / tot:5.1f} %   (inside `collide`, whose own slot then holds only compaction + contact load)")
for nm, v in zip(names, out):
    print(f"{nm:34s} {100 * v / tot:5.1f} %")
for i, nm in names2.items():
    print(f"{nm:44s} {100 * out[i] / tot:5.1f} %")
print("wave-cycles per lane-0 env-substep:", tot / max(1, out[11]))
print("mean envs at work per wave loop trip: %.2f of 4;  wave-cycles per loop trip: %.0f" % (out[12] / max(1, out[13]), tot / max(1, out[13])))
