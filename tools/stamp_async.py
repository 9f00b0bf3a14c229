"""Diagnostic: whole-GPU per-phase cycle shares of the physics kernel in the time-sliced workload (random actions). GRIP_STAMPS_LIB=hist
(a -DGRIP_STAMPS -DGRIP_HIST build) adds per-step histograms (Newton iterations, contacts, cone zones); its timings are distorted by the atomics.
Uses the -DGRIP_STAMPS build (libgrip_sim_stamps.so), never the shipped library."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, f"libgrip_sim_{os.environ.get('GRIP_STAMPS_LIB', 'stamps')}.so")
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
engine.lib().grip_debug_hist((C.c_ulonglong * 96)())
ticks(100); engine.lib().grip_debug_stamps(out)
hist = (C.c_ulonglong * 96)()
engine.lib().grip_debug_hist(hist)
names = {0: "kinematics", 1: "collide: compaction + contact load", 2: "dense (bias, mass matrix, qacc_smooth)", 3: "make_constraints (+ rows, start residuals)",
         4: "solve: other (bookkeeping)", 5: "integrate", 6: "solve: constraint pass", 7: "solve: tri-solves+gather", 8: "solve: assemble rows", 9: "solve: cholesky",
         10: "solve: line search", 14: "solve: prologue (mrow)", 15: "solve: stage logic after pricing", 16: "solve: hessian_vectors + sync",
         17: "solve: p readback, M p, J p, g0/g1", 18: "solve: loop exit (waiting for the wave-mates' iterations)", 19: "solve: hand-over",
         20: "collide: prologue (frames, sphere tests, portal rebuild)", 21: "collide: votes + cooperative support", 22: "collide: per-lane support of geom 2",
         23: "collide: per-lane support of geom 1", 27: "macro step: counters, clock, post-step transitions", 28: "macro step: first-step target pose, control hooks", 24: "collide: floor branch (neighbours inside the margin)", 25: "collide: portal phase logic", 26: "collide: loop exit"}
tot = sum(out[i] for i in names)
grp = {"collide": [1, 20, 21, 22, 23, 24, 25, 26], "dense": [0, 2, 3, 5], "macro": [27, 28], "solver": [4, 6, 7, 8, 9, 10, 14, 15, 16, 17, 18, 19]}
for g_, ids in grp.items():
    print(f"== {g_}: {100 * sum(out[i] for i in ids) / tot:5.1f} %")
    for i in ids:
        print(f"   {names[i]:62s} {100 * out[i] / tot:5.1f} %")
print("wave-cycles per lane-0 env-substep:", tot / max(1, out[11]))
print("mean envs at work per wave loop trip: %.2f of 2;  wave-cycles per loop trip: %.0f" % (out[12] / max(1, out[13]), tot / max(1, out[13])))
if not any(hist):          # the plain -DGRIP_STAMPS build keeps no per-step histograms (GRIP_STAMPS_LIB=hist does): nothing more to print
    sys.exit(0)
steps = max(1, hist[16])
print("per env and physics.step(): Newton iterations 0..6, 7+ :", " ".join(f"{100 * hist[i] / steps:.1f}%" for i in range(8)),
      "| mean %.2f" % (sum(i * hist[i] for i in range(8)) / steps))
print("per wave trip: iterations the wave ran (max over its envs) 0..6, 7+ :", " ".join(f"{100 * hist[24 + i] / max(1, sum(hist[24:32])):.1f}%" for i in range(8)),
      "| mean %.2f" % (sum(i * hist[24 + i] for i in range(8)) / max(1, sum(hist[24:32]))))
print("contacts 0, 1-2, 3-4, ... 13-14 :", " ".join(f"{100 * hist[8 + i] / steps:.1f}%" for i in range(8)))
print("constrained %.1f%%, gripper block in the solve %.1f%%, any hull contact %.1f%%, active joint limit %.1f%%" %
      tuple(100 * hist[i] / steps for i in (17, 18, 19, 20)))
for k in range(4):
    t = max(1, sum(hist[32 + 8 * k: 40 + 8 * k]))
    print(f"scaled gradient after Newton iteration {k + 1} (<1e-7, decades ..., >=1e-1):", " ".join(f"{100 * hist[32 + 8 * k + b] / t:.1f}%" for b in range(8)), f"({t} solves)")
ns = max(1, hist[73])
print("scaled gradient at the chosen start:", " ".join(f"{100 * hist[64 + b] / ns:.1f}%" for b in range(8)), f"; warm start taken {100 * hist[72] / ns:.1f}%")
for nm, o in (("hull", 74), ("floor", 81)):
    a = max(1, sum(hist[o:o + 3]))
    print(f"{nm} contacts per solve {a / ns:.2f}: zone at start (top, middle, bottom)", " ".join(f"{100 * hist[o + z] / a:.1f}%" for z in range(3)),
          "| at the end", " ".join(f"{100 * hist[o + 3 + z] / a:.1f}%" for z in range(3)), f"| zone changed {100 * hist[o + 6] / a:.1f}%")
print(f"env-steps with a gripper-object contact (H couples the two blocks): {100 * hist[22] / steps:.1f}%; wave trips with at least one such env: {100 * hist[88] / max(1, hist[23]):.1f}%")
if hist[89]:
    n = hist[89]
    print(f"solves with contacts: start zones already final {100 * hist[93] / n:.1f}%; no middle-zone contact at the end {100 * hist[94] / n:.1f}%; same contact pattern as the previous step "
          f"{100 * hist[90] / n:.1f}%; previous solution's zones = this solution's (and no middle zone) {100 * hist[91] / n:.1f}%; of those the start classification was wrong {100 * hist[92] / n:.1f}%")
