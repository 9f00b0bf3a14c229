"""Where k_wgrad23_b3 spends its cycles (diagnostic build: python tools/build_variant.py wgst -DWG_STAMPS): s_memtime stamps of wave 0 of each role's first workgroup per
phase, and every workgroup's start / end (100 MHz clock) in the last launch."""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, "libgrip_sim_wgst.so")
from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23_weight_gradients
g = th.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: th.randn(*s, device="cuda", generator=g)
cl = lambda t: t.contiguous(memory_format=th.channels_last)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
y1, y2 = cl(th.relu(rnd(n, 32, 15, 15))), cl(th.relu(rnd(n, 64, 6, 6)))
g2m, g3m = cl(rnd(n, 64, 6, 6) * (rnd(n, 64, 6, 6) > 0)), cl(rnd(n, 64, 4, 4) * (rnd(n, 64, 4, 4) > 0))
for _ in range(3): conv23_weight_gradients(y1, g2m, y2, g3m)
out = (C.c_ulonglong * 8)(); span = (C.c_ulonglong * 1024)()
engine.lib().grip_debug_wg_stamps(out, span)
reps = 20
for _ in range(reps): conv23_weight_gradients(y1, g2m, y2, g3m)
engine.lib().grip_debug_wg_stamps(out, span)
words = int(engine.lib().grip_wgrad23_scratch_floats(n))
# the split: w2 * 32768 + w3 * 36864 = words, w2 + w3 = workgroups that wrote a span
wgs = sum(1 for i in range(512) if span[2 * i])
w3 = (words - wgs * 32768) // (36864 - 32768); w2 = wgs - w3
names = ["prologue (zero fill, addresses, first request)", "staging: wait + split + store", "barrier + next requests", "MFMA loop + barrier"]
for role, (nm, units, per) in enumerate((("role 2 (dW2, image pairs)", (n + 1) // 2, w2), ("role 3 (dW3, four images)", (n + 3) // 4, w3))):
    trips = -(-units // per)
    tot = sum(out[4 * role: 4 * role + 4])
    print(f"{nm}: {per} workgroups, {trips} trips of the first one; {tot / reps:.0f} cycles per launch")
    for i, x in enumerate(names): print(f"   {x:48s} {out[4 * role + i] / reps / (1 if i == 0 else trips):8.0f} cycles per {'launch' if i == 0 else 'trip'}  {100 * out[4 * role + i] / tot:5.1f} %")
s0 = min(span[2 * i] for i in range(wgs))
for nm, lo, hi in (("role 2", 0, w2), ("role 3", w2, wgs)):
    st = [span[2 * i] - s0 for i in range(lo, hi)]; en = [span[2 * i + 1] - s0 for i in range(lo, hi)]
    print(f"{nm}: workgroup start {min(st) / 100:.2f} .. {max(st) / 100:.2f} us, end {min(en) / 100:.2f} .. {max(en) / 100:.2f} us after the first start")
