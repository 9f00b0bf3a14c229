"""Where k_trunk_bwd_b3 spends a wave's cycles (diagnostic build: python tools/build_variant.py tbst -DTB_STAMPS): s_memtime stamps of wave 0 of workgroup 0 per phase."""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, "libgrip_sim_tbst.so")
from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23_prep, trunk_backward
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = th.Generator(device="cuda").manual_seed(3)
rnd = lambda *s: th.randn(*s, device="cuda", generator=g)
cl = lambda t: t.contiguous(memory_format=th.channels_last)
w1, w2, w3 = cl(rnd(32, 4, 8, 8) * 0.05), cl(rnd(64, 32, 4, 4) * 0.05), cl(rnd(64, 64, 3, 3) * 0.05)
b2m, b3m = conv23_prep(w2, w3)
obs = th.randint(0, 256, (n, 5, 64, 64), device="cuda", dtype=th.uint8, generator=g)
g3 = cl(rnd(n, 64, 4, 4))
rbits = lambda shape, dt: th.randint(0, 2 ** 31 - 1, shape, device="cuda", dtype=th.int64, generator=g).to(dt)
m3 = (rbits((n, 16), th.int64) << 32) | rbits((n, 16), th.int64); m2 = (rbits((n, 36), th.int64) << 32) | rbits((n, 36), th.int64); m1 = (rbits((n, 225), th.int64) * 3 & 0xffffffff).to(th.int32)
run = lambda: trunk_backward(g3, m3, m2, m1, obs, b3m, b2m, w1)
for _ in range(3): run()
out = (C.c_ulonglong * 8)(); engine.lib().grip_debug_tb_stamps(out)
reps = 10
for _ in range(reps): run()
engine.lib().grip_debug_tb_stamps(out)
groups = -(-n // 2); trips = -(-groups // min(groups, 256)); tot = sum(out)
names = ["g3m pass, zero fill, barrier", "GEMM 1 + barrier", "g2m pass, zero fill, barrier", "GEMM 2 + barrier", "first layer: mask + split pass, planes (two images)", "first layer: K loops (two images)"]
print(f"n {n}: {tot / reps:.0f} cycles per launch for workgroup 0 ({trips} groups: {tot / reps / trips:.0f} per group)")
for i, nm in enumerate(names): print(f"   {nm:48s} {out[i] / reps / trips:8.0f} cycles per group  {100 * out[i] / tot:5.1f} %")
