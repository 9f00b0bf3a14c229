"""Where k_conv23_b3 spends a wave's cycles (diagnostic build: python tools/build_variant.py cbst -DCB_STAMPS): s_memtime stamps of wave 0 of workgroup 0 per phase."""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, "libgrip_sim_cbst.so")
from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23, conv23_prep
g = torch.Generator(device="cuda"); g.manual_seed(5)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
w2 = rnd(64, 32, 4, 4) / 22.0; b2 = 0.1 * rnd(64); w3 = rnd(64, 64, 3, 3) / 24.0; b3 = 0.1 * rnd(64)
mats = conv23_prep(w2, w3)
names = ["loop top / prologue", "y1: wait + split + store", "barrier 1", "GEMM 1", "y2 epilogue", "barrier 2", "GEMM 2", "output epilogue"]
for n, train in ((4096, False), (4096, True), (1536, False), (1024, False)):
    y1 = torch.relu(rnd(n, 32, 15, 15)).contiguous(memory_format=torch.channels_last)
    for _ in range(3): conv23(y1, mats[0], b2, mats[1], b3, train=train)
    out = (C.c_ulonglong * 12)(); engine.lib().grip_debug_cb_stamps(out)
    reps = 20
    for _ in range(reps): conv23(y1, mats[0], b2, mats[1], b3, train=train)
    engine.lib().grip_debug_cb_stamps(out)
    tot = sum(out[:9]); pairs = (n + 1) // 2; trips = -(-pairs // min(pairs, 256))
    print(f"n {n}{' train' if train else ''}: {tot / reps:.0f} cycles per launch for workgroup 0 ({trips} trips: {tot / reps / trips:.0f} per trip)")
    print(f"   prologue (conv3 fragments -> registers) {out[8] / reps:8.0f} cycles per launch; last launch, 100 MHz ticks after the first workgroup's start: last start {out[9]}, first end {out[10]}, last end {out[11]}")
    for i, nm in enumerate(names): print(f"   {nm:28s} {out[i] / reps / trips:8.0f} cycles per trip  {100 * out[i] / tot:5.1f} %")
