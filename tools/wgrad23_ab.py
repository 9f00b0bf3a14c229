"""grip_wgrad23 (k_wgrad23_b3 + k_wgrad23_reduce: the second and third convolution's weight gradients on the bf16 matrix pipe) against the tensor library's two
convolution_backward calls it replaces in the update: time per 4096-sample minibatch."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23_weight_gradients
th.backends.cudnn.benchmark = True
g = th.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: th.randn(*s, device="cuda", generator=g)
cl = lambda t: t.contiguous(memory_format=th.channels_last)
cb = th.ops.aten.convolution_backward
for n in (1024, 4096):
    y1, y2 = cl(th.relu(rnd(n, 32, 15, 15))), cl(th.relu(rnd(n, 64, 6, 6)))
    g2m, g3m = cl(rnd(n, 64, 6, 6) * (rnd(n, 64, 6, 6) > 0)), cl(rnd(n, 64, 4, 4) * (rnd(n, 64, 4, 4) > 0))
    w2, w3 = cl(rnd(64, 32, 4, 4)), cl(rnd(64, 64, 3, 3))
    o2, o3 = th.empty_like(w2), th.empty_like(w3)
    def mine(): conv23_weight_gradients(y1, g2m, y2, g3m, gw2_out=o2, gw3_out=o3)
    def lib():
        o3.copy_(cb(g3m, y2, w3, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1])
        o2.copy_(cb(g2m, y1, w2, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1])
    for name, f in (("grip_wgrad23", mine), ("tensor library (two convolution_backward + copies)", lib)):
        for _ in range(10): f()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        th.cuda.synchronize(); e0.record()
        for _ in range(100): f()
        e1.record(); th.cuda.synchronize()
        print(f"n {n}: {name}: {e0.elapsed_time(e1) * 10:.1f} us", flush=True)
