"""Device time of the hand-written convolution backward kernels (csrc/grip_train.hip) at the update's minibatch size, next to the tensor
library's kernels for the same work:   python tools/train_kernels_bench.py [n=4096]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch as th
from mujoco_rl_manipulate_unknown_objects_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda"
th.backends.cudnn.benchmark = True
cl = lambda t: t.contiguous(memory_format=th.channels_last)
y1 = cl(th.relu(th.randn(n, 32, 15, 15, device=dev)))
w2 = cl(th.randn(64, 32, 4, 4, device=dev) * 0.05); w3 = cl(th.randn(64, 64, 3, 3, device=dev) * 0.05)
b2 = th.zeros(64, device=dev); b3 = th.zeros(64, device=dev)
y2 = cl(th.relu(th.nn.functional.conv2d(y1, w2, b2, stride=2))); y3 = cl(th.relu(th.nn.functional.conv2d(y2, w3, b3)))
g3 = cl(th.randn_like(y3))
b2m, b3m = engine.conv23_prep(w2, w3)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def lib_dgrad():
    g3m = th.ops.aten.threshold_backward(g3, y3, 0.0)
    g2 = th.ops.aten.convolution_backward(g3m, y2, w3, [64], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    g2m = th.ops.aten.threshold_backward(g2, y2, 0.0)
    g1 = th.ops.aten.convolution_backward(g2m, y1, w2, [64], [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    return th.ops.aten.threshold_backward(g1, y1, 0.0)


obs = th.randint(0, 256, (n, 5, 64, 64), device=dev, dtype=th.uint8)
w1 = cl(th.randn(32, 4, 8, 8, device=dev) * 0.05)
pack = lambda y, dt: (((y.flatten(2) > 0).to(dt)) << th.arange(y.shape[1], device=dev, dtype=dt).view(1, -1, 1)).sum(1).contiguous()
m1, m2, m3 = pack(y1, th.int64).to(th.int32), pack(y2, th.int64), pack(y3, th.int64)


def lib_all():
    g1m = lib_dgrad()
    x, _ = engine.obs_preprocess(obs)
    return th.ops.aten.convolution_backward(g1m, x, w1, [32], [4, 4], [0, 0], [1, 1], False, [0, 0], 1, [False, True, True])


print(f"n = {n}")
print(f"conv23 data gradients + 3 masks: tensor library {timeit(lib_dgrad):8.1f} us   grip_trunk_backward (no observations) {timeit(lambda: engine.trunk_backward(g3, m3, m2, m1, None, b3m, b2m)):8.1f} us")
print(f"the same + first layer's weight and bias gradient: tensor library {timeit(lib_all):8.1f} us   grip_trunk_backward {timeit(lambda: engine.trunk_backward(g3, m3, m2, m1, obs, b3m, b2m, w1)):8.1f} us")
b1 = th.zeros(32, device=dev)


def lib_fwd():
    x, _ = engine.obs_preprocess(obs)
    a = th.relu_(th.nn.functional.conv2d(x, w1, b1, stride=4))
    a = th.relu_(th.nn.functional.conv2d(a, w2, b2, stride=2))
    return th.relu_(th.nn.functional.conv2d(a, w3, b3))


def our_fwd():
    a, _, _ = engine.conv1_u8(obs, w1, b1, with_mask=True)
    return engine.conv23(a, b2m, b2, b3m, b3, train=True)


print(f"training forward of the three layers: tensor library {timeit(lib_fwd):8.1f} us   conv1_u8 + conv23 (with y2 and the masks) {timeit(our_fwd):8.1f} us")
