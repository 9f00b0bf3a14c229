"""Distribution of physics.step() calls per macro step in the time-sliced workload (4096 envs, U(-1,1) actions, 144 / 3000 us, capacity 2048): which macro steps end in
their first, second, third ... slice, and where inside it."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine
obj = sys.argv[1] if len(sys.argv) > 1 else "acorn"
n, cap, S, bud = 4096, 2048, 144, 3000
b = engine.Batch(obj, n, auto_reset=1)
lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(0)
hist = torch.zeros(1024, dtype=torch.int64, device="cuda"); ar = torch.arange(cap, device="cuda")
for t in range(2000):
    act = torch.rand(cap, 6, device="cuda", generator=g) * 2 - 1
    out = b.advance(act, S, lst, cnt, bud)
    if t >= 1000:
        valid = (ar < cnt) & (lst >= 0)
        ns = out["n_substeps"][lst.clamp(min=0).long()].clamp(max=1023).long()
        hist.scatter_add_(0, ns, valid.long())
h = hist.cpu().numpy(); tot = h.sum()
import numpy as np
print(f"{obj}: {tot} macro steps; physics.step() calls per macro step: mean {float((h * np.arange(1024)).sum()) / tot:.1f}")
edges = [0, 48, 96, 144, 192, 240, 288, 336, 384, 432, 576, 1024]
for a, z in zip(edges[:-1], edges[1:]): print(f"  {a:4d} .. {z - 1:4d}: {100.0 * h[a:z].sum() / tot:5.1f} %")
top = np.argsort(-h)[:12]
print("  most frequent counts:", ", ".join(f"{int(i)} ({100.0 * h[i] / tot:.1f} %)" for i in top))
