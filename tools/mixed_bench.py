"""Step time of a mixed batch against its groups run alone (are the per-group launches concurrent?)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine

per = int(sys.argv[1]) if len(sys.argv) > 1 else 512
groups = [(o, per, d) for o in ("acorn", "sand_ball", "sugar_cube", "bread_crumb") for d in ((1, 0), (1, 1))]
mb = engine.MixedBatch(groups, auto_reset=1)
rng = np.random.default_rng(0)
acts = [torch.from_numpy(rng.uniform(-1, 1, (mb.n, 6)).astype(np.float32)).cuda() for _ in range(12)]

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for i in range(4):
    mb.step(acts[i])
print(f"mixed step ({len(groups)} groups x {per} envs, one launch): {timed(lambda i: mb.step(acts[4 + i]), 8):.2f} ms")
print(f"mixed observe: {timed(lambda i: mb.observe(), 4):.2f} ms")
for g, p in enumerate(mb.parts):
    lo, hi = mb.offsets[g], mb.offsets[g + 1]
    ms = timed(lambda i: p.step(acts[4 + i][lo:hi]), 4)
    print(f"  group {g} {groups[g][0]:12s} dir {groups[g][2]} alone: {ms:.2f} ms, substeps mean {p.out['n_substeps'].float().mean().item():.0f} max {p.out['n_substeps'].max().item()}")
