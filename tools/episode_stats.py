"""First-episode statistics under uniformly random actions, per object and direction, for comparison with the early episodes
of the reference's own training logs (models/trained_models/**/log_file.monitor.csv: the only physics-level data the reference holds)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine

per = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
objs = ("acorn", "sand_ball", "sugar_cube", "bread_crumb")
groups = [(o, per, d) for o in objs for d in ((1, 0), (1, 1))]
mb = engine.MixedBatch(groups, auto_reset=1)
mb.reset()
n = mb.n
g = torch.Generator(device="cuda"); g.manual_seed(0)
ret = torch.zeros(n, device="cuda"); length = torch.zeros(n, dtype=torch.int32, device="cuda")
first_ret = torch.full((n,), float("nan"), device="cuda"); first_len = torch.zeros(n, dtype=torch.int32, device="cuda"); first_status = torch.zeros(n, dtype=torch.int32, device="cuda")
open_ = torch.ones(n, dtype=torch.bool, device="cuda")
for t in range(401):
    if dist == "uniform":
        a = torch.rand(n, 6, device="cuda", generator=g) * 2 - 1
    else:       # tanh-squashed unit Gaussian: what an untrained SAC actor samples
        a = torch.tanh(torch.randn(n, 6, device="cuda", generator=g))
    out = mb.step(a)
    ret += out["reward"]; length += 1
    d = out["done"].bool() & open_
    first_ret[d] = ret[d]; first_len[d] = length[d]; first_status[d] = out["status"][d]
    open_ &= ~d
    if not open_.any():
        break
torch.cuda.synchronize()
fr, fl, fs = first_ret.cpu().numpy(), first_len.cpu().numpy(), first_status.cpu().numpy()
res = {}
for gi, (o, _, d) in enumerate(groups):
    lo, hi = mb.offsets[gi], mb.offsets[gi + 1]
    r, l, s = fr[lo:hi], fl[lo:hi], fs[lo:hi]
    key = f"{o}_dir{0 if d == (1, 0) else 45}"
    res[key] = dict(episodes=int(hi - lo), frac_time_limit=float((l == 400).mean()), frac_fail=float((s == 1).mean()), len_mean=float(l.mean()),
                    len_quartiles=[float(x) for x in np.percentile(l, [25, 50, 75])], ret_mean=float(np.nanmean(r)),
                    ret_quartiles=[float(x) for x in np.nanpercentile(r, [25, 50, 75])], ret_max=float(np.nanmax(r)))
    print(key, json.dumps(res[key]))
json.dump(dict(actions=dist, per_group=per, groups=res), open(os.path.join("gpurun_out", f"episode_stats_{dist}.json"), "w"), indent=1)
