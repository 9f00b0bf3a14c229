import sys, os; sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tools") else ".")
import torch, numpy as np
from mujoco_rl_manipulate_unknown_objects_amd import engine
engine.LIB_PATH = os.path.join(engine.CSRC, "libgrip_sim_dbg.so")
for obj in sys.argv[1:]:
    n = 4096
    b = engine.Batch(obj, n, auto_reset=1)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(40):
        b.step(torch.rand(n, 6, device="cuda", generator=g) * 2 - 1)
    full = b.observe().cpu().numpy()
    c = full.reshape(n, -1)[:, :16].copy().view(np.int32)
    for k, name in enumerate(("faces", "edges", "items", "planes")):
        print(obj, name, "mean %.1f  p50 %d  p90 %d  max %d" % (c[:, k].mean(), np.percentile(c[:, k], 50), np.percentile(c[:, k], 90), c[:, k].max()))
