"""Raw throughput of the time-sliced engine (random actions, no policy): env-steps/s vs slice length and list capacity."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine

obj = sys.argv[1] if len(sys.argv) > 1 else "acorn"
n = 4096
for cap, S, render, bud in [(1024, 32, 0, 0), (1024, 16, 0, 0), (1024, 64, 0, 1500), (1024, 64, 0, 1000), (1024, 48, 0, 700), (1024, 128, 0, 1500), (1024, 64, 1, 1000)]:
    b = engine.Batch(obj, n, auto_reset=1)
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    obs = torch.zeros(cap, 5, 64, 64, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    def tick():
        act = torch.randn(cap, 6, device="cuda", generator=g).clamp(-1, 1)
        b.advance(act, S, lst, cnt, bud)
        if render: b.observe_list(lst, cnt, obs)
        total.add_(cnt)
    for _ in range(300): tick()
    torch.cuda.synchronize(); b.kernel_time(True); t0 = time.time(); c0 = int(total.item()); T = 400
    for _ in range(T): tick()
    torch.cuda.synchronize(); dt = time.time() - t0; c1 = int(total.item())
    kms, kn = b.kernel_time(True)
    print(f"cap {cap} slice {S} budget {bud} render {render}: {(c1 - c0) / dt:9.0f} env-steps/s, {dt / T * 1e3:.2f} ms/tick, slice kernel {kms:.2f} ms, ready/tick {(c1 - c0) / T:.0f}")
    b.close()
