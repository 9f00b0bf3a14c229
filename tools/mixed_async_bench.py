"""Tick time of the time-sliced schedule over a mixed batch (advance only / with observe_list), against one batch of the same size."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mujoco_rl_manipulate_unknown_objects_amd import engine

per, cap, S, bud = 512, 1024, 96, 2000
groups = [(o, per, d) for o in ("acorn", "sand_ball", "sugar_cube", "bread_crumb") for d in ((1, 0), (1, 1))]
def run(name, b, render):
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    obs = torch.zeros(cap, 5, 64, 64, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    def tick():
        act = torch.randn(cap, 6, device="cuda", generator=g).clamp(-1, 1)
        b.advance(act, S, lst, cnt, bud)
        if render: b.observe_list(lst, cnt, obs)
        total.add_((lst >= 0).sum())
    for _ in range(100): tick()
    torch.cuda.synchronize(); b.kernel_time(True); t0 = time.time(); c0 = int(total.item()); T = 200
    for _ in range(T): tick()
    torch.cuda.synchronize(); dt = time.time() - t0; c1 = int(total.item())
    kms, kn = b.kernel_time(True)
    print(f"{name:34s} render {render}: {(c1 - c0) / dt:9.0f} env-steps/s, {dt / T * 1e3:.2f} ms/tick, slice kernel {kms:.2f} ms, ready/tick {(c1 - c0) / T:.0f}", flush=True)
import itertools
for G in (8, 4):
    gs = [(o, 4096 // G, d) for o, d in list(itertools.product(("acorn", "sand_ball", "sugar_cube", "bread_crumb"), ((1, 0), (1, 1))))[:G]]
    mb = engine.MixedBatch(gs, auto_reset=1)
    run(f"mixed {G} x {4096 // G}", mb, 0); mb.close()
gs = [("acorn", 512, (1, 0))] * 8
mb = engine.MixedBatch(gs, auto_reset=1)
run("acorn 8 x 512", mb, 0); mb.close()
