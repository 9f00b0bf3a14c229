"""Pixels of the observation kernel, build A against build B, on the same states (4096 envs after random macro steps, all rendered):
    python tools/render_ab.py <variantA|-|rays> <variantB|-|rays> [object] [macro steps of pre-roll]        (rays: the shipped build with GRIP_OBSERVE_RAYS=1)
Each build runs in its own process (the library is chosen per process); prints the time per 1024 listed rows and how many bytes differ."""
import sys, os, subprocess; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    var, obj, pre, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    n = int(sys.argv[6]) if len(sys.argv) > 6 else 4096
    if var == "rays":                 # the shipped library's ray-casting kernel (rounds 1-4) instead of its rasteriser
        os.environ["GRIP_OBSERVE_RAYS"] = "1"
    elif var != "-":
        engine.LIB_PATH = os.path.join(engine.CSRC, f"libgrip_sim_{var}.so")
    b = engine.Batch(obj, n, auto_reset=1)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(pre):
        b.step(torch.rand(n, 6, device="cuda", generator=g) * 2 - 1)
    full = b.observe().clone()
    cnt = min(1024, n)
    lst = torch.arange(cnt, dtype=torch.int32, device="cuda") * (n // cnt); c = torch.tensor([cnt], dtype=torch.int32, device="cuda")
    rows = torch.zeros(cnt, 5, 64, 64, dtype=torch.uint8, device="cuda")
    b.observe_list(lst, c, rows); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.observe_list(lst, c, rows)
    e1.record(); torch.cuda.synchronize()
    print(f"{var} {obj}: 1024 listed rows {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    torch.save(full.cpu(), out)
    sys.exit(0)
import torch
a, b_ = sys.argv[1], sys.argv[2]
obj = sys.argv[3] if len(sys.argv) > 3 else "acorn"
pre = sys.argv[4] if len(sys.argv) > 4 else "40"
outs = []
for v in (a, b_):
    out = f"/tmp/render_ab_{v.replace('-', 'shipped')}.pt"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, obj, pre, out], check=True)
    outs.append(torch.load(out).int())
d = (outs[0] - outs[1]).abs()
px = (d.amax(1) > 0)                   # per env and pixel: any channel differs
print(f"{obj}: bytes differing {int((d > 0).sum())} of {d.numel()}, by more than 1: {int((d > 1).sum())}; pixels {int(px.sum())} in {int(px.flatten(1).any(1).sum())} of {d.shape[0]} envs; "
      f"depth-channel bytes differing {int((d[:, 3] > 0).sum())}, by more than 1: {int((d[:, 3] > 1).sum())}")
