"""k_conv23_b3 (bf16 matrix pipe, three-term split, the shipped kernel) against k_conv23 (fp32 MFMA, GRIP_CONV23_F32=1): time per launch and the difference of their outputs.
    python tools/conv23_ab.py            -> runs itself twice (one process per kernel: the switch is read once per process) and prints both"""
import sys, os, subprocess; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    import torch.nn.functional as F
    from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23, conv23_prep
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    w2 = rnd(64, 32, 4, 4) / 22.0; b2 = 0.1 * rnd(64); w3 = rnd(64, 64, 3, 3) / 24.0; b3 = 0.1 * rnd(64)
    mats = conv23_prep(w2, w3)
    for n, train in ((1024, False), (1536, False), (4096, False), (4096, True)):
        y1 = torch.relu(rnd(n, 32, 15, 15)).contiguous(memory_format=torch.channels_last)
        ref = F.relu(F.conv2d(F.relu(F.conv2d(y1.double(), w2.double(), b2.double(), stride=2)), w3.double(), b3.double()))
        out = conv23(y1, mats[0], b2, mats[1], b3, train=train)
        o = out[0] if train else out
        err = float((o.double() - ref).abs().max()) / float(ref.abs().max())
        for _ in range(20): conv23(y1, mats[0], b2, mats[1], b3, train=train)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): conv23(y1, mats[0], b2, mats[1], b3, train=train)
        e1.record(); torch.cuda.synchronize()
        print(f"  n {n:5d}{' train' if train else '      '}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per launch; max |out - fp64 reference| / max |reference| = {err:.2e}")
    sys.exit(0)
for name, env in (("k_conv23_b3 (bf16 x 3 terms)", {}), ("k_conv23 (fp32 MFMA, GRIP_CONV23_F32=1)", {"GRIP_CONV23_F32": "1"})):
    print(name, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), check=True)
