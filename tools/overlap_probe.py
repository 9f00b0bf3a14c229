"""Does independent work on a second stream run WHILE the physics slice kernel occupies the chip?"""

def main():
    import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    n = 4096
    b = engine.Batch("acorn", n, auto_reset=1); b2 = engine.Batch("acorn", n, auto_reset=1)
    cap = 1024
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    act = torch.zeros(cap, 6, device="cuda")
    for _ in range(30):
        b.advance(torch.randn(cap, 6, device="cuda").clamp(-1, 1), 64, lst, cnt, 3000)
    obs = torch.zeros(512, 5, 64, 64, dtype=torch.uint8, device="cuda")
    l2 = torch.arange(512, dtype=torch.int32, device="cuda"); c2 = torch.tensor([512], dtype=torch.int32, device="cuda")
    x = torch.randn(1 << 20, device="cuda")
    side = torch.cuda.Stream()
    def side_work(kind):
        if kind == "elementwise":
            y = x
            for _ in range(20): y = y * 1.0001 + 0.1
        elif kind == "observe":
            b2.observe_list(l2, c2, obs)
        elif kind == "conv":
            global cin, conv
            conv(cin)
    conv = torch.nn.Conv2d(5, 32, 8, 4).cuda().to(memory_format=torch.channels_last); cin = torch.randn(1024, 5, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    for kind in ("elementwise", "observe", "conv"):
        side_work(kind); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record(side); side_work(kind); e1.record(side)
        torch.cuda.synchronize(); alone = e0.elapsed_time(e1)
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for mode in ("sleep", "nosleep", "event_before"):
            pre = torch.cuda.Event(); 
            b.advance(act, 8, lst, cnt, 300); pre.record()          # a short slice first; the side work depends on IT
            m0.record(); b.advance(act, 64, lst, cnt, 3000); m1.record()
            if mode == "sleep": time.sleep(0.0005)
            with torch.cuda.stream(side):
                if mode == "event_before": side.wait_event(pre)
                e0.record(side); side_work(kind); e1.record(side)
            torch.cuda.synchronize()
            print(f"{kind:12s} {mode:12s}: alone {alone:.3f} ms; next to the slice kernel ({m0.elapsed_time(m1):.2f} ms): {e0.elapsed_time(e1):.3f} ms, finished {m0.elapsed_time(e1):.2f} ms after the slice started")


if __name__ == "__main__":
    main()
