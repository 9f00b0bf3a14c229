"""Replica of the pipelined tick loop with synthetic side work, to find what stops the overlap."""

def main():
    import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    n = 4096
    b = engine.Batch("acorn", n, auto_reset=1); b2 = engine.Batch("acorn", n, auto_reset=1)
    cap = 1024
    lst = [torch.full((cap,), -1, dtype=torch.int32, device="cuda") for _ in range(2)]; cnt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
    act = [torch.randn(cap, 6, device="cuda").clamp(-1, 1) for _ in range(2)]
    obs = torch.zeros(cap, 5, 64, 64, dtype=torch.uint8, device="cuda")
    l2 = torch.arange(cap, dtype=torch.int32, device="cuda"); c2 = torch.tensor([512], dtype=torch.int32, device="cuda")
    conv = torch.nn.Conv2d(5, 32, 8, 4).cuda().to(memory_format=torch.channels_last); cin = torch.randn(1024, 5, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    small = torch.zeros(1024, device="cuda")
    side = torch.cuda.Stream()
    def side_work(kind):
        if "o" in kind: b2.observe_list(l2, c2, obs)
        if "c" in kind:
            with torch.no_grad(): conv(cin)
        if "s" in kind:
            for _ in range(40): small.add_(1.0)
    hi = torch.cuda.Stream(priority=-1)
    DELAY = 0
    NOWAIT = False
    def run(kind, pipelined, T=60, lag=2, prio=False):
        ev_side = [None, None]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        main = hi if prio else torch.cuda.current_stream()
        torch.cuda.set_stream(main)
        for t in range(T):
            p = t % 2
            if pipelined:
                if ev_side[p] is not None and not NOWAIT: main.wait_event(ev_side[p])
                b.advance(act[p], 64, lst[p], cnt[p], 3000, lag)
                ev = torch.cuda.Event(); ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    if DELAY: torch.cuda._sleep(DELAY)
                    side_work(kind)
                    e2 = torch.cuda.Event(); e2.record(side); ev_side[p] = e2
            else:
                b.advance(act[0], 64, lst[0], cnt[0], 3000, 1); side_work(kind)
        torch.cuda.synchronize(); torch.cuda.set_stream(torch.cuda.default_stream()); return (time.perf_counter() - t0) / T * 1e3
    def run_free(kind, T=60):
        """no cross-stream dependencies at all: T slices on the main stream, T side jobs on the side stream"""
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(T):
            b.advance(act[0], 64, lst[0], cnt[0], 3000, 1)
            with torch.cuda.stream(side): side_work(kind)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / T * 1e3
    for kind in ("o",):
        print(f"side work '{kind}': free-running streams {run_free(kind):.2f} ms/tick")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for cyc in (10000, 100000, 1000000):
        torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize(); print("_sleep", cyc, "->", e0.elapsed_time(e1), "ms")
    for d in (20000, 100000, 400000):
        DELAY = d
        print(f"delay {d}: 'o' pipelined {run('o', True):.2f}  'ocs' pipelined {run('ocs', True):.2f}")
    DELAY = 0
    NOWAIT = True
    print(f"main never waits for side: 'o' pipelined {run('o', True):.2f}  'ocs' pipelined {run('ocs', True):.2f}")
    NOWAIT = False
    for kind in ("o",):
        side_work(kind); torch.cuda.synchronize()
        with torch.cuda.stream(side): side_work(kind)
        torch.cuda.synchronize()
        print(f"side work '{kind}': serial {run(kind, False):.2f} ms/tick, pipelined {run(kind, True):.2f} ms/tick, pipelined + high-priority physics stream {run(kind, True, prio=True):.2f} ms/tick")


if __name__ == "__main__":
    main()
