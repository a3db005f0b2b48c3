"""Which multi-stream capture patterns does HIP (ROCm 7) survive?  Each case runs in its own process (a bad one segfaults
in hipStreamEndCapture).  usage: python probes/graph_capture_probe.py [case]"""
import subprocess
import sys

import torch

CASES = ["single", "fork_join_del", "fork_join_keep", "fork_join_x50_del", "fork_join_x50_keep", "unwaited_done_event",
         "wait_stream_x50", "two_sides", "nested_fork", "fork_join_x600_del", "rejoin_twice", "sibling_cross", "nested_join_origin", "prefork_then_nested",
         "prefork_then_nested_x30", "nested_b_joins_both"]


def run(case):
    s, a, b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 16, device="cuda")
    y = torch.zeros(1 << 16, device="cuda")
    keep = []
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()

    def fork(side, keepit):
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        if keepit:
            keep.append(ev)

    def join(side, keepit):
        ev = torch.cuda.Event()
        ev.record(side)
        torch.cuda.current_stream().wait_event(ev)
        if keepit:
            keep.append(ev)

    with torch.cuda.graph(g, stream=s):
        x.add_(1)
        if case == "single":
            pass
        elif case in ("fork_join_del", "fork_join_keep"):
            fork(a, case.endswith("keep"))
            with torch.cuda.stream(a):
                y.add_(1)
            join(a, case.endswith("keep"))
        elif case.startswith("fork_join_x"):
            n = int(case.split("_x")[1].split("_")[0])
            for _ in range(n):
                fork(a, case.endswith("keep"))
                with torch.cuda.stream(a):
                    y.add_(1)
                x.add_(1)
                join(a, case.endswith("keep"))
        elif case == "unwaited_done_event":
            for _ in range(20):
                fork(a, True)
                with torch.cuda.stream(a):
                    y.add_(1)
                    d = torch.cuda.Event()
                    d.record()
                    keep.append(d)
                x.add_(1)
            join(a, True)
        elif case == "wait_stream_x50":
            for _ in range(50):
                a.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(a):
                    y.add_(1)
                x.add_(1)
                torch.cuda.current_stream().wait_stream(a)
        elif case == "two_sides":
            for _ in range(20):
                fork(a, True); fork(b, True)
                with torch.cuda.stream(a):
                    y.add_(1)
                with torch.cuda.stream(b):
                    y2 = x * 2
                x.add_(1)
                join(a, True); join(b, True)
        elif case == "nested_fork":
            fork(a, True)
            with torch.cuda.stream(a):
                y.add_(1)
                fork(b, True)
                with torch.cuda.stream(b):
                    z = y * 2
                join(b, True)
            join(a, True)
        elif case == "sibling_cross":
            fork(a, True); fork(b, True)
            with torch.cuda.stream(a):
                y.add_(1)
                e1 = torch.cuda.Event(); e1.record(); keep.append(e1)
            b.wait_event(e1)
            with torch.cuda.stream(b):
                z = y * 2
                e2 = torch.cuda.Event(); e2.record(); keep.append(e2)
            a.wait_event(e2)
            with torch.cuda.stream(a):
                y.add_(1)
            join(a, True); join(b, True)
        elif case == "nested_join_origin":
            fork(a, True)
            with torch.cuda.stream(a):
                y.add_(1)
                fork(b, True)
                with torch.cuda.stream(b):
                    z = y * 2
            join(b, True)
            join(a, True)
        elif case == "nested_b_joins_both":
            fork(a, True)
            with torch.cuda.stream(a):
                y.add_(1)
                fork(b, True)
                with torch.cuda.stream(b):
                    z = y * 2
                join(b, True)
                y.add_(1)
            join(b, True)
            join(a, True)
        elif case.startswith("prefork_then_nested"):
            fork(a, True); fork(b, True)
            for _ in range(30 if case.endswith("x30") else 1):
                with torch.cuda.stream(a):
                    y.add_(1)
                    fork(b, True)
                    with torch.cuda.stream(b):
                        z = y * 2
                    join(b, True)
                    y.add_(1)
                x.add_(1)
            join(a, True); join(b, True)
        elif case == "rejoin_twice":
            fork(a, True)
            with torch.cuda.stream(a):
                y.add_(1)
            join(a, True)
            join(a, True)
            x.add_(1)
            fork(a, True)
            with torch.cuda.stream(a):
                y.add_(1)
            join(a, True)
        elif case == "autograd_two_streams":
            w1 = torch.randn(64, 64, device="cuda", requires_grad=True)
            w2 = torch.randn(64, 64, device="cuda", requires_grad=True)
            inp = torch.randn(8, 64, device="cuda")
            a.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(a):
                h2 = inp @ w2
            h1 = inp @ w1
            torch.cuda.current_stream().wait_stream(a)
            (h1 * h2).sum().backward()
    g.replay()
    torch.cuda.synchronize()
    print(case, "ok", float(x[0]), float(y[0]))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            print("%-24s rc=%d %s %s" % (c, r.returncode, tail, "" if r.returncode == 0 else r.stderr.strip().splitlines()[-1][:200]), flush=True)
