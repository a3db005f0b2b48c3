"""NT GEMM (128-column kernels): start-phase stagger of the first 512 workgroups (x2_tune(4, v), csrc/gemm.hip) on the shapes of
the X2VLM-base step.  v = 0 off; 1, 2 = second-slot workgroups 3.4 / 6.8 us late; 256 | u = eight start phases u x 0.43 us apart.
Interleaved rounds in one process, minimum of three.    python probes/bench_nt_stagger.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_nt256 import case, shapes, timeit, lib       # noqa: E402


def main():
    configs = [("off", 0), ("slot1", 1), ("ph x1", 257), ("ph x2", 258), ("ph x4", 260), ("ph x8", 264)]
    tot = {c[0]: 0.0 for c in configs}
    lib.x2_tune(1, 1)                                   # 128-column kernels only
    print("%-11s %6s %5s %5s %-10s | " % ("launch", "M", "N", "K", "epilogue") + " ".join("%7s" % c[0] for c in configs))
    for name, M, N, Kd, epi, n in shapes("base"):
        fn = case(M, N, Kd, epi)
        res = {c[0]: [] for c in configs}
        for _ in range(3):
            for cname, v in configs:
                lib.x2_tune(4, v)
                res[cname].append(timeit(fn))
        lib.x2_tune(4, 0)
        for c in res:
            tot[c] += n * min(res[c])
        print("%-11s %6d %5d %5d %-10s | " % (name, M, N, Kd, epi) + " ".join("%7.1f" % min(res[c[0]]) for c in configs), flush=True)
    print("per step (launch counts applied), ms: " + "  ".join("%s %.2f" % (c, tot[c] / 1e3) for c in tot))


if __name__ == "__main__":
    main()
