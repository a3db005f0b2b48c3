"""NT GEMM: tile-raster GROUP_M (x2_tune key 0) sweep per shape - how many 128-row panels are walked column by column
before moving on (L2 working set = GROUP_M A panels + the B tiles in flight)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _lib_helpers import timeit
from bench_gemm import NT, nt_case, lib  # noqa
LARGE = [("L qkv", 18464, 3072, 1024, "bias"), ("L proj", 18464, 1024, 1024, "resid"), ("L fc1", 18464, 4096, 1024, "gelu"),
         ("L fc2", 18464, 1024, 4096, "resid"), ("L dqkv", 18464, 1024, 3072, "f32")]
for name, M, N, Kd, epi in NT + LARGE:
    fn = nt_case(M, N, Kd, epi)
    res = []
    for rep in range(2):
        for g in (1, 2, 4, 8, 16, 32):
            lib.x2_tune(0, g)
            res.append((g, timeit(fn, 10)))
    lib.x2_tune(0, 0)
    best = {}
    for g, t in res:
        best[g] = min(t, best.get(g, 1e9))
    print("%-11s M=%5d N=%5d K=%4d  " % (name, M, N, Kd) + "  ".join("g%-2d %6.1f" % (g, best[g]) for g in sorted(best)), flush=True)
