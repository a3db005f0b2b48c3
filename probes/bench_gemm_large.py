"""NT GEMM tile choice (x2_tune key 3: 1 = 128x128, 2 = 192x128, 3 = 64x128, 0 = library heuristic) on the X2VLM-large shapes
(M = 32 x 577 = 18464 vision rows; 1920 / 3840 text and fusion rows)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib_helpers import timeit
from bench_gemm import nt_case, lib  # noqa
SHAPES = [("L qkv", 18464, 3072, 1024, "bias"), ("L proj", 18464, 1024, 1024, "resid"), ("L fc1", 18464, 4096, 1024, "gelu"),
          ("L fc2", 18464, 1024, 4096, "resid"), ("L dqkv", 18464, 1024, 3072, "f32"), ("L datt", 18464, 1024, 1024, "bias"),
          ("L dpre", 18464, 4096, 1024, "gelu"), ("L t.qkv", 1920, 3072, 1024, "bias"), ("L t.ffn1", 1920, 4096, 1024, "gelu"),
          ("L t.ffn2", 1920, 1024, 4096, "resid"), ("L f.ffn1", 3840, 4096, 1024, "gelu"), ("L f.ffn2", 3840, 1024, 4096, "resid"),
          ("L f.out", 3840, 1024, 1024, "resid"), ("L kv", 18464, 2048, 1024, "bias")]
for name, M, N, Kd, epi in SHAPES:
    fn = nt_case(M, N, Kd, epi)
    best = {}
    for rep in range(2):
        for g in (0, 1, 2, 3):
            lib.x2_tune(3, g)
            best[g] = min(timeit(fn, 10), best.get(g, 1e9))
    lib.x2_tune(3, 0)
    fl = 2.0 * M * N * Kd
    print("%-9s M=%5d N=%5d K=%4d  " % (name, M, N, Kd) + "  ".join("k%d %6.1fus %4.0fTF" % (g, best[g], fl / best[g] / 1e6) for g in sorted(best)), flush=True)
