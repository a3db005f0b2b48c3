"""NT epilogue A/B on the step's fp32-out shapes: default read-back (a lane = 8 columns of one row, x2_tune(2, 0)) vs
row-contiguous fp32 stores (a lane = 4 columns of two rows, x2_tune(2, 64)), interleaved, plus the main-loop-only time
(x2_tune(2, 4)).  python probes/bench_epi_f4.py [large]
(The lane-swap epilogue without the LDS round trip measured with this script's predecessor: profiles/r02e_epi_swap_probe.txt.)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def case(M, N, Kd, epi):
    A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias)
    if epi == "gelu":
        kw.update(aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16), act=1)
    elif epi == "dgelu":
        kw = dict(aux=torch.randn(M, N, device=dev).bfloat16(), act=2)
    elif epi == "lscale":
        kw.update(resid=torch.randn(M, N, device=dev), gamma=bias, aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16), out_dtype=torch.float32)
    elif epi == "resid":
        kw.update(resid=torch.randn(M, N, device=dev), out_dtype=torch.float32)
    elif epi == "f32":
        kw.update(out_dtype=torch.float32)
    return lambda: K.gemm_nt(A, B, **kw)


BASE = [("fc2 fwd", 12608, 768, 3072, "lscale"), ("proj fwd", 12608, 768, 768, "lscale"), ("fus out", 7680, 768, 768, "resid"),
        ("fus ffn2", 7680, 768, 3072, "resid"), ("text out", 3840, 768, 768, "resid"), ("text ffn2", 3840, 768, 3072, "resid"),
        ("fus dffn1", 7680, 768, 3072, "resid"), ("mlm dec f32", 768, 30528, 768, "f32")]
LARGE = [("fc2 fwd", 18464, 1024, 4096, "lscale"), ("proj fwd", 18464, 1024, 1024, "lscale")]
tot = {0: 0.0, 64: 0.0, 4: 0.0}
for name, M, N, Kd, epi in (LARGE if len(sys.argv) > 1 else BASE):
    fn = case(M, N, Kd, epi)
    best = {0: 1e9, 64: 1e9, 4: 1e9}
    for rep in range(2):
        for g in (0, 64, 4):
            lib.x2_tune(2, g)
            best[g] = min(best[g], timeit(fn))
    lib.x2_tune(2, 0)
    for g in best:
        tot[g] += best[g]
    print("%-11s M=%5d N=%5d K=%4d %-6s default %6.1f us   row-contiguous %6.1f us (%+5.1f %%)   main loop only %6.1f us" %
          (name, M, N, Kd, epi, best[0], best[64], 100.0 * (best[64] / best[0] - 1.0), best[4]), flush=True)
print("sum: default %.1f  row-contiguous %.1f  main loops %.1f us" % (tot[0], tot[64], tot[4]))
