"""gemm_nt256 at 160 rows: three-stage operand ring with pipelined fragment reads (x2_tune(10, 0)) and without (10, 2) against the two-stage kernel (10, 1), per shape of the base / large
steps that this kernel serves, outputs rotating over 12 buffer sets (not cache-resident), interleaved rounds, minimum of 3.  GPU box only."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"
NSET = 12
SHAPES = [("vit qkv", 12608, 2304, 768, "bias"), ("vit dqkv", 12608, 768, 2304, "plain"), ("vit dfc1", 12608, 768, 3072, "plain"),
          ("vit fc2", 12608, 768, 3072, "lscale"), ("vit dproj", 12608, 768, 768, "plain"), ("fus xkv", 12608, 1536, 768, "bias"),
          ("vitL qkv", 18464, 3072, 1024, "bias"), ("vitL dfc1", 18464, 1024, 4096, "plain"), ("vitL dqkv", 18464, 1024, 3072, "plain")]


def timeit(fn, iters=24):
    for i in range(NSET):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % NSET)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


tot = {0: 0.0, 1: 0.0}
for name, M, N, Kd, kind in SHAPES:
    f32 = kind == "lscale"
    As = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(NSET)]
    W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    outs = [torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(NSET)]
    bias, gamma = torch.randn(N, device=dev), torch.rand(N, device=dev)
    resid = torch.randn(M, N, device=dev) if f32 else None

    def run(i):
        if kind == "bias":
            K.gemm_nt(As[i], W, bias=bias, out=outs[i])
        elif kind == "plain":
            K.gemm_nt(As[i], W, out=outs[i])
        else:
            K.gemm_nt(As[i], W, bias=bias, gamma=gamma, resid=resid, out=outs[i])
    lib.x2_tune(1, 3); lib.x2_tune(3, 5)
    res = {0: 1e9, 1: 1e9, 2: 1e9}
    ref = None
    for rnd in range(3):
        for k in (1, 2, 0):
            lib.x2_tune(10, k)
            res[k] = min(res[k], timeit(run))
            run(0); torch.cuda.synchronize()
            if ref is None:
                ref = outs[0].clone()
            else:
                assert torch.equal(outs[0], ref), (name, k)          # same arithmetic, same order: bit-identical
    lib.x2_tune(10, 0); lib.x2_tune(1, 0); lib.x2_tune(3, 0)
    fl = 2.0 * M * N * Kd
    print("%-10s M=%6d N=%5d K=%5d %-6s | two-stage %6.1f us %5.0f TF | three-stage %6.1f us %5.0f TF (%+5.1f %%) | + pipelined reads %6.1f us %5.0f TF (%+5.1f %%)" % (
        name, M, N, Kd, kind, res[1], fl / res[1] / 1e6, res[2], fl / res[2] / 1e6, 100 * (res[2] / res[1] - 1), res[0], fl / res[0] / 1e6,
        100 * (res[0] / res[1] - 1)))
