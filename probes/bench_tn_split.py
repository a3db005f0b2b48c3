"""TN grouped GEMM: cost of the split-K fp32-atomic epilogue (split 1 vs 2 vs 4 at equal total work).  GPU box only."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, Mc, probs in [("vit block", 12608, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                        ("text layer", 3840, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])]:
    ps = [(torch.randn(Mc, N, device=dev).bfloat16(), torch.randn(Mc, Kd, device=dev).bfloat16(), torch.zeros(N, Kd, device=dev)) for N, Kd in probs]
    fl = sum(2.0 * Mc * N * Kd for N, Kd in probs)
    for rep in range(2):
        for split in (1, 2, 4):
            t = timeit(lambda: K.gemm_tn_grouped(ps, accumulate=True, split=split))
            print("%-10s split %d  %7.1f us  %5.0f TF" % (name, split, t, fl / t / 1e6))
# raw atomic rate: 14 M distinct-address fp32 atomics
x = torch.zeros(16 * 1024 * 1024, device=dev)
idx = torch.arange(x.numel(), device=dev)
v = torch.ones_like(x)
t = timeit(lambda: x.index_add_(0, idx, v), 5)
print("index_add_ 16M distinct fp32: %.1f us" % t)
