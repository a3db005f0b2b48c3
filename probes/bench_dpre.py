"""Input gradient through the GELU (dpre = (dy . W2) * GELU'(pre)), with the fused bias-gradient column sums (atomics from every
workgroup) vs without + a separate two-stage column-sum kernel."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _lib_helpers import timeit
K = importlib.import_module("x2-vlm_amd.kernels")
dev = "cuda"
for name, M, N, Kd in [("vision dpre", 12608, 3072, 768), ("fusion dpre", 7680, 3072, 768), ("text dpre", 3840, 3072, 768), ("large dpre", 18464, 4096, 1024)]:
    A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    pre = torch.randn(M, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    cs = torch.zeros(N, device=dev)
    t_fused = timeit(lambda: K.gemm_nt(A, B, aux=pre, act=2, out=out, colsum=cs))
    t_plain = timeit(lambda: K.gemm_nt(A, B, aux=pre, act=2, out=out))
    t_col = timeit(lambda: K.colsum_bf16(out, cs))
    print("%-12s fused colsum %6.1f us | without %6.1f us + separate colsum kernel %5.1f us = %6.1f us" % (name, t_fused, t_plain, t_col, t_plain + t_col))
