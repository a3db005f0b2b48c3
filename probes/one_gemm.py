"""One GEMM shape, many launches: target for rocprofv3 --pmc.  usage: python probes/one_gemm.py M N K [tn]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
M, N, Kd = (int(x) for x in sys.argv[1:4])
dev = "cuda"
if len(sys.argv) > 4:
    dY = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, Kd, device=dev).bfloat16(); dW = torch.empty(N, Kd, device=dev)
    fn = lambda: K.gemm_tn_grouped([(dY, X, dW)])
else:
    A = torch.randn(M, Kd, device=dev).bfloat16(); B = torch.randn(N, Kd, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    fn = lambda: K.gemm_nt(A, B, bias=bias, out=out)
for _ in range(10):
    fn()
torch.cuda.synchronize()
