"""LayerNorm backward micro-benchmark (vision shape).  GPU box only."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
dev = "cuda"
M, D = 12608, 768
x = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev); w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
h, _, mean, rstd = K.layernorm_fwd(x, w, b, 1e-6)
dw, db, dcol = (torch.zeros(D, device=dev) for _ in range(3))
dres = torch.randn(M, D, device=dev)
def run():
    K.layernorm_bwd(dy, x, mean, rstd, w, dw, db, dres=dres, want_bf16=True)
for _ in range(3): run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
t = a.elapsed_time(e) / 50 * 1e3
mb = M * D * (4 + 4 + 4 + 4 + 2) / 1e6   # dy, x, dres read; dx fp32 + bf16 written
print("layernorm_bwd (+ stage 2)  %.1f us  %.2f TB/s of %.0f MB" % (t, mb / t, mb))
