"""relpos bias-table gradient (batch-sum of dS + CSR gather): time vs the number of batch slices."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
beit2 = importlib.import_module("x2-vlm_amd.beit2")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib_helpers import timeit  # noqa
call, ptr = K.call, K.ptr
for name, B, H, g in [("base", 64, 12, 14), ("large", 32, 16, 24)]:
    N = g * g + 1
    ld = K.round_up(N, 64)
    T = (2 * g - 1) ** 2 + 3
    idx = beit2.relative_position_index(g, g).cuda()
    dS = torch.randn(B, H, N, ld, device="cuda").bfloat16()
    dt = torch.zeros(T, H, device="cuda")
    off, pos = K._relpos_csr(idx, ld, T)
    for slices in (1, 2, 4, 8):
        ws = K.workspace(dS.device, slices * H * N * ld)
        t = timeit(lambda: call("x2_relpos_bias_bwd", ptr(dS), ptr(off), ptr(pos), ptr(dt), B, N, H, ld, T, ptr(ws), slices))
        print("%-6s B=%d H=%d N=%d  slices=%d  %.1f us" % (name, B, H, N, slices, t))
