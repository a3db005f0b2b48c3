"""gemm_nt256_kernel: which pair of the contraction loop's three activities costs the time?  Probe build only
(X2VLM_HIP_LIB=probes/_probe/libx2vlm_hip_probe.so).  Loop ablations x2_tune(2, bits): 32 = no operand DMA (L2 -> LDS) in the loop,
64 = no fragment reads (LDS -> registers) after step 0, 128 = no MFMAs; the main-loop time per tile comes from the same wall_clock64()
stamps as probes/nt_phase_times.py (entry / prologue done / loop done / epilogue done).  Outputs are garbage under ablation: timing only."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
assert hasattr(lib, "x2_probe_set_buffer"), "needs the probe build"
lib.x2_probe_set_buffer.argtypes, lib.x2_probe_set_buffer.restype = [ctypes.c_void_p], ctypes.c_int
dev = "cuda"
NSET = 8
SHAPES = [("vit qkv", 12608, 2304, 768), ("vit dqkv", 12608, 768, 2304), ("vit dfc1", 12608, 768, 3072)]
if os.environ.get("ABL_SHAPES"):           # "M,N,K;M,N,K": other problem sizes (occupancy / cache-residency experiments)
    SHAPES = [("custom", *[int(v) for v in t.split(",")]) for t in os.environ["ABL_SHAPES"].split(";")]
VARIANTS = [("all three", 0), ("no DMA", 32), ("no reads", 64), ("no MFMA", 128), ("MFMA only", 32 + 64), ("reads only", 32 + 128), ("DMA only", 64 + 128)]
for name, M, N, Kd in SHAPES:
    PAD = int(os.environ.get("ABL_PAD", "0"))      # extra elements per operand row (leading dimension K + PAD): L2 channel mapping experiments
    As = [torch.randn(M, Kd + PAD, device=dev).bfloat16()[:, :Kd] for _ in range(NSET)]
    W = (torch.randn(N, Kd + PAD, device=dev) / Kd ** 0.5).bfloat16()[:, :Kd]
    if os.environ.get("ABL_A_ALIAS"):          # every row of A is the SAME 2 K bytes (row stride 0): the A tile never leaves L2 - what does its first-touch traffic cost the operand DMA?
        As = [torch.randn(1, Kd, device=dev).bfloat16().expand(M, Kd) for _ in range(NSET)]
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
    PP = int(os.environ.get("PP_H", "0"))          # 4 / 5 / 6: the ping-pong kernel at 32 x PP_H rows instead of gemm_nt256_kernel
    if PP:
        lib.x2_tune(15, PP)
    else:
        lib.x2_tune(1, 3)
    nbuf = torch.zeros(4096 * 4, device=dev, dtype=torch.int64)
    res = []
    for vname, bits in VARIANTS:
        assert lib.x2_tune(2, bits) == 0, lib.x2_last_error()
        for i in range(NSET):
            K.gemm_nt(As[i], W, out=outs[i])
        torch.cuda.synchronize()
        mains, spans = [], []
        for rep in range(5):
            nbuf.zero_()
            lib.x2_probe_set_buffer(ctypes.c_void_p(nbuf.data_ptr()))
            K.gemm_nt(As[rep % NSET], W, out=outs[rep % NSET])
            torch.cuda.synchronize()
            lib.x2_probe_set_buffer(None)
            t = nbuf.view(-1, 4).cpu().double()
            t = t[t[:, 3] > 0] / 100.0
            mains.append(float((t[:, 2] - t[:, 1]).mean()))
            spans.append(float(t[:, 3].max() - t[:, 0].min()))
        res.append((vname, min(mains), min(spans)))
    lib.x2_tune(2, 0); lib.x2_tune(1, 0); lib.x2_tune(15, 0)
    steps = Kd // 64
    print("%-10s M=%d N=%d K=%d (%d contraction steps per tile; MFMA issue alone = %.2f us per step at 2.4 GHz)" % (name, M, N, Kd, steps, 1280 / 2400.0))
    for vname, m, sp in res:
        print("   %-12s main loop %6.2f us per tile = %5.2f us per step   launch %6.1f us" % (vname, m, m / steps, sp))
