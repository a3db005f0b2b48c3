"""gemm_nt256_kernel: where a tile's time goes.  Probe build only (bash probes/build_probe.sh;
X2VLM_HIP_LIB=probes/_probe/libx2vlm_hip_probe.so python probes/nt_phase_times.py): wave 0 of every workgroup stamps wall_clock64()
(100 MHz) at entry, after the prologue (first operands landed), after the last contraction step and after its epilogue stores have
left.  Per launch shape: workgroups, rounds on 256 CUs, and the mean / p90 of prologue, main loop and epilogue in us, the span from the
first entry to the last exit (= the launch), and how much of the launch the average CU spends in each phase.  Buffers rotate
over 12 sets (not cache-resident: the in-step condition)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
assert hasattr(lib, "x2_probe_set_buffer"), "needs the probe build: X2VLM_HIP_LIB=probes/_probe/libx2vlm_hip_probe.so"
lib.x2_probe_set_buffer.argtypes, lib.x2_probe_set_buffer.restype = [ctypes.c_void_p], ctypes.c_int
dev = "cuda"
NSET = 12
# (name, M, N, K, kind): the launches of the base / large steps that run on the 256-column kernel, plus fc1 forced onto it
SHAPES = [("vit qkv", 12608, 2304, 768, "bias"), ("vit dqkv", 12608, 768, 2304, "plain"), ("vit dfc1", 12608, 768, 3072, "plain"),
          ("vit fc2", 12608, 768, 3072, "lscale"), ("vit dproj", 12608, 768, 768, "plain"), ("fus xkv", 12608, 1536, 768, "bias"),
          ("vit fc1 (forced)", 12608, 3072, 768, "gelu"), ("vitL qkv", 18464, 3072, 1024, "bias"), ("vitL dfc1", 18464, 1024, 4096, "plain")]


def launch(kind, A, W, out, extra):
    if kind == "bias":
        K.gemm_nt(A, W, bias=extra["bias"], out=out)
    elif kind == "plain":
        K.gemm_nt(A, W, out=out)
    elif kind == "lscale":
        K.gemm_nt(A, W, bias=extra["bias"], gamma=extra["gamma"], resid=extra["resid"], out=out)
    elif kind == "gelu":
        K.gemm_nt(A, W, bias=extra["bias"], aux=extra["aux"], act=1, out=out)


for name, M, N, Kd, kind in SHAPES:
    f32 = kind == "lscale"
    As = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(NSET)]
    W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    outs = [torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(NSET)]
    extra = dict(bias=torch.randn(N, device=dev), gamma=torch.rand(N, device=dev), resid=torch.randn(M, N, device=dev) if f32 else None,
                 aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16) if kind == "gelu" else None)
    lib.x2_tune(1, 3)                       # always the 256-column kernel; tile height from its plan
    nbuf = torch.zeros(4096 * 4, device=dev, dtype=torch.int64)
    for i in range(NSET):
        launch(kind, As[i], W, outs[i], extra)
    torch.cuda.synchronize()
    rows = []
    for rep in range(6):
        nbuf.zero_()
        lib.x2_probe_set_buffer(ctypes.c_void_p(nbuf.data_ptr()))
        launch(kind, As[rep % NSET], W, outs[rep % NSET], extra)
        torch.cuda.synchronize()
        lib.x2_probe_set_buffer(None)
        t = nbuf.view(-1, 4).cpu().double()
        t = t[t[:, 3] > 0] / 100.0                                       # us
        t0 = t[:, 0].min()
        rows.append((t.shape[0], (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2]), float(t[:, 3].max() - t0),
                     float((t[:, 0] - t0).quantile(0.9))))
    lib.x2_tune(1, 0)
    n = rows[0][0]
    pro = torch.cat([r[1] for r in rows[1:]]); main = torch.cat([r[2] for r in rows[1:]]); epi = torch.cat([r[3] for r in rows[1:]])
    span = sum(r[4] for r in rows[1:]) / len(rows[1:])
    busy = float((pro.sum() + main.sum() + epi.sum()) / len(rows[1:])) / 256.0
    print("%-18s M=%6d N=%5d K=%5d %-6s | %4d wgs (%.2f rounds) | launch %6.1f us | prologue %5.2f (p90 %5.2f)  main %6.2f (p90 %6.2f)  epilogue %5.2f (p90 %5.2f) us"
          " | per CU: %.1f us busy = %.0f %% of the launch: prologue %.0f %%, main %.0f %%, epilogue %.0f %%; entry of the 90th percentile wg %+.1f us"
          % (name, M, N, Kd, kind, n, n / 256.0, span, pro.mean(), pro.quantile(0.9), main.mean(), main.quantile(0.9), epi.mean(), epi.quantile(0.9),
             busy, 100 * busy / span, 100 * float(pro.sum()) / float(pro.sum() + main.sum() + epi.sum()),
             100 * float(main.sum()) / float(pro.sum() + main.sum() + epi.sum()), 100 * float(epi.sum()) / float(pro.sum() + main.sum() + epi.sum()),
             rows[-1][5]))
