"""Ping-pong NT kernel (gemm_nt_pp_kernel, 32x32x16 MFMAs, x2_tune(15, h)) against the library's automatic kernel choice, per shape of the
base / large steps: outputs rotate over NSET buffer sets (not cache-resident), interleaved rounds, minimum of 3; every variant is checked
against the automatic kernel's result (different accumulation order: tolerance, not bit equality) and one shape against an fp32 matmul.
GPU box only.  Usage: python probes/bench_nt_pp.py [heights, default 4,5,6]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"
NSET = 8
HEIGHTS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4,5,6".split(","))]
SHAPES = [("vit qkv", 12608, 2304, 768, "bias"), ("vit dqkv", 12608, 768, 2304, "plain"), ("vit dfc1", 12608, 768, 3072, "plain"),
          ("vit fc2", 12608, 768, 3072, "lscale"), ("vit proj", 12608, 768, 768, "lscale"), ("vit dproj", 12608, 768, 768, "plain"),
          ("vit fc1", 12608, 3072, 768, "gelu"), ("vit dfc2", 12608, 3072, 768, "dgelu"), ("fus xkv", 12608, 1536, 768, "bias"),
          ("txt ffn1", 7680, 3072, 768, "gelu"), ("txt ffn2", 7680, 768, 3072, "resid"), ("txt proj", 7680, 768, 768, "resid"),
          ("txt qkv", 3840, 768, 768, "bias"), ("txt ffn2s", 3840, 768, 3072, "resid"), ("fus dffn1", 7680, 768, 3072, "plain"), ("fus dproj", 7680, 768, 768, "plain"),
          ("txt dffn1s", 3840, 768, 3072, "plain"), ("fus q", 7680, 768, 768, "bias"), ("fus dffn2", 7680, 3072, 768, "dgelu"),
          ("vitL qkv", 18464, 3072, 1024, "bias"), ("vitL dfc1", 18464, 1024, 4096, "plain"), ("vitL fc1", 18464, 4096, 1024, "gelu")]
if os.environ.get("PP_SHAPES"):
    SHAPES = [s_ for s_ in SHAPES if s_[0] in os.environ["PP_SHAPES"].split(",")]


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


print("heights", HEIGHTS)
for name, M, N, Kd, kind in SHAPES:
    f32 = kind in ("lscale", "resid")
    As = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(NSET)]
    W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    if os.environ.get("PP_ZERO"):          # zero operands: the same instruction stream at a fraction of the switching power (is the launch power-bound?)
        for a_ in As:
            a_.zero_()
        W.zero_()
    outs = [torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(NSET)]
    bias, gamma = torch.randn(N, device=dev), torch.rand(N, device=dev)
    resid = torch.randn(M, N, device=dev) if f32 else None
    aux = torch.randn(M, N, device=dev).bfloat16() if kind in ("gelu", "dgelu") else None

    BLOCKED = bool(os.environ.get("PP_BLOCKED"))      # hand the ping-pong kernel A as [K / 64][M][64] (x2_tune(4, 1)): 20 KB of contiguous bytes per tile and step
    Ab = [a_.view(M, Kd // 64, 64).permute(1, 0, 2).contiguous().view(M, Kd) for a_ in As] if BLOCKED else None
    cur = {"v": 0}

    As_n = As

    def run(i):
        As_ = Ab if (BLOCKED and cur["v"] != 0) else As_n
        if kind == "bias":
            K.gemm_nt(As_[i], W, bias=bias, out=outs[i])
        elif kind == "plain":
            K.gemm_nt(As_[i], W, out=outs[i])
        elif kind == "gelu":
            K.gemm_nt(As_[i], W, bias=bias, aux=aux, act=1, out=outs[i])
        elif kind == "dgelu":
            K.gemm_nt(As_[i], W, aux=aux, act=2, out=outs[i])
        elif kind == "resid":
            K.gemm_nt(As_[i], W, bias=bias, resid=resid, out=outs[i])
        else:
            K.gemm_nt(As_[i], W, bias=bias, gamma=gamma, resid=resid, out=outs[i])
    variants = [0] + HEIGHTS
    res = {v: 1e9 for v in variants}
    ref = None
    err = {}
    for rnd in range(3):
        for v in variants:
            lib.x2_tune(15, v)
            res[v] = min(res[v], timeit(run))
            if rnd == 0:
                run(0); torch.cuda.synchronize()
                if v == 0:
                    ref = outs[0].float().clone()
                else:
                    d = (outs[0].float() - ref).abs().max().item()
                    err[v] = d / (ref.abs().max().item() + 1e-9)
    lib.x2_tune(15, 0)
    fl = 2.0 * M * N * Kd
    line = "%-10s M=%6d N=%5d K=%5d %-6s | auto %6.1f us %5.0f TF" % (name, M, N, Kd, kind, res[0], fl / res[0] / 1e6)
    for v in HEIGHTS:
        line += " | pp%d %6.1f us %5.0f TF (%+5.1f %%) err %.1e" % (v * 32, res[v], fl / res[v] / 1e6, 100 * (res[v] / res[0] - 1), err[v])
    print(line, flush=True)

# one shape against an fp32 matmul of the same bf16 operands
M, N, Kd = 1000, 520, 768          # ragged M, N not a multiple of 256
A = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16(); b = torch.randn(N, device=dev)
want = A.float() @ W.float().t() + b
for v in HEIGHTS:
    lib.x2_tune(15, v)
    got = K.gemm_nt(A, W, bias=b, out_dtype=torch.float32)
    torch.cuda.synchronize()
    print("ragged %dx%dx%d height %d: max |err| vs fp32 matmul %.3e" % (M, N, Kd, v * 32, (got - want).abs().max().item()))
lib.x2_tune(15, 0)
