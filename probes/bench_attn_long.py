"""Attention backward at the BEiT-2 shape of X2VLM-large (384 px: N = 577, 16 heads, batch 32): the long one-pass kernel
(attn_bwd_onepass_long_kernel: 256-key parts x 128-query chunks, one workgroup per (sequence, head)) against the dQ + dK/dV pair
(x2_tune(14, 1)) on the same buffers, alternating, min of 3 rounds.  GPU box only.
    python probes/bench_attn_long.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def case(B, H, N, nsets=3, with_ds=True):
    d = 64; HD = H * d
    sets = []
    ld = K.round_up(N, 128)
    for _ in range(nsets):
        qkv = torch.randn(B * N, 3 * HD, device=dev).bfloat16()
        out = torch.empty(B * N, HD, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B * H * N, device=dev); delta = torch.empty_like(lse)
        dout = torch.randn_like(out); dqkv = torch.empty_like(qkv)
        dS = torch.empty(B, H, N, K.round_up(N, 64), device=dev, dtype=torch.bfloat16) if with_ds else None
        sets.append((qkv, out, lse, delta, dout, dqkv, dS))
    bias = torch.randn(H, N, K.round_up(N, 64), device=dev); biasT = torch.randn(H, N, ld, device=dev)
    kw = dict(bias=bias, biasT=biasT, bias_log2=True)
    for qkv, out, lse, *_ in sets:
        K.attn_fwd(K.view3(qkv, B, N, 0), K.view3(qkv, B, N, HD), K.view3(qkv, B, N, 2 * HD), B, B, H, N, N, d ** -0.5,
                   K.view3(out, B, N), lse, bias=bias, bias_log2=True)
    it = [0]

    def bwd():
        qkv, out, lse, delta, dout, dqkv, dS = sets[it[0] % nsets]; it[0] += 1
        K.attn_bwd(K.view3(qkv, B, N, 0), K.view3(qkv, B, N, HD), K.view3(qkv, B, N, 2 * HD), K.view3(out, B, N), K.view3(dout, B, N),
                   B, B, H, N, N, d ** -0.5, lse, delta, K.view3(dqkv, B, N, 0), K.view3(dqkv, B, N, HD), K.view3(dqkv, B, N, 2 * HD),
                   dS=dS, **kw)
    return bwd, sets


print("# attention backward, relative-position bias in log2 units; us per backward (min of 3 rounds, 20 launches each)")
for name, B, H, N, with_ds in (("large step B=32 H=16 N=577, dS stream", 32, 16, 577, True), ("large step B=32 H=16 N=577, no dS", 32, 16, 577, False),
                               ("B=16 H=16 N=577, dS stream", 16, 16, 577, True), ("B=32 H=12 N=401, dS stream", 32, 12, 401, True)):
    bwd, sets = case(B, H, N, with_ds=with_ds)
    t = {0: [], 1: []}
    for _ in range(3):
        for knob in (1, 0):
            lib.x2_tune(14, knob)
            t[knob].append(timeit(bwd))
    lib.x2_tune(14, 0)
    fl = 4.0 * B * H * N * N * 64 * 2.5
    print("%-40s two kernels %7.1f   one pass %7.1f   (%+.1f %%; %4.0f -> %4.0f TFLOP/s of the 5-product count)" % (
        name, min(t[1]), min(t[0]), 100 * (min(t[0]) / min(t[1]) - 1), fl / min(t[1]) / 1e6, fl / min(t[0]) / 1e6), flush=True)
