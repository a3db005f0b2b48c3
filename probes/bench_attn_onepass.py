"""Attention backward at the BEiT-2 shapes of the step: the one-pass kernel (attn_bwd_onepass_kernel, 64 < N <= 208) against the
dQ + dK/dV pair (x2_tune(14, 1)) on the same buffers, alternating, min of `rounds`.  GPU box only.
    python probes/bench_attn_onepass.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def case(B, H, N, nsets=4):
    d = 64; HD = H * d
    sets = []
    for _ in range(nsets):          # rotating buffer sets: what a launch reads was not left in L2 by the previous one
        qkv = torch.randn(B * N, 3 * HD, device=dev).bfloat16()
        out = torch.empty(B * N, HD, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B * H * N, device=dev); delta = torch.empty_like(lse)
        dout = torch.randn_like(out); dqkv = torch.empty_like(qkv)
        dS = torch.empty(B, H, N, K.round_up(N, 64), device=dev, dtype=torch.bfloat16)
        sets.append((qkv, out, lse, delta, dout, dqkv, dS))
    bias = torch.randn(H, N, K.round_up(N, 64), device=dev); biasT = torch.randn(H, N, K.round_up(N, 64), device=dev)
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
    return bwd


print("# attention backward, relative-position bias in log2 units, dS stream written; us per backward (min of 3 rounds, 30 launches each)")
for name, B, H, N in (("base step  B=64 H=12 N=197", 64, 12, 197), ("half batch B=32 H=12 N=197", 32, 12, 197), ("B=64 H=12 N=208", 64, 12, 208),
                      ("B=64 H=12 N=101", 64, 12, 101)):
    bwd = case(B, H, N)
    t = {0: [], 1: []}
    for _ in range(3):
        for knob in (1, 0):
            lib.x2_tune(14, knob)
            t[knob].append(timeit(bwd))
    lib.x2_tune(14, 0)
    fl = 4.0 * B * H * N * N * 64 * 2.5
    print("%-28s two kernels %6.1f   one pass %6.1f   (%+.1f %%; %4.0f -> %4.0f TFLOP/s of the 5-product count)" % (
        name, min(t[1]), min(t[0]), 100 * (min(t[0]) / min(t[1]) - 1), fl / min(t[1]) / 1e6, fl / min(t[0]) / 1e6))


# cross-attention of the fusion stack: 256 text rows (4 passes x 64) on 64 images, 30 tokens each, shared K/V through the CSR
def cross_case(S=256, Bi=64, H=12, L=30, T=197, drop=None, nsets=4):
    d = 64; HD = H * d
    g = torch.Generator().manual_seed(0)
    ar = torch.arange(Bi)
    kv = torch.cat([ar, torch.randint(0, Bi, (Bi,), generator=g), ar, ar]).to(torch.int32)
    order = torch.argsort(kv, stable=True).to(torch.int32)
    off = torch.zeros(Bi + 1, dtype=torch.int32); off[1:] = torch.cumsum(torch.bincount(kv, minlength=Bi), 0)
    mask = torch.zeros(S, K.round_up(T, 64), device=dev)
    kw = dict(mask=mask, kv_idx=kv.to(dev), seq_off=off.to(dev), seq_ids=order.to(dev))
    if drop:
        kw["drop"] = drop
    sets = []
    for _ in range(nsets):
        q = torch.randn(S * L, HD, device=dev).bfloat16(); kvt = torch.randn(Bi * T, 2 * HD, device=dev).bfloat16()
        out = torch.empty_like(q); lse = torch.empty(S * H * L, device=dev); delta = torch.empty_like(lse)
        dout = torch.randn_like(q); dq = torch.empty_like(q); dkv = torch.empty_like(kvt)
        K.attn_fwd(K.view3(q, S, L), K.view3(kvt, Bi, T, 0), K.view3(kvt, Bi, T, HD), S, Bi, H, L, T, d ** -0.5, K.view3(out, S, L), lse,
                   **{k_: v_ for k_, v_ in kw.items() if k_ not in ("seq_off", "seq_ids")})
        sets.append((q, kvt, out, lse, delta, dout, dq, dkv))
    it = [0]

    def bwd():
        q, kvt, out, lse, delta, dout, dq, dkv = sets[it[0] % nsets]; it[0] += 1
        K.attn_bwd(K.view3(q, S, L), K.view3(kvt, Bi, T, 0), K.view3(kvt, Bi, T, HD), K.view3(out, S, L), K.view3(dout, S, L), S, Bi, H, L, T,
                   d ** -0.5, lse, delta, K.view3(dq, S, L), K.view3(dkv, Bi, T, 0), K.view3(dkv, Bi, T, HD), **kw)
    return bwd, int(torch.bincount(kv.long(), minlength=Bi).max())


print("# cross-attention backward (rows sharing K/V): dQ grouped + dK/dV streamed against the one-pass grouped kernel")
for name, drop in (("fusion cross, no dropout", None), ("fusion cross, dropout 0.1", K.dropout_spec(0.1, 1234, 7))):
    bwd, most = cross_case(drop=drop)
    t = {0: [], 1: []}
    for _ in range(3):
        for knob in (1, 0):
            lib.x2_tune(14, knob)
            t[knob].append(timeit(bwd))
    lib.x2_tune(14, 0)
    print("%-28s two kernels %6.1f   one pass %6.1f   (%+.1f %%; at most %d sequences on one image)" % (
        name, min(t[1]), min(t[0]), 100 * (min(t[0]) / min(t[1]) - 1), most))
