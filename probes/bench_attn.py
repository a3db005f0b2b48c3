"""Attention micro-benchmark / ablation on the shapes of the X2VLM-base step (B=64)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
dev = "cuda"

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

def case(B, H, Lq, Lk, bias, mask, drop=None):
    d = 64
    qkv = torch.randn(B * Lq, 3 * H * d, device=dev).bfloat16()
    kv = qkv if Lq == Lk else torch.randn(B * Lk, 3 * H * d, device=dev).bfloat16()
    out = torch.empty(B * Lq, H * d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H * Lq, device=dev); delta = torch.empty_like(lse)
    dout = torch.randn_like(out); dqkv = torch.empty_like(qkv); dkv = dqkv if Lq == Lk else torch.empty_like(kv)
    kw = {}
    if bias:
        kw["bias"] = torch.randn(H, Lq, K.round_up(Lk, 64), device=dev); kw["biasT"] = torch.randn(H, Lk, K.round_up(Lq, 64), device=dev)
        kw["bias_log2"] = os.environ.get("X2_BENCH_BIAS_LOG2", "1") == "1"      # as the engine hands the bias over (one-fma score path)
    if mask:
        kw["mask"] = torch.zeros(B, K.round_up(Lk, 64), device=dev)
    HD = H * d
    q3, k3, v3 = K.view3(qkv, B, Lq, 0), K.view3(kv, B, Lk, HD), K.view3(kv, B, Lk, 2 * HD)
    dS = torch.empty(B, H, Lq, K.round_up(Lk, 64), device=dev, dtype=torch.bfloat16) if bias else None
    def fwd(dbg=0):
        a = K._attn_args(q3, k3, v3, B, B, H, Lq, Lk, d ** -0.5, **{k_: v_ for k_, v_ in kw.items() if k_ != "biasT"})
        a.Out, a.o_bs, a.o_rs = K.view3(out, B, Lq); a.LSE = lse.data_ptr(); a.dbg |= dbg
        K.call("x2_attn_fwd", K.C.byref(a))
    def bwd():
        K.attn_bwd(q3, k3, v3, K.view3(out, B, Lq), K.view3(dout, B, Lq), B, B, H, Lq, Lk, d ** -0.5, lse, delta,
                   K.view3(dqkv, B, Lq, 0), K.view3(dkv, B, Lk, HD), K.view3(dkv, B, Lk, 2 * HD), dS=dS,
                   **({"drop": drop} if drop else {}), **kw)
    return fwd, bwd

for name, B, H, Lq, Lk, bias, mask in [("vision large", 32, 16, 577, 577, True, False), ("vision", 64, 12, 197, 197, True, False), ("text self", 128, 12, 30, 30, False, True),
                                       ("fusion self", 256, 12, 30, 30, False, True)]:
    fwd, bwd = case(B, H, Lq, Lk, bias, mask)
    fl = 4.0 * B * H * Lq * Lk * 64
    t = timeit(fwd); tb = timeit(bwd)
    if name == "vision large":
        a_ = K._attn_args  # per-kernel split of the backward
    print("%-12s fwd %6.1fus %5.0fTF   bwd(dq+dkv) %6.1fus %5.0fTF" % (name, t, fl / t / 1e6, tb, 2.5 * fl / tb / 1e6))
    if name == "vision":
        print("   fwd ablation: " + "  ".join("d%d %.1fus" % (g, timeit(lambda g=g: fwd(g))) for g in (0, 1, 2, 3, 4, 8, 12, 15)))

# backward with probability dropout 0.1 on (as in the training step)
DROP = K.dropout_spec(0.1, 1234, 7)
for name, B, H, N in (("vision base N=197", 64, 12, 197), ("vision large N=577", 32, 16, 577)):
    _, bwd = case(B, H, N, N, True, False, drop=DROP)
    print("%s bwd with dropout 0.1: %6.1f us" % (name, min(timeit(bwd) for _ in range(3))))

# cross-attention of the fusion stack: 256 text rows on 64 images (the 4-pass batch: positives, MLM, 2 x hard negatives)
def cross_case(S=256, Bi=64, H=12, L=30, T=197, drop=None):
    d = 64
    g = torch.Generator().manual_seed(0)
    ar = torch.arange(Bi)
    kv = torch.cat([ar, torch.randint(0, Bi, (Bi,), generator=g), ar, ar]).to(torch.int32)
    order = torch.argsort(kv, stable=True).to(torch.int32)
    off = torch.zeros(Bi + 1, dtype=torch.int32); off[1:] = torch.cumsum(torch.bincount(kv, minlength=Bi), 0)
    q = torch.randn(S * L, H * d, device=dev).bfloat16(); kvt = torch.randn(Bi * T, 2 * H * d, device=dev).bfloat16()
    out = torch.empty_like(q); lse = torch.empty(S * H * L, device=dev); delta = torch.empty_like(lse)
    dout = torch.randn_like(q); dq = torch.empty_like(q); dkv = torch.empty_like(kvt)
    mask = torch.zeros(S, K.round_up(T, 64), device=dev)
    kw = dict(mask=mask, kv_idx=kv.to(dev), seq_off=off.to(dev), seq_ids=order.to(dev))
    HD = H * d
    q3, k3, v3 = K.view3(q, S, L), K.view3(kvt, Bi, T, 0), K.view3(kvt, Bi, T, HD)
    fwd = lambda: K.attn_fwd(q3, k3, v3, S, Bi, H, L, T, d ** -0.5, K.view3(out, S, L), lse, **kw)
    bwd = lambda: K.attn_bwd(q3, k3, v3, K.view3(out, S, L), K.view3(dout, S, L), S, Bi, H, L, T, d ** -0.5, lse, delta,
                             K.view3(dq, S, L), K.view3(dkv, Bi, T, 0), K.view3(dkv, Bi, T, HD), **({"drop": drop} if drop else {}), **kw)
    return fwd, bwd

fwd, bwd = cross_case()
print("cross (256 rows on 64 images)  fwd %6.1fus   bwd(dq+dkv) %6.1fus" % (timeit(fwd), timeit(bwd)))
_, bwd = cross_case(drop=DROP)
print("cross bwd with dropout 0.1: %6.1f us" % min(timeit(bwd) for _ in range(3)))
