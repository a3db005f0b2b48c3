"""GPU parity of every HIP kernel (through the C ABI) against the oracle's primitives on the same
seeded inputs.  Inputs that the kernels take in bf16 are rounded to bf16 first, so the comparison
isolates kernel arithmetic (fp32 accumulate) from input quantisation.

Tolerances (relative to each tensor's max-abs): fp32-out kernels 1e-5; bf16-out kernels 6e-3 (one
bf16 rounding of the output is 2^-9 = 3.9e-3 of the element, plus fp32 summation-order noise)."""
import importlib
import os
import math

import numpy as np
import pytest
import torch

from oracle import x2vlm_oracle as O

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.fixture(scope="module")
def K():
    return importlib.import_module("x2-vlm_amd.kernels")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf(t):
    return t.to(torch.bfloat16)


def relerr(got, ref):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


@pytest.fixture(params=[(1, 1, 0), (1, 2, 0), (1, 3, 0), (1, 4, 0), (3, 8, 0), (3, 7, 0), (3, 6, 0), (3, 5, 0), (0, 0, 3), (0, 0, 4), (0, 0, 5), (0, 0, 6)],
                ids=["tile128x128", "tile192x128", "tile64x128", "tile160x128", "nt256x256", "nt224x256", "nt192x256", "nt160x256",
                     "pingpong96x256", "pingpong128x256", "pingpong160x256", "pingpong192x256"])
def nt_tile(request):
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    lib.x2_tune(1, request.param[0])
    lib.x2_tune(3, request.param[1])
    lib.x2_tune(15, request.param[2])           # the ping-pong kernel (32x32x16 MFMAs) at 32 x value rows; runs the same tests as every other tile
    yield request.param
    lib.x2_tune(1, 0)
    lib.x2_tune(3, 0)
    lib.x2_tune(15, 0)


@pytest.mark.parametrize("M,N,K_", [(256, 256, 128), (300, 200, 192), (788, 2304, 768), (12608, 768, 768), (100, 30528, 64)])
def test_gemm_nt_plain_and_epilogues(K, M, N, K_, nt_tile):
    A, B = bf(rnd(M, K_, seed=1)), bf(rnd(N, K_, seed=2, scale=K_ ** -0.5))
    ref = A.float() @ B.float().t()
    out = K.gemm_nt(A.to(dev), B.to(dev), out_dtype=torch.float32)
    assert relerr(out, ref) < 1e-5
    out = K.gemm_nt(A.to(dev), B.to(dev))
    assert relerr(out, ref) < 6e-3
    if N > 4096:
        return
    bias, gamma, resid = rnd(N, seed=3), rnd(N, seed=4), rnd(M, N, seed=5)
    # bias + GELU, pre-activation saved
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = K.gemm_nt(A.to(dev), B.to(dev), bias=bias.to(dev), aux=aux, act=1, out_dtype=torch.float32)
    assert relerr(out, O.gelu(ref + bias)) < 1e-5
    assert relerr(aux, ref + bias) < 6e-3
    # GELU' epilogue reads the saved pre-activation
    pre = bf(rnd(M, N, seed=6))
    cs = torch.zeros(N, device=dev)
    out = K.gemm_nt(A.to(dev), B.to(dev), aux=pre.to(dev), act=2, out_dtype=torch.float32, colsum=cs)
    x = pre.float().requires_grad_(True)
    O.gelu(x).sum().backward()
    assert relerr(out, ref * x.grad) < 1e-5
    assert relerr(cs, (ref * x.grad).sum(0)) < 2e-5         # fused bias-gradient column sums
    # layer-scale + residual (aux receives acc + bias)
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = K.gemm_nt(A.to(dev), B.to(dev), bias=bias.to(dev), gamma=gamma.to(dev), resid=resid.to(dev), aux=aux,
                    out_dtype=torch.float32)
    assert relerr(out, resid + gamma * (ref + bias)) < 1e-5
    assert relerr(aux, ref + bias) < 6e-3
    # the step's own layer-scale form (compiled feature set 6): nothing saved, the backward needs no activation
    out = K.gemm_nt(A.to(dev), B.to(dev), bias=bias.to(dev), gamma=gamma.to(dev), resid=resid.to(dev), out_dtype=torch.float32)
    assert relerr(out, resid + gamma * (ref + bias)) < 1e-5
    # the compiled feature sets of the step that the calls above do not reach (they run the generic epilogue):
    # bias + GELU -> bf16, GELU' -> bf16, bias + residual -> fp32
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = K.gemm_nt(A.to(dev), B.to(dev), bias=bias.to(dev), aux=aux, act=1)
    assert relerr(out, O.gelu(ref + bias)) < 6e-3 and relerr(aux, ref + bias) < 6e-3
    out = K.gemm_nt(A.to(dev), B.to(dev), aux=pre.to(dev), act=2)
    assert relerr(out, ref * x.grad) < 6e-3
    out = K.gemm_nt(A.to(dev), B.to(dev), bias=bias.to(dev), resid=resid.to(dev), out_dtype=torch.float32)
    assert relerr(out, resid + ref + bias) < 1e-5


@pytest.mark.parametrize("tile", [0, 1, 2, 3], ids=["auto", "tile128", "tile192", "tile64"])
@pytest.mark.parametrize("M,N,K_", [(300, 200, 192), (788, 3072, 128), (3840, 3072, 64), (130, 64, 64)])
def test_gemm_nt_dgelu_with_column_sums(K, M, N, K_, tile):
    """x2_gemm_nt_dgelu_colparts (epilogue variant 10): the GELU' input gradient equals the plain variant-3 launch bit for bit
    (same epilogue arithmetic) and the reduced per-wave partial rows equal its column sums; immediate and deferred reduction."""
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    A, B = bf(rnd(M, K_, seed=51)).to(dev), bf(rnd(N, K_, seed=52, scale=K_ ** -0.5)).to(dev)
    pre = bf(rnd(M, N, seed=53)).to(dev)
    lib.x2_tune(1, 1); lib.x2_tune(3, tile)
    try:
        ref = K.gemm_nt(A, B, aux=pre, act=2)
        cs = torch.full((N,), 3.0, device=dev)
        got = K.gemm_nt_dgelu_colsum(A, B, pre, cs)
        # same operations in the same order; only the compiler's fma contraction may differ between the two instantiations,
        # which flips a bf16 rounding here and there
        same = lambda g_, r_: float((g_.float() - r_.float()).abs().max()) <= 8e-3 * max(1.0, float(r_.float().abs().max()))
        assert same(got, ref) and float((got != ref).float().mean()) < 1e-3
        want = ref.float().sum(0).cpu().double() + 3.0          # accumulated onto what was there; fp32 values before the bf16 rounding
        assert float((cs.cpu().double() - want).abs().max()) <= 4e-3 * float(ref.float().abs().sum(0).max()) + 1e-6
        exact = (A.float() @ B.float().t() * torch.autograd.functional.jacobian(lambda t: O.gelu(t).sum(), pre.float()).to(dev)).sum(0)
        assert relerr(cs - 3.0, exact.cpu()) < 2e-4
        cs2 = torch.zeros(N, device=dev)
        K.DEFERRED = []
        try:
            got2 = K.gemm_nt_dgelu_colsum(A, B, pre, cs2)
            items, K.DEFERRED = K.DEFERRED, None
            K.reduce_partials_multi(items)
        finally:
            K.DEFERRED = None
        assert torch.equal(got2, got) and relerr(cs2, (cs - 3.0).cpu()) < 1e-6
    finally:
        lib.x2_tune(1, 0); lib.x2_tune(3, 0)


@pytest.mark.parametrize("tmw", [8, 7, 6, 5], ids=["nt256x256", "nt224x256", "nt192x256", "nt160x256"])
@pytest.mark.parametrize("M,N,K_", [(300, 200, 192), (788, 2304, 128), (1000, 768, 64), (2500, 768, 1024)])
def test_gemm_nt_256_column_kernel_matches_the_default_one(K, M, N, K_, tmw):
    """x2_tune(1, 3): the 8-wave 256-column kernel (gemm_nt256_kernel, all four tile heights) shares the epilogue code with the
    128-column kernels, so every compiled feature set - incl. dropout and DropPath row factors, whose masks are functions
    of the element index - must reproduce their outputs up to the order of the fp32 accumulation."""
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    A, B = bf(rnd(M, K_, seed=41)).to(dev), bf(rnd(N, K_, seed=42, scale=K_ ** -0.5)).to(dev)
    bias, gamma, resid = rnd(N, seed=43).to(dev), rnd(N, seed=44).to(dev), rnd(M, N, seed=45).to(dev)
    pre = bf(rnd(M, N, seed=46)).to(dev)
    rowscale = (torch.rand(M, generator=torch.Generator().manual_seed(47)) > 0.2).float().to(dev) * 1.25
    drop = K.dropout_spec(0.1, 4321, 3)

    def run(kind):
        aux = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        if kind == "bias_bf16":
            return K.gemm_nt(A, B, bias=bias), aux
        if kind == "bias_f32":
            return K.gemm_nt(A, B, bias=bias, out_dtype=torch.float32), aux
        if kind == "gelu":
            return K.gemm_nt(A, B, bias=bias, aux=aux, act=1), aux
        if kind == "dgelu":
            return K.gemm_nt(A, B, aux=pre, act=2), aux
        if kind == "bias_drop_resid":
            return K.gemm_nt(A, B, bias=bias, resid=resid, out_dtype=torch.float32, drop=drop), aux
        if kind == "layerscale":
            return K.gemm_nt(A, B, bias=bias, gamma=gamma, resid=resid, out_dtype=torch.float32), aux
        return K.gemm_nt(A, B, bias=bias, gamma=gamma, resid=resid, out_dtype=torch.float32, rowscale=rowscale), aux

    try:
        for kind in ("bias_bf16", "bias_f32", "gelu", "dgelu", "bias_drop_resid", "layerscale", "layerscale_droppath"):
            lib.x2_tune(1, 1); lib.x2_tune(3, 1)
            ref, ref_aux = run(kind)
            lib.x2_tune(1, 3); lib.x2_tune(3, tmw)
            got, got_aux = run(kind)
            for g_, r_ in ((got, ref), (got_aux, ref_aux)):
                tol = 4e-6 if g_.dtype == torch.float32 else 8e-3
                assert float((g_.float() - r_.float()).abs().max()) <= tol * max(1.0, float(r_.float().abs().max())), kind
    finally:
        lib.x2_tune(1, 0); lib.x2_tune(3, 0)


@pytest.mark.parametrize("K_", [64, 128, 192, 256, 320, 768, 3072])
def test_gemm_nt_160_row_ring_variants_are_bit_identical(K, K_):
    """The 160 x 256 tile has three main loops: the two-stage ring (x2_tune(10, 1)), the three-stage ring with one barrier per step
    (10, 2) and the shipped one - three stages + the fragment reads of the next step issued behind the current step's last MFMAs
    (10, 0).  Same products in the same order: the outputs must be bit-identical, for every prologue / tail length of the loops
    (1, 2, 3, 4, 5 contraction steps and long ones), ragged M and N, a bf16 and an fp32 + residual epilogue."""
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    M, N = 1000, 712
    A, B = bf(rnd(M, K_, seed=51)).to(dev), bf(rnd(N, K_, seed=52, scale=K_ ** -0.5)).to(dev)
    bias, gamma, resid = rnd(N, seed=53).to(dev), rnd(N, seed=54).to(dev), rnd(M, N, seed=55).to(dev)
    try:
        lib.x2_tune(1, 3); lib.x2_tune(3, 5)
        outs = {}
        for ring in (1, 2, 0):
            lib.x2_tune(10, ring)
            outs[ring] = (K.gemm_nt(A, B, bias=bias), K.gemm_nt(A, B, bias=bias, gamma=gamma, resid=resid, out_dtype=torch.float32))
        torch.cuda.synchronize()
        ref = (A.float() @ B.float().t() + bias)
        assert relerr(outs[1][0], ref.cpu()) < 6e-3
        for ring in (2, 0):
            assert torch.equal(outs[ring][0], outs[1][0]) and torch.equal(outs[ring][1], outs[1][1]), ring
    finally:
        lib.x2_tune(10, 0); lib.x2_tune(1, 0); lib.x2_tune(3, 0)


@pytest.mark.parametrize("M,N,K_,slices", [(768, 768, 30528, 0), (96, 768, 30528, 0), (40, 256, 4096, 0), (300, 200, 1024, 3),
                                           (130, 136, 640, 10), (256, 256, 128, 0), (384, 1024, 30528, 7)])
def test_gemm_nt_split_contraction(K, M, N, K_, slices):
    """few output tiles, long contraction (input gradient of the tied MLM decoder): slices of K into partial products
    added in a fixed order; equal to the one-launch product and bitwise reproducible."""
    A, B = bf(rnd(M, K_, seed=11)), bf(rnd(N, K_, seed=12, scale=K_ ** -0.5))
    ref = A.double() @ B.double().t()
    out = K.gemm_nt_splitk(A.to(dev), B.to(dev), slices=slices)
    assert relerr(out, ref) < 1e-5
    again = K.gemm_nt_splitk(A.to(dev), B.to(dev), slices=slices)
    assert torch.equal(out, again)
    wide = torch.zeros(M, N + 64, device=dev)
    K.gemm_nt_splitk(A.to(dev), B.to(dev), out=wide[:, 32:32 + N], slices=slices)          # strided output rows
    assert torch.equal(wide[:, 32:32 + N], out) and float(wide[:, :32].abs().max()) == 0.0 and float(wide[:, 32 + N:].abs().max()) == 0.0


@pytest.mark.parametrize("R,Hd,V", [(200, 128, 1000), (37, 64, 30522), (768, 768, 30522), (96, 1024, 30522), (64, 128, 512)])
def test_fused_mlm_cross_entropy(K, R, Hd, V):
    """decoder GEMM + cross-entropy with the logits kept in the accumulators (x2_mlm_ce_fwd / x2_ce_combine / x2_mlm_ce_bwd)
    against float64 log-softmax of the same bf16 operands, and against the unfused kernels (fp32 logits + x2_ce_*)."""
    Vp = K.round_up(V, 64)
    x = bf(rnd(R, Hd, seed=21))
    E = torch.zeros(Vp, Hd)
    E[:V] = rnd(V, Hd, seed=22, scale=3.0 * Hd ** -0.5)
    E = bf(E)
    bias = torch.zeros(Vp)
    bias[:V] = rnd(V, seed=23)
    g_ = torch.Generator().manual_seed(24)
    labels = torch.randint(0, V, (R,), generator=g_)
    labels[::5] = -100
    labels[1] = V - 1                                   # last valid column (the chunk that also holds the padding)
    xd, Ed, bd, ld = x.to(dev), E.to(dev), bias.to(dev), labels.to(dev)
    stat, lse = K.mlm_ce_fwd(xd, Ed, bd, ld, V)
    z = x.double() @ E.double().t()[:, :V] + bias[:V].double()
    ref_lse = torch.logsumexp(z, -1)
    valid = labels >= 0
    ref_rows = ref_lse - z[torch.arange(R), labels.clamp_min(0)]
    assert float((lse.cpu().double() - ref_lse).abs().max()) < 1e-4
    assert abs(float(stat[0]) - float(ref_rows[valid].mean())) < 1e-5 * max(1.0, float(ref_rows[valid].mean()))
    assert float(stat[1]) == float(valid.sum())
    g = torch.tensor([0.7], device=dev)
    dl = K.mlm_ce_bwd(xd, Ed, bd, ld, lse, g, stat, V)
    ref_dl = torch.softmax(z, -1)
    ref_dl[torch.arange(R), labels.clamp_min(0)] -= 1.0
    ref_dl = ref_dl * (0.7 / float(valid.sum())) * valid.double().unsqueeze(1)
    assert relerr(dl[:, :V], ref_dl) < 6e-3
    assert float(dl[:, V:].float().abs().max()) == 0.0 if Vp > V else True
    assert float(dl[~valid.to(dev)].float().abs().max()) == 0.0
    # the unfused kernels on materialised fp32 logits: same statistics, same gradient up to the bf16 rounding of dl
    logits = K.gemm_nt(xd, Ed, bias=bd, out_dtype=torch.float32)
    stat2, lse2 = K.ce_fwd(logits, ld, C_valid=V)
    assert float((lse - lse2).abs().max()) < 2e-5 and abs(float(stat[0] - stat2[0])) < 2e-6 * max(1.0, float(stat2[0]))
    dl2 = K.ce_bwd(logits, ld, lse2, g, stat2, C_valid=V, out_dtype=torch.bfloat16)
    assert relerr(dl, dl2.float().cpu()) < 6e-3


def test_gemm_nt_strided_views(K):
    """operands / outputs that are column slices of wider buffers (fused QKV layouts)."""
    M, Kd = 394, 128
    wide = bf(rnd(M, 3 * Kd, seed=7)).to(dev)
    B = bf(rnd(256, Kd, seed=8)).to(dev)
    outw = torch.zeros(M, 512, device=dev, dtype=torch.bfloat16)
    K.gemm_nt(wide[:, Kd:2 * Kd], B, out=outw[:, 256:])
    ref = wide[:, Kd:2 * Kd].float().cpu() @ B.float().cpu().t()
    assert relerr(outw[:, 256:], ref) < 6e-3
    assert float(outw[:, :256].abs().max()) == 0.0


@pytest.fixture(params=[1, 2], ids=["tn128x128", "tn256x256"])
def tn_tile(request):
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    lib.x2_tune(5, request.param)        # 1: always the 128x128 kernel, 2: 256x256 whenever the contraction allows
    yield request.param
    lib.x2_tune(5, 0)


@pytest.mark.parametrize("Mc,N,K_", [(128, 128, 128), (788, 768, 768), (1920, 256, 3072), (12608, 768, 2304), (100, 136, 72),
                                     (64, 256, 256), (3840, 520, 264), (18464, 512, 1024), (1154, 1024, 1024), (33, 256, 256)])
def test_gemm_tn_grouped(K, Mc, N, K_, tn_tile):
    dY, X = bf(rnd(Mc, N, seed=1)), bf(rnd(Mc, K_, seed=2))
    ref = dY.float().t() @ X.float()
    dW = torch.full((N, K_), 7.0, device=dev)
    K.gemm_tn_grouped([(dY.to(dev), X.to(dev), dW)])
    assert relerr(dW, ref) < 1e-5
    K.gemm_tn_grouped([(dY.to(dev), X.to(dev), dW)], accumulate=True)
    assert relerr(dW, 2 * ref) < 1e-5
    dW.zero_()
    K.gemm_tn_grouped([(dY.to(dev), X.to(dev), dW)], accumulate=True, split=3)
    assert relerr(dW, ref) < 1e-5
    if tn_tile == 2:    # deterministic split (256x256 kernel, ragged contraction lengths included): plain store, several slice counts
        for split in (1, 2, 4):
            dW.fill_(3.0)
            K.gemm_tn_grouped([(dY.to(dev), X.to(dev), dW)], split=split)
            assert relerr(dW, ref) < 1e-5


def test_gemm_tn_group_of_problems_and_padded_rows(K, tn_tile):
    probs, refs = [], []
    for i, (Mc, N, K_) in enumerate([(500, 256, 128), (500, 128, 384), (700, 64, 64), (64, 8, 8)]):
        dY, X = bf(rnd(Mc, N, seed=10 + i)), bf(rnd(Mc, K_, seed=20 + i))
        probs.append((dY.to(dev), X.to(dev), torch.empty(N, K_, device=dev)))
        refs.append(dY.float().t() @ X.float())
    # readable row width larger than N (vocabulary padded to a multiple of 64: N=250 of ld=256)
    dY, X = bf(rnd(300, 256, seed=31)), bf(rnd(300, 64, seed=32))
    probs.append((dY.to(dev), X.to(dev), torch.empty(250, 64, device=dev), 256, 64))
    refs.append(dY.float().t()[:250] @ X.float())
    K.gemm_tn_grouped(probs)
    for p, r in zip(probs, refs):
        assert relerr(p[2], r) < 1e-5
    # one layer's worth of mixed problems with 64-aligned contractions (the 256x256 path, automatic split)
    probs, refs = [], []
    for i, (Mc, N, K_) in enumerate([(1920, 768, 3072), (1920, 3072, 768), (1920, 768, 768), (1920, 2304, 768), (3136, 1536, 768),
                                     (1920, 250, 64)]):
        dY, X = bf(rnd(Mc, round_up8(N), seed=40 + i)), bf(rnd(Mc, K_, seed=50 + i))
        probs.append((dY.to(dev), X.to(dev), torch.empty(N, K_, device=dev), round_up8(N), K_))
        refs.append(dY.float().t()[:N] @ X.float())
    for split in ((0, 1, 3) if tn_tile == 2 else (0, 1)):
        for p in probs:
            p[2].fill_(9.0)
        K.gemm_tn_grouped(probs, split=split)
        for p, r in zip(probs, refs):
            assert relerr(p[2], r) < 1e-5


def round_up8(n):
    return (n + 7) // 8 * 8


def attn_ref(q, k, v, scale, add):
    """oracle attention on (B,H,L,d) fp32 leaves; returns out and grads for dout."""
    return O.attention_core(q, k, v, scale, add)


def run_attention(K, B, Bkv, H, Lq, Lk, use_bias, use_mask, kv_map, seed, bias_log2=False, narrow_biasT=False):
    d = 64
    qh = bf(rnd(B, Lq, H * d, seed=seed)); kh = bf(rnd(Bkv, Lk, H * d, seed=seed + 1)); vh = bf(rnd(Bkv, Lk, H * d, seed=seed + 2))
    doh = bf(rnd(B, Lq, H * d, seed=seed + 3))
    scale = d ** -0.5
    bias = rnd(H, Lq, Lk, seed=seed + 4) if use_bias else None
    mask = None
    if use_mask:
        keep = (torch.rand(B, Lk, generator=torch.Generator().manual_seed(seed + 5)) > 0.3).float()
        keep[:, 0] = 1
        mask = (1 - keep) * -10000.0
    # ---- oracle (CPU fp32, autograd) ----
    q = qh.float().view(B, Lq, H, d).permute(0, 2, 1, 3).requires_grad_(True)
    k0 = kh.float().view(Bkv, Lk, H, d).permute(0, 2, 1, 3).requires_grad_(True)
    v0 = vh.float().view(Bkv, Lk, H, d).permute(0, 2, 1, 3).requires_grad_(True)
    idx = torch.tensor(kv_map) if kv_map is not None else torch.arange(B)
    bias_leaf = bias.clone().requires_grad_(True) if use_bias else None
    add = None
    if use_bias:
        add = bias_leaf.unsqueeze(0)
    if use_mask:
        add = mask[:, None, None, :] if add is None else add + mask[:, None, None, :]
    out = attn_ref(q, k0[idx], v0[idx], scale, add)
    out.backward(doh.float().view(B, Lq, H, d).permute(0, 2, 1, 3))
    ref_out = out.permute(0, 2, 1, 3).reshape(B, Lq, H * d)
    # ---- HIP ----
    # (transposed bias: whole 128-query chunks for the long one-pass backward, as round_up(577, 64) = 640 happens to be; `narrow_biasT` keeps 64)
    Lkp, Lqp = K.round_up(Lk, 64), K.round_up(Lq, 128 if Lq > 208 and not narrow_biasT else 64)
    kw = {}
    if use_bias:
        bp = torch.zeros(H, Lq, Lkp); bp[:, :, :Lk] = bias
        bT = torch.zeros(H, Lk, Lqp); bT[:, :, :Lq] = bias.transpose(1, 2)
        if bias_log2:         # the bias in log2 units, as kernels.relpos_bias(log2=True) hands it over: the one-fma score path
            bp, bT = bp * K.LOG2E, bT * K.LOG2E
            kw["bias_log2"] = True
        kw.update(bias=bp.to(dev), biasT=bT.to(dev))
    if use_mask:
        mp = torch.zeros(B, Lkp); mp[:, :Lk] = mask
        kw["mask"] = mp.to(dev)
    if kv_map is not None:
        kv_idx = torch.tensor(kv_map, dtype=torch.int32)
        order = torch.argsort(kv_idx, stable=True).to(torch.int32)
        counts = torch.bincount(kv_idx, minlength=Bkv)
        off = torch.zeros(Bkv + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0)
        kw.update(kv_idx=kv_idx.to(dev), seq_off=off.to(dev), seq_ids=order.to(dev))
    qd, kd, vd, dod = (t.reshape(-1, H * d).to(dev) for t in (qh, kh, vh, doh))
    od = torch.empty_like(qd)
    lse = torch.empty(B * H * Lq, device=dev); delta = torch.empty_like(lse)
    # per-row kernels (no CSR given), then the grouped kernel (one workgroup per shared K/V batch and head) when rows share K/V
    for drop_keys in ((("biasT", "seq_off", "seq_ids"), ("biasT",)) if kv_map is not None else (("biasT",),)):
        od.fill_(float("nan")); lse.fill_(float("nan"))
        K.attn_fwd(K.view3(qd, B, Lq), K.view3(kd, Bkv, Lk), K.view3(vd, Bkv, Lk), B, Bkv, H, Lq, Lk, scale,
                   K.view3(od, B, Lq), lse, **{k_: v_ for k_, v_ in kw.items() if k_ not in drop_keys})
        assert relerr(od.view(B, Lq, H * d), ref_out) < 8e-3 and bool(torch.isfinite(lse).all())
    tol = 1.5e-2   # P and dS are rounded to bf16 before the second MFMA
    want = (q.grad.permute(0, 2, 1, 3).reshape(B, Lq, H * d), k0.grad.permute(0, 2, 1, 3).reshape(Bkv, Lk, H * d),
            v0.grad.permute(0, 2, 1, 3).reshape(Bkv, Lk, H * d))

    def backward(phase=0, into=None, ask_form=False):
        dq, dk, dv, delta = into if into is not None else (torch.full_like(qd, float("nan")), torch.full_like(kd, float("nan")),
                                                            torch.full_like(vd, float("nan")), torch.full_like(lse, float("nan")))
        dS = torch.zeros(B, H, Lq, Lkp, device=dev, dtype=torch.bfloat16) if use_bias else None
        form = K.attn_bwd(K.view3(qd, B, Lq), K.view3(kd, Bkv, Lk), K.view3(vd, Bkv, Lk), K.view3(od, B, Lq), K.view3(dod, B, Lq),
                          B, Bkv, H, Lq, Lk, scale, lse, delta, K.view3(dq, B, Lq), K.view3(dk, Bkv, Lk), K.view3(dv, Bkv, Lk),
                          dS=dS, phase=phase, ask_form=ask_form, **kw)
        return form if ask_form else (dq, dk, dv, delta, dS)

    def check(dq, dk, dv, delta, dS):
        assert relerr(dq.view(B, Lq, H * d), want[0]) < tol
        assert relerr(dk.view(Bkv, Lk, H * d), want[1]) < tol
        assert relerr(dv.view(Bkv, Lk, H * d), want[2]) < tol
        assert bool(torch.isfinite(delta).all())
        if use_bias:
            assert relerr(dS[..., :Lk].float().sum(0), bias_leaf.grad) < tol
    got = backward()
    check(*got)
    if backward(ask_form=True) in (1, 3):
        # the one-pass kernels' by-product: per-sequence column sums of the stored dQ / dV rows (the q / v bias gradient of a fused qkv projection)
        cs = torch.full((B, 2, H * d), float("nan"), device=dev)
        dq_, dk_, dv_, de_ = (torch.empty_like(t) for t in (qd, kd, vd, lse))
        assert K.attn_bwd(K.view3(qd, B, Lq), K.view3(kd, Bkv, Lk), K.view3(vd, Bkv, Lk), K.view3(od, B, Lq), K.view3(dod, B, Lq), B, Bkv, H, Lq, Lk,
                          scale, lse, de_, K.view3(dq_, B, Lq), K.view3(dk_, Bkv, Lk), K.view3(dv_, Bkv, Lk),
                          dS=torch.zeros(B, H, Lq, Lkp, device=dev, dtype=torch.bfloat16) if use_bias else None, colsum_ws=cs, **kw) in (1, 3)
        assert torch.equal(dq_, got[0]) and torch.equal(dv_, got[2])
        assert relerr(cs[:, 0], dq_.view(B, Lq, H * d).float().sum(1).cpu()) < 2e-6 and relerr(cs[:, 1], dv_.view(B, Lk, H * d).float().sum(1).cpu()) < 2e-6
    # 64 < L <= 208 without K/V sharing, and rows sharing K/V with Lq <= 128, Lk <= 208: the call above ran a ONE-PASS kernel (one
    # workgroup per (sequence, head) / (shared K/V batch, head) forms S, P, dP, dS once); x2_tune(14, 1) runs the dQ + dK/dV pair on
    # the same inputs - against the oracle as well, and the two forms against each other (same products, different summation
    # order in Delta and in the accumulators: bf16-rounding-sized differences)
    form = backward(ask_form=True)
    # ... and 208 < L (<= 640 queries, <= 768 keys) without K/V sharing: form 3, one workgroup per (sequence, head) walking 256-key parts x 128-query
    # chunks (X2VLM-large, N = 577), when the transposed bias covers whole chunks
    long_ok = kv_map is None and 208 < Lq <= 640 and 208 < Lk <= 768 and (not use_bias or Lqp >= 128 * ((Lq + 127) // 128))
    assert form == (1 if kv_map is None and 64 < Lq <= 208 and 64 < Lk <= 208 else 2 if kv_map is not None and Lq <= 128 and Lk <= 208
                    else 3 if long_ok else 0)
    one_pass = form != 0
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    if one_pass:
        lib.x2_tune(14, 1)
        try:
            two = backward()
            check(*two)
            for a_, b_ in zip(got[:3], two[:3]):
                assert relerr(a_.float(), b_.float().cpu()) < 6e-3
            assert relerr(got[3], two[3].cpu()) < 1e-5
            assert backward(ask_form=True) == 0
            if use_bias:
                assert relerr(got[4][..., :Lk].float().sum(0), two[4][..., :Lk].float().sum(0).cpu()) < 6e-3
                # pad columns of the dS stream (keys past Lk): the buffer was zeroed, neither form writes anything else there
                if Lkp > Lk:
                    assert float(got[4][..., Lk:].float().abs().max()) == 0.0 and float(two[4][..., Lk:].float().abs().max()) == 0.0
            _phase_split_matches(backward, two)
        finally:
            lib.x2_tune(14, 0)
    else:
        _phase_split_matches(backward, got)


def _phase_split_matches(backward, ref):
    """the two halves as separate calls (phase 1: dQ, dS, delta; phase 2: dK / dV from that delta) - what lets a caller put the K/V-side
    gradients on another stream - give the same bits as the two-kernel form in one call"""
    dq, dk, dv, delta, _ = ref
    into = (torch.zeros_like(dq), torch.zeros_like(dk), torch.zeros_like(dv), torch.zeros_like(delta))
    backward(phase=1, into=into)
    assert torch.equal(into[0], dq) and float(into[1].float().abs().max()) == 0.0 and float(into[2].float().abs().max()) == 0.0
    backward(phase=2, into=into)
    assert torch.equal(into[1], dk) and torch.equal(into[2], dv) and torch.equal(into[3], delta)


def test_attention_vision_bias(K):
    run_attention(K, B=3, Bkv=3, H=12, Lq=197, Lk=197, use_bias=True, use_mask=False, kv_map=None, seed=100)


def test_attention_text_self_mask(K):
    run_attention(K, B=5, Bkv=5, H=12, Lq=30, Lk=30, use_bias=False, use_mask=True, kv_map=None, seed=200)
    run_attention(K, B=3, Bkv=3, H=2, Lq=8, Lk=8, use_bias=False, use_mask=True, kv_map=None, seed=210)
    run_attention(K, B=2, Bkv=2, H=4, Lq=40, Lk=40, use_bias=False, use_mask=True, kv_map=None, seed=220)


def test_attention_vision_bias_resident_kernels(K):
    """The strip-walking resident forward / dQ kernels (one workgroup per (image, head), K / V loaded once; Lk <= 208) and the
    two-workgroup resident kernels that serve 208 < Lk <= 256, against the same oracle: N = 197 with the relative-position bias (both
    bias units), a short ragged case (one partial key tile), the full 208 rows the strips hold, and 230 / 256 keys."""
    run_attention(K, B=3, Bkv=3, H=12, Lq=197, Lk=197, use_bias=True, use_mask=False, kv_map=None, seed=100)
    run_attention(K, B=3, Bkv=3, H=12, Lq=197, Lk=197, use_bias=True, use_mask=False, kv_map=None, seed=100, bias_log2=True)
    run_attention(K, B=2, Bkv=2, H=3, Lq=70, Lk=70, use_bias=True, use_mask=True, kv_map=None, seed=110)
    run_attention(K, B=2, Bkv=2, H=2, Lq=208, Lk=208, use_bias=True, use_mask=False, kv_map=None, seed=120)
    run_attention(K, B=2, Bkv=2, H=2, Lq=208, Lk=208, use_bias=True, use_mask=False, kv_map=None, seed=120, bias_log2=True)
    run_attention(K, B=2, Bkv=2, H=2, Lq=230, Lk=230, use_bias=True, use_mask=False, kv_map=None, seed=130)
    run_attention(K, B=1, Bkv=1, H=3, Lq=256, Lk=256, use_bias=True, use_mask=True, kv_map=None, seed=140)


def test_attention_cross_shared_kv(K):
    """per-row forward + grouped dQ (one workgroup per (image, head) over all the rows sharing that image's K / V)."""
    run_attention(K, B=6, Bkv=3, H=12, Lq=30, Lk=197, use_bias=False, use_mask=True, kv_map=[0, 2, 1, 1, 0, 1], seed=300)
    run_attention(K, B=4, Bkv=2, H=2, Lq=8, Lk=5, use_bias=False, use_mask=False, kv_map=[1, 0, 1, 1], seed=310)
    # an image nobody attends to (zero gradient, nothing to do), and more rows per image than one 128-query pass holds
    run_attention(K, B=4, Bkv=3, H=2, Lq=30, Lk=197, use_bias=False, use_mask=True, kv_map=[0, 2, 2, 0], seed=320)
    run_attention(K, B=11, Bkv=2, H=3, Lq=30, Lk=70, use_bias=False, use_mask=True, kv_map=[0] * 9 + [1] * 2, seed=330)


def test_attention_one_pass_backward_shapes(K):
    """attn_bwd_onepass_kernel beyond the N = 197 / 208 / 70 cases above: 8 key strips (every wave owns exactly one), 7 (one wave
    owns none and only keeps the barriers), more queries than keys and the reverse, without a bias (no dS stream) and with bias + mask."""
    run_attention(K, B=2, Bkv=2, H=2, Lq=128, Lk=128, use_bias=True, use_mask=False, kv_map=None, seed=500)
    run_attention(K, B=2, Bkv=2, H=3, Lq=100, Lk=100, use_bias=True, use_mask=False, kv_map=None, seed=510, bias_log2=True)
    run_attention(K, B=1, Bkv=1, H=2, Lq=150, Lk=90, use_bias=False, use_mask=True, kv_map=None, seed=520)
    run_attention(K, B=2, Bkv=2, H=2, Lq=90, Lk=150, use_bias=True, use_mask=True, kv_map=None, seed=530)
    run_attention(K, B=33, Bkv=33, H=12, Lq=197, Lk=197, use_bias=True, use_mask=False, kv_map=None, seed=540, bias_log2=True)


def test_attention_one_pass_forms_agree_under_dropout(K):
    """Probability dropout (xbert.py:399) in the shared-K/V one-pass backward: the element index of (sequence, head, query, key) - hence
    the mask - is the one the forward and the two-kernel backward use, so on the same inputs the two backward forms agree to rounding
    (a wrong index would decorrelate the masks: errors of order one).  9 sequences on one image = 3 chunks of 4."""
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    d, H, Lq, Lk, kv_map = 64, 3, 30, 70, [0] * 9 + [1] * 2
    B, Bkv = len(kv_map), 2
    qd, dod = (bf(rnd(B * Lq, H * d, seed=600 + i)).to(dev) for i in range(2))
    kd, vd = (bf(rnd(Bkv * Lk, H * d, seed=610 + i)).to(dev) for i in range(2))
    keep = (torch.rand(B, Lk, generator=torch.Generator().manual_seed(7)) > 0.3).float(); keep[:, 0] = 1
    mp = torch.zeros(B, K.round_up(Lk, 64)); mp[:, :Lk] = (1 - keep) * -10000.0
    kv_idx = torch.tensor(kv_map, dtype=torch.int32)
    off = torch.zeros(Bkv + 1, dtype=torch.int32); off[1:] = torch.cumsum(torch.bincount(kv_idx, minlength=Bkv), 0)
    kw = dict(mask=mp.to(dev), kv_idx=kv_idx.to(dev), seq_off=off.to(dev), seq_ids=torch.argsort(kv_idx, stable=True).to(torch.int32).to(dev),
              drop=K.dropout_spec(0.1, 4242, 5))
    od = torch.empty_like(qd); lse = torch.empty(B * H * Lq, device=dev)
    K.attn_fwd(K.view3(qd, B, Lq), K.view3(kd, Bkv, Lk), K.view3(vd, Bkv, Lk), B, Bkv, H, Lq, Lk, d ** -0.5, K.view3(od, B, Lq), lse,
               **{k_: v_ for k_, v_ in kw.items() if k_ not in ("seq_off", "seq_ids")})
    res = []
    try:
        for knob in (0, 1):
            lib.x2_tune(14, knob)
            dq, dk, dv, delta = torch.full_like(qd, float("nan")), torch.full_like(kd, float("nan")), torch.full_like(vd, float("nan")), torch.empty_like(lse)
            args = (K.view3(qd, B, Lq), K.view3(kd, Bkv, Lk), K.view3(vd, Bkv, Lk), K.view3(od, B, Lq), K.view3(dod, B, Lq), B, Bkv, H, Lq, Lk,
                    d ** -0.5, lse, delta, K.view3(dq, B, Lq), K.view3(dk, Bkv, Lk), K.view3(dv, Bkv, Lk))
            assert K.attn_bwd(*args, ask_form=True, **kw) == (2 if knob == 0 else 0)
            K.attn_bwd(*args, **kw)
            res.append((dq, dk, dv, delta))
    finally:
        lib.x2_tune(14, 0)
    for a_, b_ in zip(res[0], res[1]):
        assert bool(torch.isfinite(a_.float()).all()) and relerr(a_.float(), b_.float().cpu()) < 6e-3


def test_attention_long_keys(K):
    """N = 577 (384 px) exercises many key tiles of the online softmax - and, in the backward, the long one-pass kernel (three 256-key
    parts x five 128-query chunks, dQ partials through the workspace) against the oracle and against the dQ + dK/dV pair."""
    run_attention(K, B=1, Bkv=1, H=2, Lq=577, Lk=577, use_bias=True, use_mask=False, kv_map=None, seed=400)
    run_attention(K, B=2, Bkv=2, H=4, Lq=577, Lk=577, use_bias=True, use_mask=False, kv_map=None, seed=410, bias_log2=True)


def test_attention_long_one_pass_backward_shapes(K):
    """attn_bwd_onepass_long_kernel beyond N = 577: the largest geometry it takes (640 x 768: full parts and chunks), one part + a ragged second
    one (one strip: seven waves idle in it), more queries than keys and the reverse, no bias (no dS stream) with a mask, a whole batch of heads
    through the XCD-aware block order - and a transposed bias that stops short of whole chunks (the pair runs, form 0)."""
    run_attention(K, B=1, Bkv=1, H=2, Lq=640, Lk=768, use_bias=True, use_mask=True, kv_map=None, seed=700)
    run_attention(K, B=2, Bkv=2, H=2, Lq=272, Lk=272, use_bias=True, use_mask=False, kv_map=None, seed=710, bias_log2=True)
    run_attention(K, B=1, Bkv=1, H=3, Lq=500, Lk=230, use_bias=False, use_mask=True, kv_map=None, seed=720)
    run_attention(K, B=2, Bkv=2, H=2, Lq=215, Lk=600, use_bias=True, use_mask=False, kv_map=None, seed=730, bias_log2=True)
    run_attention(K, B=5, Bkv=5, H=16, Lq=577, Lk=577, use_bias=True, use_mask=False, kv_map=None, seed=740, bias_log2=True)
    run_attention(K, B=1, Bkv=1, H=2, Lq=300, Lk=300, use_bias=True, use_mask=False, kv_map=None, seed=750, narrow_biasT=True)


def test_attention_long_one_pass_at_the_full_size_of_x2vlm_large(K):
    """BASELINE configs[3] on one GPU: batch 32 x 16 heads x N = 577 (512 workgroups = two rounds of the chip, 84 MB of dQ partials in the workspace,
    the XCD-aware block order over 16 heads): the oracle does not reach this size in test time, so the long one-pass backward is held against the
    dQ + dK/dV pair on the same inputs (the pair against the oracle: test_attention_long_keys) and against itself (two runs, same bits)."""
    lib = importlib.import_module("x2-vlm_amd._lib").lib()
    B, H, N, d = 32, 16, 577, 64
    HD = H * d
    g_ = torch.Generator().manual_seed(77)
    qkv = (0.5 * torch.randn(B * N, 3 * HD, generator=g_)).bfloat16().to(dev)
    dout = (0.1 * torch.randn(B * N, HD, generator=g_)).bfloat16().to(dev)
    bias = torch.randn(H, N, K.round_up(N, 64), generator=g_).to(dev) * K.LOG2E
    biasT = torch.zeros(H, N, K.round_up(N, 64), device=dev); biasT[:, :, :N] = bias[:, :, :N].transpose(1, 2)
    out = torch.empty(B * N, HD, device=dev, dtype=torch.bfloat16); lse = torch.empty(B * H * N, device=dev)
    K.attn_fwd(K.view3(qkv, B, N, 0), K.view3(qkv, B, N, HD), K.view3(qkv, B, N, 2 * HD), B, B, H, N, N, d ** -0.5, K.view3(out, B, N), lse,
               bias=bias, bias_log2=True)
    res = []
    try:
        for knob in (0, 1, 0):
            lib.x2_tune(14, knob)
            dqkv = torch.full_like(qkv, float("nan")); delta = torch.full_like(lse, float("nan"))
            dS = torch.zeros(B, H, N, K.round_up(N, 64), device=dev, dtype=torch.bfloat16)
            args = (K.view3(qkv, B, N, 0), K.view3(qkv, B, N, HD), K.view3(qkv, B, N, 2 * HD), K.view3(out, B, N), K.view3(dout, B, N), B, B, H, N, N,
                    d ** -0.5, lse, delta, K.view3(dqkv, B, N, 0), K.view3(dqkv, B, N, HD), K.view3(dqkv, B, N, 2 * HD))
            kw = dict(dS=dS, bias=bias, biasT=biasT, bias_log2=True)
            assert K.attn_bwd(*args, ask_form=True, **kw) == (3 if knob == 0 else 0)
            K.attn_bwd(*args, **kw)
            res.append((dqkv, delta, dS.float().sum(0)))
            del dS
    finally:
        lib.x2_tune(14, 0)
    one, two, again = res
    assert bool(torch.isfinite(one[0].float()).all()) and bool(torch.isfinite(one[1]).all())
    for j in range(3):            # dQ, dK, dV blocks of the fused gradient
        assert relerr(one[0][:, j * HD:(j + 1) * HD].float(), two[0][:, j * HD:(j + 1) * HD].float().cpu()) < 6e-3, j
    assert relerr(one[1], two[1].cpu()) < 1e-5 and relerr(one[2], two[2].cpu()) < 6e-3
    assert torch.equal(one[0], again[0]) and torch.equal(one[1], again[1]) and torch.equal(one[2], again[2])


@pytest.mark.parametrize("rows,D,period", [(37, 768, 0), (788, 768, 0), (4 * 196, 768, 196), (50, 128, 0), (9, 1536, 0)])
def test_layernorm(K, rows, D, period):
    total = rows if period == 0 else rows // period * (period + 1)
    x = rnd(total, D, seed=1, scale=2.0) + 0.5
    w, b = rnd(D, seed=2) * 0.1 + 1, rnd(D, seed=3) * 0.1
    dy = rnd(total, D, seed=4)
    sel = torch.arange(total) if period == 0 else torch.tensor([r + r // period + 1 for r in range(rows)])
    xl = x.clone().requires_grad_(True); wl = w.clone().requires_grad_(True); bl = b.clone().requires_grad_(True)
    ref = O.layer_norm(xl[sel], wl, bl, 1e-6)
    ref.backward(dy[sel])
    yb, yf, mean, rstd = K.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), 1e-6, rows=rows, period=period, want_f32=True)
    assert relerr(yf[sel], ref) < 1e-5 and relerr(yb[sel], ref) < 6e-3
    dw, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dres = rnd(total, D, seed=5)
    dx, dxb = K.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, w.to(dev), dw, db, dres=dres.to(dev), period=period,
                              want_bf16=True)
    assert relerr(dx[sel], xl.grad[sel] + dres[sel]) < 2e-5
    assert relerr(dxb[sel], xl.grad[sel]) < 6e-3          # bf16 copy = gradient of the LN input alone (feeds the producing linear)
    assert relerr(dw, wl.grad) < 2e-5 and relerr(db, bl.grad) < 2e-5
    # bf16 incoming gradient (input-gradient GEMMs of the pre-LN blocks write bf16): same arithmetic on the rounded values
    dyb = bf(dy)
    xl2 = x.clone().requires_grad_(True)
    O.layer_norm(xl2[sel], w, b, 1e-6).backward(dyb.float()[sel])
    dwb, dbb = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dxh, _ = K.layernorm_bwd(dyb.to(dev), x.to(dev), mean, rstd, w.to(dev), dwb, dbb, dres=dres.to(dev), period=period)
    assert relerr(dxh[sel], xl2.grad[sel] + dres[sel]) < 2e-5
    dw2, db2, dcol = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx2, _ = K.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, w.to(dev), dw2, db2, dcol=dcol, period=period)
    assert relerr(dcol, xl.grad[sel].sum(0)) < 5e-5 and relerr(dx2[sel], xl.grad[sel]) < 2e-5
    if period == 0:
        # by-products of the FINAL output for the layer scale below (post=): bf16(r * (dx + dres)) and its column sums
        for rs in (None, (torch.rand(total, generator=torch.Generator().manual_seed(9)) > 0.2).float() * 1.25):
            cs, dw3, db3 = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
            dx3, dx3b = K.layernorm_bwd(dyb.to(dev), x.to(dev), mean, rstd, w.to(dev), dw3, db3, dres=dres.to(dev),
                                        post=(None if rs is None else rs.to(dev), cs))
            want = (xl2.grad + dres) * (1.0 if rs is None else rs[:, None])
            assert relerr(dx3, dxh.cpu()) < 1e-6 and relerr(dw3, dwb.cpu()) < 1e-6 and relerr(db3, dbb.cpu()) < 1e-6
            assert relerr(dx3b, want) < 6e-3 and relerr(cs, want.sum(0)) < 2e-5


def test_colsum_layerscale_casts(K):
    y = bf(rnd(500, 768, seed=1))
    out = torch.zeros(768, device=dev)
    K.colsum_bf16(y.to(dev), out)
    assert relerr(out, y.float().sum(0)) < 1e-5
    dx = rnd(500, 768, seed=2)
    for rs in (None, (torch.rand(500, generator=torch.Generator().manual_seed(3)) > 0.2).float() * 1.25):
        cs = torch.zeros(768, device=dev)
        dxb = K.rowscale_cast_colsum(dx.to(dev), cs, rowscale=None if rs is None else rs.to(dev))
        want = dx if rs is None else dx * rs[:, None]
        assert torch.equal(dxb.cpu(), bf(want)) and relerr(cs, want.sum(0)) < 1e-5
    w = rnd(300, 200, seed=5)
    assert torch.equal(K.cast_bf16(w.to(dev)).cpu(), bf(w))
    plain, tr = K.cast_transpose_bf16(w.to(dev), ldt=320)
    assert torch.equal(plain.cpu(), bf(w)) and torch.equal(tr[:, :300].cpu(), bf(w).t()) and float(tr[:, 300:].abs().max()) == 0


@pytest.mark.parametrize("M,D,F", [(500, 768, 1024), (197, 128, 64), (1000, 256, 3072)])
def test_layerscale_backward_without_activation(K, M, D, F):
    """x_out = x_in + r * gamma * (A . W^T + b) (beit2.py:206-207 with DropPath): the pieces the vision backward runs - gamma folded
    into the transposed weight copy, the weight-gradient GEMM on bf16(r * dX), x2_layerscale_finish - against autograd."""
    eng = importlib.import_module("x2-vlm_amd.engine")
    A, W, b, gamma = bf(rnd(M, F, seed=1)), rnd(D, F, seed=2, scale=F ** -0.5), rnd(D, seed=3), rnd(D, seed=4) * 0.3 + 0.1
    dx = rnd(M, D, seed=5)
    rs = (torch.rand(M, generator=torch.Generator().manual_seed(6)) > 0.2).double() * 1.25
    Ad, Wd, bd, gd = (t.double().requires_grad_(True) for t in (A.float(), W, b, gamma))
    ((rs[:, None] * gd * (Ad @ Wd.t() + bd)) * dx.double()).sum().backward()
    Wp, gp = torch.nn.Parameter(W.to(dev)), torch.nn.Parameter(gamma.to(dev))
    bank = eng.WeightBank()
    bank.prepare([(Wp, eng.TScale(gp))])
    plain, trs = bank.linear(Wp, tscale=gp)
    assert len(bank._c) == 1 and torch.equal(plain.cpu(), bf(W)) and torch.equal(trs.cpu(), bf(W * gamma[:, None]).t())
    cs, dg, db = torch.zeros(D, device=dev), torch.full((D,), 2.0, device=dev), torch.full((D,), 3.0, device=dev)
    dxb = K.rowscale_cast_colsum(dx.to(dev), cs, rowscale=rs.float().to(dev))
    dA = K.gemm_nt(dxb, trs, out_dtype=torch.float32)
    G = torch.empty(D, F, device=dev)
    K.gemm_tn_grouped([(dxb, A.to(dev), G)])
    K.layerscale_finish([(G, Wp.detach(), b.to(dev), gp.detach(), cs, dg, db)])
    torch.cuda.synchronize()
    assert relerr(dA, Ad.grad) < 1e-2 and relerr(G, Wd.grad) < 6e-3
    assert relerr(dg - 2.0, gd.grad) < 5e-3 and relerr(db - 3.0, bd.grad) < 1e-5
    # stale-copy rule: a new gamma value (same storage) rebuilds the scaled copy on the next request after a version bump
    with torch.no_grad():
        gp.mul_(2.0)
    _, trs2 = bank.linear(Wp, tscale=gp)
    assert torch.equal(trs2.cpu(), bf(W * (2.0 * gamma)[:, None]).t())


def test_multi_tensor_cast_and_reduce(K):
    """One-launch variants: stacked (W, W^T) copies of many weights == per-weight casts; batched stage-2 reductions."""
    eng = importlib.import_module("x2-vlm_amd.engine")
    bank = eng.WeightBank()
    ws = [torch.nn.Parameter(rnd(r, c, seed=10 + i).to(dev)) for i, (r, c) in enumerate(
        [(768, 768), (768, 768), (768, 768), (3072, 768), (768, 3072), (256, 768), (64, 132)] + [(128, 64)] * 60)]
    groups = [(ws[0], ws[1], ws[2])] + [(w,) for w in ws[3:]]
    bank.prepare(groups)
    for g in groups:
        plain, tr = bank.linear(*g)                      # cache hit: built by prepare()
        ref = bf(torch.cat([w.detach().cpu() for w in g], 0))
        assert torch.equal(plain.cpu(), ref) and torch.equal(tr.cpu(), ref.t())
    assert len(bank._c) == len(groups)
    items, refs = [], []
    for i, (nblk, nk, width) in enumerate([(37, 3, 768), (5, 1, 100), (200, 2, 3072), (788, 4, 768), (13, 2, 30), (65, 1, 8)] * 4):
        part = rnd(nblk, nk, width, seed=100 + i).to(dev)
        # a null output (slot 1 of the 4-row sets) skips that partial row: how one workspace serves two gradient arenas
        outs = tuple(None if (nk == 4 and k == 1) else torch.full((width,), float(k), device=dev) for k in range(nk))
        items.append((part, nblk, nk, width, outs))
        refs.append([part[:, k].double().sum(0).cpu() + k for k in range(nk)])
    K.reduce_partials_multi(items)
    for (part, nblk, nk, width, outs), ref in zip(items, refs):
        for o, r in zip(outs, ref):
            assert o is None or relerr(o, r) < 1e-5


def test_linear_f32_split_k(K):
    """Head-sized outputs take the K-sliced (atomic) path; large ones the single-pass path: same numbers."""
    for (M, N, Kd) in [(64, 256, 768), (192, 2, 768), (64, 768, 256), (1000, 700, 96)]:
        a, b, bias = rnd(M, Kd, seed=1), rnd(N, Kd, seed=2), rnd(N, seed=3)
        out = K.linear_f32(a.to(dev), b.to(dev), bias=bias.to(dev))
        assert relerr(out, a.double() @ b.double().t() + bias.double()) < 1e-5
        acc = torch.ones(M, N, device=dev)
        K.linear_f32(a.to(dev), b.to(dev), out=acc, accumulate=True, alpha=0.5)
        assert relerr(acc, 1 + 0.5 * (a.double() @ b.double().t())) < 1e-5


def test_patch_tokens_pool_relpos(K):
    cfg = O.OracleConfig(image_res=64, vision_layers=1)
    img = rnd(3, 3, 64, 64, seed=1)
    cols = K.patchify(img.to(dev), 16)
    B, g, p = 3, 4, 16
    ref = img.view(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * p * p)
    assert torch.equal(cols.cpu(), bf(ref))
    patch, cls = rnd(B * 16, 768, seed=2), rnd(768, seed=3)
    x = K.assemble_tokens(patch.to(dev), cls.to(dev), B, 16)
    refx = torch.cat([cls.expand(B, 1, 768), patch.view(B, 16, 768)], 1)
    assert torch.equal(x.cpu(), refx)
    dcls = torch.zeros(768, device=dev)
    dpatch = K.assemble_tokens_bwd(x, dcls)
    assert torch.equal(dpatch.cpu(), bf(patch)) and relerr(dcls, cls * B) < 1e-6
    w = torch.rand(B, 16, generator=torch.Generator().manual_seed(4)).round()
    w[:, 0] = 1
    for wt in (None, w):
        xx = refx.clone().to(dev)
        K.pool_tokens(xx, None if wt is None else wt.to(dev))
        ww = torch.ones(B, 16) if wt is None else wt
        pooled = (ww.unsqueeze(-1) * refx[:, 1:]).sum(1) / ww.sum(1, keepdim=True)
        assert relerr(xx[:, 0], pooled) < 1e-6
        gg = refx.clone().to(dev)
        K.pool_tokens(gg, None if wt is None else wt.to(dev), bwd=True)
        refg = refx[:, 1:] + (ww / ww.sum(1, keepdim=True)).unsqueeze(-1) * refx[:, :1]
        assert relerr(gg[:, 1:], refg) < 1e-6 and float(gg[:, 0].abs().max()) == 0
    idx = O.relative_position_index(4)
    table = rnd(int(idx.max()) + 1, 12, seed=5)
    bias, biasT = K.relpos_bias(table.to(dev), idx.to(dev))
    refb = table[idx.reshape(-1)].view(17, 17, 12).permute(2, 0, 1)
    assert torch.equal(bias[:, :, :17].cpu(), refb) and torch.equal(biasT[:, :, :17].cpu(), refb.transpose(1, 2))
    dS = bf(rnd(2, 12, 17, 64, seed=6)); dS[..., 17:] = 0
    dtab = torch.zeros_like(table).to(dev)
    K.relpos_bias_bwd(dS.to(dev), idx.to(dev), dtab)
    tl = table.clone().requires_grad_(True)
    (tl[idx.reshape(-1)].view(17, 17, 12).permute(2, 0, 1) * dS[..., :17].float().sum(0)).sum().backward()
    assert relerr(dtab, tl.grad) < 1e-5
    idx14 = O.relative_position_index(14).to(dev)                  # the 224px window, batch cut into slices
    T14 = int(idx14.max()) + 1
    dS = bf(rnd(40, 3, 197, 256, seed=7)); dS[..., 197:] = 0
    dtab = torch.ones(T14, 3, device=dev)
    K.relpos_bias_bwd(dS.to(dev), idx14, dtab)
    ref = torch.ones(T14, 3, dtype=torch.float64).index_add_(0, idx14.reshape(-1).cpu(),
                                                             dS[..., :197].double().sum(0).permute(1, 2, 0).reshape(-1, 3))
    assert relerr(dtab, ref) < 1e-5


def test_embed_gather_linear_l2norm(K):
    V, D, L, Bn = 100, 128, 8, 5
    ids = torch.randint(0, V, (Bn, L), generator=torch.Generator().manual_seed(1))
    word, pos, typ = rnd(V, D, seed=2), rnd(16, D, seed=3), rnd(2, D, seed=4)
    out = K.embed_fwd(ids.to(dev), word.to(dev), pos.to(dev), typ.to(dev))
    ref = word[ids] + pos[:L] + typ[0]
    assert relerr(out.view(Bn, L, D), ref) < 1e-6
    g = rnd(Bn * L, D, seed=5)
    dword, dpos, dtyp = torch.zeros(V, D, device=dev), torch.zeros(16, D, device=dev), torch.zeros(2, D, device=dev)
    K.embed_bwd(ids.to(dev), g.to(dev), dword, dpos, dtyp)
    rw = torch.zeros(V, D).index_add_(0, ids.reshape(-1), g)
    assert relerr(dword, rw) < 1e-5 and relerr(dpos[:L], g.view(Bn, L, D).sum(0)) < 1e-5 and relerr(dtyp[0], g.sum(0)) < 1e-5
    # the step's shapes: 7680 token rows of a region batch with [CLS] / [SEP] / [MASK]-like ids on hundreds of rows (the row list of a leader is walked by four
    # row lanes, eight rows in flight each, combined in a fixed order: same bits every time), more rows than one 8192-row bitmap pass, D = 1024 and a D
    # that needs a second round of columns
    for Rr, Lr, Dr, Vr in ((7680, 30, 768, 3000), (9000, 30, 128, 50), (640, 32, 1024, 40), (96, 8, 1536, 7)):
        g_ = torch.Generator().manual_seed(Rr)
        idr = torch.randint(0, Vr, (Rr // Lr, Lr), generator=g_)
        idr[:, 0] = 1; idr[:, -1] = 2; idr[torch.rand(Rr // Lr, Lr, generator=g_) < 0.3] = 3
        gr = torch.randn(Rr, Dr, generator=g_)
        outs = []
        for _ in range(2):
            dw, dp, dt = torch.zeros(Vr, Dr, device=dev), torch.zeros(Lr, Dr, device=dev), torch.zeros(2, Dr, device=dev)
            K.embed_bwd(idr.to(dev), gr.to(dev), dw, dp, dt)
            outs.append((dw, dp, dt))
        assert relerr(outs[0][0], torch.zeros(Vr, Dr).index_add_(0, idr.reshape(-1), gr)) < 2e-6
        assert relerr(outs[0][1], gr.view(-1, Lr, Dr).sum(0)) < 2e-6 and relerr(outs[0][2][0], gr.sum(0)) < 1e-5
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs[0], outs[1]))
    src = rnd(6, 40, seed=6); idx = torch.tensor([3, 3, 0, 5], dtype=torch.int32)
    d32, d16 = K.gather_rows(src.to(dev), idx.to(dev), 40, want_bf16=True)
    assert torch.equal(d32.cpu(), src[idx.long()]) and torch.equal(d16.cpu(), bf(src[idx.long()]))
    acc = K.scatter_rows(d32, idx.to(dev), 6, 40)                # every row written: rows nothing points at are zeros, not what was there
    assert relerr(acc, torch.zeros(6, 40).index_add_(0, idx.long(), src[idx.long()])) < 1e-6
    assert float(acc[[1, 2, 4]].abs().max()) == 0.0
    # the two shapes of the step: 256 sequences of 30 x 768 onto 128 (several per row, some rows none), 768 masked rows onto 1920
    for R, Dn, ln in ((256, 128, 30 * 768), (768, 1920, 768), (5000, 37, 8)):
        g_ = torch.Generator().manual_seed(R)
        ix = torch.randint(0, Dn, (R,), generator=g_).to(torch.int32)
        sr = torch.randn(R, ln, generator=g_)
        got = K.scatter_rows(sr.to(dev), ix.to(dev), Dn, ln)
        assert relerr(got, torch.zeros(Dn, ln).index_add_(0, ix.long(), sr)) < 1e-6
        assert torch.equal(got, K.scatter_rows(sr.to(dev), ix.to(dev), Dn, ln))      # ascending-r sums: the same bits every time
    A, Bm, bias = rnd(70, 50, seed=7), rnd(30, 50, seed=8), rnd(30, seed=9)
    assert relerr(K.linear_f32(A.to(dev), Bm.to(dev), bias=bias.to(dev)), A @ Bm.t() + bias) < 1e-5
    assert relerr(K.linear_f32(A.t().contiguous().to(dev), Bm.t().contiguous().to(dev), transA=True, transB=True, alpha=0.5),
                  0.5 * A @ Bm.t()) < 1e-5
    x = rnd(9, 32, seed=10); dy = rnd(9, 32, seed=11)
    xl = x.clone().requires_grad_(True)
    y = torch.nn.functional.normalize(xl, dim=-1); y.backward(dy)
    assert relerr(K.l2norm(x.to(dev)), y) < 1e-6 and relerr(K.l2norm(x.to(dev), dy.to(dev)), xl.grad) < 1e-5


def test_cross_entropy_and_sampling(K):
    R, Cv, ld = 20, 1000, 1024
    z = rnd(R, ld, seed=1, scale=3.0)
    lab = torch.randint(0, Cv, (R,), generator=torch.Generator().manual_seed(2)); lab[3] = -100; lab[7] = -100
    zl = z[:, :Cv].clone().requires_grad_(True)
    loss = O.cross_entropy(zl, lab); loss.backward()
    stat, lse = K.ce_fwd(z.to(dev), lab.to(dev), C_valid=Cv)
    assert abs(float(stat[0]) - float(loss)) < 1e-5 * float(loss) and float(stat[1]) == R - 2
    g = torch.tensor([0.7], device=dev)
    dl = K.ce_bwd(z.to(dev), lab.to(dev), lse, g, stat, C_valid=Cv)
    assert relerr(dl[:, :Cv], 0.7 * zl.grad) < 1e-5 and float(dl[:, Cv:].abs().max()) == 0
    n = 64
    sim = rnd(n, n, seed=3, scale=2.0)
    u = torch.rand(n, generator=torch.Generator().manual_seed(4))
    pick = K.sample_negatives(sim.to(dev), u.to(dev)).cpu().long()
    w = torch.softmax(sim, 1) + 1e-5; w.fill_diagonal_(0)
    cdf = torch.cumsum(w.double(), 1)
    refpick = (cdf > (u.double() * cdf[:, -1]).unsqueeze(1)).float().argmax(1)
    assert (pick != torch.arange(n)).all()
    # inverse-CDF draw in fp32 on the device vs float64 here: a different pick is only acceptable where the target u * total
    # sits within fp32 rounding of the CDF step that separates the two candidates
    target = u.double() * cdf[:, -1]
    for r in (pick != refpick).nonzero().flatten().tolist():
        lo = min(int(pick[r]), int(refpick[r]))
        assert abs(int(pick[r]) - int(refpick[r])) == 1 or w[r, lo + 1:max(int(pick[r]), int(refpick[r]))].sum() == 0, (r, pick[r], refpick[r])
        assert abs(float(cdf[r, lo] - target[r])) <= 1e-5 * float(cdf[r, -1]), (r, float(cdf[r, lo]), float(target[r]))
    x = rnd(1000, seed=5); dy = rnd(1000, seed=6)
    xl = x.clone().requires_grad_(True); O.gelu(xl).backward(dy)
    assert relerr(K.gelu_f32(x.to(dev)), O.gelu(x)) < 1e-6 and relerr(K.gelu_f32(x.to(dev), dy.to(dev)), xl.grad) < 1e-5


@pytest.mark.parametrize("S,Bi", [(256, 64), (11, 2), (40, 300), (2048, 1000), (5, 5)])
def test_kv_csr_and_additive_mask(K, S, Bi):
    """x2_kv_csr == stable argsort + bincount + cumsum (the torch ops it replaces); x2_additive_mask == (1 - m) * neg padded."""
    g = torch.Generator().manual_seed(S * 31 + Bi)
    kv = torch.randint(0, Bi, (S,), generator=g).to(torch.int32)
    off, order = K.kv_csr(kv.to(dev), Bi)
    want_order = torch.argsort(kv, stable=True).to(torch.int32)
    want_off = torch.zeros(Bi + 1, dtype=torch.int32)
    want_off[1:] = torch.cumsum(torch.bincount(kv, minlength=Bi), 0)
    assert torch.equal(off.cpu(), want_off) and torch.equal(order.cpu(), want_order)
    L = 30 if S % 2 == 0 else 197
    atts = (torch.rand(S, L, generator=g) > 0.3).long()
    for neg in (-10000.0, -1e9):
        m = K.additive_mask(atts.to(dev), neg)
        assert m.shape == (S, K.round_up(L, 64))
        assert torch.equal(m[:, :L].cpu(), (1.0 - atts.float()) * neg) and float(m[:, L:].abs().max()) == 0.0


def test_tail_index_tables(K):
    """x2_tail_index: row tables of the 4B-row fusion batch == the torch cat / index expressions they replace."""
    g = torch.Generator().manual_seed(5)
    B, L, T = 7, 9, 13
    ineg = torch.randint(0, B, (B,), generator=g, dtype=torch.int32)
    tneg = torch.randint(0, B, (B,), generator=g, dtype=torch.int32)
    ta = (torch.rand(B, L, generator=g) > 0.3).long()
    ia = (torch.rand(B, T, generator=g) > 0.2).long()
    ar = torch.arange(B, dtype=torch.int32)
    for with_match in (True, False):
        t_idx, kv, atts, enc = K.tail_index(ineg.to(dev) if with_match else None, tneg.to(dev) if with_match else None, ta.to(dev), ia.to(dev),
                                            with_match=with_match)
        ti = torch.cat([ar, ar, tneg, ar + B]) if with_match else ar + B
        ki = torch.cat([ar, ineg, ar, ar]) if with_match else ar
        assert torch.equal(t_idx.cpu(), ti) and torch.equal(kv.cpu(), ki)
        assert torch.equal(atts.cpu(), torch.cat([ta, ta])[ti.long()]) and torch.equal(enc.cpu(), ia[ki.long()])


def test_droppath_rows_match_host_mirror_and_rates(K):
    """x2_droppath_rows: keeps are the host mirror's (hash of (seed, epoch, block, branch, sample)), constant over a sample's T rows,
    scaled by 1 / (1 - rate); rate 0 keeps everything; the empirical drop frequency follows the rate; a new epoch redraws."""
    rates = [0.0, 0.05, 0.1, 0.5]
    r = torch.tensor(rates, device=dev)
    B, T = 6, 5
    old = K.DROP_EPOCH
    try:
        K.DROP_EPOCH = None
        rows = K.droppath_rows(r, 1234, B, T).cpu().view(len(rates), 2, B, T)
        keep = K.droppath_keep(rates, 1234, B)
        for l, rate in enumerate(rates):
            assert torch.equal(rows[l], (keep[l] / (1.0 - torch.tensor(rate, dtype=torch.float32))).unsqueeze(-1).expand(2, B, T).contiguous()), l
        assert float(rows[0].min()) == 1.0
        K.DROP_EPOCH = torch.tensor([7], dtype=torch.int32, device=dev)
        rows7 = K.droppath_rows(r, 1234, B, T).cpu().view(len(rates), 2, B, T)
        assert torch.equal((rows7[..., 0] > 0).float(), K.droppath_keep(rates, 1234, B, epoch=7))
        K.DROP_EPOCH = None
        big = K.droppath_rows(r, 99, 4096, 1).cpu().view(len(rates), 2, 4096)
        for l, rate in enumerate(rates):
            frac = float((big[l] == 0).float().mean())
            assert abs(frac - rate) < 0.03, (l, frac)
    finally:
        K.DROP_EPOCH = old


def test_frame_mean_forward_backward(K):
    """x2_frame_mean (video path, xvlm.py:627-645) against autograd of the tensor expression."""
    ops = importlib.import_module("x2-vlm_amd.ops")
    Bc, F, T, D = 3, 4, 5, 64
    x = rnd(Bc * F, T, D, seed=1).requires_grad_(True)
    pos = rnd(1, F, 1, D, seed=2).requires_grad_(True)
    dy = rnd(Bc, T, D, seed=3)
    ref = (x.view(Bc, F, T, D) + pos.view(1, F, 1, D)).mean(1)
    ref.backward(dy)
    xd, pd = x.detach().to(dev).requires_grad_(True), pos.detach().to(dev).requires_grad_(True)
    out = ops.frame_mean(xd, pd, F)
    out.backward(dy.to(dev))
    assert relerr(out, ref.detach()) < 1e-6 and relerr(xd.grad, x.grad) < 1e-6 and relerr(pd.grad, pos.grad) < 1e-5
