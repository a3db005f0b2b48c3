"""Tensor-level wrappers over the C ABI (one Python function per entry point of
include/x2vlm_hip.h).  They only check shapes/dtypes, allocate outputs with torch (device memory
is PyTorch's job) and launch on the current HIP stream.  No arithmetic happens in Python.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import AttnArgs, call, ptr, raw_stream

BF16, F32 = torch.bfloat16, torch.float32


GEMM_TIMER = None   # bench.py: list collecting (start_event, end_event, flops) per GEMM launch


class _timed:
    """HIP events around one launch on the current stream (only while bench.py's GEMM_TIMER is set)."""

    def __init__(self, flops, name="gemm_nt"):
        self.flops = flops
        self.name = name

    def __enter__(self):
        if GEMM_TIMER is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if GEMM_TIMER is not None:
            self.b.record()
            GEMM_TIMER.append((self.a, self.b, self.flops, self.name))


NO_DROP = (0, 0, 1.0, None)

# int32 [1] device tensor or None.  When set (graph.GraphedStep does, for a hipGraph-captured step whose kernel arguments are
# frozen at capture time), every dropout site mixes this device-resident step counter into its seed, so each replay of
# the graph draws new masks; None = eager launches, seeds drawn on the host per call (xbert.next_dropout_seed).
DROP_EPOCH = None


def dropout_spec(p, seed, site):
    """(thr16, site seed, scale, epoch tensor or None) for the kernels' counter-based dropout: element e is dropped iff
    the 16-bit uniform hashed from (e, site seed [, epoch]) is < thr16 = round(p * 65536); survivors are scaled by 1/(1-p)."""
    if p <= 0.0:
        return NO_DROP
    return (int(round(p * 65536.0)), _hash32_int((int(seed) * 0x9E3779B1 + int(site) * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF),
            1.0 / (1.0 - p), DROP_EPOCH)


def _hash32_int(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_keep(spec, index):
    """Host mirror of csrc/x2_common.h drop_mul(): multiplier (0 or scale) for int64 element indices."""
    thr, seed, scale = spec[:3]
    if len(spec) > 3 and spec[3] is not None:            # csrc/x2_common.h drop_at_epoch()
        seed = _hash32_int((seed + 0x9E3779B1 * (int(spec[3].item()) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    e = index.to(torch.int64) & 0xFFFFFFFF
    x = ((e >> 1) ^ seed) & 0xFFFFFFFF
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & 0xFFFFFFFF
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    u = torch.where((e & 1) == 1, x >> 16, x & 0xFFFF)
    return (u >= thr).to(torch.float32) * scale


_WS = {}
_WS_RETIRED = []
_WS_CAPTURED = set()        # data_ptr of every workspace a stream capture has been handed


def workspace(device, nfloats):
    """Scratch for the two-stage column reductions: one growing fp32 buffer per (device, stream); kernels on one
    stream use it strictly in order (stage 1 writes, stage 2 reads, next kernel overwrites).
    Growth is geometric (a run whose shapes creep upwards - variable batch / caption length, several configurations in one
    process - re-allocates O(log) times, not once per new size), and an outgrown buffer is kept alive only if a stream
    capture has seen it (a captured hipGraph has its address baked in); otherwise the caching allocator gets it back."""
    key = (device, raw_stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None and buf.data_ptr() in _WS_CAPTURED:
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(nfloats, 1 << 22, 0 if buf is None else 2 * buf.numel()), device=device, dtype=F32)
        _WS[key] = buf
    if buf.is_cuda and torch.cuda.is_current_stream_capturing():
        _WS_CAPTURED.add(buf.data_ptr())
    return buf


DEFERRED = None     # engine sets this to a list while a layer's backward runs: stage-2 reductions of the parameter
                    # gradients are collected here and launched later on the weight-gradient side stream


def _ws_and_defer(device, nfloats):
    """(workspace, defer flag): a private buffer when the reduction is deferred (the shared per-stream workspace
    would be overwritten by the next kernel before the side stream reads it)."""
    if DEFERRED is not None:
        return torch.empty(nfloats, device=device, dtype=F32), 1
    return workspace(device, nfloats), 0


def reduce_partials(ws, nblk, nk, width, outs):
    o = list(outs) + [None] * (3 - len(outs))
    call("x2_reduce_partials", ptr(ws), nblk, nk, width, ptr(o[0]), ptr(o[1]), ptr(o[2]))


def reduce_partials_multi(items):
    """Several stage-2 reductions (tuples as collected in DEFERRED; up to 4 outputs each) in one launch."""
    flat = []
    for ws, nblk, nk, width, outs in items:
        o = list(outs) + [None] * (4 - len(outs))
        flat += [ws.data_ptr(), nblk, nk, width] + [0 if t is None else t.data_ptr() for t in o]
    call("x2_reduce_partials_multi", (C.c_int64 * len(flat))(*flat), len(items))


def cast_transpose_multi(desc):
    """desc: tuples (src_ptr, dst_ptr, dstT_ptr, R, C, ldt, row_offset[, tscale_ptr or 0]), see csrc/rowwise.hip."""
    flat = [v for d in desc for v in (tuple(d) + (0,) if len(d) == 7 else d)]
    call("x2_cast_transpose_multi", (C.c_int64 * len(flat))(*flat), len(desc))


def copy_f32_multi(desc):
    """desc: tuples (src_ptr or 0 for zeros, dst_ptr, n)."""
    flat = [v for d in desc for v in d]
    call("x2_copy_f32_multi", (C.c_int64 * len(flat))(*flat), len(desc))


def _rows(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor (got strides %s)" % (t.stride(),)
    return t.stride(0)


def round_up(x, m):
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- GEMMs

def gemm_nt(A, B, *, bias=None, gamma=None, resid=None, aux=None, act=0, out=None, out_dtype=BF16, drop=NO_DROP,
            rowscale=None, colsum=None):
    """out[M,N] = epilogue(A[M,K] @ B[N,K]^T).  act: 0 none (aux, if given, receives acc+bias),
    1 GELU (aux receives the pre-activation), 2 multiply by GELU'(aux).  Then dropout(drop), *gamma,
    *rowscale[m], +resid.  colsum (fp32 [N], accumulated): column sums of the stored result."""
    assert A.dtype == BF16 and B.dtype == BF16 and A.shape[1] == B.shape[1]
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=out_dtype)
    assert out.shape == (M, N) and out.dtype in (BF16, F32)
    for t in (bias, gamma):
        assert t is None or (t.dtype == F32 and t.numel() == N and t.is_contiguous())
    assert resid is None or (resid.dtype == F32 and resid.shape == (M, N))
    assert aux is None or (aux.dtype == BF16 and aux.shape == (M, N))
    assert rowscale is None or (rowscale.dtype == F32 and rowscale.numel() == M and rowscale.is_contiguous())
    with _timed(2.0 * M * N * K):
        call("x2_gemm_nt", ptr(A), ptr(B), ptr(out), M, N, K, _rows(A), _rows(B), _rows(out), ptr(bias), ptr(gamma),
             ptr(resid), _rows(resid) if resid is not None else 0, ptr(aux), _rows(aux) if aux is not None else 0,
             act, 1 if out.dtype == F32 else 0, drop[0], drop[1], drop[2], ptr(drop[3]), ptr(rowscale), ptr(colsum))
    return out


def gemm_nt_dgelu_colsum(A, B, pre, colsum):
    """out[M,N] bf16 = (A @ B^T) * GELU'(pre) and colsum[N] += column sums of out: the GELU input gradient of an MLP and
    the bias gradient of its first linear from ONE kernel (epilogue variant 10: one partial row per wave row, reduced with the
    layer's other deferred reductions or right here)."""
    assert A.dtype == BF16 and B.dtype == BF16 and A.shape[1] == B.shape[1] and pre.dtype == BF16
    M, K = A.shape
    N = B.shape[0]
    assert pre.shape == (M, N) and colsum.dtype == F32 and colsum.numel() == N
    out = torch.empty(M, N, device=A.device, dtype=BF16)
    ws, defer = _ws_and_defer(A.device, 2 * ((M + 63) // 64) * N)
    nrows = C.c_int(0)
    with _timed(2.0 * M * N * K):
        call("x2_gemm_nt_dgelu_colparts", ptr(A), ptr(B), ptr(out), M, N, K, _rows(A), _rows(B), _rows(out), ptr(pre), _rows(pre),
             ptr(ws), C.byref(nrows))
    if defer:
        DEFERRED.append((ws, nrows.value, 1, N, (colsum,)))
    else:
        reduce_partials(ws, nrows.value, 1, N, (colsum,))
    return out


def gemm_nt_splitk(A, B, out=None, slices=0):
    """out[M,N] fp32 = A[M,K] @ B[N,K]^T with the contraction cut into slices (0 = the library fills the CUs): few output
    tiles, long K (input gradient of the tied MLM decoder).  Partials pass through the stream workspace and are added in
    a fixed order."""
    assert A.dtype == BF16 and B.dtype == BF16 and A.shape[1] == B.shape[1]
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=F32)
    assert out.shape == (M, N) and out.dtype == F32
    ws = workspace(A.device, min(32, max(1, K // 64)) * M * N)       # room for up to 32 slices
    with _timed(2.0 * M * N * K):
        call("x2_gemm_nt_splitk", ptr(A), ptr(B), ptr(out), M, N, K, _rows(A), _rows(B), _rows(out), slices, ptr(ws), ws.numel())
    return out


def gemm_tn_grouped(problems, accumulate=False, split=0):
    """Weight gradients of one layer in one launch.  problems: list of (dY[Mc,N] bf16, X[Mc,K] bf16,
    dW[N,K] fp32) or 5-tuples with (n_ld, k_ld) = readable row widths when they exceed N / K.
    split: slices of the contraction (0 = the library picks what fills the CUs; partial tiles go through the stream
    workspace and are added deterministically; small launches use the 128x128 kernel, where split > 1 adds atomically and
    needs accumulate=True)."""
    rows = []
    for pr in problems:
        dY, X, dW = pr[:3]
        assert dY.dtype == BF16 and X.dtype == BF16 and dW.dtype == F32 and dY.shape[0] == X.shape[0]
        N, K = dW.shape
        n_ld, k_ld = (pr[3], pr[4]) if len(pr) == 5 else (dY.shape[1], X.shape[1])
        assert dY.shape[1] >= N or n_ld >= N
        rows.append([dY.data_ptr(), X.data_ptr(), dW.data_ptr(), dY.shape[0], N, K, _rows(dY), _rows(X), _rows(dW),
                     n_ld, k_ld])
    dev = problems[0][0].device
    for i in range(0, len(rows), 8):
        chunk = rows[i:i + 8]
        arr = (C.c_int64 * (11 * len(chunk)))(*[v for r in chunk for v in r])
        tiles256 = sum(((r[4] + 255) // 256) * ((r[5] + 255) // 256) for r in chunk)
        want = (split if split else 4) * tiles256 * 65536
        ws = workspace(dev, want) if split != 1 else None
        with _timed(sum(2.0 * r[3] * r[4] * r[5] for r in chunk), "gemm_tn"):
            call("x2_gemm_tn_grouped", arr, len(chunk), 1 if accumulate else 0, split, ptr(ws), 0 if ws is None else ws.numel())


# ----------------------------------------------------------------------------- attention

def _attn_args(q, k, v, B, Bkv, H, Lq, Lk, scale, bias=None, biasT=None, mask=None, kv_idx=None, seq_off=None,
               seq_ids=None, drop=NO_DROP, head_dim=64, bias_log2=False):
    """q/k/v: (tensor, batch_stride, row_stride) views in elements; data pointer already at head 0."""
    a = AttnArgs()
    a.Q, a.q_bs, a.q_rs = q
    a.K, a.k_bs, a.k_rs = k
    a.V, a.v_bs, a.v_rs = v
    a.B, a.Bkv, a.H, a.Lq, a.Lk, a.scale = B, Bkv, H, Lq, Lk, scale
    a.head_dim = head_dim
    a.dbg = 16 if bias_log2 else 0
    a.drop_thr16, a.drop_seed, a.drop_scale = drop[:3]
    a.drop_epoch = ptr(drop[3])
    if bias is not None:
        assert bias.dtype == F32 and bias.dim() == 3 and bias.is_contiguous()
        a.bias, a.bias_ld = bias.data_ptr(), bias.shape[2]
    if biasT is not None:
        assert biasT.dtype == F32 and biasT.dim() == 3 and biasT.is_contiguous()
        a.biasT, a.biasT_ld = biasT.data_ptr(), biasT.shape[2]
    if mask is not None:
        assert mask.dtype == F32 and mask.dim() == 2 and mask.is_contiguous() and mask.shape[0] == B
        a.mask, a.mask_ld = mask.data_ptr(), mask.shape[1]
    if kv_idx is not None:
        assert kv_idx.dtype == torch.int32 and kv_idx.numel() == B
        a.kv_idx = kv_idx.data_ptr()
    if seq_off is not None:
        assert seq_off.dtype == torch.int32 and seq_ids.dtype == torch.int32
        a.seq_off, a.seq_ids = seq_off.data_ptr(), seq_ids.data_ptr()
    return a


def view3(t, B, L, col0=0):
    """(ptr, batch stride, row stride) of a [B*L, W] row-major bf16 buffer read from column col0."""
    assert t.dtype == BF16 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == B * L
    return (t.data_ptr() + 2 * col0, L * t.stride(0), t.stride(0))


def attn_fwd(q, k, v, B, Bkv, H, Lq, Lk, scale, out, lse, **kw):
    a = _attn_args(q, k, v, B, Bkv, H, Lq, Lk, scale, **kw)
    a.Out, a.o_bs, a.o_rs = out
    assert lse.dtype == F32 and lse.numel() == B * H * Lq
    a.LSE = lse.data_ptr()
    call("x2_attn_fwd", C.byref(a))


def attn_bwd_form(Lq, Lk, shared_kv=False, dropout=False, workspace=False):
    """Which backward x2_attn_bwd runs for this geometry (x2_attn_bwd_one_pass on an argument block with placeholder pointers - the
    library reads sizes, strides and which pointers are null): 0 two kernels, 1 / 2 / 3 one pass.  For callers that decide BEFORE the
    backward whether a second stream has anything to do (engine.BertLayersFn.forward: the K/V projections ahead of the layers)."""
    a = AttnArgs()
    a.B = a.Bkv = a.H = 1
    a.Lq, a.Lk, a.head_dim, a.scale = Lq, Lk, 64, 0.125
    for f in ("q_bs", "q_rs", "k_bs", "k_rs", "v_bs", "v_rs", "o_bs", "o_rs", "dq_bs", "dq_rs", "dk_bs", "dk_rs", "dv_bs", "dv_rs", "do_bs", "do_rs"):
        setattr(a, f, 64)
    if shared_kv:
        a.kv_idx = a.seq_off = a.seq_ids = 8        # non-null, never dereferenced by the query
    a.drop_thr16 = 1 if dropout else 0
    if workspace:                                  # as attn_bwd hands one over for long self-attention (form 3)
        a.ws, a.ws_floats = 8, ((Lq + 127) // 128) * 8192
    return _lib.lib().x2_attn_bwd_one_pass(C.byref(a))


def attn_bwd(q, k, v, o, do, B, Bkv, H, Lq, Lk, scale, lse, delta, dq, dk, dv, dS=None, phase=0, ask_form=False, colsum_ws=None, **kw):
    """phase: 0 both halves; 1 the dQ half (+ dS, delta); 2 the dK / dV half (needs the delta of a phase-1 call).
    ask_form: launch nothing, return which backward a phase-0 call with these arguments runs (x2_attn_bwd_one_pass: 0 = the two
    kernels, 1 / 2 = one pass) - a caller that would put the dK / dV half on a second stream has nothing to put there in one pass."""
    a = _attn_args(q, k, v, B, Bkv, H, Lq, Lk, scale, **kw)
    a.phase = phase
    a.O, a.o_bs, a.o_rs = o
    a.dO, a.do_bs, a.do_rs = do
    a.dQ, a.dq_bs, a.dq_rs = dq
    a.dK, a.dk_bs, a.dk_rs = dk
    a.dV, a.dv_bs, a.dv_rs = dv
    a.LSE, a.Delta = lse.data_ptr(), delta.data_ptr()
    if dS is not None:
        assert dS.dtype == BF16 and dS.is_contiguous() and dS.shape[:3] == (B, H, Lq)
        a.dS, a.ds_ld = dS.data_ptr(), dS.shape[3]
    if phase == 0 and Lq > 208 and Lk > 208 and a.kv_idx is None and not a.drop_thr16:
        # long self-attention (X2VLM-large, N = 577): the one-pass kernel carries the dQ of a query strip from key part to key part as an
        # fp32 partial in this scratch (include/x2vlm_hip.h, X2AttnArgs.ws)
        ws = workspace(lse.device, B * H * ((Lq + 127) // 128) * 8192)
        a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    if ask_form:
        return _lib.lib().x2_attn_bwd_one_pass(C.byref(a))
    if colsum_ws is not None:
        # fp32 [B, 2, H * 64]: the one-pass kernels of forms 1 / 3 leave per-sequence column sums of the stored dQ / dV rows there (the q / v bias gradient of a
        # fused qkv projection after a sum over B); returns the form that ran - any other form has NOT written the buffer
        assert colsum_ws.dtype == F32 and colsum_ws.is_contiguous() and colsum_ws.shape == (B, 2, H * 64)
        form = _lib.lib().x2_attn_bwd_one_pass(C.byref(a))
        if form in (1, 3):
            a.colsum_ws = colsum_ws.data_ptr()
        call("x2_attn_bwd", C.byref(a))
        return form
    call("x2_attn_bwd", C.byref(a))


# ----------------------------------------------------------------------------- row-wise

def layernorm_fwd(x, w, b, eps, *, rows=None, period=0, want_bf16=True, want_f32=False, y_bf16=None, y_f32=None,
                  drop=NO_DROP):
    """x fp32 [R_total, D]; with period>0 only rows r + r//period + 1 (token 0 of each sample skipped)."""
    assert x.dtype == F32 and x.is_contiguous()
    D = x.shape[-1]
    x2 = x.view(-1, D)
    R = x2.shape[0] if rows is None else rows
    if want_bf16 and y_bf16 is None:
        y_bf16 = torch.empty_like(x2, dtype=BF16)
    if want_f32 and y_f32 is None:
        y_f32 = torch.empty_like(x2)
    mean = torch.empty(R, device=x.device, dtype=F32)
    rstd = torch.empty(R, device=x.device, dtype=F32)
    call("x2_layernorm_fwd", ptr(x2), ptr(w), ptr(b), ptr(y_bf16), ptr(y_f32), ptr(mean), ptr(rstd), R, D, eps, period,
         drop[0], drop[1], drop[2], ptr(drop[3]))
    return y_bf16, y_f32, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, w, dw, db, *, dres=None, dcol=None, period=0, want_f32=True, want_bf16=False, dx=None,
                  drop_in=NO_DROP, drop_out=NO_DROP, post=None):
    """dcol (fp32 [D], accumulated): column sums of the LN-input gradient = bias gradient of the producing linear.
    post = (rowscale or None, colsum fp32 [D]): the bf16 copy is rowscale[m] * (dx + dres) and `colsum` receives its column sums -
    the two things the layer scale BELOW this LayerNorm needs of its incoming gradient (layerscale_finish)."""
    assert dy.dtype in (F32, BF16) and x.dtype == F32 and dy.is_contiguous() and x.is_contiguous()
    D = x.shape[-1]
    R = mean.numel()
    if want_f32 and dx is None:
        dx = torch.empty_like(x) if period == 0 else torch.zeros_like(x)
    prs = None
    if post is not None:
        assert dcol is None and dy.dtype == BF16 and period == 0
        prs, dcol = post
        assert prs is None or (prs.dtype == F32 and prs.numel() == R and prs.is_contiguous())
        want_bf16 = True
    dxb = torch.empty(x.shape, device=x.device, dtype=BF16) if want_bf16 else None
    nblk = (R + 15) // 16
    ws, defer = _ws_and_defer(x.device, nblk * 3 * D)
    call("x2_layernorm_bwd", ptr(dy), 1 if dy.dtype == BF16 else 0, ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(dres), ptr(dx), ptr(dxb), ptr(dw), ptr(db),
         ptr(dcol), R, D, period, drop_in[0], drop_in[1], drop_in[2], drop_out[0], drop_out[1], drop_out[2],
         ptr(drop_in[3] if drop_in[3] is not None else drop_out[3]), ptr(ws), defer, 0 if post is None else 1, ptr(prs))
    if defer:
        DEFERRED.append((ws, nblk, 3, D, (dw, db, dcol)))
    return dx, dxb


def colsum_bf16(y, out):
    nblk = (y.shape[0] + 63) // 64
    ws, defer = _ws_and_defer(y.device, nblk * y.shape[1])
    call("x2_colsum_bf16", ptr(y), ptr(out), y.shape[0], y.shape[1], _rows(y), ptr(ws), defer)
    if defer:
        DEFERRED.append((ws, nblk, 1, y.shape[1], (out,)))


def rowscale_cast_colsum(dx, colsum, rowscale=None):
    """bf16(rowscale[m] * dx) and colsum += its column sums: what a layer scale's backward needs of its incoming gradient when no
    LayerNorm backward of the same stage produced it (layernorm_bwd(post=...) otherwise)."""
    assert dx.dtype == F32 and dx.is_contiguous() and dx.dim() == 2
    dxb = torch.empty_like(dx, dtype=BF16)
    nblk = (dx.shape[0] + 31) // 32
    ws, defer = _ws_and_defer(dx.device, nblk * dx.shape[1])
    call("x2_rowscale_cast_colsum", ptr(dx), ptr(rowscale), ptr(dxb), ptr(colsum), dx.shape[0], dx.shape[1], ptr(ws), defer)
    if defer:
        DEFERRED.append((ws, nblk, 1, dx.shape[1], (colsum,)))
    return dxb


def layerscale_finish(items):
    """items: (G [N,K] fp32 = dX'^T . A as the weight-gradient GEMM left it, W [N,K] fp32, bias or None, gamma, cs, dgamma, dbias or
    None).  dgamma += rowdot(G, W) + bias * cs, dbias += gamma * cs, G <- diag(gamma) G; one launch."""
    flat = []
    for G, W, bias, gamma, cs, dgamma, dbias in items:
        N, Kd = G.shape
        assert G.dtype == F32 and W.dtype == F32 and G.is_contiguous() and W.is_contiguous() and W.numel() == G.numel() and Kd % 4 == 0
        flat += [G.data_ptr(), W.data_ptr(), 0 if bias is None else bias.data_ptr(), gamma.data_ptr(), cs.data_ptr(), dgamma.data_ptr(),
                 0 if dbias is None else dbias.data_ptr(), N, Kd]
    call("x2_layerscale_finish", (C.c_int64 * len(flat))(*flat), len(items))


def cast_bf16(src, out=None):
    assert src.dtype == F32 and src.is_contiguous() and src.numel() % 4 == 0
    if out is None:
        out = torch.empty_like(src, dtype=BF16)
    call("x2_cast_bf16", ptr(src), ptr(out), src.numel())
    return out


def cast_transpose_bf16(src, want_plain=True, ldt=None):
    """fp32 [R,C] -> (bf16 [R,C] or None, bf16 [C, ldt]) with ldt >= R (pad columns zero)."""
    assert src.dtype == F32 and src.dim() == 2 and src.is_contiguous()
    R, Cc = src.shape
    ldt = R if ldt is None else ldt
    plain = torch.empty(R, Cc, device=src.device, dtype=BF16) if want_plain else None
    tr = (torch.zeros if ldt > R else torch.empty)(Cc, ldt, device=src.device, dtype=BF16)
    call("x2_cast_transpose_bf16", ptr(src), ptr(plain), ptr(tr), R, Cc, ldt)
    return plain, tr


def patchify(image, ps):
    assert image.dtype == F32 and image.is_contiguous() and image.dim() == 4 and image.shape[1] == 3
    B, _, R, _ = image.shape
    g = R // ps
    cols = torch.empty(B * g * g, 3 * ps * ps, device=image.device, dtype=BF16)
    call("x2_patchify", ptr(image), ptr(cols), B, R, ps)
    return cols


def assemble_tokens(patch, cls, B, P):
    D = patch.shape[1]
    x = torch.empty(B, P + 1, D, device=patch.device, dtype=F32)
    call("x2_assemble_tokens", ptr(patch), ptr(cls), ptr(x), B, P, D)
    return x


def assemble_tokens_bwd(dx, dcls):
    B, T, D = dx.shape
    dpatch = torch.empty(B * (T - 1), D, device=dx.device, dtype=BF16)
    call("x2_assemble_tokens_bwd", ptr(dx), ptr(dpatch), ptr(dcls), B, T - 1, D)
    return dpatch


def pool_tokens(x, w=None, bwd=False):
    """In place on x fp32 [B, 1+P, D]: fwd writes token 0 = (weighted) mean of patch tokens; bwd spreads the
    token-0 gradient onto the patch rows and zeroes it."""
    B, T, D = x.shape
    assert x.dtype == F32 and x.is_contiguous() and (w is None or (w.dtype == F32 and w.shape == (B, T - 1) and w.is_contiguous()))
    call("x2_pool_tokens", ptr(x), ptr(w), B, T - 1, D, 1 if bwd else 0)
    return x


LOG2E = 1.4426950408889634


def relpos_bias(table, index, want_T=True, log2=False):
    """log2: the gathered bias times log2(e) - the unit the attention kernels' softmax works in; pass bias_log2=True to
    attn_fwd / attn_bwd with it (one fma per score instead of four VALU operations)."""
    N = index.shape[0]
    H = table.shape[1]
    ld = round_up(N, 64)
    # pad columns stay undefined: the attention kernels select on key < Lk / query < Lq before using a bias value
    bias = torch.empty(H, N, ld, device=table.device, dtype=F32)
    biasT = torch.empty(H, N, ld, device=table.device, dtype=F32) if want_T else None
    call("x2_relpos_bias", ptr(table), ptr(index), ptr(_relpos_index_t(index)) if want_T else None, ptr(bias), ptr(biasT), N, H, ld, ld,
         LOG2E if log2 else 1.0)
    return bias, biasT


_RELPOS_CSR = {}
_RELPOS_T = {}


def _relpos_index_t(index):
    """Transposed copy of the (static) relative_position_index, built once per buffer: lets x2_relpos_bias write the
    transposed bias row-contiguously."""
    key = (index.data_ptr(), index._version)
    ent = _RELPOS_T.get(key)
    if ent is None:
        ent = _RELPOS_T[key] = (index.t().contiguous(), index)      # keeps `index` alive: the key holds its address
    return ent[0]



def _relpos_csr(index, ld, T):
    """CSR inverse of the (static) relative_position_index: for table entry t the flat positions i * ld + j that
    read it.  Built once per (index tensor, ld) with torch ops on the device."""
    key = (index.data_ptr(), index._version, ld, T)
    ent = _RELPOS_CSR.get(key)
    if ent is None:
        N = index.shape[0]
        flat = index.reshape(-1)
        order = torch.argsort(flat, stable=True)
        off = torch.zeros(T + 1, device=index.device, dtype=torch.int32)
        off[1:] = torch.cumsum(torch.bincount(flat, minlength=T), 0)
        pos = ((order // N) * ld + order % N).to(torch.int32).contiguous()
        ent = _RELPOS_CSR[key] = (off, pos, index)       # keeps `index` alive: the key holds its address
    return ent[0], ent[1]


def relpos_bias_bwd(dS, index, dtable):
    """dtable[index[i][j]][h] += sum_b dS[b][h][i][j]: batch-slice sums into the stream workspace, then a gather."""
    B, H, N, ld = dS.shape
    T = dtable.shape[0]
    assert dS.dtype == BF16 and dS.is_contiguous() and dtable.dtype == F32 and dtable.is_contiguous() and dtable.shape[1] == H
    off, pos = _relpos_csr(index, ld, T)
    slices = 1      # probes/bench_relpos.py: 21 / 109 us (base / large) with one slice vs 51 / 217 us with 8 / 4: the gather reads S x
    ws = workspace(dS.device, slices * H * N * ld)
    call("x2_relpos_bias_bwd", ptr(dS), ptr(off), ptr(pos), ptr(dtable), B, N, H, ld, T, ptr(ws), slices)


# ----------------------------------------------------------------------------- embeddings, heads, losses

def embed_fwd(ids, word, pos, type_emb):
    R, L = ids.numel(), ids.shape[-1]
    D = word.shape[1]
    out = torch.empty(R, D, device=word.device, dtype=F32)
    call("x2_embed_fwd", ptr(ids), ptr(word), ptr(pos), ptr(type_emb), ptr(out), R, L, D)
    return out


def embed_bwd(ids, g, dword, dpos, dtype):
    R, L = ids.numel(), ids.shape[-1]
    D = g.shape[-1]
    scratch = workspace(g.device, min(L, R) * D)          # per-position totals on their way to the type-0 row (fixed-order sums, no atomics)
    call("x2_embed_bwd", ptr(ids), ptr(g), ptr(dword), ptr(dpos), ptr(dtype), R, L, D, ptr(scratch))


def gather_rows(src, idx, row_len, want_f32=True, want_bf16=False):
    R = idx.numel()
    assert src.dtype == F32 and src.is_contiguous() and idx.dtype == torch.int32
    dst = torch.empty(R, row_len, device=src.device, dtype=F32) if want_f32 else None
    dstb = torch.empty(R, row_len, device=src.device, dtype=BF16) if want_bf16 else None
    call("x2_gather_rows", ptr(src), ptr(idx), ptr(dst), ptr(dstb), R, row_len)
    return dst, dstb


def scatter_rows(src, idx, rows, row_len):
    """-> dst [rows, row_len] fp32: dst[d] = sum of src[r] over idx[r] == d (zeros where nothing points): the gather's backward."""
    assert src.dtype == F32 and src.is_contiguous() and idx.dtype == torch.int32 and src.numel() == idx.numel() * row_len
    dst = torch.empty(rows, row_len, device=src.device, dtype=F32)
    call("x2_scatter_rows", ptr(src), ptr(idx), ptr(dst), idx.numel(), rows, row_len)
    return dst


def linear_f32(A, B, *, bias=None, transA=False, transB=False, alpha=1.0, alpha_ptr=None, out=None, accumulate=False):
    """out[M,N] (+)= alpha * op(A) @ op(B)^T + bias, fp32.  op(A) is [M,K] (A is [K,M] if transA);
    op(B) is [N,K] (B is [K,N] if transB)."""
    assert A.dtype == F32 and B.dtype == F32 and A.dim() == 2 and B.dim() == 2
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    N = B.shape[1] if transB else B.shape[0]
    sam, sak = (A.stride(1), A.stride(0)) if transA else (A.stride(0), A.stride(1))
    sbn, sbk = (B.stride(1), B.stride(0)) if transB else (B.stride(0), B.stride(1))
    # head-sized outputs (a handful of 64x64 tiles) with a long contraction: slice K over gridDim.z; the partial tiles
    # pass through the stream workspace and are added in slice order (deterministic)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    ksplit = 1 if tiles >= 64 or K < 128 else max(1, min(16, K // 64, 256 // tiles))
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, device=A.device, dtype=F32)
    ws = workspace(A.device, ksplit * M * N) if ksplit > 1 else None
    call("x2_linear_f32", ptr(A), ptr(B), ptr(out), ptr(bias), ptr(alpha_ptr), alpha, M, N, K, sam, sak, sbn, sbk,
         out.stride(0), 1 if accumulate else 0, ksplit, ptr(ws))
    return out


def l2norm(x, dy=None):
    out = torch.empty_like(x)
    call("x2_l2norm", ptr(x), ptr(dy), ptr(out), x.shape[0], x.shape[1], 0 if dy is None else 1)
    return out


def ce_fwd(logits, labels, C_valid=None):
    """logits fp32 [R, ld]; returns (stat[2] = (mean loss, #valid rows), lse[R])."""
    R, ld = logits.shape
    Cv = ld if C_valid is None else C_valid
    lse = torch.empty(R, device=logits.device, dtype=F32)
    rows = torch.empty(R, device=logits.device, dtype=F32)
    stat = torch.empty(2, device=logits.device, dtype=F32)
    call("x2_ce_fwd", ptr(logits), ld, ptr(labels), R, Cv, ptr(lse), ptr(rows), ptr(stat))
    return stat, lse


def ce_bwd(logits, labels, lse, g, stat, C_valid=None, gscale=1.0, out_dtype=F32):
    R, ld = logits.shape
    Cv = ld if C_valid is None else C_valid
    dl = torch.empty(R, ld, device=logits.device, dtype=out_dtype)
    call("x2_ce_bwd", ptr(logits), ld, ptr(labels), ptr(lse), ptr(g), ptr(stat), gscale, R, Cv,
         ptr(dl) if out_dtype == F32 else None, ptr(dl) if out_dtype == BF16 else None, ld)
    return dl


def mlm_ce_fwd(x, E, bias, labels, V):
    """Cross-entropy of z = x @ E^T + bias over the first V columns without storing z (x [R,Hd] bf16, E [Vp,Hd] bf16,
    Vp % 64 == 0).  Returns (stat[2] = (mean loss, #counted rows), lse[R])."""
    assert x.dtype == BF16 and E.dtype == BF16 and x.shape[1] == E.shape[1] and bias.dtype == F32 and bias.numel() == E.shape[0]
    assert labels.dtype == torch.int64 and labels.numel() == x.shape[0]
    R, Hd = x.shape
    Vp = E.shape[0]
    dev = x.device
    part = torch.empty(R, Vp // 64, 2, device=dev, dtype=F32)
    small = torch.empty(3 * R + 2, device=dev, dtype=F32)
    zlab, lse, rows, stat = small[:R], small[R:2 * R], small[2 * R:3 * R], small[3 * R:]
    with _timed(2.0 * R * Vp * Hd):
        call("x2_mlm_ce_fwd", ptr(x), ptr(E), ptr(bias), ptr(labels), R, Vp, V, Hd, _rows(x), _rows(E), ptr(part), ptr(zlab))
    call("x2_ce_combine", ptr(part), Vp // 64, ptr(zlab), ptr(labels), R, ptr(lse), ptr(rows), ptr(stat))
    return stat, lse


def mlm_ce_bwd(x, E, bias, labels, lse, g, stat, V, gscale=1.0):
    """dlogits bf16 [R, Vp] of the loss above (logits recomputed in the GEMM, never stored)."""
    R, Hd = x.shape
    Vp = E.shape[0]
    dl = torch.empty(R, Vp, device=x.device, dtype=BF16)
    with _timed(2.0 * R * Vp * Hd, "gemm_nt_recompute"):        # not algorithmic work: kept out of bench.py's NT roofline sums
        call("x2_mlm_ce_bwd", ptr(x), ptr(E), ptr(bias), ptr(labels), ptr(lse), ptr(g), ptr(stat), gscale, R, Vp, V, Hd,
             _rows(x), _rows(E), ptr(dl), Vp)
    return dl


def mask_words(seed, epoch, batch, count):
    """Host mirror of csrc/masking.hip mk_word(): the 32-bit words caption b consumes when none are injected, [batch, count] int64 (CPU)."""
    seed = int(seed) & 0xFFFFFFFF
    if epoch is not None:
        seed = _hash32_int((seed + 0x9E3779B1 * (int(epoch) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    out = torch.empty(batch, count, dtype=torch.int64)
    for b in range(batch):
        hb = _hash32_int((seed + 0x9E3779B1 * (b + 1)) & 0xFFFFFFFF)
        for k in range(count):
            out[b, k] = _hash32_int(hb ^ ((0x85EBCA6B * (k + 1)) & 0xFFFFFFFF))
    return out


def mask_tokens(text_ids, text_atts, is_subword, *, words=None, seed=0, epoch=None, mask_prob=0.5, max_masks=12, skipgram_prb=0.2, skipgram_size=3,
                mask_whole_word=True, cls_id=101, mask_id=103, pad_id=0, pad_mask=-100, out=None):
    """MLM masking of a padded batch on the device (pretrain_dataset.py:59-130, 242-275): (text_ids_masked [B, L], masked_pos [B, max_masks],
    masked_ids [B, max_masks]), int64.  is_subword: uint8 [vocab], 1 = WordPiece continuation ('##...').  words: injected uint32 stream [B, W]
    stored as int32 / uint32 bits (parity tests) or None: hashed from (seed, *epoch, caption, draw) - epoch is a device uint32 word (DROP_EPOCH)."""
    assert text_ids.dtype == torch.int64 and text_atts.dtype == torch.int64 and text_ids.shape == text_atts.shape and text_ids.dim() == 2
    assert text_ids.is_contiguous() and text_atts.is_contiguous() and is_subword.dtype == torch.uint8 and is_subword.is_contiguous()
    B, L = text_ids.shape
    dev = text_ids.device
    if out is not None:                              # static tensors of a captured step
        idm, mpos, mids = out
        assert idm.shape == (B, L) and mpos.shape == (B, max_masks) and mids.shape == (B, max_masks)
        assert all(t.dtype == torch.int64 and t.is_contiguous() for t in out)
    else:
        idm = torch.empty(B, L, device=dev, dtype=torch.int64)
        mpos = torch.empty(B, max_masks, device=dev, dtype=torch.int64)
        mids = torch.empty(B, max_masks, device=dev, dtype=torch.int64)
    if words is not None:
        assert words.dtype in (torch.int32, torch.uint32) and words.dim() == 2 and words.shape[0] == B and words.is_contiguous()
    call("x2_mask_tokens", ptr(text_ids), ptr(text_atts), B, L, ptr(is_subword), is_subword.numel(), ptr(words), words.shape[1] if words is not None else 0,
         int(seed) & 0xFFFFFFFF, ptr(epoch), float(mask_prob), int(max_masks), float(skipgram_prb), int(skipgram_size), 1 if mask_whole_word else 0,
         int(cls_id), int(mask_id), int(pad_id), int(pad_mask), ptr(idm), ptr(mpos), ptr(mids))
    return idm, mpos, mids


def sample_negatives(sim, u, group=None):
    n = sim.shape[0]
    out = torch.empty(n, device=sim.device, dtype=torch.int32)
    call("x2_sample_negatives", ptr(sim), n, ptr(group), ptr(u), ptr(out))
    return out


def additive_mask(atts, neg):
    """[S, L] 0/1 attention mask (int64) -> [S, round_up(L, 64)] fp32 additive key mask (1 - m) * neg, pad columns 0."""
    assert atts.dtype == torch.int64 and atts.dim() == 2 and atts.is_contiguous()
    S, L = atts.shape
    out = torch.empty(S, round_up(L, 64), device=atts.device, dtype=F32)
    call("x2_additive_mask", ptr(atts), ptr(out), S, L, out.shape[1], float(neg))
    return out


def kv_csr(kv, Bi):
    """kv int32 [S] (K/V batch of every query sequence) -> (off int32 [Bi + 1], order int32 [S]): the sequences grouped by the
    K/V batch they use, ascending inside a group."""
    assert kv.dtype == torch.int32 and kv.dim() == 1 and kv.is_contiguous()
    S = kv.numel()
    off = torch.empty(Bi + 1, device=kv.device, dtype=torch.int32)
    order = torch.empty(S, device=kv.device, dtype=torch.int32)
    call("x2_kv_csr", ptr(kv), S, Bi, ptr(off), ptr(order))
    return off, order


def tail_index(ineg, tneg, text_atts, image_atts, with_match=True):
    """Row tables of the 4B-row fusion batch (x2_tail_index): -> (t_idx int32 [R], kv int32 [R], atts int64 [R, L], enc_atts
    int64 [R, T]), R = 4B (with_match) or B."""
    B, L = text_atts.shape
    T = image_atts.shape[1]
    assert text_atts.dtype == torch.int64 and image_atts.dtype == torch.int64 and text_atts.is_contiguous() and image_atts.is_contiguous()
    assert not with_match or (ineg.dtype == torch.int32 and tneg.dtype == torch.int32 and ineg.numel() == B and tneg.numel() == B)
    R = 4 * B if with_match else B
    dev = text_atts.device
    t_idx = torch.empty(R, device=dev, dtype=torch.int32)
    kv = torch.empty(R, device=dev, dtype=torch.int32)
    atts = torch.empty(R, L, device=dev, dtype=torch.int64)
    enc = torch.empty(R, T, device=dev, dtype=torch.int64)
    call("x2_tail_index", ptr(ineg) if with_match else None, ptr(tneg) if with_match else None, ptr(text_atts), ptr(image_atts), B, L, T,
         1 if with_match else 0, ptr(t_idx), ptr(kv), ptr(atts), ptr(enc))
    return t_idx, kv, atts, enc


def droppath_rows(rates, seed, B, T):
    """fp32 [depth, 2, B * T] per-row keep / (1 - rate) factors of every block's two residual branches (x2_droppath_rows); the
    uniforms are hashed from (seed, DROP_EPOCH) like the dropout masks."""
    depth = rates.numel()
    out = torch.empty(depth, 2, B * T, device=rates.device, dtype=F32)
    call("x2_droppath_rows", ptr(rates), int(seed) & 0xFFFFFFFF, ptr(DROP_EPOCH), depth, B, T, ptr(out))
    return out


def droppath_keep(rates, seed, B, epoch=None):
    """Host mirror of droppath_rows_kernel: 0/1 keep tensor [depth, 2, B] for (seed [, epoch])."""
    s_ = int(seed) & 0xFFFFFFFF
    if epoch is not None:
        s_ = _hash32_int((s_ + 0x9E3779B1 * (int(epoch) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    depth = len(rates)
    keep = torch.zeros(depth, 2, B)
    for lb in range(2 * depth):
        for b in range(B):
            h = _hash32_int(((lb * B + b) & 0xFFFFFFFF) ^ s_)
            u = torch.tensor(float(h >> 8), dtype=torch.float32) * torch.tensor(1.0 / 16777216.0, dtype=torch.float32)
            keep[lb >> 1, lb & 1, b] = 1.0 if float(u) >= float(torch.tensor(rates[lb >> 1], dtype=torch.float32)) else 0.0
    return keep


def frame_mean(x, pos, frames):
    """(B * F, T, D) fp32 [+ pos (F, D)] -> (B, T, D): mean over the frames (x2_frame_mean forward)."""
    BF, T, D = x.shape
    assert x.dtype == F32 and x.is_contiguous() and BF % frames == 0 and (pos is None or (pos.is_contiguous() and pos.numel() == frames * D))
    out = torch.empty(BF // frames, T, D, device=x.device, dtype=F32)
    call("x2_frame_mean", ptr(x), ptr(pos), None, ptr(out), None, BF // frames, frames, T, D, 0)
    return out


def frame_mean_bwd(dy, frames, want_dpos=True):
    """-> (dx (B * F, T, D), dpos (F, D) or None)."""
    B, T, D = dy.shape
    assert dy.dtype == F32 and dy.is_contiguous()
    dx = torch.empty(B * frames, T, D, device=dy.device, dtype=F32)
    dpos = torch.empty(frames, D, device=dy.device, dtype=F32) if want_dpos else None
    call("x2_frame_mean", None, None, ptr(dy), ptr(dx), ptr(dpos), B, frames, T, D, 1)
    return dx, dpos


def gelu_f32(x, dy=None):
    out = torch.empty_like(x)
    call("x2_gelu_f32", ptr(x), ptr(dy), ptr(out), x.numel())
    return out


def colsum_f32(x, out):
    call("x2_colsum_f32", ptr(x), ptr(out), x.shape[0], x.shape[1])
