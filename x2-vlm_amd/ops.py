"""Differentiable functional ops on the HIP kernels for the small heads and glue of the step
(projection + F.normalize, build_mlp heads, cross-entropies, row gathers, region pooling)."""
import torch

from . import kernels as K
from .engine import (CrossEntropyFn, GatherRowsFn, GeluF32Fn, L2NormFn, LayerNormF32Fn, LinearBf16Fn, LinearF32Fn)


def linear(x, w, b=None):
    return LinearF32Fn.apply(x, w, b)


def layer_norm(x, w, b, eps):
    return LayerNormF32Fn.apply(x, w, b, eps)


def gelu(x):
    return GeluF32Fn.apply(x)


def normalize(x):
    return L2NormFn.apply(x)


def cross_entropy(logits, labels):
    return CrossEntropyFn.apply(logits, labels)


def gather_rows(src, idx):
    return GatherRowsFn.apply(src, idx.to(torch.int32))


def mlp_head(seq, x):
    """nn.Sequential(Linear, LayerNorm(1e-5), GELU, Linear) of xvlm.py:163-169 on fp32 rows."""
    w0 = seq[0].weight
    if w0.shape[1] % 64 == 0 and w0.shape[0] % 8 == 0 and x.numel() % 4 == 0:
        h = LinearBf16Fn.apply(x, w0, seq[0].bias)          # 768 -> 1536: MFMA GEMM
    else:
        h = linear(x, w0, seq[0].bias)
    h = gelu(layer_norm(h, seq[1].weight, seq[1].bias, seq[1].eps))
    return linear(h, seq[3].weight, seq[3].bias)


class _MaskedMeanToken0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        out = x.contiguous().clone()
        w = w.contiguous()
        K.pool_tokens(out, w)
        ctx.save_for_backward(w)
        return out

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        g = dy.contiguous().clone()
        K.pool_tokens(g, w, bwd=True)
        return g, None


def masked_mean_token0(x, w):
    """token 0 <- sum_p w[b,p] x[b,1+p] / sum_p w[b,p]   (region pooling, beit2.py:430-436)."""
    return _MaskedMeanToken0.apply(x, w)


class _FrameMean(torch.autograd.Function):
    """(B*F, T, D) + pos (1,F,1,D) -> mean over frames (B, T, D).  xvlm.py:627-645; x2_frame_mean forward / backward kernels
    (HBM-bound row work)."""

    @staticmethod
    def forward(ctx, x, pos, frames):
        BF, T, D = x.shape
        ctx.frames, ctx.pos_shape = frames, pos.shape
        if not x.is_cuda:                                        # host use (unit tests of the wrapper logic): plain tensor maths
            return (x.view(BF // frames, frames, T, D).sum(1) + pos.view(1, frames, 1, D).sum(1)) / frames
        return K.frame_mean(x.contiguous(), pos.detach().reshape(frames, D).contiguous(), frames)

    @staticmethod
    def backward(ctx, dy):
        f = ctx.frames
        B, T, D = dy.shape
        if not dy.is_cuda:
            g = (dy / f).unsqueeze(1).expand(B, f, T, D).reshape(B * f, T, D)
            return g, (dy.sum((0, 1)) / f).view(1, 1, 1, D).expand(1, f, 1, D).clone(), None
        dx, dpos = K.frame_mean_bwd(dy.contiguous(), f)
        return dx, dpos.view(ctx.pos_shape), None


def frame_mean(x, pos, frames):
    return _FrameMean.apply(x, pos, frames)
