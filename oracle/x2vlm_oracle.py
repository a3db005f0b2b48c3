"""ORACLE — test infrastructure, not product code.

CPU fp32 restatement of the X^2-VLM pre-training step (the hot path of SURVEY.md section 8a),
written from the reference's behaviour as a flat functional program over a state dict with the
reference's key names.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product path (x2-vlm_amd/) never does and fails loudly without its HIP library.

Pinned: tests/test_oracle_golden.py checks every function here against golden vectors produced
by the real reference in the build container (tests/golden/make_golden.py, fixtures
tests/golden/*.npz): losses, activations, logits, bbox coordinates and per-parameter gradients.

Also here: rerank_scores (Retrieval.py:113-160), pinned by tests/golden/tiny_retrieval.npz.  NOT covered by any golden
vector: the apex DDP / AMP semantics around the step (un-vendored third-party code): parity unpinned there, see DESIGN.md 6.

Operand-rounding-aware mode (round 5): `xvlm_forward(..., round_operands=torch.bfloat16)` (or the `rounding(...)` context)
runs the SAME program with the operands of every linear / attention product that the HIP path feeds to the matrix cores as
bf16 rounded to bf16 at the same sites (and the tensors it stores as bf16 - qkv, contexts, GELU outputs, the gradients it
hands on as bf16 - rounded where they are stored); accumulation and everything else stays fp32.  With the mode off (the
default) every helper below is the identity and the program is bit-identical to the one the golden vectors pin
(tests/test_oracle_golden.py asserts both).  The mode exists so that the model-level GPU tests can hold POINTWISE values and
per-tensor gradients to ~1e-3 instead of the 1-3 % that bf16-vs-fp32 operand rounding through 36 GEMMs allows.

Deviations from the pinned third-party stack, all result-neutral:
  * image (encoder) attention mask uses transformers==4.12.5's fp32 constant (1-m)*-1e9
    (modeling_utils.invert_attention_mask); the golden run used transformers 5's finfo.min.
    Identical softmax for every row with >= 1 visible key (token 0 is always visible).
  * GELU is the exact erf form (ACT2FN["gelu"], nn.GELU()).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    image_res: int = 224
    patch_size: int = 16
    vision_width: int = 768
    vision_heads: int = 12
    vision_layers: int = 12
    hidden: int = 768
    heads: int = 12
    ffn: int = 3072
    vocab: int = 30522
    text_layers: int = 18      # includes the fusion layers (text_num_hidden_layers)
    fusion_at: int = 12        # text_fusion_start_at
    embed_dim: int = 256
    frames: int = 0            # video_encoding == 'avgpool' when > 0
    max_pos: int = 512         # max_position_embeddings

    @property
    def grid(self):
        return self.image_res // self.patch_size

    @property
    def n_tokens(self):
        return self.grid * self.grid + 1


def config_from_case(c):
    vw = c.get("vision_width", 768)
    return OracleConfig(image_res=c["image_res"], vision_width=vw, vision_heads=vw // 64, vision_layers=c["vision_layers"], hidden=c["hidden"],
                        heads=c["heads"], ffn=c["ffn"], vocab=c["vocab"], text_layers=c["text_layers"],
                        fusion_at=c["fusion_at"], embed_dim=c["embed_dim"], frames=c["frames"],
                        max_pos=c["max_pos"])


# --------------------------------------------------------------------------- parameters

def relative_position_index(grid):
    """(N,N) int64 index into the (2g-1)^2+3 entry bias table. beit2.py:93-113."""
    n_rel = (2 * grid - 1) ** 2 + 3
    ys, xs = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
    py, px = ys.reshape(-1), xs.reshape(-1)
    dy = py[:, None] - py[None, :] + grid - 1
    dx = px[:, None] - px[None, :] + grid - 1
    idx = torch.zeros(grid * grid + 1, grid * grid + 1, dtype=torch.int64)
    idx[1:, 1:] = dy * (2 * grid - 1) + dx
    idx[0, :] = n_rel - 3
    idx[:, 0] = n_rel - 2
    idx[0, 0] = n_rel - 1
    return idx


def parameter_shapes(cfg):
    """name -> shape for every trainable tensor, in the reference's state-dict naming
    (SURVEY.md section 8b).  The tied MLM decoder weight is not listed separately."""
    D, Hd, Ff, V, E = cfg.vision_width, cfg.hidden, cfg.ffn, cfg.vocab, cfg.embed_dim
    n_rel = (2 * cfg.grid - 1) ** 2 + 3
    s = {"temp": ()}
    ve = "vision_encoder."
    s[ve + "cls_token"] = (1, 1, D)
    s[ve + "patch_embed.proj.weight"] = (D, 3, cfg.patch_size, cfg.patch_size)
    s[ve + "patch_embed.proj.bias"] = (D,)
    for i in range(cfg.vision_layers):
        b = ve + "blocks.%d." % i
        s[b + "gamma_1"] = (D,); s[b + "gamma_2"] = (D,)
        s[b + "norm1.weight"] = (D,); s[b + "norm1.bias"] = (D,)
        s[b + "attn.q_bias"] = (D,); s[b + "attn.v_bias"] = (D,)
        s[b + "attn.relative_position_bias_table"] = (n_rel, cfg.vision_heads)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.proj.weight"] = (D, D); s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,); s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (4 * D, D); s[b + "mlp.fc1.bias"] = (4 * D,)
        s[b + "mlp.fc2.weight"] = (D, 4 * D); s[b + "mlp.fc2.bias"] = (D,)
    s[ve + "fc_norm.weight"] = (D,); s[ve + "fc_norm.bias"] = (D,)
    te = "text_encoder.bert."
    s[te + "embeddings.word_embeddings.weight"] = (V, Hd)
    s[te + "embeddings.position_embeddings.weight"] = (cfg.max_pos, Hd)
    s[te + "embeddings.token_type_embeddings.weight"] = (2, Hd)
    s[te + "embeddings.LayerNorm.weight"] = (Hd,); s[te + "embeddings.LayerNorm.bias"] = (Hd,)
    for i in range(cfg.text_layers):
        b = te + "encoder.layer.%d." % i
        atts = ["attention"] + (["crossattention"] if i >= cfg.fusion_at else [])
        for a in atts:
            kin = D if a == "crossattention" else Hd
            s[b + a + ".self.query.weight"] = (Hd, Hd); s[b + a + ".self.query.bias"] = (Hd,)
            s[b + a + ".self.key.weight"] = (Hd, kin); s[b + a + ".self.key.bias"] = (Hd,)
            s[b + a + ".self.value.weight"] = (Hd, kin); s[b + a + ".self.value.bias"] = (Hd,)
            s[b + a + ".output.dense.weight"] = (Hd, Hd); s[b + a + ".output.dense.bias"] = (Hd,)
            s[b + a + ".output.LayerNorm.weight"] = (Hd,); s[b + a + ".output.LayerNorm.bias"] = (Hd,)
        s[b + "intermediate.dense.weight"] = (Ff, Hd); s[b + "intermediate.dense.bias"] = (Ff,)
        s[b + "output.dense.weight"] = (Hd, Ff); s[b + "output.dense.bias"] = (Hd,)
        s[b + "output.LayerNorm.weight"] = (Hd,); s[b + "output.LayerNorm.bias"] = (Hd,)
    c = "text_encoder.cls.predictions."
    s[c + "bias"] = (V,)
    s[c + "transform.dense.weight"] = (Hd, Hd); s[c + "transform.dense.bias"] = (Hd,)
    s[c + "transform.LayerNorm.weight"] = (Hd,); s[c + "transform.LayerNorm.bias"] = (Hd,)
    if cfg.frames:
        s["absolute_frame_pos_embed"] = (1, cfg.frames, 1, D)
    s["vision_proj.weight"] = (E, D); s["vision_proj.bias"] = (E,)
    s["text_proj.weight"] = (E, Hd); s["text_proj.bias"] = (E,)
    for h, o in (("itm_head", 2), ("bbox_head", 4)):
        s[h + ".0.weight"] = (2 * Hd, Hd); s[h + ".0.bias"] = (2 * Hd,)
        s[h + ".1.weight"] = (2 * Hd,); s[h + ".1.bias"] = (2 * Hd,)
        s[h + ".3.weight"] = (o, 2 * Hd); s[h + ".3.bias"] = (o,)
    return s


def make_params(cfg, seed, synth_tensor, requires_grad=True):
    """Seeded parameters (x2-vlm_amd/synthetic.synth_tensor) as a name -> leaf tensor dict."""
    sd = {}
    for name, shape in parameter_shapes(cfg).items():
        t = synth_tensor(name, shape, seed)
        sd[name] = t.requires_grad_(requires_grad)
    return sd


# --------------------------------------------------------------------------- operand rounding (off by default)

_ROUND = None       # None, or the dtype (torch.bfloat16) the matrix-core operands are rounded to


class rounding:
    """Context: `with rounding(torch.bfloat16): ...` - forward AND backward of everything built inside round at the sites
    marked below (the backward rounding is fixed when the forward node is created)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _ROUND
        self.prev, _ROUND = _ROUND, self.dtype
        return self

    def __exit__(self, *a):
        global _ROUND
        _ROUND = self.prev


class _RoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, fwd, bwd):
        ctx.dtype, ctx.bwd = dtype, bwd
        return x.to(dtype).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.bwd else g), None, None, None


def _q(x):
    """A value the HIP path reads as a bf16 matrix operand while its gradient stays fp32 (weights, the fp32 hidden states'
    bf16 copies, softmax probabilities): rounded in the forward, straight-through in the backward."""
    return x if _ROUND is None else _RoundFn.apply(x, _ROUND, True, False)


def _qg(x):
    """An fp32 GEMM result whose incoming GRADIENT the HIP path consumes as a bf16 operand (dY of the input- and
    weight-gradient GEMMs): identity in the forward, the gradient rounded in the backward."""
    return x if _ROUND is None else _RoundFn.apply(x, _ROUND, False, True)


def _qq(x):
    """A tensor the HIP path STORES as bf16 and whose gradient it stores as bf16 too (qkv, attention contexts, the pre-LN
    vision blocks' normalised rows): rounded both ways."""
    return x if _ROUND is None else _RoundFn.apply(x, _ROUND, True, True)


class _GeluSavedRounded(torch.autograd.Function):
    """GELU whose forward sees the fp32 pre-activation (the GEMM epilogue applies it to the accumulators) and whose backward
    evaluates GELU' at the bf16 copy of it that was saved (csrc/gemm.hip act = 1 / act = 2)."""

    @staticmethod
    def forward(ctx, v, dtype):
        ctx.save_for_backward(v.to(dtype).to(v.dtype))
        return 0.5 * v * (1.0 + torch.erf(v * (1.0 / math.sqrt(2.0))))

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        cdf = 0.5 * (1.0 + torch.erf(v * (1.0 / math.sqrt(2.0))))
        pdf = torch.exp(-0.5 * v * v) * (1.0 / math.sqrt(2.0 * math.pi))
        return g * (cdf + v * pdf), None


class _AttentionRounded(torch.autograd.Function):
    """Attention with the roundings of csrc/attention.hip, forward AND backward (the generic helpers cannot place them: the kernels
    round P BEFORE normalising, and their backward is not the autograd of their forward):
      forward   keys in tiles of 64 with a running maximum m_t; P_t = bf16(exp(s - m_t) [x dropout]) - the largest probability of a
                row is exactly 1, whatever the partition sum -; O = sum_t exp(m_t - m) (P_t V_t) / l, l from the unrounded, undropped
                exponentials; O stored as bf16.
      backward  p = exp(s - lse) recomputed in fp32; dV = bf16(p [x dropout])^T dO; dP = dO V^T; Delta = sum_d dO x O with the STORED
                bf16 O; dS = p (dP - Delta) rounded to bf16 for dQ = dS K, dK = dS^T Q (both x scale, stored as bf16) and for the
                gradient of the additive term (the relative-position bias is reduced from the bf16 dS stream)."""

    @staticmethod
    def forward(ctx, q, k, v, add, scale, pmul, dtype):
        rd = lambda t: t.to(dtype).to(t.dtype)
        s = (q @ k.transpose(-1, -2)) * scale
        if add is not None:
            s = s + add
        m = s.max(-1, keepdim=True).values
        l = torch.exp(s - m).sum(-1, keepdim=True)
        o = torch.zeros(q.shape[:-1] + (v.shape[-1],), dtype=q.dtype)
        m_run = torch.full_like(m, -float("inf"))
        for t0 in range(0, s.shape[-1], 64):
            st = s[..., t0:t0 + 64]
            m_run = torch.maximum(m_run, st.max(-1, keepdim=True).values)
            p = torch.exp(st - m_run)
            if pmul is not None:
                p = p * pmul[..., t0:t0 + 64]
            o = o + torch.exp(m_run - m) * (rd(p) @ v[..., t0:t0 + 64, :])
        out = rd(o / l)
        ctx.save_for_backward(q, k, v, add if add is not None else torch.zeros(()), out, m + torch.log(l),
                              pmul if pmul is not None else torch.zeros(()))
        ctx.meta = (scale, dtype, add is not None, pmul is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, add, out, lse, pmul = ctx.saved_tensors
        scale, dtype, has_add, has_drop = ctx.meta
        rd = lambda t: t.to(dtype).to(t.dtype)
        s = (q @ k.transpose(-1, -2)) * scale
        if has_add:
            s = s + add
        p = torch.exp(s - lse)
        do = rd(g)
        dv = rd(p * pmul if has_drop else p).transpose(-1, -2) @ do
        dp = do @ v.transpose(-1, -2)
        if has_drop:
            dp = dp * pmul
        ds = rd(p * (dp - (do * out).sum(-1, keepdim=True)))
        dq = rd((ds @ k) * scale)
        dk = rd((ds.transpose(-1, -2) @ q) * scale)
        dadd = None
        if has_add and ctx.needs_input_grad[3]:
            dadd = ds
            while dadd.dim() > add.dim():
                dadd = dadd.sum(0)
            for d in range(add.dim()):
                if add.shape[d] == 1 and dadd.shape[d] != 1:
                    dadd = dadd.sum(d, keepdim=True)
        return dq, dk, rd(dv), dadd, None, None, None


def gelu_mm(v):
    """GELU behind a matrix product (fc1 / intermediate / MLM transform)."""
    return gelu(v) if _ROUND is None else _GeluSavedRounded.apply(v, _ROUND)


def mm_linear(x, w, b=None, x_grad=False, out=None):
    """A linear layer the HIP path runs on the matrix cores (bf16 operands, fp32 accumulation).  x_grad: the gradient w.r.t.
    x is stored as bf16 (vision blocks); out: None = fp32 result whose gradient is consumed as a bf16 operand, "bf16" = the
    result itself is stored as bf16 (qkv / q / kv projections), "raw" = fp32 result, gradient handled by the caller."""
    y = linear(_qq(x) if x_grad else _q(x), _q(w), b)
    return _qq(y) if out == "bf16" else y if out == "raw" else _qg(y)


# --------------------------------------------------------------------------- primitive ops

def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def split_heads(x, heads):
    B, N, C = x.shape
    return x.view(B, N, heads, C // heads).permute(0, 2, 1, 3)


def merge_heads(x):
    B, H, N, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, N, H * d)


def attention_core(q, k, v, scale, add, pmul=None):
    """softmax(q k^T * scale + add) v over (B,H,Lq,d)/(B,H,Lk,d); `add` broadcastable or None.
    pmul: training-mode dropout multiplier (0 or 1/(1-p)) on the probabilities (xbert.py:399)."""
    if _ROUND is not None:      # the kernels' own rounding sites, forward and backward
        return _AttentionRounded.apply(q, k, v, add, scale, pmul, _ROUND)
    s = (q @ k.transpose(-1, -2)) * scale
    if add is not None:
        s = s + add
    p = torch.softmax(s, dim=-1)
    if pmul is not None:
        p = p * pmul
    return p @ v


def cross_entropy(logits, labels, ignore_index=-100):
    """Mean over non-ignored rows of -log softmax(logits)[label] (nn.CrossEntropyLoss default)."""
    keep = labels != ignore_index
    lse = torch.logsumexp(logits, dim=-1)
    picked = logits.gather(-1, labels.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    return ((lse - picked) * keep).sum() / keep.sum()


# --------------------------------------------------------------------------- vision encoder

def patch_embed(sd, cfg, image):
    """Conv2d(3->D, k=s=16) as a GEMM over flattened patches. beit2.py:225-232."""
    B = image.shape[0]
    p, g = cfg.patch_size, image.shape[-1] // cfg.patch_size
    cols = image.view(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
    w = sd["vision_encoder.patch_embed.proj.weight"].reshape(cfg.vision_width, -1)
    return mm_linear(cols, w, sd["vision_encoder.patch_embed.proj.bias"])


def vision_block(sd, cfg, i, x, rel_index, dp=None):
    """Pre-LN block with fused-QKV attention + rel-pos bias + layer scale. beit2.py:125-166, 191-209.
    dp: training-mode DropPath multipliers (m1, m2), each (B,1,1) with values 0 or 1/(1-p) (timm drop_path)."""
    p = "vision_encoder.blocks.%d." % i
    H = cfg.vision_heads
    D = cfg.vision_width
    h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv_bias = torch.cat([sd[p + "attn.q_bias"], torch.zeros(D), sd[p + "attn.v_bias"]])
    qkv = mm_linear(h, sd[p + "attn.qkv.weight"], qkv_bias, x_grad=True, out="bf16")
    q, k, v = (split_heads(t, H) for t in qkv.split(D, dim=-1))
    N = x.shape[1]
    bias = sd[p + "attn.relative_position_bias_table"][rel_index.reshape(-1)].view(N, N, H).permute(2, 0, 1)
    ctx = merge_heads(attention_core(q, k, v, (D // H) ** -0.5, bias.unsqueeze(0)))
    m1, m2 = dp if dp is not None else (1.0, 1.0)
    # layer scale: the gradient entering gamma * u (after the DropPath factor) is what the HIP backward rounds to bf16
    x = x + m1 * _qg(sd[p + "gamma_1"] * mm_linear(ctx, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], out="raw"))
    h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = _q(gelu_mm(mm_linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], x_grad=True)))
    return x + m2 * _qg(sd[p + "gamma_2"] * mm_linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], out="raw"))


def vision_encoder(sd, cfg, image, idx_to_group_img=None, image_atts=None, drop_path=None):
    """beit2.py:378-436.  Returns (B,1+P,D); with idx_to_group_img also the region-pooled copy:
    (region_embeds, full_embeds) where full_embeds is still per image (not yet gathered)."""
    B = image.shape[0]
    x = torch.cat([sd["vision_encoder.cls_token"].expand(B, -1, -1), patch_embed(sd, cfg, image)], dim=1)
    rel_index = relative_position_index(image.shape[-1] // cfg.patch_size)
    for i in range(cfg.vision_layers):
        x = vision_block(sd, cfg, i, x, rel_index, None if drop_path is None else drop_path[i])
    patches = layer_norm(x[:, 1:], sd["vision_encoder.fc_norm.weight"], sd["vision_encoder.fc_norm.bias"], 1e-6)
    full = torch.cat([patches.mean(dim=1, keepdim=True), patches], dim=1)
    if idx_to_group_img is None:
        return full
    per_row = patches[idx_to_group_img]                              # (bsz, P, D)
    w = image_atts[:, 1:].unsqueeze(-1).to(per_row.dtype)            # (bsz, P, 1)
    pooled = (w * per_row).sum(dim=1, keepdim=True) / w.sum(dim=1, keepdim=True)
    return torch.cat([pooled, per_row], dim=1), full


def frame_embeds(sd, cfg, frames):
    """Video: frames through the image encoder, + frame position, mean over frames. xvlm.py:615-661."""
    B, Fr = frames.shape[:2]
    per_frame = vision_encoder(sd, cfg, frames.reshape(B * Fr, *frames.shape[2:]))
    e = per_frame.view(B, Fr, per_frame.shape[1], per_frame.shape[2]) + sd["absolute_frame_pos_embed"]
    return e.mean(dim=1), per_frame


# --------------------------------------------------------------------------- text / fusion encoder

def text_embeddings(sd, cfg, ids, drop=None):
    """word + type(0) + position, LayerNorm eps 1e-12, dropout multiplier drop("emb") if given. xbert.py:189-216."""
    p = "text_encoder.bert.embeddings."
    L = ids.shape[1]
    e = sd[p + "word_embeddings.weight"][ids] + sd[p + "token_type_embeddings.weight"][0] \
        + sd[p + "position_embeddings.weight"][:L]
    y = layer_norm(e, sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"], 1e-12)
    return y * drop("emb") if drop is not None else y


def bert_attention(sd, cfg, prefix, h, kv_src, add_mask, drop=None, site=""):
    """Self- or cross-attention sub-block with post-LN residual. xbert.py:322-431.
    drop(name) -> dropout multiplier tensor for site+".probs" / site+".out" (training mode) or None."""
    q = split_heads(mm_linear(h, sd[prefix + "self.query.weight"], sd[prefix + "self.query.bias"], out="bf16"), cfg.heads)
    k = split_heads(mm_linear(kv_src, sd[prefix + "self.key.weight"], sd[prefix + "self.key.bias"], out="bf16"), cfg.heads)
    v = split_heads(mm_linear(kv_src, sd[prefix + "self.value.weight"], sd[prefix + "self.value.bias"], out="bf16"), cfg.heads)
    ctx = merge_heads(attention_core(q, k, v, 1.0 / math.sqrt(cfg.hidden // cfg.heads), add_mask,
                                     drop(site + ".probs") if drop is not None else None))
    o = mm_linear(ctx, sd[prefix + "output.dense.weight"], sd[prefix + "output.dense.bias"])
    if drop is not None:
        o = o * drop(site + ".out")
    return layer_norm(o + h, sd[prefix + "output.LayerNorm.weight"], sd[prefix + "output.LayerNorm.bias"], 1e-12)


def bert_layer(sd, cfg, i, h, self_mask, enc, enc_mask, drop=None):
    """xbert.py:566-625: self-attn, [cross-attn iff i >= fusion_at and enc given], FFN."""
    p = "text_encoder.bert.encoder.layer.%d." % i
    h = bert_attention(sd, cfg, p + "attention.", h, h, self_mask, drop, "L%d.self" % i)
    if i >= cfg.fusion_at and enc is not None:
        h = bert_attention(sd, cfg, p + "crossattention.", h, enc, enc_mask, drop, "L%d.cross" % i)
    f = _q(gelu_mm(mm_linear(h, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"])))
    o = mm_linear(f, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    if drop is not None:
        o = o * drop("L%d.ffn.out" % i)
    return layer_norm(o + h, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], 1e-12)


def bert_encoder(sd, cfg, h, text_atts, enc=None, enc_atts=None, mode="multi_modal", drop=None):
    """Layer range by mode (xbert.py:674-686); additive masks (xbert.py:1071-1072 and
    transformers 4.12.5 invert_attention_mask, fp32 branch)."""
    lo, hi = {"text": (0, cfg.fusion_at), "fusion": (cfg.fusion_at, cfg.text_layers),
              "multi_modal": (0, cfg.text_layers)}[mode]
    self_mask = (1.0 - text_atts.to(h.dtype))[:, None, None, :] * -10000.0
    enc_mask = None
    if enc is not None:
        enc_mask = (1.0 - enc_atts.to(h.dtype))[:, None, None, :] * -1e9
    for i in range(lo, hi):
        h = bert_layer(sd, cfg, i, h, self_mask, enc, enc_mask, drop)
    return h


def text_embeds(sd, cfg, ids, atts):
    return bert_encoder(sd, cfg, text_embeddings(sd, cfg, ids), atts, mode="text")


def cross_embeds(sd, cfg, image_embeds, image_atts, text_atts, text_embeds=None, text_ids=None):
    """xvlm.py:753-783."""
    if text_embeds is not None:
        return bert_encoder(sd, cfg, text_embeds, text_atts, image_embeds, image_atts, mode="fusion")
    return bert_encoder(sd, cfg, text_embeddings(sd, cfg, text_ids), text_atts, image_embeds, image_atts)


# --------------------------------------------------------------------------- retrieval re-ranking

def rerank_scores(sd, cfg, image_feats, image_embeds, text_feats, text_atts, text_embeds, k_test):
    """Retrieval.py:113-160 on one rank, one query at a time as the reference does: for every image the k_test texts
    with the highest ITC similarity are re-scored by a fusion pass + itm_head (logit of class 1), and symmetrically for
    every text; all other entries stay at -100."""
    sims = image_embeds @ text_embeds.t()
    score_i2t = torch.full((image_feats.shape[0], text_feats.shape[0]), -100.0)
    for i in range(sims.shape[0]):
        idx = sims[i].topk(k=k_test, dim=0).indices
        enc = image_feats[i].repeat(k_test, 1, 1)
        out = cross_embeds(sd, cfg, enc, torch.ones(enc.shape[:2], dtype=torch.long), text_atts[idx], text_embeds=text_feats[idx])
        score_i2t[i, idx] = head_mlp(sd, "itm_head", out[:, 0])[:, 1]
    score_t2i = torch.full((text_feats.shape[0], image_feats.shape[0]), -100.0)
    sims_t = sims.t()
    for i in range(sims_t.shape[0]):
        idx = sims_t[i].topk(k=k_test, dim=0).indices
        enc = image_feats[idx]
        out = cross_embeds(sd, cfg, enc, torch.ones(enc.shape[:2], dtype=torch.long), text_atts[i].repeat(k_test, 1),
                           text_embeds=text_feats[i].repeat(k_test, 1, 1))
        score_t2i[i, idx] = head_mlp(sd, "itm_head", out[:, 0])[:, 1]
    return score_i2t, score_t2i


# --------------------------------------------------------------------------- heads and losses

def head_mlp(sd, name, x):
    """build_mlp: Linear -> LayerNorm(1e-5) -> GELU -> Linear. xvlm.py:163-169."""
    w0 = sd[name + ".0.weight"]
    on_mfma = w0.shape[1] % 64 == 0 and w0.shape[0] % 8 == 0 and x.numel() % 4 == 0      # ops.mlp_head's own rule
    h = (mm_linear if on_mfma else linear)(x, w0, sd[name + ".0.bias"])
    h = gelu(layer_norm(h, sd[name + ".1.weight"], sd[name + ".1.bias"], 1e-5))
    return linear(h, sd[name + ".3.weight"], sd[name + ".3.bias"])


def features(sd, image_embeds, text_embeds_):
    """xvlm.py:785-792."""
    fi = F.normalize(linear(image_embeds[:, 0], sd["vision_proj.weight"], sd["vision_proj.bias"]), dim=-1)
    ft = F.normalize(linear(text_embeds_[:, 0], sd["text_proj.weight"], sd["text_proj.bias"]), dim=-1)
    return fi, ft


def contrastive_loss(sd, image_feat, text_feat, all_image_feat=None, all_text_feat=None):
    """xvlm.py:794-826 (idx=None branch).  `all_*` are the all-gathered features (concatenated in
    rank order, this rank's rows being the autograd-connected ones); default: single rank."""
    fi = image_feat if all_image_feat is None else all_image_feat
    ft = text_feat if all_text_feat is None else all_text_feat
    logits = fi @ ft.t() / sd["temp"]
    labels = torch.arange(logits.shape[0])
    return 0.5 * (cross_entropy(logits, labels) + cross_entropy(logits.t(), labels)), logits


def matching_loss(sd, cfg, image_embeds, image_atts, text_embeds_, text_atts, neg_idx):
    """xvlm.py:859-899 with the hard-negative indices given (image_neg_idx, text_neg_idx)."""
    ineg, tneg = (torch.as_tensor(n, dtype=torch.int64) for n in neg_idx)
    B = image_embeds.shape[0]
    pos = cross_embeds(sd, cfg, image_embeds, image_atts, text_atts, text_embeds=text_embeds_)[:, 0]
    t_all = torch.cat([text_embeds_, text_embeds_[tneg]]); ta_all = torch.cat([text_atts, text_atts[tneg]])
    i_all = torch.cat([image_embeds[ineg], image_embeds]); ia_all = torch.cat([image_atts[ineg], image_atts])
    neg = cross_embeds(sd, cfg, i_all, ia_all, ta_all, text_embeds=t_all)[:, 0]
    logits = head_mlp(sd, "itm_head", torch.cat([pos, neg]))
    labels = torch.cat([torch.ones(B, dtype=torch.int64), torch.zeros(2 * B, dtype=torch.int64)])
    return cross_entropy(logits, labels), logits


def mlm_logits(sd, cfg, ids_masked, text_atts, image_embeds, image_atts, masked_pos, return_hidden=False):
    """xbert.py:1591-1661: full multi_modal pass, gather masked positions, transform, tied decoder."""
    seq = cross_embeds(sd, cfg, image_embeds, image_atts, text_atts, text_ids=ids_masked)
    if return_hidden:
        return mlm_logits_from_hidden(sd, seq, masked_pos), seq
    return mlm_logits_from_hidden(sd, seq, masked_pos)


def mlm_logits_from_hidden(sd, h, masked_pos):
    h = h.gather(1, masked_pos.unsqueeze(-1).expand(-1, -1, h.shape[-1]))
    p = "text_encoder.cls.predictions."
    h = gelu_mm(mm_linear(h, sd[p + "transform.dense.weight"], sd[p + "transform.dense.bias"]))
    h = layer_norm(h, sd[p + "transform.LayerNorm.weight"], sd[p + "transform.LayerNorm.bias"], 1e-12)
    return mm_linear(h, sd["text_encoder.bert.embeddings.word_embeddings.weight"], sd[p + "bias"])


def box_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def giou_pairs(a, b):
    """GIoU of matched pairs = diagonal of box_ops.generalized_box_iou. box_ops.py:24-57."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, 2:], b[:, 2:]) - torch.max(a[:, :2], b[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    whc = (torch.max(a[:, 2:], b[:, 2:]) - torch.min(a[:, :2], b[:, :2])).clamp(min=0)
    hull = whc[:, 0] * whc[:, 1]
    return inter / union - (hull - union) / hull


def bbox_loss(coord, target, is_image):
    """xvlm.py:927-957."""
    l1 = (coord - target).abs()
    a, b = box_cxcywh_to_xyxy(coord), box_cxcywh_to_xyxy(target)
    if (a[:, 2:] < a[:, :2]).any() or (b[:, 2:] < b[:, :2]).any():
        giou = torch.zeros(coord.shape[0])
    else:
        giou = 1 - giou_pairs(a, b)
    keep = (1 - is_image).to(coord.dtype)
    n = keep.sum()
    return (l1 * keep[:, None]).sum() / n, (giou * keep).sum() / n


# --------------------------------------------------------------------------- the step

def xvlm_forward_text(sd, cfg, batch):
    """model_pretrain.py:67-72 (XVLM.forward_text, the image=None branch of forward: Pretrain.run_text_iter): the masked ids
    through all text_layers layers WITHOUT cross-attention (xbert.py:595: no encoder states), MLM head, {'loss_mlm'}."""
    logits, seq = mlm_logits(sd, cfg, batch["text_ids_masked"], batch["text_atts"], None, None, batch["masked_pos"], return_hidden=True)
    # text_embeds: what the reference's bert forward hook sees in this iteration - the full-depth sequence output
    ex = {"mlm_logits": logits, "mlm_lse": torch.logsumexp(logits.double(), dim=-1).float(), "text_embeds": seq}
    return {"loss_mlm": cross_entropy(logits.reshape(-1, cfg.vocab), batch["masked_ids"].reshape(-1))}, ex


def xvlm_forward(sd, cfg, batch, neg_idx, ret_bbox_loss=False, ret_match_loss=True, gather=None, round_operands=None):
    """model_pretrain.py:30-88 (XVLM.forward / forward_multimodal; image absent or None: forward_text).  Returns (losses, extras).
    `gather(t)` stands in for the ITC all-gather (identity when None).  round_operands=torch.bfloat16: the operand-rounding-aware
    mode (module docstring)."""
    if round_operands is not None:
        with rounding(round_operands):
            return xvlm_forward(sd, cfg, batch, neg_idx, ret_bbox_loss, ret_match_loss, gather)
    image = batch.get("image")
    if image is None:
        return xvlm_forward_text(sd, cfg, batch)
    ex = {}
    if ret_bbox_loss:
        image_embeds, full = vision_encoder(sd, cfg, image, batch["idx_to_group_img"], batch["image_atts"])
        image_atts = batch["image_atts"]
        full = full[batch["idx_to_group_img"]]
        ex["image_embeds"] = image_embeds
    else:
        if image.dim() == 5:
            image_embeds, ex["image_embeds"] = frame_embeds(sd, cfg, image)   # extras keep the per-frame output
        else:
            image_embeds = ex["image_embeds"] = vision_encoder(sd, cfg, image)
        image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.int64)
    t_emb = text_embeds(sd, cfg, batch["text_ids"], batch["text_atts"])
    fi, ft = features(sd, image_embeds, t_emb)
    if gather is None:
        loss_itc, itc_logits = contrastive_loss(sd, fi, ft)
    else:
        loss_itc, itc_logits = contrastive_loss(sd, fi, ft, gather(fi), gather(ft))
    losses = {"loss_itc": loss_itc}
    ex.update(text_embeds=t_emb, image_feat=fi, text_feat=ft, itc_logits=fi @ ft.t() / sd["temp"])
    if ret_match_loss:
        losses["loss_itm"], ex["itm_logits"] = matching_loss(sd, cfg, image_embeds, image_atts, t_emb,
                                                             batch["text_atts"], neg_idx)
    else:
        losses["loss_itm"] = torch.tensor(0.0)
    logits = mlm_logits(sd, cfg, batch["text_ids_masked"], batch["text_atts"], image_embeds, image_atts,
                        batch["masked_pos"])
    ex["mlm_logits"] = logits
    ex["mlm_lse"] = torch.logsumexp(logits.double(), dim=-1).float()
    losses["loss_mlm"] = cross_entropy(logits.reshape(-1, cfg.vocab), batch["masked_ids"].reshape(-1))
    if ret_bbox_loss:
        ones = torch.ones(full.shape[:2], dtype=torch.int64)
        cls = cross_embeds(sd, cfg, full, ones, batch["text_atts"], text_embeds=t_emb)[:, 0]
        coord = head_mlp(sd, "bbox_head", cls).sigmoid()
        ex["bbox_coord"] = coord
        losses["loss_bbox"], losses["loss_giou"] = bbox_loss(coord, batch["target_bbox"], batch["is_image"])
    return losses, ex
