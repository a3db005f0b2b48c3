"""BEiT-v2 vision encoder with the reference's module tree and state-dict keys (models/beit2.py),
executed by the HIP stage VisionEncoderFn.  The nn.Modules below are parameter containers: the
arithmetic lives in csrc/ and is sequenced by engine.py.
"""
import json
import math
from functools import partial

import torch
import torch.nn as nn

from . import kernels as K
from .engine import VisionEncoderFn, vision_param_names


def trunc_normal_(t, std=0.02):
    # timm 0.4.9 trunc_normal_(std=.02) truncates at absolute +-2: effectively an un-truncated normal
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def relative_position_index(gh, gw):
    """beit2.py:93-113: index (1+gh*gw, 1+gh*gw) into the (2gh-1)(2gw-1)+3 entry bias table."""
    n_rel = (2 * gh - 1) * (2 * gw - 1) + 3
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    py, px = ys.reshape(-1), xs.reshape(-1)
    rel = (py[:, None] - py[None, :] + gh - 1) * (2 * gw - 1) + (px[:, None] - px[None, :] + gw - 1)
    idx = torch.zeros(gh * gw + 1, gh * gw + 1, dtype=torch.int64)
    idx[1:, 1:] = rel
    idx[0, :] = n_rel - 3
    idx[:, 0] = n_rel - 2
    idx[0, 0] = n_rel - 1
    return idx


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, window_size):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.window_size = window_size
        self.num_relative_distance = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(*window_size))
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, init_values, norm_layer, window_size, drop_path):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, window_size)
        self.drop_path_rate = drop_path
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    """beit2.py:274-436 (use_rel_pos_bias=True, use_abs_pos_emb=False, use_mean_pooling=True,
    init_values=0.1, qkv_bias=True: the only configuration X^2-VLM builds, xvlm.py:254-261)."""

    def __init__(self, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 drop_path_rate=0.1, init_values=0.1, vision_num_hidden_layers=-1, eps=1e-6):
        super().__init__()
        if vision_num_hidden_layers > 0:
            depth = vision_num_hidden_layers
        self.depth, self.embed_dim, self.num_heads, self.eps = depth, embed_dim, num_heads, eps
        self.num_features = embed_dim
        self.use_rel_pos_bias, self.pos_embed = True, None        # per-block tables, no absolute embedding (checkpoint.py reads these)
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        norm_layer = partial(nn.LayerNorm, eps=eps)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, init_values, norm_layer,
                                           self.patch_embed.patch_shape, dpr[i]) for i in range(depth)])
        self.fc_norm = norm_layer(embed_dim)
        trunc_normal_(self.cls_token)
        self.apply(self._init_weights)
        for layer_id, blk in enumerate(self.blocks):          # fix_init_weight, beit2.py:332-338
            blk.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            blk.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    def _params(self, lo=0, hi=None):
        # resolved once: walking named_parameters() costs ~1 ms of host time per call (Parameter objects are stable:
        # .to() / load_state_dict() update them in place)
        hi = self.depth if hi is None else hi
        cache = self.__dict__.setdefault("_param_lists", {})
        cached = cache.get((lo, hi))
        if cached is None:
            sd = dict(self.named_parameters())
            cached = cache[(lo, hi)] = [sd[n] for n in vision_param_names(self.depth, lo, hi)]
        return cached

    def drop_path_scales(self, B, T, device, keep=None):
        """Per-row keep/(1-p) factors of every block's two residual branches (timm drop_path: one Bernoulli per
        sample and branch).  keep: optional (depth, 2, B) 0/1 tensor (tests); default draws it with torch.rand."""
        cache = self.__dict__.setdefault("_dp_rates", {})     # built once per device: a host->device copy cannot be hipGraph-captured
        rates = cache.get(device)
        if rates is None:
            rates = cache[device] = torch.tensor([b.drop_path_rate for b in self.blocks], device=device, dtype=torch.float32)
        if keep is None and rates.is_cuda:
            # one kernel: Bernoulli keeps hashed from (seed, device epoch word) like the dropout masks, scaled and spread over the rows
            from .xbert import next_dropout_seed
            rows = K.droppath_rows(rates, next_dropout_seed(), B, T)
        else:
            if keep is None:
                keep = (torch.rand(self.depth, 2, B, device=device) >= rates.view(-1, 1, 1)).float()
            scale = keep.to(device).float() / (1.0 - rates).view(-1, 1, 1)
            rows = scale.repeat_interleave(T, dim=2)                   # (depth, 2, B*T)
        return [(rows[i, 0].contiguous(), rows[i, 1].contiguous()) if float(self.blocks[i].drop_path_rate) > 0 else (None, None)
                for i in range(self.depth)]

    def _run(self, x, pool_w=None):
        dp = None
        if self.training and any(b.drop_path_rate > 0 for b in self.blocks):
            T = self.patch_embed.num_patches + 1 if x.shape[-1] == self.patch_embed.img_size[0] else (x.shape[-1] // self.patch_embed.patch_size[0]) ** 2 + 1
            dp = self.drop_path_scales(x.shape[0], T, x.device, getattr(self, "fixed_drop_path_keep", None))
        meta = dict(depth=self.depth, heads=self.num_heads, patch=self.patch_embed.patch_size[0], eps=self.eps,
                    rel_index=self.blocks[0].attn.relative_position_index, pool_w=pool_w, drop_path=dp)
        cuts = [c for c in (getattr(self, "chunk_at", None) or ()) if 0 < c < self.depth]
        if not cuts:
            return VisionEncoderFn.apply(x.float(), meta, *self._params())
        # the tower as a chain of stages over block ranges (same arithmetic): `chunk_hook(i, tokens)` sees the residual stream
        # between two stages and returns what the next one consumes (graph.SegmentedStep cuts autograd there)
        x = x.float()
        hook = getattr(self, "chunk_hook", None)
        bounds = [0] + sorted(cuts) + [self.depth]
        for ci, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            x = VisionEncoderFn.apply(x, dict(meta, lo=lo, hi=hi), *self._params(lo, hi))
            if hook is not None and hi < self.depth:
                x = hook(ci, x)
        return x

    def forward(self, x, idx_to_group_img=None, image_atts=None, output_attentions=None, output_hidden_states=None):
        """beit2.py:378-436.  Returns (B,1+P,D); with idx_to_group_img the pair
        (region-pooled embeds per text row, full embeds per image)."""
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention maps / hidden states are not materialised by the fused kernels")
        full = self._run(x)
        if idx_to_group_img is None:
            return full
        from .engine import GatherRowsFn
        from . import ops
        per_row = GatherRowsFn.apply(full, idx_to_group_img.to(torch.int32))
        return ops.masked_mean_token0(per_row, image_atts[:, 1:].float()), full


def beit_base_patch16(img_size, **kw):
    return VisionTransformer(img_size=img_size, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, **kw)


def beit_large_patch16(img_size, **kw):
    return VisionTransformer(img_size=img_size, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, **kw)


def read_json(path):
    with open(path) as f:
        return json.load(f)
