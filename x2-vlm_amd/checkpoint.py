"""Checkpoint compatibility: released X2-VLM / BEiT-2 / BERT checkpoints load into the
MI355X modules, including a change of image resolution.  Load-time host code only; nothing here touches the GPU.

Organisation (what the reference spreads over if-ladders in models/xvlm.py:66-118, 318-460 and models/beit2.py:473-754):

  * ONE resampler, `resample_rel_pos_table`, for relative-position-bias tables (BEiT's geometric source grid, bicubic
    spline), and `resample_abs_pos_embed` for encoders with an absolute table;
  * ONE key-rewrite engine, `rewrite_keys`, driven by small rule tables:
        TEXT_RECIPES      text-encoder family -> (TF LayerNorm names?, {model depth: layer map, keep sources?})
        X2VLM_INFIXES     infixes dropped from text / cross encoder keys of an X2-VLM checkpoint
        TIMESFORMER_COPIES spatial -> temporal block names (init_timesformer)
  * thin entry points with the reference's names and call signatures (used by xvlm.XVLMBase and the fine-tune scripts):
    interpolate_pos_embed, load_pretrained_beit2, rename_tf_layernorm, load_params_choose_layers, load_text_params,
    load_pretrained, init_timesformer_keys.

The spline: the reference calls scipy.interpolate.interp2d(kind='cubic') (removed in SciPy 1.14).  On a rectilinear grid
that is the interpolating bicubic tensor-product spline with not-a-knot ends; it is evaluated here as two passes of 1-D
not-a-knot cubic splines (scipy.interpolate.CubicSpline along one axis, then the other), a formulation independent of the
FITPACK routine the golden vectors were produced with (tests/golden/make_golden_ckpt.py drove the reference's own
function through SciPy's documented interp2d replacement); tests/test_checkpoint_cpu.py holds the two to 1e-6.
"""
import os
import re

import numpy as np
import torch

# --------------------------------------------------------------------------------------------- resampling

def _geometric_axis(src_size, dst_size):
    """Source coordinates of a (2w-1)-point relative-offset axis laid out on a geometric progression that spans the
    target half-width, and the integer target coordinates (BEiT; models/beit2.py:688-713).  The ratio is found by the
    same bisection (bracket [1.01, 1.5], tolerance 1e-6) so that the coordinates agree to the last bit."""
    half_src, half_dst = src_size // 2, dst_size // 2
    lo, hi = 1.01, 1.5
    ratio = (lo + hi) / 2.0
    while hi - lo > 1e-6:
        ratio = (lo + hi) / 2.0
        span = (1.0 - ratio ** half_src) / (1.0 - ratio)          # 1 + r + ... + r^(half_src - 1)
        lo, hi = (lo, ratio) if span > half_dst else (ratio, hi)
    steps = np.cumsum([1.0] + [ratio ** (i + 1) for i in range(half_src - 1)]) if half_src else np.zeros(0)
    src = np.concatenate([-steps[::-1], [0.0], steps])
    dst = np.arange(-(dst_size // 2.0), dst_size // 2.0 + 0.1, 1.0)
    return src, dst


def _bicubic_resample(grid, src, dst):
    """grid[iy][ix] sampled at src x src -> values at dst x dst: tensor-product not-a-knot cubic spline."""
    from scipy.interpolate import CubicSpline
    # the bisection leaves the outermost source offset within 1e-6 of the outermost target, on either side: a target a hair
    # outside the source range is evaluated AT the boundary (what FITPACK's gridded evaluation does), not extrapolated
    dst = np.clip(dst, src[0], src[-1])
    along_x = CubicSpline(src, grid, axis=1, bc_type="not-a-knot")(dst)          # [src_y][dst_x]
    return CubicSpline(src, along_x, axis=0, bc_type="not-a-knot")(dst)          # [dst_y][dst_x]


def resample_rel_pos_table(table, dst_rows, dst_grid):
    """[(2s-1)^2 + extra, heads] -> [dst_rows, heads]; the trailing cls/extra rows are carried over unchanged.
    Returns the input object itself when the grids already agree."""
    gh, gw = dst_grid
    if gh != gw:
        raise NotImplementedError("square patch grids only")
    extra = dst_rows - (2 * gh - 1) * (2 * gw - 1)
    src_size, dst_size = int((table.shape[0] - extra) ** 0.5), int((dst_rows - extra) ** 0.5)
    if src_size == dst_size:
        return table
    src, dst = _geometric_axis(src_size, dst_size)
    body = table[:table.shape[0] - extra].detach().float().cpu().numpy().astype(np.float64)
    heads = body.reshape(src_size, src_size, -1)                                  # [iy][ix][head]
    out = np.stack([_bicubic_resample(heads[:, :, h], src, dst) for h in range(heads.shape[2])], axis=-1)
    new_body = torch.from_numpy(out.reshape(dst_size * dst_size, -1).astype(np.float32)).to(table.device)
    return torch.cat([new_body, table[table.shape[0] - extra:]], dim=0)


def resample_abs_pos_embed(pos, num_patches, num_extra):
    """[1, extra + s*s, D] absolute table -> the model's grid (torch bicubic; encoders with pos_embed only)."""
    dim = pos.shape[-1]
    src, dst = int((pos.shape[-2] - num_extra) ** 0.5), int(num_patches ** 0.5)
    if src == dst:
        return pos
    grid = pos[:, num_extra:].reshape(-1, src, src, dim).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(dst, dst), mode="bicubic", align_corners=False)
    return torch.cat([pos[:, :num_extra], grid.permute(0, 2, 3, 1).flatten(1, 2)], dim=1)


def interpolate_rel_pos_bias(rel_pos_bias, dst_num_pos, dst_patch_shape):
    return resample_rel_pos_table(rel_pos_bias, dst_num_pos, dst_patch_shape)


def interpolate_pos_embed(model, checkpoint_model):
    """Vision-encoder state dict (no prefix) fitted to `model`'s patch grid, in place: index buffers dropped (they are
    rebuilt by the module), every bias table / pos_embed resampled.  models/beit2.py:653-754."""
    own = model.state_dict()
    grid = model.patch_embed.patch_shape
    for key in [k for k in checkpoint_model if "relative_position_index" in k]:
        del checkpoint_model[key]
    for key in [k for k in checkpoint_model if "relative_position_bias_table" in k]:
        if key not in own:
            print("vision encoder has no parameter %s (left in the state dict)" % key)
            continue
        fitted = resample_rel_pos_table(checkpoint_model[key], own[key].shape[0], grid)
        if fitted is not checkpoint_model[key]:
            print("resampled %s to a %dx%d patch grid" % (key, grid[0], grid[1]))
            checkpoint_model[key] = fitted
    if "pos_embed" in checkpoint_model and getattr(model, "pos_embed", None) is not None:
        n = model.patch_embed.num_patches
        checkpoint_model["pos_embed"] = resample_abs_pos_embed(checkpoint_model["pos_embed"], n, model.pos_embed.shape[-2] - n)
    return checkpoint_model


# --------------------------------------------------------------------------------------------- key rewriting

def rewrite_keys(state_dict, rename=None, drop=None):
    """In place: keys for which drop(key) holds are removed, every other key k becomes rename(k) (None / k = unchanged;
    a later key wins when two map to the same name)."""
    for key in list(state_dict.keys()):
        if drop is not None and drop(key):
            del state_dict[key]
            continue
        new = rename(key) if rename is not None else key
        if new is not None and new != key:
            state_dict[new] = state_dict.pop(key)
    return state_dict


_TF_LAYERNORM = (("LayerNorm.beta", "LayerNorm.bias"), ("LayerNorm.gamma", "LayerNorm.weight"))


def rename_tf_layernorm(state_dict):
    """TF-era LayerNorm.gamma / .beta -> .weight / .bias (models/xvlm.py:66-73)."""
    def fix(key):
        for old, new in _TF_LAYERNORM:
            key = key.replace(old, new)
        return key
    return rewrite_keys(state_dict, rename=lambda k: fix(k) if "LayerNorm." in k else k)


def load_params_choose_layers(prefix, state_dict, mapper, do_expand=False):
    """Re-number encoder layers: keys `<prefix>.<src>.*` become `<prefix>.<mapper[src]>.*`.  With do_expand the source
    layers are kept as well (12 -> 18 layers: 6..11 are ALSO used as 12..17); without it every layer outside the map is
    dropped (pick 12 of 24).  models/xvlm.py:76-118."""
    targets = list(mapper.values())
    assert len(set(targets)) == len(targets), "layer map is not one-to-one: %s" % (mapper,)
    mapper = {int(s): int(d) for s, d in mapper.items()}
    if not mapper:
        return state_dict
    # values are taken from the checkpoint as loaded: a map whose target is itself a LATER source would need the
    # reference's in-place chaining semantics, which no recipe uses
    if any(d in mapper and d > s for s, d in mapper.items()):
        raise NotImplementedError("chained layer map %s" % (mapper,))
    layer_of = re.compile(re.escape(prefix) + r"\.(\d+)\.")
    original = dict(state_dict)
    family = [(key, layer_of.match(key)) for key in original]
    if not do_expand:                      # picking layers: everything of the family goes, the chosen ones come back below
        for key, m in family:
            if m is not None:
                del state_dict[key]
    for key, m in family:
        if m is not None and int(m.group(1)) in mapper:
            state_dict["%s.%d.%s" % (prefix, mapper[int(m.group(1))], key[m.end():])] = original[key]
    return state_dict


_UPPER_HALF_TWICE = {6: 12, 7: 13, 8: 14, 9: 15, 10: 16, 11: 17}     # fusion layers start as copies of text layers 6..11
# text-encoder family (substring of config['text_encoder'], first match wins) ->
#   tf_names: checkpoint uses LayerNorm.gamma/beta;  depths: {model depth: (layer map, keep sources)}, None = load as is;
#   other depths: other_depths_ok or NotImplementedError (as the reference's ladder, models/xvlm.py:333-365)
TEXT_RECIPES = (
    ("bert-base-uncased", dict(tf_names=True, depths={18: (_UPPER_HALF_TWICE, True)}, other_depths_ok=True)),
    ("bert-large-uncased-12l", dict(tf_names=False, depths={18: (_UPPER_HALF_TWICE, True)}, other_depths_ok=False)),
    ("bert-large-uncased", dict(tf_names=True, depths={12: ({src: i for i, src in enumerate(range(1, 25, 2))}, False)},
                                other_depths_ok=False)),
)


def load_text_params(text_encoder, config, config_text, use_mlm_loss):
    """<text_encoder dir>/pytorch_model.bin into the text encoder (models/xvlm.py:318-385, BERT families).
    Returns the keys the checkpoint lacks: they join XVLMBase.init_params (trained with lr * lr_mult)."""
    name = config["text_encoder"]
    path = os.path.join(name, "pytorch_model.bin")
    print("text encoder weights <- %s" % path)
    state_dict = torch.load(path, map_location="cpu")
    state_dict = state_dict.get("model", state_dict)
    if "roberta" in name:                  # the reference loads those into models/xroberta.py modules: not built on this path
        raise NotImplementedError("RoBERTa text encoders are outside the X2VLM-base/large hot path")
    prefix = "bert.encoder.layer"
    if not use_mlm_loss:                   # bare BertModel: no `bert.` level in its own keys
        state_dict = {k.replace("roberta.", "").replace("bert.", ""): v for k, v in state_dict.items()}
        prefix = "encoder.layer"
    recipe = next((r for tag, r in TEXT_RECIPES if tag in name), None)
    if recipe is None:
        raise NotImplementedError("no loading recipe for text encoder %s" % name)
    if recipe["tf_names"]:
        rename_tf_layernorm(state_dict)
    depth = config_text.num_hidden_layers
    if depth in recipe["depths"]:
        layer_map, keep = recipe["depths"][depth]
        if layer_map is _UPPER_HALF_TWICE:
            assert config["text_fusion_start_at"] == 12
        load_params_choose_layers(prefix, state_dict, layer_map, do_expand=keep)
    elif not recipe["other_depths_ok"]:
        raise NotImplementedError("%s with %d layers" % (name, depth))
    if config.get("init_word_embeddings", False):
        print("word embeddings (and the tied MLM decoder) are left at their initialisation")
        rewrite_keys(state_dict, drop=lambda k: "word_embeddings" in k or k in ("cls.predictions.decoder.weight", "cls.predictions.bias"))
    result = text_encoder.load_state_dict(state_dict, strict=False)
    print("text encoder: %d keys missing, %d unexpected" % (len(result.missing_keys), len(result.unexpected_keys)), flush=True)
    return list(result.missing_keys)


# --------------------------------------------------------------------------------------------- whole-model checkpoints

X2VLM_INFIXES = ("roberta.", "bert.")          # dropped from text_encoder.* / cross_encoder.* keys when load_text
TIMESFORMER_COPIES = (("norm1", "temporal_norm1"), ("attn", "time_attn"), ("norm2", "temporal_norm2"), ("mlp", "temporal_mlp"),
                      ("gamma_1", "time_gamma_1"), ("gamma_2", "time_gamma_2"))


def init_timesformer_keys(state_dict):
    """Temporal blocks start as copies of the spatial ones (models/xvlm.py:442-454, 585-597)."""
    for spatial, temporal in TIMESFORMER_COPIES:
        for key in [k for k in state_dict if spatial in k]:
            state_dict[key.replace(spatial, temporal)] = state_dict[key].clone()
    return state_dict


def _model_state(ckpt_rpath):
    blob = torch.load(ckpt_rpath, map_location="cpu")
    return blob["model"] if "model" in blob else blob


def load_pretrained(model, ckpt_rpath, config, is_eval=False, load_text=False):
    """An X2-VLM checkpoint as a state dict for `model` (models/xvlm.py:390-456, use_beit_v2 branch).  Unless is_eval the
    vision tables are resampled to the model's resolution and, with load_text, text / cross encoder keys lose their
    `bert.` / `roberta.` infix."""
    state_dict = _model_state(ckpt_rpath)
    if is_eval:
        return state_dict
    if not config.get("use_beit_v2", False):
        raise ValueError("only use_beit_v2 vision encoders are built by the MI355X path")
    tag = "vision_encoder."
    vision = {k[len(tag):]: state_dict.pop(k) for k in [k for k in state_dict if k.startswith(tag)]}
    for k, v in interpolate_pos_embed(model.vision_encoder, vision).items():
        state_dict[tag + k] = v
    if load_text:
        def strip(key):
            if not key.startswith(("text_encoder.", "cross_encoder.")):
                return key
            for infix in X2VLM_INFIXES:
                key = key.replace(infix, "")
            return key.strip()
        rewrite_keys(state_dict, rename=strip)
    if config.get("init_timesformer", False):
        init_timesformer_keys(state_dict)
    return state_dict


def load_state_dict_lenient(model, state_dict, prefix="", ignore_missing="relative_position_index"):
    """Non-strict load that does not report the index buffers (models/beit2.py:604-650).
    Returns (missing keys that matter, unexpected keys)."""
    if prefix:
        state_dict = {k[len(prefix):] if k.startswith(prefix) else k: v for k, v in state_dict.items()}
    result = model.load_state_dict(state_dict, strict=False)
    quiet = ignore_missing.split("|")
    missing = [k for k in result.missing_keys if not any(q in k for q in quiet)]
    if missing:
        print("%s: not in the checkpoint: %s" % (type(model).__name__, missing))
    if result.unexpected_keys:
        print("%s: unused checkpoint entries: %s" % (type(model).__name__, list(result.unexpected_keys)))
    return missing, list(result.unexpected_keys)


def load_pretrained_beit2(model, ckpt_rpath):
    """A stand-alone BEiT-2 checkpoint ('model' | 'module' | bare state dict) into the vision encoder: classifier head
    dropped, a shared rel_pos_bias table copied to every block, tables resampled.  models/beit2.py:473-601."""
    print("vision encoder weights <- %s" % ckpt_rpath)
    blob = torch.load(ckpt_rpath, map_location="cpu")
    state = next((blob[k] for k in ("model", "module") if k in blob), blob)
    for k in ("head.weight", "head.bias"):
        del state[k]                                    # a BEiT-2 file always carries its classifier (KeyError otherwise)
    shared = "rel_pos_bias.relative_position_bias_table"
    if getattr(model, "use_rel_pos_bias", False) and shared in state:
        table = state.pop(shared)
        for i in range(model.get_num_layers()):
            state["blocks.%d.attn.relative_position_bias_table" % i] = table.clone()
    interpolate_pos_embed(model, state)
    return load_state_dict_lenient(model, state)
