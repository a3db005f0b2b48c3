"""Checkpoint compatibility: released X2-VLM / BEiT-2 / BERT checkpoints load into
the MI355X modules, including a change of image resolution.

Pure load-time host code, as in the reference (nothing here touches the GPU):

  interpolate_pos_embed    models/beit2.py:653-754   relative-position-bias tables (and abs. pos_embed) -> new grid
  load_pretrained_beit2    models/beit2.py:473-601   stand-alone BEiT-2 checkpoint -> vision encoder
  load_state_dict_lenient  models/beit2.py:604-650   non-strict load that ignores relative_position_index
  rename_tf_layernorm      models/xvlm.py:66-73
  load_params_choose_layers models/xvlm.py:76-118    BERT layer remapping (12 -> 18 layers: 6..11 copied to 12..17)
  load_text_params         models/xvlm.py:318-385    pytorch_model.bin -> text encoder
  load_pretrained          models/xvlm.py:390-456    X2-VLM checkpoint -> state dict for XVLMBase.load_state_dict

The relative-position interpolation follows BEiT: source offsets sit on a geometric progression (ratio q found by
bisection so that the progression spans the target half-width), and every head's (2w-1)x(2w-1) table is resampled with a
bicubic spline at integer target offsets.  The reference calls scipy.interpolate.interp2d(kind='cubic'), which was
removed in SciPy 1.14; on a rectilinear grid that was FITPACK's regrid_smth, i.e. RectBivariateSpline(kx=ky=3, s=0),
SciPy's documented replacement - used directly here (tests/golden/ckpt_interp.npz pins it against the reference's own
function driven through that replacement).
"""
import copy
import os

import numpy as np
import torch


def _geometric_positions(src_size, dst_size):
    """Source coordinates (geometric progression, symmetric around 0) and integer target coordinates.
    models/beit2.py:688-713."""
    def geometric_progression(a, r, n):
        return a * (1.0 - r ** n) / (1.0 - r)

    left, right = 1.01, 1.5
    q = (left + right) / 2.0
    while right - left > 1e-6:
        q = (left + right) / 2.0
        gp = geometric_progression(1, q, src_size // 2)
        if gp > dst_size // 2:
            right = q
        else:
            left = q
    dis, cur = [], 1
    for i in range(src_size // 2):
        dis.append(cur)
        cur += q ** (i + 1)
    r_ids = [-d for d in reversed(dis)]
    x = np.asarray(r_ids + [0] + dis, dtype=np.float64)
    t = dst_size // 2.0
    dx = np.arange(-t, t + 0.1, 1.0)
    return x, dx


def interpolate_rel_pos_bias(rel_pos_bias, dst_num_pos, dst_patch_shape):
    """[src_num_pos, heads] -> [dst_num_pos, heads]; the trailing cls/extra rows are carried over unchanged.
    Returns the input object itself when no resampling is needed (models/beit2.py:676-731)."""
    from scipy.interpolate import RectBivariateSpline

    if dst_patch_shape[0] != dst_patch_shape[1]:
        raise NotImplementedError()
    src_num_pos, num_heads = rel_pos_bias.size()
    num_extra_tokens = dst_num_pos - (dst_patch_shape[0] * 2 - 1) * (dst_patch_shape[1] * 2 - 1)
    src_size = int((src_num_pos - num_extra_tokens) ** 0.5)
    dst_size = int((dst_num_pos - num_extra_tokens) ** 0.5)
    if src_size == dst_size:
        return rel_pos_bias
    extra_tokens = rel_pos_bias[-num_extra_tokens:, :]
    body = rel_pos_bias[:-num_extra_tokens, :]
    x, dx = _geometric_positions(src_size, dst_size)
    heads = []
    for h in range(num_heads):
        z = body[:, h].view(src_size, src_size).float().numpy().astype(np.float64)       # z[iy][ix]
        spline = RectBivariateSpline(x, x, z.T, kx=3, ky=3, s=0)                          # interp2d(x, y, z, 'cubic')
        heads.append(torch.Tensor(spline(dx, dx).T.copy()).contiguous().view(-1, 1).to(rel_pos_bias.device))
    return torch.cat((torch.cat(heads, dim=-1), extra_tokens), dim=0)


def interpolate_pos_embed(model, checkpoint_model):
    """models/beit2.py:653-754.  model: the vision encoder; checkpoint_model: its state dict (no prefix), edited in
    place and returned: relative_position_index buffers dropped, every relative_position_bias_table (and pos_embed, for
    encoders that have one) resampled to the model's patch grid."""
    own = model.state_dict()
    for key in list(checkpoint_model.keys()):
        if "relative_position_index" in key:
            checkpoint_model.pop(key)
        if "relative_position_bias_table" in key:
            if key not in own:
                print("Note that vision encoder does not have: ", key)
                continue
            new = interpolate_rel_pos_bias(checkpoint_model[key], own[key].size(0), model.patch_embed.patch_shape)
            if new is not checkpoint_model[key]:
                print("Position interpolate for %s to %dx%d" % (key, *model.patch_embed.patch_shape))
                checkpoint_model[key] = new
    if "pos_embed" in checkpoint_model and getattr(model, "pos_embed", None) is not None:
        pos = checkpoint_model["pos_embed"]
        dim = pos.shape[-1]
        num_patches = model.patch_embed.num_patches
        num_extra = model.pos_embed.shape[-2] - num_patches
        orig, new = int((pos.shape[-2] - num_extra) ** 0.5), int(num_patches ** 0.5)
        if orig != new:
            extra = pos[:, :num_extra]
            tok = pos[:, num_extra:].reshape(-1, orig, orig, dim).permute(0, 3, 1, 2)
            tok = torch.nn.functional.interpolate(tok, size=(new, new), mode="bicubic", align_corners=False)
            checkpoint_model["pos_embed"] = torch.cat((extra, tok.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)
    return checkpoint_model


def load_state_dict_lenient(model, state_dict, prefix="", ignore_missing="relative_position_index"):
    """models/beit2.py:604-650: non-strict load; returns (missing keys that matter, unexpected keys)."""
    msg = model.load_state_dict({k[len(prefix):] if prefix and k.startswith(prefix) else k: v for k, v in state_dict.items()},
                                strict=False)
    ignore = ignore_missing.split("|")
    missing = [k for k in msg.missing_keys if not any(i in k for i in ignore)]
    if missing:
        print("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if msg.unexpected_keys:
        print("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, list(msg.unexpected_keys)))
    return missing, list(msg.unexpected_keys)


def load_pretrained_beit2(model, ckpt_rpath):
    """models/beit2.py:473-601: a stand-alone BEiT-2 checkpoint ('model' | 'module' | bare state dict) into the vision
    encoder: classifier head dropped, a shared rel_pos_bias table expanded to every block, tables resampled."""
    print("Load BEIT-V2 ckpt from %s" % ckpt_rpath)
    checkpoint = torch.load(ckpt_rpath, map_location="cpu")
    checkpoint_model = None
    for model_key in ("model", "module"):
        if model_key in checkpoint:
            checkpoint_model = checkpoint[model_key]
            print("Load state_dict by model_key = %s" % model_key)
            break
    if checkpoint_model is None:
        checkpoint_model = checkpoint
    for k in ("head.weight", "head.bias"):
        del checkpoint_model[k]                         # KeyError if absent, as in the reference
    if getattr(model, "use_rel_pos_bias", False) and "rel_pos_bias.relative_position_bias_table" in checkpoint_model:
        print("Expand the shared relative position embedding to each transformer block. ")
        shared = checkpoint_model.pop("rel_pos_bias.relative_position_bias_table")
        for i in range(model.get_num_layers()):
            checkpoint_model["blocks.%d.attn.relative_position_bias_table" % i] = shared.clone()
    interpolate_pos_embed(model, checkpoint_model)
    return load_state_dict_lenient(model, checkpoint_model)


def rename_tf_layernorm(state_dict):
    """models/xvlm.py:66-73: TF-era LayerNorm.gamma / .beta -> .weight / .bias, in place."""
    for k in list(state_dict.keys()):
        if "LayerNorm." in k:
            new_k = k.strip().replace("LayerNorm.beta", "LayerNorm.bias").strip().replace("LayerNorm.gamma", "LayerNorm.weight")
            state_dict[new_k] = state_dict[k]
            if new_k != k:
                del state_dict[k]


def load_params_choose_layers(prefix, state_dict, mapper, do_expand=False):
    """models/xvlm.py:76-118.  mapper: {old_layer: new_layer}; keys `<prefix>.<old>.…` are copied to `<prefix>.<new>.…`
    walking from the lowest layer up; the source keys are kept only when do_expand."""
    assert len(set(mapper.values())) == len(mapper), f"{set(mapper.values())} != {len(mapper)}"
    mapper = {k: mapper[k] for k in sorted(int(k) for k in mapper.keys())}
    if not len(mapper):
        return state_dict
    keyed = []
    for k in list(state_dict.keys()):
        keyed.append((k, int(k[len(prefix) + 1:].strip().split(".")[0]) if k.startswith(prefix) else -1))
    for k in [p[0] for p in sorted(keyed, key=lambda p: p[1])]:
        if not k.startswith(prefix):
            continue
        new_k = None
        for i in mapper.keys():
            if k.startswith(f"{prefix}.{i}."):
                new_k = k.replace(f"{prefix}.{i}.", f"{prefix}.{mapper[i]}.")
                break
        if new_k:
            state_dict[new_k] = state_dict[k]
        if (new_k != k) and (not do_expand):
            del state_dict[k]
    return state_dict


def load_text_params(text_encoder, config, config_text, use_mlm_loss):
    """models/xvlm.py:318-385 (BERT branches): <text_encoder dir>/pytorch_model.bin into the text encoder.
    Returns the missing keys (they join XVLMBase.init_params: trained with lr * lr_mult)."""
    path = os.path.join(config["text_encoder"], "pytorch_model.bin")
    print("### Initializing text encoder from ", path)
    state_dict = torch.load(path, map_location="cpu")
    if "model" in state_dict.keys():
        state_dict = state_dict["model"]
    if "roberta" in config["text_encoder"]:
        raise NotImplementedError("RoBERTa text encoders are outside the X2VLM-base/large hot path")
    prefix = "bert.encoder.layer"
    if not use_mlm_loss:
        state_dict = {k.replace("roberta.", "").replace("bert.", ""): v for k, v in state_dict.items()}
        prefix = "encoder.layer"
    expand = {6: 12, 7: 13, 8: 14, 9: 15, 10: 16, 11: 17}
    if "bert-base-uncased" in config["text_encoder"]:
        rename_tf_layernorm(state_dict)
        if config_text.num_hidden_layers == 18:
            assert config["text_fusion_start_at"] == 12
            load_params_choose_layers(prefix, state_dict, expand, do_expand=True)
    elif "bert-large-uncased-12l" in config["text_encoder"]:
        if config["text_num_hidden_layers"] == 18:
            assert config["text_fusion_start_at"] == 12
            load_params_choose_layers(prefix, state_dict, expand, do_expand=True)
        else:
            raise NotImplementedError
    elif "bert-large-uncased" in config["text_encoder"]:
        rename_tf_layernorm(state_dict)
        if config_text.num_hidden_layers == 12:
            load_params_choose_layers(prefix, state_dict, {layer: i for i, layer in enumerate(range(1, 24 + 1, 2))})
        else:
            raise NotImplementedError
    elif "chinese-roberta-wwm-ext" in config["text_encoder"]:
        if config_text.num_hidden_layers == 6:
            load_params_choose_layers(prefix, state_dict, {1: 0, 3: 1, 5: 2, 7: 3, 9: 4, 11: 5})
    else:
        raise NotImplementedError
    if config.get("init_word_embeddings", False):
        print("### Train word_embeddings from scratch...", flush=True)
        for k in list(state_dict.keys()):
            if "word_embeddings" in k or k in ("cls.predictions.decoder.weight", "cls.predictions.bias"):
                del state_dict[k]
    msg = text_encoder.load_state_dict(state_dict, strict=False)
    print("missing_keys: ", msg.missing_keys, flush=True)
    print("unexpected_keys: ", msg.unexpected_keys, flush=True)
    return list(msg.missing_keys)


_TIMESFORMER_MAP = {"temporal_norm1": "norm1", "time_attn": "attn", "temporal_norm2": "norm2", "temporal_mlp": "mlp",
                    "time_gamma_1": "gamma_1", "time_gamma_2": "gamma_2"}


def init_timesformer_keys(state_dict):
    """models/xvlm.py:442-454 / 585-597: temporal blocks start as copies of the spatial ones."""
    for from_key, to_key in _TIMESFORMER_MAP.items():
        for key in list(state_dict.keys()):
            if to_key in key:
                state_dict[key.replace(to_key, from_key)] = copy.deepcopy(state_dict[key])


def load_pretrained(model, ckpt_rpath, config, is_eval=False, load_text=False):
    """models/xvlm.py:390-456 (use_beit_v2 branch): an X2-VLM checkpoint as a state dict for `model`; unless is_eval,
    the vision tables are resampled to the model's resolution and (load_text) text keys lose their bert. infix."""
    checkpoint = torch.load(ckpt_rpath, map_location="cpu")
    state_dict = checkpoint["model"] if "model" in checkpoint.keys() else checkpoint
    if is_eval:
        return state_dict
    print("### Loading pretrained vision encoder", flush=True)
    if not config.get("use_beit_v2", False):
        raise ValueError("only use_beit_v2 vision encoders are built by the MI355X path")
    vision_state_dict = {}
    for k in list(state_dict.keys()):
        if k.startswith("vision_encoder."):
            vision_state_dict[k[15:]] = state_dict.pop(k)
    vision_state_dict = interpolate_pos_embed(model.vision_encoder, vision_state_dict)
    for k in vision_state_dict.keys():
        state_dict["vision_encoder." + k] = vision_state_dict[k]
    if load_text:
        print("### Loading pretrained text encoder", flush=True)
        for key in list(state_dict.keys()):
            if key.startswith("text_encoder.") or key.startswith("cross_encoder."):
                encoder_key = key.replace("roberta.", "").replace("bert.", "").strip()
                state_dict[encoder_key] = state_dict[key]
                if encoder_key != key:
                    del state_dict[key]
    if config.get("init_timesformer", False):
        init_timesformer_keys(state_dict)
    return state_dict
