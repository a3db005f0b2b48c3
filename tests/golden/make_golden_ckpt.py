#!/usr/bin/env python3
"""Golden vectors for checkpoint loading, produced by the REAL reference functions
(models/beit2.py: interpolate_pos_embed; models/xvlm.py: rename_tf_layernorm, load_params_choose_layers,
load_pretrained) on seeded synthetic state dicts.  Build container only.

scipy.interpolate.interp2d, which the reference calls (beit2.py:720), no longer exists in this image's SciPy 1.15
(removed in 1.14).  The stand-in installed below is SciPy's documented replacement for interp2d on a rectilinear grid
(interpolate transition guide: `RectBivariateSpline(x, y, z.T)` and `r(xnew, ynew).T`), the same FITPACK routine
interp2d(kind='cubic') dispatched to for gridded data - a third-party name, not reference code.

writes tests/golden/ckpt_interp.npz
usage:  python tests/golden/make_golden_ckpt.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_shims  # noqa: E402


class _Interp2dCubic:
    def __init__(self, x, y, z, kind="cubic"):
        from scipy.interpolate import RectBivariateSpline
        assert kind == "cubic"
        self.r = RectBivariateSpline(np.asarray(x, float), np.asarray(y, float), np.asarray(z, float).T, kx=3, ky=3, s=0)

    def __call__(self, xnew, ynew):
        return self.r(np.asarray(xnew, float), np.asarray(ynew, float)).T


def table(rows, heads, seed):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal((rows, heads)).astype(np.float32))


def main():
    reference_shims.install()
    import scipy.interpolate
    scipy.interpolate.interp2d = _Interp2dCubic
    os.chdir(reference_shims.REFERENCE_ROOT)
    from models import beit2 as rb
    from models import xvlm as rx
    rb.interpolate.interp2d = _Interp2dCubic

    out = {}
    # --- 1. rel-pos tables: 224 px (14x14 -> 732 rows) into a 384 px model (24x24 -> 2212 rows) and back -----------
    for tag, src_res, dst_res in (("up", 224, 384), ("down", 384, 224)):
        model = rb.beit_base_patch16(img_size=dst_res, drop_rate=0.0, drop_path_rate=0.1, attn_drop_rate=0.0, use_mean_pooling=True,
                                     init_scale=0.001, use_rel_pos_bias=True, use_abs_pos_emb=False, init_values=0.1, qkv_bias=True,
                                     vision_num_hidden_layers=2)
        w = src_res // 16
        rows = (2 * w - 1) ** 2 + 3
        sd = {"blocks.%d.attn.relative_position_bias_table" % i: table(rows, 12, 100 + i) for i in range(2)}
        sd["blocks.0.attn.relative_position_index"] = torch.zeros(3, 3, dtype=torch.long)
        sd["cls_token"] = torch.ones(1, 1, 768)
        got = rb.interpolate_pos_embed(model, dict(sd))
        assert "blocks.0.attn.relative_position_index" not in got
        for i in range(2):
            k = "blocks.%d.attn.relative_position_bias_table" % i
            out["%s_src_%d" % (tag, i)] = sd[k].numpy()
            out["%s_dst_%d" % (tag, i)] = got[k].numpy()
        out[tag + "_keys"] = np.array(json.dumps(sorted(got.keys())))

    # --- 2. BERT key surgery ----------------------------------------------------------------------------------------
    keys = ["bert.embeddings.LayerNorm.gamma", "bert.embeddings.LayerNorm.beta", "bert.embeddings.word_embeddings.weight"]
    for i in range(12):
        keys += ["bert.encoder.layer.%d.attention.output.LayerNorm.gamma" % i, "bert.encoder.layer.%d.attention.output.LayerNorm.beta" % i,
                 "bert.encoder.layer.%d.attention.self.query.weight" % i, "bert.encoder.layer.%d.output.dense.bias" % i]
    keys += ["cls.predictions.transform.LayerNorm.gamma", "cls.predictions.bias"]
    sd = {k: torch.tensor([float(i)]) for i, k in enumerate(keys)}
    rx.rename_tf_layernorm(sd)
    rx.load_params_choose_layers("bert.encoder.layer", sd, {6: 12, 7: 13, 8: 14, 9: 15, 10: 16, 11: 17}, do_expand=True)
    out["surgery_in"] = np.array(json.dumps(keys))
    out["surgery_expand"] = np.array(json.dumps(sorted((k, float(v)) for k, v in sd.items())))
    sd = {k: torch.tensor([float(i)]) for i, k in enumerate(keys)}
    rx.load_params_choose_layers("bert.encoder.layer", sd, {layer: i for i, layer in enumerate(range(1, 12, 2))})
    out["surgery_pick"] = np.array(json.dumps(sorted((k, float(v)) for k, v in sd.items())))

    # --- 3. load_pretrained on an X2-VLM-style checkpoint file -------------------------------------------------------
    model = types.SimpleNamespace(vision_encoder=rb.beit_base_patch16(
        img_size=384, drop_rate=0.0, drop_path_rate=0.1, attn_drop_rate=0.0, use_mean_pooling=True, init_scale=0.001,
        use_rel_pos_bias=True, use_abs_pos_emb=False, init_values=0.1, qkv_bias=True, vision_num_hidden_layers=1))
    ck = {"model": {"vision_encoder.blocks.0.attn.relative_position_bias_table": table(732, 12, 7),
                    "vision_encoder.blocks.0.attn.relative_position_index": torch.zeros(2, 2, dtype=torch.long),
                    "vision_encoder.cls_token": torch.full((1, 1, 768), 0.5),
                    "text_encoder.bert.encoder.layer.0.output.dense.bias": torch.arange(4.0),
                    "text_encoder.cls.predictions.bias": torch.arange(3.0),
                    "temp": torch.tensor(0.07), "itm_head.0.weight": torch.ones(2, 2)}}
    path = os.path.join(tempfile.mkdtemp(), "ck.th")
    torch.save(ck, path)
    cfg = {"use_beit_v2": True, "image_res": 384, "patch_size": 16}
    sd = rx.load_pretrained(model, path, cfg, is_eval=False, load_text=True)
    out["lp_keys"] = np.array(json.dumps(sorted(sd.keys())))
    out["lp_table"] = sd["vision_encoder.blocks.0.attn.relative_position_bias_table"].numpy()
    out["lp_keys_eval"] = np.array(json.dumps(sorted(rx.load_pretrained(model, path, cfg, is_eval=True).keys())))
    np.savez_compressed(os.path.join(HERE, "ckpt_interp.npz"), **out)
    print("wrote ckpt_interp.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
