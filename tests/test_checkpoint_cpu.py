"""Checkpoint compatibility against golden vectors made by the reference's own functions
(tests/golden/make_golden_ckpt.py).  Host-only code: runs without a GPU."""
import importlib
import json
import os
import types

import numpy as np
import pytest
import torch

ck = importlib.import_module("x2-vlm_amd.checkpoint")
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ckpt_interp.npz"))


def fake_encoder(res, depth):
    """What interpolate_pos_embed reads of a vision encoder: state_dict() table shapes, patch grid, pos_embed."""
    w = res // 16
    sd = {"blocks.%d.attn.relative_position_bias_table" % i: torch.zeros((2 * w - 1) ** 2 + 3, 12) for i in range(depth)}
    return types.SimpleNamespace(state_dict=lambda: sd, patch_embed=types.SimpleNamespace(patch_shape=(w, w), num_patches=w * w),
                                 pos_embed=None)


@pytest.mark.parametrize("tag,dst_res", [("up", 384), ("down", 224)])
def test_rel_pos_tables_resampled_like_reference(tag, dst_res):
    sd = {"blocks.%d.attn.relative_position_bias_table" % i: torch.from_numpy(GOLD["%s_src_%d" % (tag, i)]) for i in range(2)}
    sd["blocks.0.attn.relative_position_index"] = torch.zeros(3, 3, dtype=torch.long)
    sd["cls_token"] = torch.ones(1, 1, 768)
    out = ck.interpolate_pos_embed(fake_encoder(dst_res, 2), sd)
    assert sorted(out.keys()) == json.loads(str(GOLD[tag + "_keys"]))
    for i in range(2):
        got = out["blocks.%d.attn.relative_position_bias_table" % i].numpy()
        ref = GOLD["%s_dst_%d" % (tag, i)]
        assert got.shape == ref.shape and got.dtype == ref.dtype
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
        np.testing.assert_array_equal(got[-3:], GOLD["%s_src_%d" % (tag, i)][-3:])      # cls rows carried over verbatim


def test_same_resolution_is_untouched():
    t = torch.randn(732, 12)
    out = ck.interpolate_pos_embed(fake_encoder(224, 1), {"blocks.0.attn.relative_position_bias_table": t})
    assert out["blocks.0.attn.relative_position_bias_table"] is t


def test_bert_key_surgery():
    keys = json.loads(str(GOLD["surgery_in"]))
    sd = {k: torch.tensor([float(i)]) for i, k in enumerate(keys)}
    ck.rename_tf_layernorm(sd)
    ck.load_params_choose_layers("bert.encoder.layer", sd, {6: 12, 7: 13, 8: 14, 9: 15, 10: 16, 11: 17}, do_expand=True)
    assert sorted([k, float(v)] for k, v in sd.items()) == json.loads(str(GOLD["surgery_expand"]))
    sd = {k: torch.tensor([float(i)]) for i, k in enumerate(keys)}
    ck.load_params_choose_layers("bert.encoder.layer", sd, {layer: i for i, layer in enumerate(range(1, 12, 2))})
    assert sorted([k, float(v)] for k, v in sd.items()) == json.loads(str(GOLD["surgery_pick"]))
    with pytest.raises(AssertionError):
        ck.load_params_choose_layers("bert.encoder.layer", {}, {0: 1, 2: 1})


def test_load_pretrained_state_dict(tmp_path):
    src = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).standard_normal((732, 12)).astype(np.float32))
    path = str(tmp_path / "ck.th")
    torch.save({"model": {"vision_encoder.blocks.0.attn.relative_position_bias_table": src,
                          "vision_encoder.blocks.0.attn.relative_position_index": torch.zeros(2, 2, dtype=torch.long),
                          "vision_encoder.cls_token": torch.full((1, 1, 768), 0.5),
                          "text_encoder.bert.encoder.layer.0.output.dense.bias": torch.arange(4.0),
                          "text_encoder.cls.predictions.bias": torch.arange(3.0),
                          "temp": torch.tensor(0.07), "itm_head.0.weight": torch.ones(2, 2)}}, path)
    model = types.SimpleNamespace(vision_encoder=fake_encoder(384, 1))
    cfg = {"use_beit_v2": True, "image_res": 384, "patch_size": 16}
    sd = ck.load_pretrained(model, path, cfg, is_eval=False, load_text=True)
    assert sorted(sd.keys()) == json.loads(str(GOLD["lp_keys"]))
    np.testing.assert_allclose(sd["vision_encoder.blocks.0.attn.relative_position_bias_table"].numpy(), GOLD["lp_table"], rtol=0, atol=1e-6)
    assert sorted(ck.load_pretrained(model, path, cfg, is_eval=True).keys()) == json.loads(str(GOLD["lp_keys_eval"]))
    with pytest.raises(ValueError):
        ck.load_pretrained(model, path, {"use_swin": True})


def test_model_level_round_trip(tmp_path):
    """A 224 px checkpoint of the MI355X model loads into a 384 px model through XVLMBase.load_pretrained; a stand-alone
    BEiT-2 file (shared rel_pos_bias, classifier head) and a TF-named 12-layer BERT initialise the 18-layer towers."""
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    cfgs = importlib.import_module("x2-vlm_amd.configs")

    def build(res, root, **kw):
        cfg = cfgs.pretrain_config(str(root), "base", res)
        cfg.update(vision_num_hidden_layers=1, text_num_hidden_layers=2, text_fusion_start_at=1)
        cfg.update(kw)
        return cfg

    torch.manual_seed(0)
    cfg224 = build(224, tmp_path / "a")
    m224 = mp.XVLM(config=cfg224, load_vision_params=False, load_text_params=False, pretraining=True)
    path = str(tmp_path / "x2vlm_224.th")
    torch.save({"model": m224.state_dict()}, path)
    cfg384 = build(384, tmp_path / "b")
    m384 = mp.XVLM(config=cfg384, load_vision_params=False, load_text_params=False, pretraining=True)
    msg = m384.load_pretrained(path, cfg384, is_domain_pretrain=False)
    t224 = m224.vision_encoder.blocks[0].attn.relative_position_bias_table
    t384 = m384.vision_encoder.blocks[0].attn.relative_position_bias_table
    assert t384.shape == (47 * 47 + 3, 12) and torch.equal(t384[-3:], t224[-3:])
    ref = ck.interpolate_rel_pos_bias(t224.detach(), 47 * 47 + 3, (24, 24))
    assert torch.equal(t384.detach(), ref)
    assert torch.equal(m384.vision_proj.weight, m224.vision_proj.weight) and float(m384.temp) == float(m224.temp)
    # load_text=True strips the `bert.` infix (xvlm.py:433-439): the MLM-headed text encoder is then left to its init
    assert any(k.startswith("text_encoder.bert.") for k in msg.missing_keys)
    assert all(n in dict(m384.named_parameters()) for n in m384.init_params)
    # domain pre-training keeps the keys as they are: everything but the resolution-bound tables must match
    m224b = mp.XVLM(config=cfg224, load_vision_params=False, load_text_params=False, pretraining=True)
    msg = m224b.load_pretrained(path, cfg224, is_domain_pretrain=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    assert torch.equal(m224b.text_encoder.bert.encoder.layer[1].output.dense.weight, m224.text_encoder.bert.encoder.layer[1].output.dense.weight)

    # stand-alone BEiT-2 checkpoint + pytorch_model.bin at construction time
    enc = m224.vision_encoder
    beit = {k: v.clone() for k, v in enc.state_dict().items() if "relative_position" not in k}
    beit["rel_pos_bias.relative_position_bias_table"] = t224.detach().clone()
    beit["head.weight"], beit["head.bias"] = torch.zeros(10, 768), torch.zeros(10)
    vcfg = json.load(open(cfg384["vision_config"]))
    vcfg["ckpt"] = str(tmp_path / "beit2.pth")
    torch.save({"model": beit}, vcfg["ckpt"])
    json.dump(vcfg, open(cfg384["vision_config"], "w"))
    bert = {}
    for k, v in m224.text_encoder.state_dict().items():
        if "crossattention" in k or "position_ids" in k:
            continue
        bert[k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")] = v.clone()
    torch.save(bert, os.path.join(cfg384["text_encoder"], "pytorch_model.bin"))
    m = mp.XVLM(config=cfg384, load_vision_params=True, load_text_params=True, pretraining=True)
    assert torch.equal(m.vision_encoder.blocks[0].attn.relative_position_bias_table.detach(), ref)
    assert torch.equal(m.vision_encoder.patch_embed.proj.weight, enc.patch_embed.proj.weight)
    assert torch.equal(m.text_encoder.bert.embeddings.LayerNorm.weight, m224.text_encoder.bert.embeddings.LayerNorm.weight)
    assert torch.equal(m.text_encoder.bert.encoder.layer[0].attention.self.query.weight,
                       m224.text_encoder.bert.encoder.layer[0].attention.self.query.weight)
    assert any("crossattention" in n for n in m.init_params)           # not in BERT: trained from scratch, lr * lr_mult
