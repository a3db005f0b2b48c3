"""Retrieval: oracle pinned to the reference's golden vectors on CPU; the HIP path
(x2-vlm_amd/model_retrieval.py) against both on the GPU."""
import importlib
import os

import numpy as np
import pytest
import torch

from cases import CASES, model_config
from oracle import x2vlm_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_retrieval.npz"))
NI, NT, K_TEST, WSEED, BSEED = 6, 10, 4, 51, 52          # as tests/golden/make_golden_retrieval.py


def data(synthetic):
    c = CASES["tiny"]
    img = synthetic.synth_batch(BSEED, NI, c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=True)["image"]
    txt = synthetic.synth_batch(BSEED + 1, NT, c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=True)
    return img, txt["text_ids"], txt["text_atts"]


def oracle_params(synthetic, requires_grad=False):
    """The retrieval model's text encoder is a bare BertModel: its state-dict names have no `bert.` infix, and the
    synthetic weights are seeded by name."""
    cfg = O.config_from_case(CASES["tiny"])
    sd = {}
    for name, shape in O.parameter_shapes(cfg).items():
        if name.startswith("text_encoder.cls.") or name.startswith("bbox_head."):
            continue
        sd[name] = synthetic.synth_tensor(name.replace("text_encoder.bert.", "text_encoder."), shape, WSEED).requires_grad_(requires_grad)
    return cfg, sd


def test_oracle_rerank_matches_reference(synthetic):
    cfg, sd = oracle_params(synthetic)
    image, ids, atts = data(synthetic)
    with torch.no_grad():
        tf = O.text_embeds(sd, cfg, ids, atts)
        vf = O.vision_encoder(sd, cfg, image)
        fi, ft = O.features(sd, vf, tf)
        np.testing.assert_allclose(fi.numpy(), GOLD["image_embeds"], atol=2e-6)
        np.testing.assert_allclose(ft.numpy(), GOLD["text_embeds"], atol=2e-6)
        s_i2t, s_t2i = O.rerank_scores(sd, cfg, vf, fi, tf, atts, ft, K_TEST)
    assert np.array_equal(s_i2t.numpy() == -100.0, GOLD["score_i2t"] == -100.0)
    assert np.array_equal(s_t2i.numpy() == -100.0, GOLD["score_t2i"] == -100.0)
    np.testing.assert_allclose(s_i2t.numpy(), GOLD["score_i2t"], atol=2e-5)
    np.testing.assert_allclose(s_t2i.numpy(), GOLD["score_t2i"], atol=2e-5)
    assert int((GOLD["score_i2t"] != -100.0).sum()) == NI * K_TEST and int((GOLD["score_t2i"] != -100.0).sum()) == NT * K_TEST


@pytest.mark.gpu
def test_hip_retrieval_matches_reference(synthetic, tmp_path):
    mr = importlib.import_module("x2-vlm_amd.model_retrieval")
    dev = torch.device("cuda:0")
    model = mr.XVLMForRetrieval(config=model_config("tiny", str(tmp_path)))
    synthetic.synth_state_dict(model, WSEED)
    model = model.to(dev).eval()
    image, ids, atts = (t.to(dev) for t in data(synthetic))
    with torch.no_grad():
        tf = model.get_text_embeds(ids, atts)
        te = model.get_features(text_embeds=tf)
        vf, _ = model.get_vision_embeds(image)
        ve = model.get_features(image_embeds=vf)
    # bf16 GEMM operands: features within 2e-2 of fp32 (unit vectors), ITM logits within 5e-2
    assert np.abs(ve.cpu().numpy() - GOLD["image_embeds"]).max() < 2e-2
    assert np.abs(te.cpu().numpy() - GOLD["text_embeds"]).max() < 2e-2
    # candidate sets come from the top-k of the similarities: re-rank the REFERENCE's candidates (feed its fp32 features)
    # so that a near-tie in bf16 cannot swap a candidate and turn the comparison into a set mismatch
    gi, gt = torch.from_numpy(GOLD["image_embeds"]).to(dev), torch.from_numpy(GOLD["text_embeds"]).to(dev)
    for qpp in (16, 3):
        s_i2t, s_t2i = mr.rerank_scores(model, vf, gi, tf, atts, gt, K_TEST, queries_per_pass=qpp)
        for got, ref in ((s_i2t.cpu().numpy(), GOLD["score_i2t"]), (s_t2i.cpu().numpy(), GOLD["score_t2i"])):
            assert np.array_equal(got == -100.0, ref == -100.0)
            assert np.abs(got - ref).max() < 5e-2 * max(1.0, np.abs(ref[ref != -100.0]).max())
    # sharded over two ranks: the two halves tile the matrix exactly as Retrieval.py:118-120 cuts it
    parts = [mr.rerank_scores(model, vf, gi, tf, atts, gt, K_TEST, rank=r, world_size=2, reduce=False) for r in range(2)]
    step = NI // 2 + 1
    assert float((parts[0][0][step:] != -100.0).sum()) == 0 and float((parts[1][0][:step] != -100.0).sum()) == 0
    merged = torch.where(parts[0][0] != -100.0, parts[0][0], parts[1][0])
    assert torch.allclose(merged, s_i2t, atol=1e-6)
    # fine-tuning forward with idx soft labels and injected negatives
    model.train(False)
    model.injected_negatives = tuple(GOLD["neg_idx"].tolist())
    B = 4
    loss_itc, loss_itm = model(image[:B], ids[:B], atts[:B], idx=torch.from_numpy(GOLD["idx"]).to(dev))
    assert abs(float(loss_itc) - float(GOLD["loss_itc"])) < 5e-3 * float(GOLD["loss_itc"])
    assert abs(float(loss_itm) - float(GOLD["loss_itm"])) < 5e-3 * max(1.0, float(GOLD["loss_itm"]))
    (loss_itc + loss_itm).backward()
    sq = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5
    assert abs(sq - float(GOLD["total_grad_norm"])) < 3e-2 * float(GOLD["total_grad_norm"])
