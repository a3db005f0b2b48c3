"""Checkpoint compatibility ON THE DEVICE (reference models/xvlm.py:579-613, models/beit2.py:473-601,
653-754): a 224-px checkpoint is loaded into a 384-px HIP model through XVLMBase.load_pretrained (relative-position tables
resampled 27x27 -> 47x47 offsets per block), the model runs forward + backward on the GPU, and losses / parameter gradients
are compared with the CPU oracle fed the SAME loaded state dict.  A second load_state_dict then changes every weight: the
bf16 copies the kernels read (engine.WeightBank) must follow.  Tolerances: those of the large_shallow case in
tests/test_model_gpu.py (toy batch, N = 577)."""
import importlib
import os

import pytest
import torch

from cases import CASES, model_config

pytestmark = pytest.mark.gpu


def _losses(model, b, neg, **kw):
    model.injected_negatives = neg
    return model(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                 masked_ids=b["masked_ids"], **kw)


def _oracle(O, cfg, state, batch, neg, backward=True):
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = {n: state[n].detach().cpu().float().clone().requires_grad_(backward) for n in O.parameter_shapes(cfg)}
    with torch.set_grad_enabled(backward):
        loss, _ = O.xvlm_forward(sd, cfg, batch, neg)
        if backward:
            sum(loss.values()).backward()
    return {k: float(v) for k, v in loss.items()}, sd


def test_224px_checkpoint_in_384px_model_matches_oracle_on_the_loaded_weights(tmp_path, synthetic):
    from oracle import x2vlm_oracle as O
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    ck = importlib.import_module("x2-vlm_amd.checkpoint")
    eng = importlib.import_module("x2-vlm_amd.engine")
    c224 = CASES["base_shallow"]
    CASES["_ckpt384"] = c384 = dict(c224, image_res=384, batch=2)
    try:
        cfg224, cfg384 = model_config("base_shallow", str(tmp_path / "a")), model_config("_ckpt384", str(tmp_path / "b"))
    finally:
        del CASES["_ckpt384"]
    m224 = mp.XVLM(config=cfg224, load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(m224, 101)
    path = str(tmp_path / "x2vlm_224.th")
    torch.save({"model": m224.state_dict()}, path)
    m384 = mp.XVLM(config=cfg384, load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(m384, 202)            # what the checkpoint does not provide keeps these values
    m384 = m384.cuda().eval()
    msg = m384.load_pretrained(path, cfg384, is_domain_pretrain=False)
    # the resampled tables are the product's own resampler applied to the checkpoint's (goldens: tests/test_checkpoint_cpu.py)
    t224 = m224.vision_encoder.blocks[1].attn.relative_position_bias_table.detach()
    t384 = m384.vision_encoder.blocks[1].attn.relative_position_bias_table.detach().cpu()
    assert t384.shape == (47 * 47 + 3, 12) and torch.equal(t384, ck.interpolate_rel_pos_bias(t224, 47 * 47 + 3, (24, 24)))
    assert torch.equal(m384.vision_encoder.blocks[0].mlp.fc1.weight.detach().cpu(), m224.vision_encoder.blocks[0].mlp.fc1.weight.detach())
    assert torch.equal(m384.itm_head[0].weight.detach().cpu(), m224.itm_head[0].weight.detach())
    assert any(k.startswith("text_encoder.bert.") for k in msg.missing_keys)        # xvlm.py:433-439: infix stripped, MLM-headed encoder keeps its init

    cfg = O.config_from_case(c384)
    batch = synthetic.synth_batch(303, c384["batch"], c384["seq_len"], 384, c384["vocab"], c384["max_masks"], ragged=True)
    neg = synthetic.synth_negatives(303, c384["batch"])
    gb = {k: v.cuda() for k, v in batch.items()}
    state = {k: v.detach().cpu().clone() for k, v in m384.state_dict().items()}
    loss = _losses(m384, gb, neg)
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    ref, sd = _oracle(O, cfg, state, batch, neg)
    for k, v in ref.items():
        assert abs(float(loss[k]) - v) <= 5e-3 * max(abs(v), 1e-6), (k, float(loss[k]), v)
    total = float(torch.sqrt(sum((t.grad.double() ** 2).sum() for t in sd.values() if t.grad is not None)))
    got = dict(m384.named_parameters())
    sq, bad = 0.0, []
    for n, t in sd.items():
        if t.grad is None:
            continue
        g = got[n].grad
        assert g is not None, "no gradient for " + n
        gn, rn = float(g.double().norm()), float(t.grad.double().norm())
        sq += gn * gn
        if abs(gn - rn) > 3e-2 * max(rn, 1e-2 * total):
            bad.append((n, gn, rn))
    assert not bad, bad[:5]
    assert abs(sq ** 0.5 - total) <= 3e-3 * total, (sq ** 0.5, total)

    # ---- weights replaced through load_state_dict while bf16 copies of the old ones are cached
    with torch.no_grad():
        before = {k: float(v) for k, v in _losses(m384, gb, neg).items()}     # forward only: the bank now holds fresh copies
    assert eng.BANK._c, "the weight bank is empty: nothing would be stale"
    state2 = {k: (v * 1.25 if v.is_floating_point() and v.dim() >= 2 else v) for k, v in state.items()}
    m384.load_state_dict(state2)
    with torch.no_grad():
        after = {k: float(v) for k, v in _losses(m384, gb, neg).items()}
    ref2, _ = _oracle(O, cfg, state2, batch, neg, backward=False)
    for k, v in ref2.items():
        assert abs(after[k] - v) <= 5e-3 * max(abs(v), 1e-6), ("stale bf16 weight copies?", k, after[k], v, before[k])
    assert any(abs(after[k] - before[k]) > 1e-2 * max(abs(before[k]), 1e-6) for k in after), (before, after)
