"""Pins the oracle (oracle/x2vlm_oracle.py) to golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only.  Tolerances: both sides are fp32 on CPU, differences
come from summation order only -> 2e-5 relative to each tensor's scale (losses 1e-5 rel)."""
import os

import numpy as np
import pytest
import torch

from cases import CASES, reduce_out
from cases import make_batch as case_batch
from oracle import x2vlm_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_batch(synthetic, c):
    b = case_batch(synthetic, c)
    if c.get("text_only"):
        b = {k: v for k, v in b.items() if k != "image"}          # Pretrain.run_text_iter: model(None, text_ids, ...)
    return b


def oracle_step(c, synthetic, round_operands=None):
    cfg = O.config_from_case(c)
    sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
    batch = make_batch(synthetic, c)
    neg = synthetic.synth_negatives(c["bseed"], c["batch"])
    losses, ex = O.xvlm_forward(sd, cfg, batch, neg, ret_bbox_loss=c["region"], ret_match_loss=c.get("match", True),
                                round_operands=round_operands)
    sum(losses.values()).backward()
    return sd, losses, ex, neg


def close(a, b, rtol, what, floor=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), floor)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "%s: max err / scale = %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("case", ["tiny", "tiny_region", "tiny_video", "base_shallow", "large_shallow", "base_region",
                                  "tiny_text", "base_shallow_text", "tiny_nomatch", "base_shallow_nomatch", "tiny_region_degenerate",
                                  pytest.param("base_full", marks=pytest.mark.slow),
                                  pytest.param("base_full_b64", marks=pytest.mark.slow),
                                  pytest.param("large_full", marks=pytest.mark.slow),
                                  pytest.param("video_full", marks=pytest.mark.slow)])
def test_oracle_matches_reference(case, synthetic):
    c = CASES[case]
    gold = np.load(os.path.join(GOLD, case + ".npz"))
    torch.set_num_threads(8)
    sd, losses, ex, neg = oracle_step(c, synthetic)
    assert np.array_equal(np.array(neg), gold["neg_idx"])
    assert sorted(losses) == sorted(k for k in gold.files if k.startswith("loss_"))
    if c.get("degenerate"):
        assert float(gold["loss_giou"]) == 0.0 and float(losses["loss_giou"]) == 0.0        # the reference's early-out, xvlm.py:943-946
    if not c.get("match", True):
        assert float(gold["loss_itm"]) == 0.0 and float(losses["loss_itm"]) == 0.0          # model_pretrain.py:52
    for k, v in losses.items():
        assert abs(v.item() - float(gold[k])) <= 1e-5 * max(1.0, abs(float(gold[k]))), (k, v.item(), float(gold[k]))
    full = case.startswith("tiny")
    checked = 0
    for k in gold.files:
        if not k.startswith("act/"):
            continue
        _, name, kind = k.split("/")
        got = reduce_out(ex[name], full)[kind]
        close(got, gold[k], 2e-5 if kind != "moments" else 1e-4, k)
        checked += 1
    assert checked >= (2 if c.get("text_only") else 6)
    sq = 0.0
    # key biases have an analytically zero gradient (softmax shift invariance): their stored
    # grads are ~1e-8 rounding noise, so every grad is compared on a floor tied to the total norm
    gfloor = 1e-3 * float(gold["total_grad_norm"])
    for k in gold.files:
        if k.startswith("gradnorm/"):
            name = k[len("gradnorm/"):]
            if name == "text_encoder.cls.predictions.decoder.weight":
                continue
            g = sd[name].grad
            if float(gold[k]) < 0:
                assert g is None or float(g.abs().max()) == 0.0, name
                continue
            assert g is not None, name
            n = g.double().norm().item()
            sq += n * n
            assert abs(n - float(gold[k])) <= 5e-5 * max(float(gold[k]), gfloor), (name, n, float(gold[k]))
        elif k.startswith("grad/"):
            name = k[len("grad/"):]
            close(sd[name].grad.numpy(), gold[k], 5e-5, k, floor=gfloor)
    assert abs(sq ** 0.5 - float(gold["total_grad_norm"])) <= 2e-5 * float(gold["total_grad_norm"])


@pytest.mark.parametrize("case", ["tiny", "tiny_region", "tiny_text"])
def test_operand_rounding_mode(case, synthetic):
    """The rounded mode is the SAME program with bf16 roundings at the HIP path's operand sites: switched off (the default, or
    an explicit rounding(None)) it is bit-identical to the pinned fp32 program; switched on it moves every loss by a
    bf16-sized amount (not zero, not large) and leaves finite gradients for the same parameters."""
    c = CASES[case]
    torch.set_num_threads(8)
    sd0, l0, _, _ = oracle_step(c, synthetic)
    with O.rounding(None):
        sd1, l1, _, _ = oracle_step(c, synthetic)
    for k in l0:
        assert float(l0[k]) == float(l1[k])
    for n in sd0:
        assert (sd0[n].grad is None) == (sd1[n].grad is None)
        if sd0[n].grad is not None:
            assert torch.equal(sd0[n].grad, sd1[n].grad), n
    sd2, l2, _, _ = oracle_step(c, synthetic, round_operands=torch.bfloat16)
    assert O._ROUND is None                                  # the context restored the default
    moved = 0
    for k in l0:
        a, b = float(l0[k]), float(l2[k])
        assert abs(a - b) <= 2e-2 * max(abs(a), 1e-3), (k, a, b)
        moved += a != b
    assert moved >= 1
    for n in sd0:
        if sd0[n].grad is not None:
            g = sd2[n].grad
            assert g is not None and bool(torch.isfinite(g).all()), n
    tot0 = sum(float(t.grad.double().pow(2).sum()) for t in sd0.values() if t.grad is not None) ** 0.5
    tot2 = sum(float(t.grad.double().pow(2).sum()) for t in sd2.values() if t.grad is not None) ** 0.5
    assert 0 < abs(tot0 - tot2) <= 3e-2 * tot0
