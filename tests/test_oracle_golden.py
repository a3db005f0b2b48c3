"""Pins the oracle (oracle/x2vlm_oracle.py) to golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only.  Tolerances: both sides are fp32 on CPU, differences
come from summation order only -> 2e-5 relative to each tensor's scale (losses 1e-5 rel)."""
import os

import numpy as np
import pytest
import torch

from cases import CASES, reduce_out
from oracle import x2vlm_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_batch(synthetic, c):
    if c["region"]:
        return synthetic.synth_region_batch(c["bseed"], c["n_images"], c["batch"], c["seq_len"],
                                            c["image_res"], 16, c["vocab"], c["max_masks"])
    return synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"],
                                 c["max_masks"], ragged=c["ragged"], frames=c["frames"])


def close(a, b, rtol, what, floor=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), floor)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "%s: max err / scale = %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("case", ["tiny", "tiny_region", "tiny_video", "base_shallow", "large_shallow", "base_region",
                                  pytest.param("base_full", marks=pytest.mark.slow),
                                  pytest.param("base_full_b64", marks=pytest.mark.slow),
                                  pytest.param("large_full", marks=pytest.mark.slow),
                                  pytest.param("video_full", marks=pytest.mark.slow)])
def test_oracle_matches_reference(case, synthetic):
    c = CASES[case]
    gold = np.load(os.path.join(GOLD, case + ".npz"))
    cfg = O.config_from_case(c)
    sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
    batch = make_batch(synthetic, c)
    neg = synthetic.synth_negatives(c["bseed"], c["batch"])
    assert np.array_equal(np.array(neg), gold["neg_idx"])
    torch.set_num_threads(8)
    losses, ex = O.xvlm_forward(sd, cfg, batch, neg, ret_bbox_loss=c["region"])
    sum(losses.values()).backward()
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
    assert checked >= 8
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
