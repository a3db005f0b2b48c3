"""End-to-end GPU parity of the HIP training step (x2-vlm_amd.model_pretrain.XVLM, same call as
Pretrain.py:59 / :90 / :143) against the golden vectors the REAL reference produced (tests/golden/*.npz),
on identical seeded weights, batches and injected hard negatives.

Two sets of bounds per case:

(A) against the reference's fp32 CPU run (the golden vectors: the pin).  bf16 GEMM / attention operands against fp32 ones through
    ~36 GEMMs: each bound is >= 1.3x the worst deviation measured on MI355X over all 16 cases (round 6: profiles/r12c_parity_worst.txt; the step is
    bit-reproducible run to run since that round - test_two_runs_give_the_same_bits - so the measured values are constants of the build, and the
    margin is what a different compiler / ROCm release may move), so that a regression shows:
  losses              1e-3 relative at the configurations' own per-GPU batches (cases base_full_b64 = BASELINE.json configs[1] and
                      large_full_b32 = configs[3]: the north-star tolerance; measured 3.1e-4 worst); 6.5e-3 for the 2..8-sample toy batches, whose losses average the same per-sample
                      bf16 operand-rounding noise over 16x fewer samples (measured 4.7e-3, tiny_video)
  activations/logits  pointwise, of the tensor's max-abs (one bf16 rounding is 2^-9 of an element; ~36 GEMMs deep): vision tokens /
                      features 8e-3 (4.5e-3), text tokens 1.8e-2 (1.36e-2) / features 1.6e-2 (1.07e-2), ITC logits 2.4e-2 (1.80e-2), MLM logits 2e-2 (1.35e-2), ITM logits
                      2.5e-2 (1.59e-2); whole-tensor moments 1e-3 (7.1e-4); MLM log-partition 2e-4 (1.18e-4); bbox coordinates 5e-3
                      (2.4e-3 at the real geometry, case base_region)
  parameter grads     per-tensor norm error <= 3.8e-2 of max(its norm, 1e-2 x total gradient norm) (measured 2.83e-2): tensors whose
                      true gradient is ~0 by cancellation (q/k projections of saturated attention, key biases) are held to
                      3.8e-4 of the total norm instead of to their own norm; small tensors stored in full: 4.5e-2 pointwise (3.24e-2);
                      total gradient norm: 2.6e-3 at batch 64 (1.95e-3), 3e-3 for the other full-geometry cases (1.7e-3), 6e-3 for
                      X2VLM-large at batch 32 (4.0e-3: 24 + 18 layers deep), 1.2e-2 for the 32-px toy models (8.7e-3)

(B) against the oracle in its OPERAND-ROUNDING-AWARE mode (oracle.xvlm_forward(round_operands=torch.bfloat16): the same fp32
    program with the matrix-core operands and the bf16-stored tensors rounded at the HIP path's sites, attention rounded the way the
    kernels round it), run on this box's host cores for the CPU-cheap cases (ROUNDED below).  Measured (profiles/r09_parity_worst.txt):
    the AVERAGED quantities come 2-10x closer - losses 9.6e-4 worst (toy batches: 4.7e-3 against the fp32 goldens), per-tensor
    gradient norms 7.4e-3 (3e-2), total gradient norm 1.1e-3 (8.7e-3), moments 3.3e-4 - and are held to ~1.5x that (RB below).
    POINTWISE values do not come closer (1.4e-2 vs 1.6e-2 on ITM logits): with depth the two bf16 runs decorrelate - every fp32-level
    difference flips a bf16 rounding somewhere, a post-LN stack amplifies the flips - so pointwise tightness is asserted where it
    exists: per layer, teacher-forced, at ~1e-3 (tests/test_layerwise_gpu.py, which is also where a wrong epsilon, a dropped bias or
    a one-position mask slip in one of 42 layers shows).
"""
import importlib
import os

import numpy as np
import pytest
import torch

from cases import CASES, forward_kwargs, make_batch, model_config, reduce_out

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# (A) pointwise bounds per tensor against the fp32 goldens, as a fraction of its max-abs (see the table above)
POINTWISE = dict(mlm_lse=2e-4, bbox_coord=5e-3, image_embeds=8e-3, image_feat=8e-3, text_embeds=1.8e-2, text_feat=1.6e-2,
                 itc_logits=2.4e-2, mlm_logits=2e-2, itm_logits=2.5e-2)
# (B) bounds against the operand-rounding-aware oracle
ROUNDED = ("tiny", "tiny_region", "tiny_video", "tiny_text", "tiny_nomatch", "tiny_region_degenerate", "base_shallow",
           "base_shallow_text", "base_shallow_nomatch", "large_shallow", "base_region")
RB = dict(loss=1.5e-3, moments=6e-4, mlm_lse=2e-4, gradnorm=1.2e-2, total=2.2e-3)


def run_case(case, tmpdir, synthetic):
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES[case]
    cfg = model_config(case, str(tmpdir))
    torch.manual_seed(0)
    model = mp.XVLM(config=cfg, load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.cuda().eval()
    cpu_batch = make_batch(synthetic, c)
    batch = {k: v.cuda() for k, v in cpu_batch.items()}
    model.injected_negatives = synthetic.synth_negatives(c["bseed"], c["batch"])
    kw = forward_kwargs(c, batch)
    eng = importlib.import_module("x2-vlm_amd.engine")
    eng.KEEP_MLM_LOGITS = True          # inspection copy of the MLM logits; the loss and its gradient still take the fused path
    try:
        loss = model(None if c.get("text_only") else batch["image"], batch["text_ids"], batch["text_atts"], **kw)
    finally:
        eng.KEEP_MLM_LOGITS = False
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    return model, loss, c, cpu_batch


def model_acts(model, c):
    acts = dict(model.last)
    if "image_feat" in acts:
        acts["itc_logits"] = acts["image_feat"] @ acts["text_feat"].t() / model.temp.detach()
    if c.get("match", True) and not c.get("text_only"):
        acts["itm_logits"] = model.last_itm_logits
    V = c["vocab"]
    ml = model.last_mlm_logits[:, :V].reshape(c["batch"], c["max_masks"], V)
    acts["mlm_logits"] = ml
    # the log-partition the loss was computed from (softmax statistics reduced in the decoder GEMM's epilogue when fused)
    acts["mlm_lse"] = model.last_mlm_lse.reshape(c["batch"], c["max_masks"])
    assert float((acts["mlm_lse"].double() - torch.logsumexp(ml.double(), dim=-1)).abs().max()) < 2e-4
    if c["frames"]:
        acts.pop("image_embeds")        # fixture holds the per-frame encoder output; pooled output is checked via the losses
    return acts


def rounded_oracle_report(case, c, synthetic, cpu_batch, model, loss, acts):
    """(B): the HIP step against the oracle with bf16 roundings at the same operand sites, on this box's host cores."""
    from oracle import x2vlm_oracle as O
    cfg = O.config_from_case(c)
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    osd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
    ob = {k: v for k, v in cpu_batch.items() if not (c.get("text_only") and k == "image")}
    neg = synthetic.synth_negatives(c["bseed"], c["batch"])
    ol, ex = O.xvlm_forward(osd, cfg, ob, neg, ret_bbox_loss=c["region"], ret_match_loss=c.get("match", True),
                            round_operands=torch.bfloat16)
    sum(ol.values()).backward()
    rep = []
    for k, v in loss.items():
        ref = float(ol[k])
        rep.append(("R loss " + k, abs(float(v) - ref) / max(abs(ref), 1e-6) if ref != 0.0 else abs(float(v)), RB["loss"]))
    for name, got in acts.items():
        if name not in ex:
            continue
        ref = ex[name].detach().double()
        g = got.detach().cpu().double().reshape(ref.shape)
        scale = max(float(ref.abs().max()), 1e-6)
        if name == "mlm_lse":
            rep.append(("R act " + name, float((g - ref).abs().max()) / scale, RB["mlm_lse"]))
        if ref.numel() > 4096:
            mg, mr = (np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]) for t in (g, ref))
            rep.append(("R act " + name + "/moments", float(np.abs(mg - mr).max() / max(np.abs(mr).max(), 1e-6)), RB["moments"]))
    sd = dict(model.named_parameters())
    sq_m = sq_o = 0.0
    norms = {}
    for name, t in osd.items():
        if t.grad is not None:
            norms[name] = float(t.grad.double().norm())
            sq_o += norms[name] ** 2
    total = sq_o ** 0.5
    for name, t in osd.items():
        g = sd[name].grad
        if t.grad is None or float(t.grad.abs().max()) == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, name
            continue
        assert g is not None, "no gradient for " + name
        gd = g.detach().cpu().double()
        n = float(gd.norm())
        sq_m += n * n
        if name != "temp":      # d loss / d temp is a difference of two large sums over the B x B similarity matrix: 3e-2 measured, left to (A)
            rep.append(("R gradnorm " + name, abs(n - norms[name]) / max(norms[name], 1e-2 * total), RB["gradnorm"]))
    rep.append(("R total_grad_norm", abs(sq_m ** 0.5 - total) / total, RB["total"]))
    return rep


@pytest.mark.parametrize("case", ["tiny", "tiny_region", "tiny_video", "tiny_text", "tiny_nomatch", "tiny_region_degenerate",
                                  "base_shallow", "base_shallow_text", "base_shallow_nomatch", "large_shallow", "base_full",
                                  "base_full_b64", "large_full", "large_full_b32", "video_full", "base_region"])
def test_step_matches_reference(case, tmp_path, synthetic):
    gold = np.load(os.path.join(GOLD, case + ".npz"))
    model, loss, c, cpu_batch = run_case(case, tmp_path, synthetic)
    assert sorted(loss) == sorted(k for k in gold.files if k.startswith("loss_")), sorted(loss)
    report = []
    for k, v in loss.items():
        ref = float(gold[k])
        if ref == 0.0:            # loss_itm with ret_match_loss=False (model_pretrain.py:52), loss_giou of a degenerate batch (xvlm.py:945)
            assert float(v) == 0.0, (k, float(v))
            continue
        # north_star's 1e-3 at the configurations' own per-GPU batches (base: 64, large: 32); 5e-3 for the toy batches
        report.append(("loss " + k, abs(v.item() - ref) / max(abs(ref), 1e-6), 1e-3 if c["batch"] >= 32 else 6.5e-3))
    full = case.startswith("tiny")
    acts = model_acts(model, c)
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
        if ref <= 0:              # no gradient in the reference (-1), or an exactly zero one
            assert g is None or float(g.abs().max()) == 0.0, name
            continue
        assert g is not None, "no gradient for " + name
        assert bool(torch.isfinite(g).all()), "non-finite gradient for " + name
        n = float(g.double().norm())
        sq += n * n
        # `temp`: d loss_itc / d temp sums the B x B similarity matrix with weights of both signs amplified by 1 / temp^2 = 200 - at some
        # (weights, batch) it all but cancels (base_shallow_nomatch: 0.016 of a total norm of 22.1) and what is left is noise: held to 6e-4
        # of the total norm (measured 4.3e-4) instead of 3e-4 like the other cancellation-dominated tensors
        report.append(("gradnorm " + name, abs(n - ref) / max(ref, (2e-2 if name == "temp" else 1e-2) * total), 3.8e-2))
    for k in gold.files:
        if k.startswith("grad/"):
            name = k[len("grad/"):]
            ref = gold[k].astype(np.float64)
            if sd[name].grad is None:
                assert float(np.abs(ref).max()) == 0.0, name
                continue
            got = sd[name].grad.detach().cpu().double().numpy()
            fl = (2e-2 if name == "temp" else 1e-2) * total / max(ref.size, 1) ** 0.5
            report.append(("grad " + name, float(np.abs(got - ref).max() / max(np.abs(ref).max(), fl)), 4.5e-2))
    # the 42-layer X2VLM-large at batch 32 measured 4.0e-3 (profiles/r05e_parity_large_b32.txt): its own bound, 6e-3
    # ret_match_loss=False at the real geometry (round 5): two loss terms instead of three feed the total - 3.2e-3 measured, bound 5e-3
    gtol = (2.6e-3 if c["batch"] >= 64 else 1.2e-2 if c["image_res"] < 64 else 6e-3 if case == "large_full_b32" else
            5e-3 if not c.get("match", True) else 3e-3)
    report.append(("total_grad_norm", abs(sq ** 0.5 - total) / total, gtol))
    if case in ROUNDED and os.environ.get("X2_ROUNDED_ORACLE", "1") == "1":
        report += rounded_oracle_report(case, c, synthetic, cpu_batch, model, loss, acts)
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
    if os.environ.get("X2_PARITY_NO_ASSERT") == "1":      # measuring run (profiles/*_parity_worst.txt): report, do not gate
        return
    assert not bad, "%d checks out of tolerance, worst: %s" % (len(bad), bad[:5])


@pytest.mark.parametrize("case", ["base_shallow", "base_region", "tiny_video", "large_shallow"])
def test_two_runs_give_the_same_bits(case, tmp_path, synthetic):
    """Run-to-run reproducibility (round 6): no accumulation of the step depends on the order in which workgroups or atomics arrive - the embedding
    scatter, the cls-token gradient and the patch-embedding weight gradient were the last ones - so two runs of the same step from the same state
    give bit-identical losses and gradients (eval mode, injected negatives: nothing random left).  large_shallow (N = 577): the long one-pass
    attention backward, whose dQ partials travel through a workspace in a fixed order."""
    runs = []
    for i in range(2):
        model, loss, c, _ = run_case(case, tmp_path / str(i), synthetic)
        runs.append(({k: v.detach().clone() for k, v in loss.items()},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        del model, loss
    (l0, g0), (l1, g1) = runs
    assert l0.keys() == l1.keys() and g0.keys() == g1.keys()
    for k in l0:
        assert torch.equal(l0[k], l1[k]), k
    differ = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not differ, "%d gradient tensors differ between two runs, e.g. %s" % (len(differ), differ[:5])
