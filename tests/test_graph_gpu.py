"""graph.GraphedStep: the captured step must BE the eager step.

(1) eval mode (no randomness): five optimisation steps driven by hipGraph replays + an optimizer outside the graph follow
    the eagerly launched loop to the run-to-run noise of the step itself (same kernels, same order per stream) - static input copies, gradients
    living in the captured arenas, bf16 weight copies re-cast inside the graph after every weight update.  The update is
    plain SGD through `p.data` (linear in the gradient): AdamW's first steps are ~lr * sign(g), which turns the 1e-7
    run-to-run noise of atomically accumulated gradients into sign flips and would need a loose tolerance.
(2) train mode: replays draw a new dropout mask each time (device-resident epoch), losses stay finite and differ from
    replay to replay on constant inputs; launch mode is reported."""
import importlib
import tempfile

import pytest
import torch

from cases import CASES, model_config

pytestmark = pytest.mark.gpu
dev = "cuda"


def _build(synthetic, train):
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES["tiny"]
    model = mp.XVLM(config=model_config("tiny", tempfile.mkdtemp()), load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.to(dev).train(train)
    # device tensors: a host list would be copied host->device inside the step, which cannot be captured
    model.injected_negatives = tuple(torch.tensor(n, dtype=torch.int32, device=dev) for n in synthetic.synth_negatives(c["bseed"], c["batch"]))
    return model, c


def _batches(synthetic, c, n):
    return [{k: v.to(dev) for k, v in synthetic.synth_batch(c["bseed"] + i, c["batch"], c["seq_len"], c["image_res"], c["vocab"],
                                                            c["max_masks"], ragged=False).items()} for i in range(n)]


def _loop(synthetic, use_graph):
    graph = importlib.import_module("x2-vlm_amd.graph")
    model, c = _build(synthetic, train=False)
    data = _batches(synthetic, c, 5)
    static = {k: v.clone() for k, v in data[0].items()}
    params = list(model.parameters())

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = model(static["image"], static["text_ids"], static["text_atts"], text_ids_masked=static["text_ids_masked"],
                     masked_pos=static["masked_pos"], masked_ids=static["masked_ids"])
        sum(loss.values()).backward()
        return loss

    step = graph.GraphedStep(fwd_bwd, enabled=use_graph)
    assert step.mode == ("hipgraph" if use_graph else "eager"), step.error
    # the warm-up / capture passes ran fwd+bwd on batch 0 without an optimizer step: parameters are still the initial ones
    out = []
    for b in data:
        graph.GraphedStep.copy_inputs(static, b)
        loss = step()
        with torch.no_grad():
            for p in params:                       # through .data: no version bump (what transformers' AdamW does)
                if p.grad is not None:
                    p.data.add_(p.grad, alpha=-0.02)
        out.append({k: float(v) for k, v in loss.items()})
    return out, [p.detach().clone() for p in params]


def test_graph_replay_is_the_eager_step(synthetic):
    eager, pe = _loop(synthetic, use_graph=False)
    graphed, pg = _loop(synthetic, use_graph=True)
    for i, (a, b) in enumerate(zip(eager, graphed)):
        for k in a:
            # step 0 (identical weights): forward is deterministic; later steps inherit the run-to-run noise of the gradients
            # (fp32 atomics in the row scatter-adds and bias sums flip bf16 roundings downstream: probes/race_probe.py shows the
            # same 1e-4..2e-3 spread between two EAGER runs of one schedule)
            tol = 1e-6 if i == 0 else 2e-3
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(a[k])), (i, k, a[k], b[k])
    assert eager[0] != eager[-1]                                       # the loop really trains (weights and batches change)
    for a, b in zip(pe, pg):
        assert float((a - b).abs().max()) <= 2e-3 * max(1.0, float(a.abs().max()))


def test_replays_draw_new_dropout_masks(synthetic):
    graph = importlib.import_module("x2-vlm_amd.graph")
    model, c = _build(synthetic, train=True)
    b = _batches(synthetic, c, 1)[0]
    params = list(model.parameters())

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = model(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                     masked_ids=b["masked_ids"])
        sum(loss.values()).backward()
        return loss

    step = graph.GraphedStep(fwd_bwd)
    assert step.mode == "hipgraph", step.error
    seen = []
    for _ in range(4):
        loss = step()
        seen.append(tuple(round(float(v), 6) for v in loss.values()))
    assert all(all(x == x and abs(x) < 1e4 for x in s) for s in seen)   # finite
    assert len(set(seen)) == 4, seen                                    # same inputs, four different masks
