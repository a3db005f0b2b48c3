"""graph.GraphedStep: the captured step must BE the eager step.

(1) eval mode (no randomness): at five successive (weights, batch) states the hipGraph replay reproduces the eagerly launched
    step - static input copies, gradients living in the captured arenas, bf16 weight copies re-cast inside the graph after
    every weight update made behind autograd's back (`p.data`).
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


def test_graph_replay_is_the_eager_step(synthetic):
    """Five (weights, batch) states: at each one the eager launch sequence and the hipGraph replay run from the SAME state
    (comparing two separately trained trajectories instead only measures how fast a 4-sample model amplifies gradient
    noise).  Loss: forward is deterministic -> 1e-6.  Gradients: the run-to-run noise of the step itself (fp32 atomics)."""
    graph = importlib.import_module("x2-vlm_amd.graph")
    model, c = _build(synthetic, train=False)
    data = _batches(synthetic, c, 5)
    static = {k: v.clone() for k, v in data[0].items()}
    params = list(model.parameters())
    names = [n for n, _ in model.named_parameters()]

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = model(static["image"], static["text_ids"], static["text_atts"], text_ids_masked=static["text_ids_masked"],
                     masked_pos=static["masked_pos"], masked_ids=static["masked_ids"])
        sum(loss.values()).backward()
        return loss

    step = graph.GraphedStep(fwd_bwd)
    assert step.mode == "hipgraph", step.error
    captured = [p.grad for p in params]                      # the arenas the graph writes on every replay
    seen = []
    for i, b in enumerate(data):
        graph.GraphedStep.copy_inputs(static, b)
        le = {k: float(v) for k, v in fwd_bwd().items()}     # eager launches (re-binds .grad to fresh tensors)
        torch.cuda.synchronize()
        eager = [None if p.grad is None else p.grad.detach().clone() for p in params]
        lg = {k: float(v) for k, v in step().items()}        # the same state through the captured graph
        torch.cuda.synchronize()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-6 * max(1.0, abs(le[k])), (i, k, le[k], lg[k])
        total = sum(float(g.double().pow(2).sum()) for g in eager if g is not None) ** 0.5
        for n, ge, gg in zip(names, eager, captured):
            assert (ge is None) == (gg is None), n
            if ge is not None and "key.bias" not in n:      # key biases: analytically zero gradient, pure rounding noise
                err = float((ge.double() - gg.double()).norm()) / max(float(ge.double().norm()), 1e-2 * total)
                assert err <= 5e-3, (i, n, err)
        seen.append(le)
        with torch.no_grad():                                # move to the next state through p.data (no version bump: what
            for p, g in zip(params, eager):                  # transformers' AdamW does); the graph re-casts its bf16 copies itself
                if g is not None:
                    p.data.add_(g, alpha=-0.02)
    assert seen[0] != seen[-1]


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


def test_segmented_replay_is_the_eager_step(synthetic):
    """graph.SegmentedStep (linear hipGraph segments on two streams, autograd cut at the tower outputs, tied word-embedding
    gradient in one buffer) against the plain eager model(...) + backward at five successive (weights, batch) states."""
    graph = importlib.import_module("x2-vlm_amd.graph")
    model, c = _build(synthetic, train=False)
    data = _batches(synthetic, c, 5)
    static = {k: v.clone() for k, v in data[0].items()}
    params = list(model.parameters())
    names = [n for n, _ in model.named_parameters()]

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = model(static["image"], static["text_ids"], static["text_atts"], text_ids_masked=static["text_ids_masked"],
                     masked_pos=static["masked_pos"], masked_ids=static["masked_ids"])
        sum(loss.values()).backward()
        return loss

    step = graph.SegmentedStep(model, static, clamp_temp=False, vision_cuts=[1])
    assert step.mode == "hipgraph-segments", step.error
    # towers | features | tail + its backward | vision backward in two stages, text backward, deferred weight gradients
    assert sorted(step.graphs) == ["F1", "F2", "Fw", "P", "T", "Tb", "V", "Vb", "Vb1", "Vw"]
    captured = [p.grad for p in params]                      # static tensors the segments write on every replay
    seen = []
    for i, b in enumerate(data):
        graph.SegmentedStep.copy_inputs(static, b)
        le = {k: float(v) for k, v in fwd_bwd().items()}     # eager launches (re-binds .grad to fresh tensors)
        torch.cuda.synchronize()
        eager = [None if p.grad is None else p.grad.detach().clone() for p in params]
        lg = {k: float(v) for k, v in step().items()}        # the same state through the segments (re-attaches .grad)
        torch.cuda.synchronize()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-6 * max(1.0, abs(le[k])), (i, k, le[k], lg[k])
        total = sum(float(g.double().pow(2).sum()) for g in eager if g is not None) ** 0.5
        for n, p, ge, gg in zip(names, params, eager, captured):
            assert (ge is None) == (gg is None), n
            assert p.grad is gg, n
            if ge is not None and "key.bias" not in n:      # key biases: analytically zero gradient, pure rounding noise
                err = float((ge.double() - gg.double()).norm()) / max(float(ge.double().norm()), 1e-2 * total)
                assert err <= 5e-3, (i, n, err)
        seen.append(le)
        with torch.no_grad():
            for p, g in zip(params, eager):
                if g is not None:
                    p.data.add_(g, alpha=-0.02)
    assert seen[0] != seen[-1]


def test_segmented_replays_draw_new_dropout_masks(synthetic):
    graph = importlib.import_module("x2-vlm_amd.graph")
    model, c = _build(synthetic, train=True)
    b = _batches(synthetic, c, 1)[0]
    step = graph.SegmentedStep(model, b)
    assert step.mode == "hipgraph-segments", step.error
    seen = []
    for _ in range(4):
        loss = step()
        seen.append(tuple(round(float(v), 6) for v in loss.values()))
    assert all(all(x == x and abs(x) < 1e4 for x in s) for s in seen)   # finite
    assert len(set(seen)) == 4, seen                                    # same inputs, four different masks


def test_segmented_step_masks_raw_captions_on_the_device(synthetic):
    """SegmentedStep(masking=...): the text segment starts with x2_mask_tokens on the raw (text_ids, text_atts); every replay draws the mask of the
    step counter (= the oracle's masking on the host mirror of the hashed words), and the losses are those of the eager model on that mask."""
    from oracle import masking_oracle as mo
    graph = importlib.import_module("x2-vlm_amd.graph")
    K = importlib.import_module("x2-vlm_amd.kernels")
    model, c = _build(synthetic, train=False)
    full = _batches(synthetic, c, 1)[0]
    raw = {k: full[k].clone() for k in ("image", "text_ids", "text_atts")}
    sub = synthetic.synth_subword_flags(c["vocab"]).to(raw["text_ids"].device)
    mk = synthetic.masking_config(dict(max_masks=c["max_masks"]), sub, seed=77, big_vocab=c["vocab"] > 2000)
    step = graph.SegmentedStep(model, raw, clamp_temp=False, masking=mk)
    assert step.mode == "hipgraph-segments", step.error
    seen = []
    for _ in range(3):
        lg = {k: float(v) for k, v in step().items()}
        torch.cuda.synchronize()
        ep = int(K.DROP_EPOCH.item())
        B, L = raw["text_ids"].shape
        words = K.mask_words(77, ep, B, 4 * L + 64).numpy().astype("uint32")
        kw = {k: v for k, v in mk.items() if k not in ("is_subword", "seed")}
        want = mo.mask_tokens(raw["text_ids"].cpu().numpy(), raw["text_atts"].cpu().numpy(), sub.cpu().numpy(), words, vocab_size=c["vocab"], **kw)
        for key, w_ in zip(("text_ids_masked", "masked_pos", "masked_ids"), want):
            assert (raw[key].cpu().numpy() == w_).all(), key
        le = model(raw["image"], raw["text_ids"], raw["text_atts"], text_ids_masked=raw["text_ids_masked"], masked_pos=raw["masked_pos"],
                   masked_ids=raw["masked_ids"])
        for k in lg:
            if k != "loss_itm":                         # ITM: hard negatives are re-drawn by the eager call
                assert abs(float(le[k]) - lg[k]) <= 1e-5 * max(1.0, abs(lg[k])), (k, float(le[k]), lg[k])
        seen.append(tuple(raw["masked_pos"].flatten().tolist()))
    assert len(set(seen)) == 3                          # a new mask on every replay


def _mixed_parts(synthetic, model, c, shift=0):
    """An image part and a region part (weight 0.5) for the tiny_region geometry, with their injected negatives (device tensors)."""
    bi = synthetic.synth_batch(c["bseed"] + shift, 4, c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=True)
    br = synthetic.synth_region_batch(c["bseed"] + 7 + shift, c["n_images"], c["batch"], c["seq_len"], c["image_res"], 16, c["vocab"], c["max_masks"])
    ni, nr = synthetic.synth_negatives(c["bseed"] + shift, 4), synthetic.synth_negatives(c["bseed"] + 7 + shift, c["batch"])
    return (bi, ni), (br, nr)


def test_mixed_iteration_replayed_accumulates_like_the_oracle(synthetic):
    """graph.MixedStep = Pretrain.run_mixed_iter (Pretrain.py:189-252): an image sub-iteration and a region sub-iteration
    (iter_perc 0.5) of ONE optimizer step, each a chain of replayed hipGraph segments, the second's gradients ADDED to the first's
    static buffers.  Against the CPU oracle's accumulated gradients, for two successive iterations through the same graphs (a
    replay that overwrote instead of accumulating, or accumulated across iterations, fails the second one)."""
    import tempfile
    from oracle import x2vlm_oracle as O
    graph = importlib.import_module("x2-vlm_amd.graph")
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES["tiny_region"]
    model = mp.XVLM(config=model_config("tiny_region", tempfile.mkdtemp()), load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.to(dev).eval()
    (bi, ni), (br, nr) = _mixed_parts(synthetic, model, c)
    si, sr = {k: v.to(dev) for k, v in bi.items()}, {k: v.to(dev) for k, v in br.items()}
    negi = tuple(torch.tensor(n, dtype=torch.int32, device=dev) for n in ni)
    negr = tuple(torch.tensor(n, dtype=torch.int32, device=dev) for n in nr)
    step = graph.MixedStep(model, [dict(batch=si, negatives=negi), dict(batch=sr, negatives=negr, weight=0.5, ret_bbox_loss=True)],
                           clamp_temp=False)
    assert step.mode == "hipgraph-segments", step.error
    # every step object shares ONE set of streams: a parameter's AccumulateGrad node can be pinned to a stream only once, and the
    # region part (fusion layers run twice: in-place accumulation) otherwise accumulates on the image part's streams, outside its
    # own captures - a race that corrupted memory at random before round 4's fix
    assert step.steps[0].sA is step.steps[1].sA and step.steps[0].sB is step.steps[1].sB
    cfg = O.config_from_case(c)
    torch.set_num_threads(8)
    for it in range(2):
        if it:
            (bi, ni), (br, nr) = _mixed_parts(synthetic, model, c, shift=100)
            step.copy_inputs(0, {k: v.to(dev) for k, v in bi.items()})
            step.copy_inputs(1, {k: v.to(dev) for k, v in br.items()})
            for t, n in zip(negi + negr, list(ni) + list(nr)):
                t.copy_(torch.tensor(n, dtype=torch.int32))
            for p in model.parameters():
                p.grad = None                                # optimizer.zero_grad(set_to_none=True) between iterations
        li, lr = step()
        torch.cuda.synchronize()
        sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
        ri, _ = O.xvlm_forward(sd, cfg, bi, ni)
        sum(ri.values()).backward()
        rr, _ = O.xvlm_forward(sd, cfg, br, nr, ret_bbox_loss=True)
        (0.5 * sum(rr.values())).backward()                  # accumulates into the same .grad
        for got, ref in ((li, ri), (lr, rr)):
            assert set(got) == set(ref)
            for k, v in ref.items():
                assert abs(float(got[k]) - float(v)) <= 5e-3 * max(1.0, abs(float(v))), (it, k, float(got[k]), float(v))
        total = sum(float(t.grad.double().pow(2).sum()) for t in sd.values() if t.grad is not None) ** 0.5
        got = dict(model.named_parameters())
        sq = 0.0
        for n, t in sd.items():
            if t.grad is None:
                continue
            g = got[n].grad
            assert g is not None, n
            gn, rn = float(g.double().norm()), float(t.grad.double().norm())
            sq += gn * gn
            assert abs(gn - rn) <= 3e-2 * max(rn, 1e-2 * total), (it, n, gn, rn)
        assert abs(sq ** 0.5 - total) <= 1.2e-2 * total, (it, sq ** 0.5, total)
        assert got["bbox_head.0.weight"].grad is not None    # only the region part produces it


def _text_batch(synthetic, c, seed):
    b = synthetic.synth_batch(seed, 5, c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=True)
    return {k: v for k, v in b.items() if k != "image"}


def _compare_grads(model, sd, tol_norm=3e-2, tol_total=1.2e-2, what=""):
    total = sum(float(t.grad.double().pow(2).sum()) for t in sd.values() if t.grad is not None) ** 0.5
    got = dict(model.named_parameters())
    sq = 0.0
    for n, t in sd.items():
        g = got[n].grad
        if t.grad is None:
            assert g is None or float(g.abs().max()) == 0.0, (what, n)
            continue
        assert g is not None, (what, n)
        gn, rn = float(g.double().norm()), float(t.grad.double().norm())
        sq += gn * gn
        assert abs(gn - rn) <= tol_norm * max(rn, 1e-2 * total), (what, n, gn, rn)
    assert abs(sq ** 0.5 - total) <= tol_total * total, (what, sq ** 0.5, total)


def test_text_only_iteration_replayed_matches_oracle(synthetic):
    """Pretrain.run_text_iter (Pretrain.py:139-157; XVLM.forward(image=None), model_pretrain.py:67-72) as graph.TextOnlyStep: three
    linear segments on the image step's two streams.  Eager forward_text, the replayed step and the oracle's text branch (pinned to
    the reference by tests/golden/tiny_text.npz) agree on the loss and on every gradient; parameters the branch does not touch
    (vision tower, cross-attention, ITC / ITM / bbox heads) get none.  Two iterations through the same graphs."""
    from oracle import x2vlm_oracle as O
    graph = importlib.import_module("x2-vlm_amd.graph")
    model, c = _build(synthetic, train=False)
    cfg = O.config_from_case(c)
    b = _text_batch(synthetic, c, 301)
    static = {k: v.to(dev) for k, v in b.items()}
    step = graph.TextOnlyStep(model, static)
    assert step.mode == "hipgraph-segments", step.error
    assert sorted(step.graphs) == ["XF", "XT", "XTb"]
    torch.set_num_threads(8)
    for it in range(2):
        if it:
            b = _text_batch(synthetic, c, 302)
            step.copy_inputs(static, {k: v.to(dev) for k, v in b.items()})
            for p in model.parameters():
                p.grad = None
        got = step()
        torch.cuda.synchronize()
        sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
        ref, _ = O.xvlm_forward(sd, cfg, b, None)                # no "image" in the batch: forward_text
        assert set(ref) == {"loss_mlm"} == set(got)
        ref["loss_mlm"].backward()
        assert abs(float(got["loss_mlm"]) - float(ref["loss_mlm"])) <= 5e-3 * abs(float(ref["loss_mlm"])), (it, float(got["loss_mlm"]), float(ref["loss_mlm"]))
        _compare_grads(model, sd, what="text it %d" % it)
        named = dict(model.named_parameters())
        assert named["vision_encoder.blocks.0.attn.qkv.weight"].grad is None and named["itm_head.0.weight"].grad is None
        assert named["text_encoder.bert.encoder.layer.%d.crossattention.self.query.weight" % c["fusion_at"]].grad is None
    # the plain eager call of the same branch (what a caller without the step object runs)
    for p in model.parameters():
        p.grad = None
    el = model(None, static["text_ids"], static["text_atts"], text_ids_masked=static["text_ids_masked"], masked_pos=static["masked_pos"],
               masked_ids=static["masked_ids"])
    assert set(el) == {"loss_mlm"}
    assert abs(float(el["loss_mlm"]) - float(got["loss_mlm"])) <= 1e-5 * abs(float(got["loss_mlm"]))


def test_mixed_iteration_with_text_part_and_bbox_only_regions(synthetic):
    """run_mixed_iter with all three kinds of parts the XVLM API can express (Pretrain.py:203-235): image part, region part under
    `regions_use_bbox_only` (only loss_bbox + loss_giou of the region forward enter the backward, Pretrain.py:220-222: per-part
    `loss_keys`), text part (t_loss['loss_mlm'] * iter_perc, :232-235: `text_only`).  Accumulated gradients against the oracle; the
    parts' injected negatives do not leak into the model; per-part settings passed as whole-iteration keywords are refused."""
    import tempfile
    from oracle import x2vlm_oracle as O
    graph = importlib.import_module("x2-vlm_amd.graph")
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES["tiny_region"]
    model = mp.XVLM(config=model_config("tiny_region", tempfile.mkdtemp()), load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.to(dev).eval()
    (bi, ni), (br, nr) = _mixed_parts(synthetic, model, c)
    bt = _text_batch(synthetic, c, 411)
    si, sr, st = ({k: v.to(dev) for k, v in b.items()} for b in (bi, br, bt))
    negi = tuple(torch.tensor(n, dtype=torch.int32, device=dev) for n in ni)
    negr = tuple(torch.tensor(n, dtype=torch.int32, device=dev) for n in nr)
    with pytest.raises(TypeError):
        graph.MixedStep(model, [dict(batch=si, negatives=negi)], total_loss=lambda l: sum(l.values()))
    assert model.injected_negatives is None
    bbox_only = ("loss_bbox", "loss_giou")
    step = graph.MixedStep(model, [dict(batch=si, negatives=negi),
                                   dict(batch=sr, negatives=negr, weight=0.5, ret_bbox_loss=True, loss_keys=bbox_only),
                                   dict(batch=st, weight=0.25, text_only=True)], clamp_temp=False)
    assert step.mode == "hipgraph-segments", step.error
    assert model.injected_negatives is None                  # the parts' negatives are the parts' only
    li, lr, lt = step()
    torch.cuda.synchronize()
    cfg = O.config_from_case(c)
    torch.set_num_threads(8)
    sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
    ri, _ = O.xvlm_forward(sd, cfg, bi, ni)
    rr, _ = O.xvlm_forward(sd, cfg, br, nr, ret_bbox_loss=True)
    rt, _ = O.xvlm_forward(sd, cfg, bt, None)
    (sum(ri.values()) + 0.5 * (rr["loss_bbox"] + rr["loss_giou"]) + 0.25 * rt["loss_mlm"]).backward()      # Pretrain.py:203-247
    for got, ref in ((li, ri), (lr, rr), (lt, rt)):
        assert set(got) == set(ref)                          # every part still REPORTS all its losses (metric_logger)
        for k, v in ref.items():
            assert abs(float(got[k]) - float(v)) <= 5e-3 * max(1.0, abs(float(v))), (k, float(got[k]), float(v))
    _compare_grads(model, sd, what="mixed image + bbox-only region + text")
