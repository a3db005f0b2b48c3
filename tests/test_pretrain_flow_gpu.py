"""The reference's UNPATCHED training iteration (Pretrain.run_image_iter, /root/reference/Pretrain.py:54-76: model(...) -> optimizer.zero_grad() ->
accelerator.backward_step(sum of the losses) -> accelerator.optimizer_step) through what RocmDDPAccelerator.set_up returns: from the second call with a
shape signature the wrapper replays graph.SegmentedStep segments (forward AND backward) and backward_step only publishes the gradients.  Checked against
the eager module at the same weights / batch; a weighted sum of the losses falls back to an eager recomputation."""
import importlib
import json
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from tests.golden.cases import CASES, model_config, bert_config_dict

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_image_iter(model, image_batch, optimizer, accelerator, weights=(1.0, 1.0, 1.0)):
    """The call sequence of Pretrain.run_image_iter (no optimizer.step: the test compares gradients at fixed weights)."""
    image, batch = image_batch[0].to("cuda", non_blocking=True), [t.to("cuda") for t in image_batch[1:]]
    text_ids, text_atts, text_ids_masked, masked_pos, masked_ids = batch
    loss = model(image, text_ids, text_atts, text_ids_masked=text_ids_masked, masked_pos=masked_pos, masked_ids=masked_ids, ret_match_loss=True)
    optimizer.zero_grad()
    if weights == (1.0, 1.0, 1.0):
        loss_in_total = loss["loss_itc"] + loss["loss_itm"] + loss["loss_mlm"]
    else:
        loss_in_total = weights[0] * loss["loss_itc"] + weights[1] * loss["loss_itm"] + weights[2] * loss["loss_mlm"]
    accelerator.backward_step(loss_in_total, optimizer)
    norm = accelerator.optimizer_step(optimizer, model, 1.0)
    return {k: v.item() for k, v in loss.items()}, norm


def _worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    synthetic = importlib.import_module("x2-vlm_amd.synthetic")
    mp_ = importlib.import_module("x2-vlm_amd.model_pretrain")
    acc = importlib.import_module("x2-vlm_amd.accelerator")
    optim = importlib.import_module("x2-vlm_amd.optim")
    c = CASES["tiny"]
    wd = tempfile.mkdtemp()
    cfg = model_config("tiny", wd)
    bc = dict(bert_config_dict(c), hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)     # train mode without randomness: exact comparisons
    with open(os.path.join(cfg["text_encoder"], "config.json"), "w") as f:
        json.dump(bc, f)
    cfg.update(drop_path_rate=0.0, dropout=0.0)
    model = mp_.XVLM(config=cfg, load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model.train()
    opt = optim.create_optimizer(dict(lr=1e-4, weight_decay=0.01, lr_mult=2), model)
    a = acc.RocmDDPAccelerator(dict(RNG_SEED=7), None)
    ddp, opt, _ = a.set_up(model, opt, None, local_rank=0, world_size=1, rank=0)
    ddp.module.injected_negatives = tuple(torch.tensor(n, dtype=torch.int32, device="cuda") for n in synthetic.synth_negatives(c["bseed"], c["batch"]))
    names = [n for n, _ in ddp.module.named_parameters()]
    params = [p for _, p in ddp.module.named_parameters()]
    out = dict(modes=[], loss_err=[], grad_err=[], norms=[])
    try:
        for it in range(5):
            b = synthetic.synth_batch(c["bseed"] + it, c["batch"], c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=False)
            image_batch = [b[k] for k in ("image", "text_ids", "text_atts", "text_ids_masked", "masked_pos", "masked_ids")]
            w = (1.0, 0.5, 2.0) if it == 4 else (1.0, 1.0, 1.0)
            losses, norm = run_image_iter(ddp, image_batch, opt, a, w)
            torch.cuda.synchronize()
            out["modes"].append(ddp.last_mode)
            got = [None if p.grad is None else p.grad.detach().clone() for p in params]
            # the same state through the eager module
            for p in params:
                p.grad = None
            d = {k: v.cuda() for k, v in b.items()}
            le = ddp.module(d["image"], d["text_ids"], d["text_atts"], text_ids_masked=d["text_ids_masked"], masked_pos=d["masked_pos"],
                            masked_ids=d["masked_ids"], ret_match_loss=True)
            (w[0] * le["loss_itc"] + w[1] * le["loss_itm"] + w[2] * le["loss_mlm"]).backward()
            torch.cuda.synchronize()
            out["loss_err"].append(max(abs(float(le[k]) - losses[k]) / max(1.0, abs(float(le[k]))) for k in losses))
            total = sum(float(p.grad.double().pow(2).sum()) for p in params if p.grad is not None) ** 0.5
            worst = 0.0
            for n, p, g in zip(names, params, got):
                assert (p.grad is None) == (g is None), n
                if g is not None and "key.bias" not in n:
                    worst = max(worst, float((p.grad.double() - g.double()).norm()) / max(float(p.grad.double().norm()), 1e-2 * total))
            out["grad_err"].append(worst)
            out["norms"].append((norm, min(total, 1e30)))
            del le                                     # no autograd graph of the model may outlive the iteration: the capture in the next call pins every
            #                                            AccumulateGrad node to its segment's stream (in Pretrain.py every model call goes through the wrapper)
            with torch.no_grad():                      # move the weights: the replays must follow them (bf16 copies re-cast inside the step)
                for p in params:
                    if p.grad is not None:
                        p.add_(p.grad, alpha=-0.02)
            importlib.import_module("x2-vlm_amd.engine").BANK.invalidate()
        ret[0] = out
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def test_unpatched_run_image_iter_replays_segments():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), ret), nprocs=1, join=True)
    r = ret[0]
    # first sight of the shapes: eager; from the second call on: replayed segments; the weighted iteration: replay + eager recomputation
    assert r["modes"] == ["eager-fused", "hipgraph-segments", "hipgraph-segments", "hipgraph-segments", "hipgraph-segments"], r["modes"]
    assert max(r["loss_err"]) <= 1e-5, r["loss_err"]
    assert max(r["grad_err"]) <= 5e-3, r["grad_err"]                  # the bound of test_segmented_replay_is_the_eager_step
    for norm, total in r["norms"][:4]:
        assert abs(norm - total) <= 2e-3 * total, (norm, total)       # optimizer_step saw the published gradients
