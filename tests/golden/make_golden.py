#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference) on CPU fp32.

Runs only in the build container (the reference never travels to the GPU box).  For each case
in cases.py it builds models.model_pretrain.XVLM, fills it with the seeded synthetic weights,
puts it in eval() (dropout / DropPath off; grads still flow), injects the seeded hard-negative
indices in place of the torch.multinomial draws (xvlm.py:845-855), runs forward + backward of
the summed losses exactly as Pretrain.py:run_image_iter / run_region_iter do, and stores the
outputs as tests/golden/<case>.npz.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so these fixtures are what pins the oracle.

usage:  python tests/golden/make_golden.py [case ...]
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import reference_shims  # noqa: E402
from cases import CASES, SMALL, forward_kwargs, make_batch, model_config, reduce_out  # noqa: E402

synthetic = importlib.import_module("x2-vlm_amd.synthetic")

def run_case(name):
    c = CASES[name]
    torch.set_num_threads(8)
    cfg = model_config(name, "/tmp/x2golden")
    from models.model_pretrain import XVLM
    model = XVLM(config=cfg, load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model.eval()
    if c.get("checkpoint_blocks"):
        # memory only (the B = 32 X2VLM-large case keeps ~70 GB of fp32 activations otherwise): every vision block of the
        # reference runs under torch.utils.checkpoint - the same modules, the same arithmetic (eval mode: deterministic),
        # their internals recomputed in the backward instead of stored
        from torch.utils.checkpoint import checkpoint
        for blk in model.vision_encoder.blocks:
            blk.forward = (lambda *a, _f=blk.forward, **k: checkpoint(_f, *a, use_reentrant=False, **k))

    batch = make_batch(synthetic, c)
    neg = synthetic.synth_negatives(c["bseed"], c["batch"])
    model.get_hard_negatives = lambda *a, **k: neg

    cap = {}
    full = name.startswith("tiny")

    def hook(key, sel=lambda o: o, first_only=True):
        def fn(_m, _i, o):
            if first_only and key in cap:
                return
            cap[key] = sel(o).detach().clone()
        return fn
    model.vision_encoder.register_forward_hook(
        hook("image_embeds", lambda o: o[0] if isinstance(o, tuple) else o))
    model.text_encoder.bert.register_forward_hook(hook("text_embeds", lambda o: o.last_hidden_state))
    model.vision_proj.register_forward_hook(hook("vision_proj"))
    model.text_proj.register_forward_hook(hook("text_proj"))
    model.itm_head.register_forward_hook(hook("itm_logits"))
    model.text_encoder.cls.register_forward_hook(hook("mlm_logits"))
    model.bbox_head.register_forward_hook(hook("bbox_logits"))

    t0 = time.time()
    kw = forward_kwargs(c, batch)
    # image=None: Pretrain.run_text_iter's call (Pretrain.py:143)
    loss = model(None if c.get("text_only") else batch["image"], batch["text_ids"], batch["text_atts"], **kw)
    total = sum(loss.values())
    total.backward()
    dt = time.time() - t0

    out = {}
    for k, v in loss.items():
        out[k] = np.array(v.item(), dtype=np.float64)
    if "vision_proj" in cap:                     # absent in a text-only iteration
        img_feat = torch.nn.functional.normalize(cap["vision_proj"], dim=-1)
        txt_feat = torch.nn.functional.normalize(cap["text_proj"], dim=-1)
        cap["image_feat"], cap["text_feat"] = img_feat, txt_feat
        cap["itc_logits"] = img_feat @ txt_feat.t() / model.temp.detach()
        del cap["vision_proj"], cap["text_proj"]
    if "bbox_logits" in cap:
        cap["bbox_coord"] = cap.pop("bbox_logits").sigmoid()
    if "mlm_logits" in cap:
        cap["mlm_lse"] = torch.logsumexp(cap["mlm_logits"].double(), dim=-1).float()
    for k, v in cap.items():
        for kk, vv in reduce_out(v, full).items():
            out["act/%s/%s" % (k, kk)] = vv
    sq = 0.0
    for n, p in model.named_parameters():
        if p.grad is None:
            out["gradnorm/" + n] = np.array(-1.0)
            continue
        g = p.grad.detach().double()
        out["gradnorm/" + n] = np.array(g.norm().item())
        sq += float((g * g).sum())
        if g.numel() <= SMALL:
            out["grad/" + n] = g.numpy().astype(np.float32)
    out["total_grad_norm"] = np.array(sq ** 0.5)
    out["neg_idx"] = np.array(neg, dtype=np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-13s %.1fs  %s  |grad|=%.5f  -> %s (%.0f KB)" % (
        name, dt, {k: round(float(v), 6) for k, v in loss.items()}, sq ** 0.5,
        os.path.relpath(path, ROOT), os.path.getsize(path) / 1024), flush=True)


if __name__ == "__main__":
    reference_shims.install()
    reference_shims.ensure_process_group()
    for case in (sys.argv[1:] or list(CASES)):
        run_case(case)
