"""Teacher-forced, layer-by-layer parity of the HIP stages against the oracle in its operand-rounding-aware mode.

Why: the whole-model tests (test_model_gpu.py) compare against fp32 goldens through up to 42 layers; bf16 operand rounding alone moves
pointwise values by 1-2 % there, and even with the roundings placed identically (the oracle's rounded mode) the two runs decorrelate with
depth - every fp32-level difference flips a bf16 rounding somewhere, each flip is a bf16-sized perturbation, and a post-LN stack
amplifies them (measured: per layer the two agree to ~1e-3, after 18 layers to ~1e-2; profiles/r09_parity_worst.txt).  A 2 % gate cannot
see a wrong epsilon, a dropped bias or a one-position mask slip in one layer.  Here EVERY layer of every tower gets the ORACLE's own
(rounded) input and the same cotangent, forward and backward, so nothing accumulates and the bounds are ~1e-3:

  forward           max |y - y_ref| <= 4e-3 of max |y_ref|  (measured 2.3e-3; one flipped bf16 rounding on one element is 4e-3 of that
                    element) and ||y - y_ref|| <= 1.5e-3 ||y_ref||  (8.8e-4)
  input gradient    ||dx - dx_ref|| <= 6e-3 ||dx_ref||  (3.9e-3: the image-token gradient of a fusion layer)
  parameter grads   ||dw - dw_ref|| <= 1e-2 max(||dw_ref||, 1e-2 x the layer's total gradient norm) for every parameter tensor (6.2e-3);
                    3e-2 for the query / key projections of the BERT self-attentions (1.8e-2: dS = P (dP - Delta) is a difference of
                    nearly equal terms where attention saturates, so the bf16 dS operand carries a larger relative error)
(bounds ~1.7x the worst measured over the cases below, profiles/r09_parity_worst.txt.)  Mutation check of the gate itself at the
bottom: ONE more masked key position in ONE sequence on the oracle's side moves a text layer's output by 4.8e-2 - 12x the bound.

Covers: patch embedding + cls (stem), every vision block (rel-pos bias attention, layer scale), fc_norm + pooling (head), BERT
embeddings, every text layer (padding mask), every fusion layer (cross-attention to the image tokens, d(image tokens)), MLM head + tied
decoder + CE, ITM / bbox MLP heads, projection + normalise + ITC."""
import importlib
import math
import os
import tempfile

import pytest
import torch

from cases import CASES, make_batch, model_config

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
FWD_MAX, FWD_L2, DX_L2, DW_L2, DW_QK_L2 = 4e-3, 1.5e-3, 6e-3, 1e-2, 3e-2


def _l2(a, b, floor=0.0):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm()) / max(float(b.norm()), floor, 1e-30)


def _mx(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


class Report:
    def __init__(self):
        self.rows = []

    def add(self, what, kind, err, tol):
        self.rows.append((what, kind, err, tol))

    def finish(self, case, dump=True):
        worst = {}
        for what, kind, err, tol in self.rows:
            if err > worst.get(kind, (0, ""))[0]:
                worst[kind] = (err, what)
        print("\n[%s] teacher-forced worst deviations:" % case)
        for kind, (err, what) in sorted(worst.items()):
            print("   %-18s %.3e  (%s)" % (kind, err, what))
        if dump and os.environ.get("X2_PARITY_DUMP"):
            os.makedirs(os.environ["X2_PARITY_DUMP"], exist_ok=True)
            with open(os.path.join(os.environ["X2_PARITY_DUMP"], "layerwise_" + case + ".txt"), "w") as f:
                for what, kind, err, tol in sorted(self.rows, key=lambda r: -r[2] / r[3]):
                    f.write("%-70s %-18s %.4e  tol %.1e\n" % (what, kind, err, tol))
        bad = [(w, k, e, t) for w, k, e, t in self.rows if not e <= t]
        if os.environ.get("X2_PARITY_NO_ASSERT") == "1":
            return
        assert not bad, "%d layer checks out of tolerance, worst: %s" % (len(bad), sorted(bad, key=lambda r: -r[2] / r[3])[:5])


def _zero(sd):
    for t in sd.values():
        t.grad = None


def _cmp_layer(rep, what, y, y_ref, x_hip, x_ref, hip_params, ref_params, extra=()):
    """forward output, input gradient, every parameter gradient of one teacher-forced layer.  hip_params / ref_params: name -> tensor
    with .grad (same names); extra: further (name, hip tensor with grad, ref tensor with grad) pairs (the image tokens of a fusion layer)."""
    rep.add(what, "forward max", _mx(y, y_ref), FWD_MAX)
    rep.add(what, "forward l2", _l2(y, y_ref), FWD_L2)
    if x_hip is not None:
        rep.add(what, "input grad l2", _l2(x_hip.grad, x_ref.grad), DX_L2)
    for n, h, r in extra:
        rep.add(what + " " + n, "input grad l2", _l2(h.grad, r.grad), DX_L2)
    tot = math.sqrt(sum(float(t.grad.double().pow(2).sum()) for t in ref_params.values() if t.grad is not None))
    for n, r in ref_params.items():
        if r.grad is None:
            continue
        g = hip_params[n].grad
        assert g is not None, (what, n)
        qk = "attention.self.query" in n or "attention.self.key" in n
        rep.add(what + " " + n, "param grad l2", _l2(g, r.grad, floor=1e-2 * tot), DW_QK_L2 if qk else DW_L2)


def run_layers(case, synthetic, mutate=None):
    """mutate: None, or a function (name of the check, oracle sd / kwargs) used by the mutation test."""
    from oracle import x2vlm_oracle as O
    eng = importlib.import_module("x2-vlm_amd.engine")
    xbert = importlib.import_module("x2-vlm_amd.xbert")
    ops = importlib.import_module("x2-vlm_amd.ops")
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES[case]
    cfg = O.config_from_case(c)
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
    model = mp.XVLM(config=model_config(case, tempfile.mkdtemp()), load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.cuda().eval()
    named = dict(model.named_parameters())
    b = make_batch(synthetic, c)
    rep = Report()
    gen = torch.Generator().manual_seed(1234)
    cot = lambda t: torch.randn(t.shape, generator=gen) * (1.0 / math.sqrt(t.shape[-1]))

    def hip_params(prefix):
        return {n[len(prefix):]: p for n, p in named.items() if n.startswith(prefix)}

    def ref_params(prefix):
        return {n[len(prefix):]: p for n, p in sd.items() if n.startswith(prefix)}

    def clear_hip():
        for p in named.values():
            p.grad = None

    with O.rounding(BF):
        # ------------------------------------------------------------------ vision tower
        image = b["image"]
        if image.dim() == 5:
            image = image.reshape(-1, *image.shape[2:])
        B = image.shape[0]
        enc = model.vision_encoder
        depth = cfg.vision_layers
        rel_index = O.relative_position_index(image.shape[-1] // cfg.patch_size)
        meta = dict(depth=depth, heads=enc.num_heads, patch=cfg.patch_size, eps=enc.eps, rel_index=enc.blocks[0].attn.relative_position_index,
                    pool_w=None, drop_path=None)
        with torch.no_grad():
            x = torch.cat([sd["vision_encoder.cls_token"].expand(B, -1, -1), O.patch_embed(sd, cfg, image)], dim=1)
        for i in range(depth):
            stem, head = i == 0, i == depth - 1
            _zero(sd); clear_hip()
            x_ref = None if stem else x.detach().clone().requires_grad_()
            xin = torch.cat([sd["vision_encoder.cls_token"].expand(B, -1, -1), O.patch_embed(sd, cfg, image)], dim=1) if stem else x_ref
            y_blk = O.vision_block(sd, cfg, i, xin, rel_index)
            if head:
                patches = O.layer_norm(y_blk[:, 1:], sd["vision_encoder.fc_norm.weight"], sd["vision_encoder.fc_norm.bias"], 1e-6)
                y_ref = torch.cat([patches.mean(dim=1, keepdim=True), patches], dim=1)
            else:
                y_ref = y_blk
            g = cot(y_ref)
            y_ref.backward(g)
            x_hip = None if stem else x.detach().clone().cuda().requires_grad_()
            y = eng.VisionEncoderFn.apply(image.cuda() if stem else x_hip, dict(meta, lo=i, hi=i + 1), *enc._params(i, i + 1))
            y.backward(g.cuda())
            names = eng.vision_param_names(depth, i, i + 1)
            _cmp_layer(rep, "vision block %d%s%s" % (i, " + stem" if stem else "", " + head" if head else ""), y, y_ref, x_hip, x_ref,
                       {n: named["vision_encoder." + n] for n in names}, {n: sd["vision_encoder." + n] for n in names})
            with torch.no_grad():
                x = y_blk.detach()
        with torch.no_grad():
            image_embeds = y_ref.detach()                       # the tower's output (fc_norm + pooled token 0)
        if c["frames"]:
            Fr = c["frames"]
            image_embeds = (image_embeds.view(B // Fr, Fr, *image_embeds.shape[1:]) + sd["absolute_frame_pos_embed"]).mean(1).detach()
        # ------------------------------------------------------------------ text embeddings + text layers
        ids, atts = b["text_ids"], b["text_atts"]
        S, L = ids.shape
        bert = model._bert
        with torch.no_grad():
            h = O.text_embeddings(sd, cfg, ids)
            rep.add("embeddings", "forward max", _mx(bert.embeddings(ids.cuda()), h), FWD_MAX)
        self_mask = (1.0 - atts.float())[:, None, None, :] * -10000.0
        if mutate == "mask":                                   # mutation test: the oracle masks one more key of sequence 0
            self_mask = self_mask.clone()
            self_mask[0, 0, 0, 1] = -10000.0
        enc_atts = torch.ones(image_embeds.shape[:2], dtype=torch.int64)
        hip_mask = xbert._key_mask(atts.cuda(), -10000.0)
        enc_names = dict(bert.encoder.named_parameters())
        for i in range(cfg.text_layers):
            cross = i >= cfg.fusion_at
            _zero(sd); clear_hip()
            h_ref = h.detach().clone().requires_grad_()
            e_ref = image_embeds[:S].detach().clone().requires_grad_() if cross else None
            e_mask = (1.0 - enc_atts[:S].float())[:, None, None, :] * -1e9 if cross else None
            y_ref = O.bert_layer(sd, cfg, i, h_ref, self_mask, e_ref, e_mask)
            g = cot(y_ref)
            y_ref.backward(g)
            h_hip = h.detach().clone().cuda().requires_grad_()
            e_hip = image_embeds[:S].detach().clone().cuda().requires_grad_() if cross else None
            m = dict(lo=i, hi=i + 1, fusion_at=cfg.fusion_at, heads=cfg.heads, eps=bert.config.layer_norm_eps, self_mask=hip_mask,
                     enc_mask=xbert._key_mask(enc_atts[:S].cuda(), -1e9) if cross else None, kv_idx=None, seq_off=None, seq_ids=None, drop=None)
            params = [enc_names[n] for n in eng.bert_layer_param_names(i, i + 1, cfg.fusion_at, cross)]
            y = eng.BertLayersFn.apply(h_hip, e_hip, m, *params)
            y.backward(g.cuda())
            pre = "text_encoder.bert.encoder.layer.%d." % i
            _cmp_layer(rep, "%s layer %d" % ("fusion" if cross else "text", i), y, y_ref, h_hip, h_ref, hip_params(pre), ref_params(pre),
                       extra=[("image tokens", e_hip, e_ref)] if cross else ())
            with torch.no_grad():
                h = y_ref.detach()
        # ------------------------------------------------------------------ MLM head (tied decoder) + CE
        if hasattr(model.text_encoder, "cls"):
            _zero(sd); clear_hip()
            seq_ref = h.detach().clone().requires_grad_()
            logits = O.mlm_logits_from_hidden(sd, seq_ref, b["masked_pos"])
            loss_ref = O.cross_entropy(logits.reshape(-1, cfg.vocab), b["masked_ids"].reshape(-1))
            loss_ref.backward()
            seq_hip = h.detach().clone().cuda().requires_grad_()
            loss, lse, _ = model.text_encoder.mlm_loss_from_hidden(seq_hip, b["masked_pos"].cuda(), b["masked_ids"].cuda())
            loss.backward()
            rep.add("MLM head", "forward max", abs(float(loss) - float(loss_ref)) / abs(float(loss_ref)), 2e-4)
            hp = hip_params("text_encoder.cls.predictions."); hp["word"] = named["text_encoder.bert.embeddings.word_embeddings.weight"]
            rp = ref_params("text_encoder.cls.predictions."); rp["word"] = sd["text_encoder.bert.embeddings.word_embeddings.weight"]
            _cmp_layer(rep, "MLM head", loss, loss_ref, seq_hip, seq_ref, hp, rp)
        # ------------------------------------------------------------------ MLP heads on the fused [CLS] rows
        for head_name, mod in (("itm_head", model.itm_head), ("bbox_head", model.bbox_head)):
            _zero(sd); clear_hip()
            rows_ref = h[:, 0].detach().clone().requires_grad_()
            y_ref = O.head_mlp(sd, head_name, rows_ref)
            g = cot(y_ref)
            y_ref.backward(g)
            rows_hip = h[:, 0].detach().clone().cuda().requires_grad_()
            y = ops.mlp_head(mod, rows_hip)
            y.backward(g.cuda())
            _cmp_layer(rep, head_name, y, y_ref, rows_hip, rows_ref, hip_params(head_name + "."), ref_params(head_name + "."))
    rep.finish(case, dump=mutate is None)
    return rep


@pytest.mark.parametrize("case", ["tiny", "tiny_video", "base_shallow", "large_shallow", "base_full"])
def test_every_layer_teacher_forced(case, synthetic):
    run_layers(case, synthetic)


def test_the_gate_sees_a_one_position_mask_slip(synthetic):
    """Mutation check: the oracle's side masks ONE more key position in ONE sequence - the layer bounds must fail (a 2 % whole-model
    gate would not notice)."""
    keep = os.environ.get("X2_PARITY_NO_ASSERT")
    os.environ["X2_PARITY_NO_ASSERT"] = "1"
    try:
        rep = run_layers("base_shallow", synthetic, mutate="mask")
    finally:
        if keep is None:
            os.environ.pop("X2_PARITY_NO_ASSERT", None)
        else:
            os.environ["X2_PARITY_NO_ASSERT"] = keep
    bad = [r for r in rep.rows if r[0].startswith(("text layer", "fusion layer")) and r[1] == "forward max" and r[2] > r[3]]
    assert len(bad) >= 1, "a one-position mask slip went unnoticed: %s" % [r for r in rep.rows if r[1] == "forward max"][:6]
