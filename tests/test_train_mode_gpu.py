"""Training-mode (dropout / DropPath) parity on the GPU.  The kernels' dropout is counter-based:
keep(element) = f(site seed, element index), regenerated in the backward.  kernels.dropout_keep()
is the host mirror of that function, so the oracle can be run with EXACTLY the masks the HIP path
used and compared element-wise (outputs and all gradients), stage by stage:
  * BERT layers (self + cross attention-probability dropout, three hidden dropouts per layer),
  * embeddings dropout,
  * vision blocks with per-sample stochastic depth.
Plus whole-model sanity: determinism under a fixed seed, train != eval, finite gradients, keep rate."""
import importlib
import math
import tempfile

import pytest
import torch

from cases import CASES, model_config
from oracle import x2vlm_oracle as O

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.detach().double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-9))


@pytest.fixture(scope="module")
def mods():
    K = importlib.import_module("x2-vlm_amd.kernels")
    xbert = importlib.import_module("x2-vlm_amd.xbert")
    beit2 = importlib.import_module("x2-vlm_amd.beit2")
    return K, xbert, beit2


def test_keep_rate_and_host_mirror(mods):
    K, _, _ = mods
    spec = K.dropout_spec(0.1, 1234, 7)
    keep = K.dropout_keep(spec, torch.arange(1 << 20))
    rate = float((keep == 0).float().mean())
    assert abs(rate - 0.1) < 2e-3 and float(keep.max()) == pytest.approx(1 / 0.9)
    # device side through the GEMM epilogue: identity weights, dropout on the output
    M, N = 256, 128
    A = torch.eye(N).repeat(2, 1).bfloat16().to(dev)          # [256,128] rows = one-hot
    B = torch.eye(N).bfloat16().to(dev)
    out = K.gemm_nt(A, B, out_dtype=torch.float32, drop=spec).cpu()
    idx = torch.arange(M).unsqueeze(1) * N + torch.arange(N).unsqueeze(0)
    ref = (A.float().cpu() @ B.float().cpu().t()) * K.dropout_keep(spec, idx)
    assert torch.equal(out, ref)
    # device-resident epoch (hipGraph replays): same kernel arguments, the mask follows the counter in device memory
    epoch = torch.tensor([5], dtype=torch.int32, device=dev)
    spec_e = spec[:3] + (epoch,)
    outs = []
    for e in (5, 6, 5):
        epoch.fill_(e)
        o = K.gemm_nt(A, B, out_dtype=torch.float32, drop=spec_e).cpu()
        assert torch.equal(o, (A.float().cpu() @ B.float().cpu().t()) * K.dropout_keep(spec_e, idx))
        outs.append(o)
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], out)


def test_bert_layers_with_dropout_match_oracle_with_same_masks(mods, synthetic):
    K, xbert, _ = mods
    cfg_o = O.OracleConfig(image_res=32, vision_layers=1, hidden=128, heads=2, ffn=256, vocab=512, text_layers=3, fusion_at=1,
                           embed_dim=32, max_pos=64)
    conf = xbert.BertConfig(vocab_size=512, hidden_size=128, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64)
    conf.num_hidden_layers, conf.fusion_layer, conf.encoder_width = 3, 1, 768
    model = xbert.BertModel(conf)
    synthetic.synth_state_dict(model, 77)
    sd_all = O.make_params(cfg_o, 77, lambda n, sh, s: synthetic.synth_tensor(n[len("text_encoder.bert."):] if n.startswith("text_encoder.bert.") else n, sh, s))
    model = model.to(dev).train()
    S, L, Hd, H, T, Bi = 4, 8, 128, 2, 5, 2
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(5, 512, (S, L), generator=g)
    atts = torch.ones(S, L, dtype=torch.long); atts[1, 6:] = 0; atts[3, 5:] = 0
    enc = torch.randn(Bi, T, 768, generator=g)
    enc_atts = torch.ones(S, T, dtype=torch.long); enc_atts[2, 3:] = 0
    kv = torch.tensor([0, 1, 1, 0])
    seed_emb, seed_enc = 111, 222
    xbert._FIXED_SEEDS[:] = [seed_emb, seed_enc]
    enc_d = enc.to(dev).requires_grad_(True)
    out = model(ids.to(dev), attention_mask=atts.to(dev), encoder_hidden_states=enc_d, encoder_attention_mask=enc_atts.to(dev),
                kv_idx=kv.to(dev)).last_hidden_state
    dout = torch.randn(S, L, Hd, generator=g)
    out.backward(dout.to(dev))
    # ---- oracle with the same masks ----
    Lp, Tp = K.round_up(L, 64), K.round_up(T, 64)

    def drop(name):
        if name == "emb":
            spec = K.dropout_spec(0.1, seed_emb, 1000)
            return K.dropout_keep(spec, torch.arange(S * L * Hd)).view(S, L, Hd)
        layer = int(name[1:name.index(".")])
        kind = {"self.probs": 0, "self.out": 1, "cross.probs": 2, "cross.out": 3, "ffn.out": 4}[name[name.index(".") + 1:]]
        spec = K.dropout_spec(0.1, seed_enc, 8 * layer + kind)
        if kind in (1, 3, 4):
            return K.dropout_keep(spec, torch.arange(S * L * Hd)).view(S, L, Hd)
        Lk, Lkp = (L, Lp) if kind == 0 else (T, Tp)
        b, h, q, k = torch.meshgrid(torch.arange(S), torch.arange(H), torch.arange(L), torch.arange(Lk), indexing="ij")
        return K.dropout_keep(spec, ((b * H + h) * L + q) * Lkp + k)

    sd = {k: v for k, v in sd_all.items()}
    enc_o = enc.clone().requires_grad_(True)
    h0 = O.text_embeddings(sd, cfg_o, ids, drop)
    ref = O.bert_encoder(sd, cfg_o, h0, atts, enc_o[kv], enc_atts, "multi_modal", drop)
    ref.backward(dout)
    assert rel(out, ref) < 1.2e-2                    # measured 6.8e-3 (round 6)
    assert rel(enc_d.grad, enc_o.grad) < 3e-2
    worst = 0.0
    tot = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in model.parameters()))
    for n, p in model.named_parameters():
        r = sd["text_encoder.bert." + n].grad
        err = float((p.grad.cpu().double() - r.double()).norm()) / max(float(r.double().norm()), 1e-2 * tot)
        worst = max(worst, err)
        assert err < 2.5e-2, (n, err)                # measured 1.49e-2
    print("bert train-mode: out err %.3e, worst param-grad err %.3e" % (rel(out, ref), worst))


def test_vision_blocks_with_drop_path_match_oracle(mods, synthetic):
    K, _, beit2 = mods
    cfg_o = O.OracleConfig(image_res=32, vision_layers=3)
    vit = beit2.beit_base_patch16(32, drop_path_rate=0.3, vision_num_hidden_layers=3)
    synthetic.synth_state_dict(vit, 9)
    sd = {"vision_encoder." + n: synthetic.synth_tensor(n, p.shape, 9).requires_grad_(True) for n, p in vit.named_parameters()}
    vit = vit.to(dev).train()
    B = 6
    keep = torch.tensor([[[1, 0, 1, 1, 0, 1], [1, 1, 1, 0, 1, 1]], [[0, 1, 1, 1, 1, 0], [1, 1, 0, 1, 1, 1]], [[1, 1, 1, 1, 0, 0], [0, 1, 1, 1, 1, 1]]])
    vit.fixed_drop_path_keep = keep
    img = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    out = vit(img.to(dev))
    dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    out.backward(dout.to(dev))
    rates = [b.drop_path_rate for b in vit.blocks]
    assert rates[0] == 0.0 and abs(rates[2] - 0.3) < 1e-6
    # rate 0 -> nn.Identity in the reference (beit2.py:180): block 0 ignores its keep row
    dp = [tuple(((keep[i, j].float() if rates[i] > 0 else torch.ones(B)) / (1 - rates[i])).view(B, 1, 1) for j in range(2))
          for i in range(3)]
    ref = O.vision_encoder(sd, cfg_o, img, drop_path=dp)
    ref.backward(dout)
    assert rel(out, ref) < 2e-2
    tot = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in vit.parameters()))
    for n, p in vit.named_parameters():
        r = sd["vision_encoder." + n].grad
        err = float((p.grad.cpu().double() - r.double()).norm()) / max(float(r.double().norm()), 1e-2 * tot)
        assert err < 4e-2, (n, err)


def test_whole_model_train_mode_sanity(synthetic):
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES["tiny"]
    model = mp.XVLM(config=model_config("tiny", tempfile.mkdtemp()), load_vision_params=False, load_text_params=False)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.to(dev)
    batch = {k: v.to(dev) for k, v in synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"],
                                                            c["max_masks"], ragged=True).items()}
    model.injected_negatives = synthetic.synth_negatives(c["bseed"], c["batch"])

    def run(train, seed):
        model.train(train)
        torch.manual_seed(seed)
        model.zero_grad(set_to_none=True)
        loss = model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"])
        sum(loss.values()).backward()
        gn = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None))
        return {k: float(v) for k, v in loss.items()}, gn

    ev, gn_e = run(False, 0)
    t1, gn1 = run(True, 1)
    t1b, gn1b = run(True, 1)
    t2, gn2 = run(True, 2)
    assert t1 == t1b and abs(gn1 - gn1b) < 1e-3 * gn1          # same host seed -> same masks (up to fp32 atomics order)
    assert t1 != t2 and t1 != ev                                # different masks, and dropout really is on
    for d in (t1, t2):
        for k in d:
            assert math.isfinite(d[k]) and abs(d[k] - ev[k]) < 2.0 * max(1.0, abs(ev[k])), (k, d[k], ev[k])
    assert math.isfinite(gn1) and math.isfinite(gn2)


def test_whole_model_train_mode_matches_oracle_with_same_masks(synthetic):
    """The WHOLE training step in train mode (what bench.py times): every dropout site of the text tower, the fusion stack
    and the embeddings plus per-sample DropPath, against the oracle's stage functions run with exactly the masks the HIP
    path drew (host mirror of the counter-based generator).  The oracle stages are composed in the order the product
    batches them (clean + masked ids as one 2B-row text batch; positives, two negative sets and the MLM rows as one 4B-row
    fusion batch), because a mask element is addressed by its position in that batch; in eval mode this composition is
    the pinned oracle.xvlm_forward (tests/test_oracle_golden.py, test_model_gpu.py)."""
    K = importlib.import_module("x2-vlm_amd.kernels")
    xbert = importlib.import_module("x2-vlm_amd.xbert")
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    c = CASES["tiny"]
    cfg = O.config_from_case(c)
    model = mp.XVLM(config=model_config("tiny", tempfile.mkdtemp()), load_vision_params=False, load_text_params=False)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.to(dev).train()
    sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor)
    batch = synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=True)
    ineg, tneg = synthetic.synth_negatives(c["bseed"], c["batch"])
    model.injected_negatives = (ineg, tneg)
    B, L, Hd, H, V = c["batch"], c["seq_len"], c["hidden"], c["heads"], c["vocab"]
    depth = c["vision_layers"]
    keep = (torch.rand(depth, 2, B, generator=torch.Generator().manual_seed(9)) > 0.3).float()
    model.vision_encoder.fixed_drop_path_keep = keep
    s_emb, s_text, s_fus = 1001, 2002, 3003
    xbert._FIXED_SEEDS[:] = [s_emb, s_text, s_fus]
    gb = {k: v.to(dev) for k, v in batch.items()}
    loss = model(gb["image"], gb["text_ids"], gb["text_atts"], text_ids_masked=gb["text_ids_masked"], masked_pos=gb["masked_pos"],
                 masked_ids=gb["masked_ids"])
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    assert not xbert._FIXED_SEEDS                      # exactly three draws: embeddings, text layers, fusion layers

    # ---- the same step on the oracle, same masks ----
    bc = model.text_encoder.config
    p_h, p_a = bc.hidden_dropout_prob, bc.attention_probs_dropout_prob
    T = cfg.n_tokens

    def masks(seed, S, Lk_cross):
        def drop(name):
            if name == "emb":
                return K.dropout_keep(K.dropout_spec(p_h, s_emb, 1000), torch.arange(S * L * Hd)).view(S, L, Hd)
            layer = int(name[1:name.index(".")])
            kind = {"self.probs": 0, "self.out": 1, "cross.probs": 2, "cross.out": 3, "ffn.out": 4}[name[name.index(".") + 1:]]
            spec = K.dropout_spec(p_a if kind in (0, 2) else p_h, seed, 8 * layer + kind)
            if kind in (1, 3, 4):
                return K.dropout_keep(spec, torch.arange(S * L * Hd)).view(S, L, Hd)
            Lk = L if kind == 0 else Lk_cross
            b, h, q, k = torch.meshgrid(torch.arange(S), torch.arange(H), torch.arange(L), torch.arange(Lk), indexing="ij")
            return K.dropout_keep(spec, ((b * H + h) * L + q) * K.round_up(Lk, 64) + k)
        return drop

    rates = [blk.drop_path_rate for blk in model.vision_encoder.blocks]
    dp = [tuple(((keep[i, j] if rates[i] > 0 else torch.ones(B)) / (1 - rates[i])).view(B, 1, 1) for j in range(2)) for i in range(depth)]
    ids2 = torch.cat([batch["text_ids"], batch["text_ids_masked"]]); atts2 = torch.cat([batch["text_atts"], batch["text_atts"]])
    d_text = masks(s_text, 2 * B, 0)
    both = O.bert_encoder(sd, cfg, O.text_embeddings(sd, cfg, ids2, d_text), atts2, mode="text", drop=d_text)
    image_embeds = O.vision_encoder(sd, cfg, batch["image"], drop_path=dp)
    fi, ft = O.features(sd, image_embeds, both[:B])
    ref = {"loss_itc": O.contrastive_loss(sd, fi, ft)[0]}
    ar = torch.arange(B)
    t_idx = torch.cat([ar, ar, torch.tensor(tneg), ar + B]); kv = torch.cat([ar, torch.tensor(ineg), ar, ar])
    ones = torch.ones(4 * B, T, dtype=torch.int64)
    fused = O.bert_encoder(sd, cfg, both[t_idx], atts2[t_idx], image_embeds[kv], ones, "fusion", masks(s_fus, 4 * B, T))
    labels = torch.cat([torch.ones(B, dtype=torch.int64), torch.zeros(2 * B, dtype=torch.int64)])
    ref["loss_itm"] = O.cross_entropy(O.head_mlp(sd, "itm_head", fused[:3 * B, 0]), labels)
    h = fused[-B:].gather(1, batch["masked_pos"].unsqueeze(-1).expand(-1, -1, Hd))
    pr = "text_encoder.cls.predictions."
    h = O.gelu(O.linear(h, sd[pr + "transform.dense.weight"], sd[pr + "transform.dense.bias"]))
    h = O.layer_norm(h, sd[pr + "transform.LayerNorm.weight"], sd[pr + "transform.LayerNorm.bias"], 1e-12)
    logits = O.linear(h, sd["text_encoder.bert.embeddings.word_embeddings.weight"], sd[pr + "bias"])
    ref["loss_mlm"] = O.cross_entropy(logits.reshape(-1, V), batch["masked_ids"].reshape(-1))
    sum(ref.values()).backward()

    for k in ref:
        # train mode, identical masks, toy batch: measured 1.5e-3 (loss_itc); the eval-mode bound of toy batches is 5e-3 at 4.7e-3 measured
        assert abs(float(loss[k]) - float(ref[k])) <= 3e-3 * max(1.0, abs(float(ref[k]))), (k, float(loss[k]), float(ref[k]))
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    tot = math.sqrt(sum(float(t.grad.double().pow(2).sum()) for t in sd.values() if t.grad is not None))
    gn = math.sqrt(sum(float(g.double().pow(2).sum()) for g in got.values()))
    print("whole-model train mode: total gradient norm %.5f, oracle %.5f (%.2e)" % (gn, tot, abs(gn - tot) / tot))
    # measured 9.7e-4 with the test file run on its own, 2.4e-3 inside the whole suite: the masks are a function of (seed, epoch word) and the replayed
    # steps of earlier tests have advanced the process-wide epoch word (the oracle mirrors it: K.dropout_keep) - another mask sample, another error sample
    assert abs(gn - tot) <= 5e-3 * tot, (gn, tot)
    worst = ("", 0.0)
    for n, g in got.items():
        r = sd[n].grad
        if n == "text_encoder.cls.predictions.decoder.weight" or r is None:
            continue
        err = float((g.cpu().double() - r.double()).norm()) / max(float(r.double().norm()), 1e-2 * tot)
        worst = max(worst, (n, err), key=lambda t: t[1])
        assert err < 3e-2, (n, err)                  # per-tensor gradient ERROR norm (stricter than eval mode's norm difference): measured 1.63e-2
    print("whole-model train mode: losses", {k: (round(float(loss[k]), 5), round(float(ref[k]), 5)) for k in ref}, "worst grad", worst,
          "total norm err %.3e" % (abs(gn - tot) / tot))
