"""Where does the HIP text layer leave the operand-rounding-aware oracle?  One BERT layer at a time (and its sub-steps: fused
QKV GEMM, attention, output projection + LayerNorm, FFN), each fed the ORACLE's rounded input, compared with the oracle's
rounded output - errors do not accumulate across layers, so a mismatch shows where it is made.  Also the same for one vision
block.  GPU box only (the oracle runs on the host cores)."""
import importlib, math, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from cases import CASES, make_batch, model_config
from oracle import x2vlm_oracle as O
synthetic = importlib.import_module("x2-vlm_amd.synthetic")
K = importlib.import_module("x2-vlm_amd.kernels")
eng = importlib.import_module("x2-vlm_amd.engine")
xbert = importlib.import_module("x2-vlm_amd.xbert")
mp = importlib.import_module("x2-vlm_amd.model_pretrain")
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9), float((a - b).norm()) / max(float(b.norm()), 1e-9)


for case in sys.argv[1:] or ["tiny_text", "base_shallow_text"]:
    c = CASES[case]
    cfg = O.config_from_case(c)
    sd = O.make_params(cfg, c["wseed"], synthetic.synth_tensor, requires_grad=False)
    model = mp.XVLM(config=model_config(case, tempfile.mkdtemp()), load_vision_params=False, load_text_params=False, pretraining=True)
    synthetic.synth_state_dict(model, c["wseed"])
    model = model.cuda().eval()
    b = make_batch(synthetic, c)
    ids, atts = b["text_ids_masked"], b["text_atts"]
    S, L = ids.shape
    Hd, H = cfg.hidden, cfg.heads
    print("== %s: S=%d L=%d hidden=%d heads=%d" % (case, S, L, Hd, H))
    with torch.no_grad(), O.rounding(BF):
        h = O.text_embeddings(sd, cfg, ids)
        e_hip = model._bert.embeddings(ids.cuda())
        print("  embeddings                         max %.2e  l2 %.2e" % rel(e_hip, h))
        self_mask = (1.0 - atts.float())[:, None, None, :] * -10000.0
        bert = model._bert
        names_all = dict(bert.encoder.named_parameters())
        for i in range(cfg.text_layers):
            p = "text_encoder.bert.encoder.layer.%d." % i
            a_ = p + "attention."
            # ---- oracle sub-steps (rounded)
            q = O.mm_linear(h, sd[a_ + "self.query.weight"], sd[a_ + "self.query.bias"], out="bf16")
            k = O.mm_linear(h, sd[a_ + "self.key.weight"], sd[a_ + "self.key.bias"], out="bf16")
            v = O.mm_linear(h, sd[a_ + "self.value.weight"], sd[a_ + "self.value.bias"], out="bf16")
            ctx = O.merge_heads(O.attention_core(O.split_heads(q, H), O.split_heads(k, H), O.split_heads(v, H), 1.0 / math.sqrt(Hd // H), self_mask))
            o = O.mm_linear(ctx, sd[a_ + "output.dense.weight"], sd[a_ + "output.dense.bias"])
            h1 = O.layer_norm(o + h, sd[a_ + "output.LayerNorm.weight"], sd[a_ + "output.LayerNorm.bias"], 1e-12)
            f = O._q(O.gelu_mm(O.mm_linear(h1, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"])))
            o2 = O.mm_linear(f, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
            h2 = O.layer_norm(o2 + h1, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], 1e-12)
            # ---- HIP sub-steps on the oracle's inputs
            M = S * L
            hg = h.cuda().contiguous().view(M, Hd)
            hb = K.cast_bf16(hg)
            lp = {n[len("layer.%d." % i):]: t for n, t in names_all.items() if n.startswith("layer.%d." % i)}
            wqkv, _ = eng.BANK.linear(lp["attention.self.query.weight"], lp["attention.self.key.weight"], lp["attention.self.value.weight"])
            bqkv = eng.BANK.vector(lp["attention.self.query.bias"], lp["attention.self.key.bias"], lp["attention.self.value.bias"])
            qkv = K.gemm_nt(hb, wqkv, bias=bqkv)
            ref_qkv = torch.cat([q, k, v], -1).view(M, 3 * Hd)
            print("  layer %d  qkv GEMM                   max %.2e  l2 %.2e" % ((i,) + rel(qkv.float(), ref_qkv)))
            # attention on the ORACLE's q, k, v
            qkv_o = ref_qkv.to(BF).cuda().contiguous()
            att = torch.empty(M, Hd, device="cuda", dtype=BF)
            lse = torch.empty(S * H * L, device="cuda")
            mask = xbert._key_mask(atts.cuda(), -10000.0)
            K.attn_fwd(K.view3(qkv_o, S, L, 0), K.view3(qkv_o, S, L, Hd), K.view3(qkv_o, S, L, 2 * Hd), S, S, H, L, L, 1.0 / math.sqrt(Hd // H),
                       K.view3(att, S, L), lse, mask=mask)
            print("  layer %d  attention                  max %.2e  l2 %.2e" % ((i,) + rel(att.float().view(S, L, Hd), ctx)))
            # output projection + LN on the oracle's context
            ctx_b = ctx.to(BF).cuda().contiguous().view(M, Hd)
            wo, _ = eng.BANK.linear(lp["attention.output.dense.weight"])
            s1 = K.gemm_nt(ctx_b, wo, bias=lp["attention.output.dense.bias"], resid=hg, out_dtype=torch.float32)
            print("  layer %d  out proj + residual        max %.2e  l2 %.2e" % ((i,) + rel(s1.view(S, L, Hd), o + h)))
            _, h1g, _, _ = K.layernorm_fwd(s1, lp["attention.output.LayerNorm.weight"], lp["attention.output.LayerNorm.bias"], 1e-12, want_f32=True)
            print("  layer %d  LayerNorm 1 (HIP chain)    max %.2e  l2 %.2e" % ((i,) + rel(h1g.view(S, L, Hd), h1)))
            h1o = h1.cuda().contiguous().view(M, Hd)
            wi, _ = eng.BANK.linear(lp["intermediate.dense.weight"]); wout, _ = eng.BANK.linear(lp["output.dense.weight"])
            pre = torch.empty(M, wi.shape[0], device="cuda", dtype=BF)
            act = K.gemm_nt(K.cast_bf16(h1o), wi, bias=lp["intermediate.dense.bias"], aux=pre, act=1)
            print("  layer %d  intermediate + GELU        max %.2e  l2 %.2e" % ((i,) + rel(act.float().view(S, L, -1), f)))
            s3 = K.gemm_nt(f.to(BF).cuda().contiguous().view(M, -1), wout, bias=lp["output.dense.bias"], resid=h1o, out_dtype=torch.float32)
            print("  layer %d  output dense + residual    max %.2e  l2 %.2e" % ((i,) + rel(s3.view(S, L, Hd), o2 + h1)))
            # the whole layer as the stage runs it
            params = [names_all[n] for n in eng.bert_layer_param_names(i, i + 1, cfg.fusion_at, False)]
            meta = dict(lo=i, hi=i + 1, fusion_at=cfg.fusion_at, heads=H, eps=1e-12, self_mask=mask, enc_mask=None, kv_idx=None,
                        seq_off=None, seq_ids=None, drop=None)
            out = eng.BertLayersFn.apply(h.cuda(), None, meta, *params)
            print("  layer %d  WHOLE LAYER (stage)        max %.2e  l2 %.2e" % ((i,) + rel(out, h2)))
            h = h2
