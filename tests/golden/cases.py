"""Parity cases shared by make_golden.py (reference run, build container only) and the tests.

Each case fixes a model geometry, a weight seed and a batch seed; weights and inputs are
regenerated from the seeds (x2-vlm_amd/synthetic.py), so fixtures hold outputs only.
Vision width/heads are fixed at 768/12 by the reference's beit_base_patch16 (beit2.py:439-446).
"""

CASES = {
    # 2x2 patch grid (N=5), narrow text tower: every op of the path, seconds on CPU, full tensors kept
    "tiny": dict(image_res=32, vision_layers=2, hidden=128, heads=2, ffn=256, vocab=512, max_pos=64,
                 text_layers=4, fusion_at=2, embed_dim=32, batch=4, seq_len=8, max_masks=3,
                 ragged=True, region=False, frames=0, wseed=11, bseed=12),
    # same geometry, region/bbox path: idx_to_group_img, masked mean pooling, 5th fusion pass, GIoU
    "tiny_region": dict(image_res=32, vision_layers=2, hidden=128, heads=2, ffn=256, vocab=512, max_pos=64,
                        text_layers=4, fusion_at=2, embed_dim=32, batch=6, n_images=3, seq_len=8,
                        max_masks=3, ragged=True, region=True, frames=0, wseed=13, bseed=14),
    # video path: frames folded into the batch, frame position embedding, frame mean (xvlm.py:615-661)
    "tiny_video": dict(image_res=32, vision_layers=1, hidden=128, heads=2, ffn=256, vocab=512, max_pos=64,
                       text_layers=3, fusion_at=2, embed_dim=32, batch=3, seq_len=8, max_masks=3,
                       ragged=True, region=False, frames=2, wseed=15, bseed=16),
    # the real token geometry (N=197, L=30, d_h=64, 12 heads, V=30522) on a shallow stack
    "base_shallow": dict(image_res=224, vision_layers=2, hidden=768, heads=12, ffn=3072, vocab=30522,
                         max_pos=512, text_layers=3, fusion_at=2, embed_dim=256, batch=4, seq_len=30,
                         max_masks=12, ragged=True, region=False, frames=0, wseed=21, bseed=22),
    # full X2VLM-base (configs/pretrain/x2vlm_base_4m.yaml): 12 + 12 + 6 layers, 254.76 M params
    "base_full": dict(image_res=224, vision_layers=12, hidden=768, heads=12, ffn=3072, vocab=30522,
                      max_pos=512, text_layers=18, fusion_at=12, embed_dim=256, batch=4, seq_len=30,
                      max_masks=12, ragged=False, region=False, frames=0, wseed=31, bseed=32),
    # the geometry of X2VLM-large at 384 px (BEiT2-large: D=1024, 16 heads, N=577; BERT-large widths) on a shallow stack
    "large_shallow": dict(image_res=384, vision_layers=2, vision_width=1024, hidden=1024, heads=16, ffn=4096, vocab=30522,
                          max_pos=512, text_layers=3, fusion_at=2, embed_dim=256, batch=2, seq_len=30,
                          max_masks=12, ragged=True, region=False, frames=0, wseed=61, bseed=62),
    # BASELINE.json configs[1]: full X2VLM-base at the headline per-GPU batch 64 (ragged captions)
    "base_full_b64": dict(image_res=224, vision_layers=12, hidden=768, heads=12, ffn=3072, vocab=30522,
                          max_pos=512, text_layers=18, fusion_at=12, embed_dim=256, batch=64, seq_len=30,
                          max_masks=12, ragged=True, region=False, frames=0, wseed=41, bseed=42),
    # BASELINE.json configs[3]: full X2VLM-large (BEiT2-large 24 blocks, BERT-large-12l 18 layers, 593 M params) at 384 px;
    # CPU-feasible batch (the per-GPU batch 32 is covered by the bench and by property checks)
    "large_full": dict(image_res=384, vision_layers=24, vision_width=1024, hidden=1024, heads=16, ffn=4096, vocab=30522,
                       max_pos=512, text_layers=18, fusion_at=12, embed_dim=256, batch=2, seq_len=30,
                       max_masks=12, ragged=True, region=False, frames=0, wseed=71, bseed=72),
    # BASELINE.json configs[3] at ITS OWN per-GPU batch 32: the losses of X2VLM-large are held to north_star's 1e-3 here
    # (the B = 2 case above averages the same per-sample bf16 noise over 16x fewer samples); one-off ~1 h CPU run of the reference
    "large_full_b32": dict(image_res=384, vision_layers=24, vision_width=1024, hidden=1024, heads=16, ffn=4096, vocab=30522,
                           max_pos=512, text_layers=18, fusion_at=12, embed_dim=256, batch=32, seq_len=30,
                           max_masks=12, ragged=True, region=False, frames=0, wseed=73, bseed=74, checkpoint_blocks=True),
    # the region / bbox iteration (Pretrain.run_region_iter, Pretrain.py:79-111) at the REAL geometry: full X2VLM-base, 224 px,
    # 8 region texts over 4 images, image_atts with masked-out patches (masked mean pooling over 197 tokens, beit2.py:426-436),
    # the 5th fusion pass of predict_bbox at base width, L1 + GIoU (xvlm.py:688-698, 910-957)
    "base_region": dict(image_res=224, vision_layers=12, hidden=768, heads=12, ffn=3072, vocab=30522,
                        max_pos=512, text_layers=18, fusion_at=12, embed_dim=256, batch=8, n_images=4, seq_len=30,
                        max_masks=12, ragged=True, region=True, frames=0, wseed=91, bseed=92),
    # BASELINE.json configs[4]: full X2VLM-base video path, 8-frame 224 px clips (avgpool + frame position embedding)
    "video_full": dict(image_res=224, vision_layers=12, hidden=768, heads=12, ffn=3072, vocab=30522,
                       max_pos=512, text_layers=18, fusion_at=12, embed_dim=256, batch=2, seq_len=30,
                       max_masks=12, ragged=True, region=False, frames=8, wseed=81, bseed=82),
    # ---- API branches of XVLM.forward that the headline iteration does not take (round 5) ------------------------------------
    # image=None: Pretrain.run_text_iter -> XVLM.forward_text (model_pretrain.py:67-72): the 18-layer multi_modal pass WITHOUT
    # cross-attention + MLM head; only loss_mlm, no vision tower, no ITC / ITM
    "tiny_text": dict(image_res=32, vision_layers=2, hidden=128, heads=2, ffn=256, vocab=512, max_pos=64,
                      text_layers=4, fusion_at=2, embed_dim=32, batch=4, seq_len=8, max_masks=3,
                      ragged=True, region=False, frames=0, wseed=17, bseed=18, text_only=True),
    "base_shallow_text": dict(image_res=224, vision_layers=2, hidden=768, heads=12, ffn=3072, vocab=30522,
                              max_pos=512, text_layers=3, fusion_at=2, embed_dim=256, batch=4, seq_len=30,
                              max_masks=12, ragged=True, region=False, frames=0, wseed=23, bseed=24, text_only=True),
    # ret_match_loss=False (model_pretrain.py:49-52): loss_itm = tensor(0.0), the fusion batch is B rows (MLM only), not 4B
    "tiny_nomatch": dict(image_res=32, vision_layers=2, hidden=128, heads=2, ffn=256, vocab=512, max_pos=64,
                         text_layers=4, fusion_at=2, embed_dim=32, batch=4, seq_len=8, max_masks=3,
                         ragged=True, region=False, frames=0, wseed=19, bseed=20, match=False),
    "base_shallow_nomatch": dict(image_res=224, vision_layers=2, hidden=768, heads=12, ffn=3072, vocab=30522,
                                 max_pos=512, text_layers=3, fusion_at=2, embed_dim=256, batch=4, seq_len=30,
                                 max_masks=12, ragged=True, region=False, frames=0, wseed=25, bseed=26, match=False),
    # a degenerate TARGET box (negative width) in a region batch: the reference's early-out zeroes every row's GIoU term and
    # never evaluates generalized_box_iou (xvlm.py:940-946); L1 and all other losses unchanged
    "tiny_region_degenerate": dict(image_res=32, vision_layers=2, hidden=128, heads=2, ffn=256, vocab=512, max_pos=64,
                                   text_layers=4, fusion_at=2, embed_dim=32, batch=6, n_images=3, seq_len=8,
                                   max_masks=3, ragged=True, region=True, frames=0, wseed=13, bseed=14, degenerate=True),
}


def make_batch(synthetic, c):
    """The seeded batch of a case (CPU tensors), incl. the case's own twists (degenerate target box)."""
    if c["region"]:
        b = synthetic.synth_region_batch(c["bseed"], c["n_images"], c["batch"], c["seq_len"], c["image_res"], 16, c["vocab"],
                                         c["max_masks"])
        if c.get("degenerate"):
            b["target_bbox"][1, 2] = -0.25          # row 1 is a region row (is_image == 0): negative width
    else:
        b = synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"], c["max_masks"],
                                  ragged=c["ragged"], frames=c["frames"])
    return b


def forward_kwargs(c, batch):
    """Keyword arguments of XVLM.forward for a case (reference and HIP model take the same ones)."""
    kw = dict(text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"])
    if c["region"]:
        kw.update(image_atts=batch["image_atts"], idx_to_group_img=batch["idx_to_group_img"],
                  target_bbox=batch["target_bbox"], is_image=batch["is_image"], ret_bbox_loss=True)
    if not c.get("match", True):
        kw.update(ret_match_loss=False)
    return kw


def model_config(case, workdir):
    """Write the two small JSON files the reference's builders read (xvlm.py:122-137, 245-249)
    and return the config dict XVLM(config=...) takes."""
    import json
    import os
    c = CASES[case]
    os.makedirs(workdir, exist_ok=True)
    vw = c.get("vision_width", 768)
    vis = os.path.join(workdir, "config_beit2_%s.json" % ("large" if vw == 1024 else "base"))   # the builders key on the file name
    with open(vis, "w") as f:
        json.dump({"ckpt": "", "vision_width": vw, "patch_size": 16}, f)
    tdir = os.path.join(workdir, "bert-base-uncased-%s" % case)
    os.makedirs(tdir, exist_ok=True)
    with open(os.path.join(tdir, "config.json"), "w") as f:
        json.dump(bert_config_dict(c), f)
    cfg = dict(use_beit_v2=True, vision_config=vis, image_res=c["image_res"], patch_size=16,
               vision_num_hidden_layers=c["vision_layers"], text_encoder=tdir,
               text_num_hidden_layers=c["text_layers"], text_fusion_start_at=c["fusion_at"],
               embed_dim=c["embed_dim"], temp=0.07, max_tokens=c["seq_len"], max_masks=c["max_masks"],
               accelerator={"FP16_OPT_LEVEL": "O0"})
    if c["frames"]:
        cfg.update(video_encoding="avgpool", frame_len=c["frames"], add_frame_pos=True)
    return cfg


def bert_config_dict(c):
    return dict(vocab_size=c["vocab"], hidden_size=c["hidden"], num_attention_heads=c["heads"],
                intermediate_size=c["ffn"], max_position_embeddings=c["max_pos"], type_vocab_size=2,
                hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                layer_norm_eps=1e-12, initializer_range=0.02, pad_token_id=0, model_type="bert")


SMALL = 4096  # grads with at most this many elements are stored in full


def reduce_out(t, full):
    """What a fixture keeps of an activation tensor: everything (tiny cases) or a corner slice
    plus three global moments (big cases).  Tests apply the same function to the tested path."""
    import numpy as np
    import torch
    t = t.detach().to("cpu", torch.float64)
    if full or t.numel() <= SMALL:
        return {"full": t.numpy().astype(np.float32)}
    sl = tuple(slice(0, min(s, 4 if i < t.dim() - 1 else 16)) for i, s in enumerate(t.shape))
    return {"slice": t[sl].numpy().astype(np.float32),
            "moments": np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])}
