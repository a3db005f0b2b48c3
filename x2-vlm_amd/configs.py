"""Model configurations of the reference's pre-training YAMLs (configs/pretrain/x2vlm_base_4m.yaml,
x2vlm_large_4m.yaml) as dicts, plus the two small JSON files its builders read
(configs/config_beit2_{base,large}.json and <text_encoder>/config.json)."""
import json
import os

BERT = {
    "base": dict(vocab_size=30522, hidden_size=768, num_attention_heads=12, intermediate_size=3072),
    "large": dict(vocab_size=30522, hidden_size=1024, num_attention_heads=16, intermediate_size=4096),
}


def pretrain_config(workdir, size="base", image_res=224, dropout=None, drop_path_rate=None, **over):
    """XVLM(config=...) dict for X2VLM-{base,large}: BEiT2 + BERT 18 layers with fusion at 12, embed 256."""
    os.makedirs(workdir, exist_ok=True)
    vis = os.path.join(workdir, "config_beit2_%s.json" % size)
    with open(vis, "w") as f:
        json.dump({"ckpt": "", "vision_width": 768 if size == "base" else 1024, "patch_size": 16}, f)
    tdir = os.path.join(workdir, "bert-%s-uncased" % size)
    os.makedirs(tdir, exist_ok=True)
    with open(os.path.join(tdir, "config.json"), "w") as f:
        json.dump(dict(BERT[size], max_position_embeddings=512, type_vocab_size=2, hidden_act="gelu",
                       hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12,
                       initializer_range=0.02, pad_token_id=0), f)
    cfg = dict(use_beit_v2=True, vision_config=vis, image_res=image_res, patch_size=16, text_encoder=tdir,
               text_num_hidden_layers=18, text_fusion_start_at=12, embed_dim=256, temp=0.07, max_tokens=40,
               max_masks=12, accelerator={"FP16_OPT_LEVEL": "O1"})
    if dropout is not None:
        cfg["dropout"] = dropout
    if drop_path_rate is not None:
        cfg["drop_path_rate"] = drop_path_rate
    cfg.update(over)
    return cfg
