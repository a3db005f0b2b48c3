"""XVLMBase: model assembly, encoders' public getters and the ITC / ITM / MLM / bbox losses with
the reference's API and state-dict keys (models/xvlm.py:140-169, 463-957), running on the HIP
stages.  Beyond the reference: `get_cross_embeds(..., kv_idx=...)` lets several text rows share one
image's K/V (used by the pre-training step to run the 4 fusion passes as one batch).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import box_ops, checkpoint, ops
from . import kernels as K
from .beit2 import beit_base_patch16, beit_large_patch16, read_json, trunc_normal_
from .xbert import BertForMaskedLM, BertModel, get_bert_config


class AllGather(torch.autograd.Function):
    """all_gather whose backward keeps only the local rows of the incoming gradient (no
    reduce-scatter): xvlm.py:140-160.  RCCL when the process group is 'nccl' on ROCm."""

    @staticmethod
    def forward(ctx, tensor, rank, world_size):
        out = [torch.empty_like(tensor) for _ in range(world_size)]
        dist.all_gather(out, tensor.contiguous())
        ctx.rank, ctx.batch_size = rank, tensor.shape[0]
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None, None


def allgather(t):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    return AllGather.apply(t, dist.get_rank(), dist.get_world_size())


def build_mlp(input_dim, output_dim):
    return nn.Sequential(nn.Linear(input_dim, input_dim * 2), nn.LayerNorm(input_dim * 2), nn.GELU(),
                         nn.Linear(input_dim * 2, output_dim))


def build_vision_encoder(config, load_params=False):
    """xvlm.py:172-283, use_beit_v2 branch (the only one on this path)."""
    if not config.get("use_beit_v2", False):
        raise ValueError("only use_beit_v2 vision encoders are built by the MI355X path")
    vc = read_json(config["vision_config"])
    assert config["patch_size"] == vc["patch_size"]
    if "base" in config["vision_config"]:
        fn = beit_base_patch16
    elif "large" in config["vision_config"]:
        fn = beit_large_patch16
    else:
        raise ValueError
    enc = fn(img_size=config["image_res"], drop_path_rate=config.get("drop_path_rate", 0.1), init_values=0.1,
             vision_num_hidden_layers=config.get("vision_num_hidden_layers", -1))
    if load_params:
        checkpoint.load_pretrained_beit2(enc, vc["ckpt"])          # xvlm.py:264-266
    enc.vision_width = vc["vision_width"]
    return enc


def build_text_encoder(config, vision_width, load_text_params=False, use_mlm_loss=False, config_text=None):
    """xvlm.py:286-387."""
    if config_text is None:
        config_text = get_bert_config(config["text_encoder"], num_hidden_layers=config["text_num_hidden_layers"],
                                      cross_start_at=config["text_fusion_start_at"])
    config_text.hidden_dropout_prob = config.get("dropout", config_text.hidden_dropout_prob)
    config_text.encoder_width = vision_width
    enc = BertForMaskedLM(config_text) if use_mlm_loss else BertModel(config_text)
    missing = checkpoint.load_text_params(enc, config, config_text, use_mlm_loss) if load_text_params else []
    return enc, missing


class XVLMBase(nn.Module):
    def __init__(self, config=None, load_vision_params=False, load_text_params=False, load_cross_params=False,
                 use_contrastive_loss=False, use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False,
                 config_text=None, pretraining=False):
        super().__init__()
        self.init_params = []
        self.vision_encoder = build_vision_encoder(config, load_params=load_vision_params)
        self.vision_width = self.vision_encoder.vision_width
        self.text_encoder, missing = build_text_encoder(config, self.vision_width, load_text_params, use_mlm_loss, config_text)
        self.update_init_params(["text_encoder.%s" % k for k in missing])          # xvlm.py:477
        tc = self.text_encoder.config
        self.vocab_size, self.num_text_layers, self.text_width = tc.vocab_size, tc.fusion_layer, tc.hidden_size
        self.num_cross_layers, self.cross_width = tc.num_hidden_layers - tc.fusion_layer, tc.hidden_size
        self.cross_encoder = None
        self.video_encoding = config.get("video_encoding", "")
        if self.video_encoding not in ("", "avgpool"):
            raise ValueError("Not Supported video_encoding == %s" % config["video_encoding"])
        if self.video_encoding:
            self.frame_len, self.add_frame_pos = config["frame_len"], config["add_frame_pos"]
            if self.add_frame_pos:
                self.absolute_frame_pos_embed = nn.Parameter(torch.zeros(1, self.frame_len, 1, self.vision_width))
                trunc_normal_(self.absolute_frame_pos_embed)
                self.update_init_params(["absolute_frame_pos_embed"])
        self.use_contrastive_loss = use_contrastive_loss
        if use_contrastive_loss:
            self.embed_dim = config["embed_dim"]
            self.vision_proj = nn.Linear(self.vision_width, self.embed_dim)
            self.text_proj = nn.Linear(self.text_width, self.embed_dim)
            self.update_init_params(["vision_proj." + n for n, _ in self.vision_proj.named_parameters()])
            self.update_init_params(["text_proj." + n for n, _ in self.text_proj.named_parameters()])
            if config.get("fix_temp", False):
                self.temp = torch.ones([]) * config["temp"]
            else:
                self.temp = nn.Parameter(torch.ones([]) * config["temp"])
                self.update_init_params(["temp"])
        self.use_matching_loss = use_matching_loss
        if use_matching_loss:
            self.itm_head = build_mlp(self.text_width, 2)
            self.update_init_params(["itm_head." + n for n, _ in self.itm_head.named_parameters()])
        self.use_bbox_loss = use_bbox_loss
        if use_bbox_loss:
            self.bbox_head = build_mlp(self.text_width, 4)
            self.update_init_params(["bbox_head." + n for n, _ in self.bbox_head.named_parameters()])
        if not pretraining:
            self.init_params = []
        self.injected_negatives = None     # parity tests: (image_neg_idx, text_neg_idx) instead of sampling

    def update_init_params(self, missing_keys=None):
        for k in missing_keys or []:
            if k not in self.init_params:
                self.init_params.append(k)
        named = set(n for n, _ in self.named_parameters())
        self.init_params = [n for n in self.init_params if n in named]

    def load_pretrained(self, ckpt_rpath, config, is_eval=False, is_domain_pretrain=False):
        """Load an X2-VLM checkpoint (models/xvlm.py:579-613).  Plain pre-training / fine-tuning start: vision tables are
        resampled to this model's resolution (unless is_eval) and text keys are normalised by checkpoint.load_pretrained;
        is_domain_pretrain continues from a checkpoint of this very architecture and takes the keys as they are.
        Parameters the checkpoint does not provide join init_params (the lr * lr_mult group).  Returns the
        load_state_dict result."""
        if is_domain_pretrain:
            state = checkpoint._model_state(ckpt_rpath)
            if config.get("init_timesformer", False):
                checkpoint.init_timesformer_keys(state)
        else:
            state = checkpoint.load_pretrained(self, ckpt_rpath, config, is_eval=is_eval, load_text=True)
        own = getattr(self, "absolute_frame_pos_embed", None)
        given = state.get("absolute_frame_pos_embed")
        if own is not None and given is not None and given.shape != own.shape:
            # a checkpoint trained with another clip length: keep the frame positions both have
            n = min(given.shape[1], own.shape[1])
            with torch.no_grad():
                own[:, :n] = given[:, :n]
            del state["absolute_frame_pos_embed"]
            print("frame position embedding: %d of %d checkpoint frames used" % (n, given.shape[1]), flush=True)
        result = self.load_state_dict(state, strict=False)
        self.update_init_params(list(result.missing_keys))
        print("checkpoint %s: %d unexpected keys; trained from scratch: %s"
              % (ckpt_rpath, len(result.unexpected_keys), sorted(self.init_params)), flush=True)
        return result

    # ------------------------------------------------------------------ encoders
    @property
    def _bert(self):
        return self.text_encoder.bert if hasattr(self.text_encoder, "bert") else self.text_encoder

    def get_frame_embeds(self, frame, **unused):
        assert frame.dim() == 5
        B, Fr = frame.shape[:2]
        e = self.vision_encoder(frame.reshape(B * Fr, *frame.shape[2:]))
        pos = self.absolute_frame_pos_embed if self.add_frame_pos else torch.zeros(1, Fr, 1, e.shape[-1], device=e.device)
        e = ops.frame_mean(e, pos, Fr)
        return e, torch.ones(e.shape[:2], dtype=torch.long, device=e.device)

    def get_image_embeds(self, image, image_atts=None, idx_to_group_img=None, **unused):
        assert image.dim() == 4
        if idx_to_group_img is None:
            e = self.vision_encoder(image)
            return e, torch.ones(e.shape[:2], dtype=torch.long, device=image.device)
        if image_atts is None:
            full = ops.gather_rows(self.vision_encoder(image), idx_to_group_img)
            return full, torch.ones(full.shape[:2], dtype=torch.long, device=image.device)
        assert image_atts.size(0) == idx_to_group_img.size(0)
        e, full = self.vision_encoder(image, idx_to_group_img=idx_to_group_img, image_atts=image_atts)
        return e, image_atts, ops.gather_rows(full, idx_to_group_img)

    def get_vision_embeds(self, image, image_atts=None, idx_to_group_img=None, output_hidden_states=None, output_attentions=None):
        assert output_hidden_states == output_attentions
        if image.dim() == 5:
            assert idx_to_group_img is None, "not supported"
            return self.get_frame_embeds(image)
        return self.get_image_embeds(image, image_atts=image_atts, idx_to_group_img=idx_to_group_img)

    def get_text_embeds(self, text_ids, text_atts, **unused):
        return self._bert(text_ids, attention_mask=text_atts, mode="text").last_hidden_state

    def get_text_embeds_12L(self, text_ids, text_atts, **unused):
        return self._bert(text_ids, attention_mask=text_atts).last_hidden_state

    def get_cross_embeds(self, image_embeds, image_atts, text_ids=None, text_embeds=None, text_atts=None, kv_idx=None, **unused):
        assert text_atts is not None
        if text_embeds is not None:
            return self._bert(encoder_embeds=text_embeds, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                              encoder_attention_mask=image_atts, mode="fusion", kv_idx=kv_idx).last_hidden_state
        if text_ids is not None:
            return self._bert(text_ids, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                              encoder_attention_mask=image_atts, kv_idx=kv_idx).last_hidden_state
        raise ValueError

    # ------------------------------------------------------------------ heads and losses
    def get_features(self, image_embeds=None, text_embeds=None):
        fi = ft = None
        if image_embeds is not None:
            fi = ops.normalize(ops.linear(image_embeds[:, 0, :], self.vision_proj.weight, self.vision_proj.bias))
        if text_embeds is not None:
            ft = ops.normalize(ops.linear(text_embeds[:, 0, :], self.text_proj.weight, self.text_proj.bias))
        if image_embeds is None:
            return ft
        if text_embeds is None:
            return fi
        return fi, ft

    def _sim(self, a, b):
        """a @ b^T / temp (fp32)."""
        return ops.linear(a, b) / self.temp

    def get_contrastive_loss(self, image_feat, text_feat, idx=None, gathered=None):
        """xvlm.py:794-826.  gathered: (image_feat_all, text_feat_all) already exchanged between the ranks by the caller
        (graph.SegmentedStep issues the all-gather between two hipGraph segments); None: all-gather here."""
        assert image_feat.size(-1) == self.embed_dim and text_feat.size(-1) == self.embed_dim
        fi, ft = gathered if gathered is not None else (allgather(image_feat), allgather(text_feat))
        logits = self._sim(fi, ft)
        n = logits.shape[0]
        if idx is None:
            labels = torch.arange(n, device=logits.device)
            return (ops.cross_entropy(logits, labels) + ops.cross_entropy(logits.t(), labels)) / 2
        idx = idx.view(-1, 1)
        idx_all = allgather(idx)
        pos = torch.eq(idx_all, idx_all.t()).float()
        lab = pos / pos.sum(1, keepdim=True)
        # soft-label branch (fine-tuning only, not on the pre-training hot path): plain torch log-softmax
        li = -torch.sum(torch.log_softmax(logits, dim=1) * lab, dim=1).mean()
        lt = -torch.sum(torch.log_softmax(logits.t(), dim=1) * lab, dim=1).mean()
        return (li + lt) / 2

    def get_hard_negatives(self, image_feat, text_feat, idx=None):
        """xvlm.py:828-857 without the 2*B host syncs: one batched inverse-CDF draw per direction on
        the device.  Returns int32 device tensors (image_neg_idx, text_neg_idx)."""
        if self.injected_negatives is not None:
            return tuple(torch.as_tensor(n, dtype=torch.int32, device=image_feat.device) for n in self.injected_negatives)
        with torch.no_grad():
            fi, ft = image_feat.detach(), text_feat.detach()
            temp = self.temp.detach().reshape(1).float()
            bs = fi.shape[0]
            sim_i2t = K.linear_f32(fi, ft, alpha_ptr=None) / temp
            sim_t2i = K.linear_f32(ft, fi, alpha_ptr=None) / temp
            u = torch.rand(2, bs, device=fi.device)
            grp = idx.view(-1).long().contiguous() if idx is not None else None
            if grp is not None and bool((grp.view(-1, 1) == grp.view(1, -1)).all(1).any()):
                # fine-tuning path only (one host sync; the reference has 2*B there): a row whose every candidate shares its
                # `idx` has no negative to draw - torch.multinomial raises on the all-zero weights (xvlm.py:845-855)
                raise RuntimeError("get_hard_negatives: a row has no candidate with a different idx (invalid multinomial distribution)")
            return K.sample_negatives(sim_t2i, u[0].contiguous(), grp), K.sample_negatives(sim_i2t, u[1].contiguous(), grp)

    def get_matching_loss(self, image_embeds, image_atts, image_feat, text_embeds, text_atts, text_feat, idx=None):
        """xvlm.py:859-899: B positives + 2B hard negatives as ONE 3B-row fusion pass."""
        ineg, tneg = self.get_hard_negatives(image_feat, text_feat, idx=idx)
        B = image_embeds.shape[0]
        ar = torch.arange(B, device=image_embeds.device, dtype=torch.int32)
        t_idx = torch.cat([ar, ar, tneg])                 # text rows: pos, (text b, image neg), (text neg, image b)
        kv = torch.cat([ar, ineg, ar])
        cls = self._fusion_cls(image_embeds, image_atts, text_embeds, text_atts, t_idx, kv)
        return self._itm_loss(cls, B)

    def _fusion_cls(self, image_embeds, image_atts, text_embeds, text_atts, t_idx, kv):
        h0 = ops.gather_rows(text_embeds, t_idx)
        out = self.get_cross_embeds(image_embeds, image_atts[kv.long()], text_embeds=h0, text_atts=text_atts[t_idx.long()], kv_idx=kv)
        return out[:, 0, :]

    def _itm_loss(self, cls, B):
        logits = ops.mlp_head(self.itm_head, cls)
        cache = self.__dict__.setdefault("_itm_labels", {})                    # constant per (B, device): built once, no launch per step
        labels = cache.get((B, cls.device))
        if labels is None:
            labels = torch.zeros(3 * B, dtype=torch.long, device=cls.device)
            labels[:B] = 1
            cache[(B, cls.device)] = labels
        self.last_itm_logits = logits.detach()
        return ops.cross_entropy(logits, labels)

    def get_mlm_loss(self, text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids):
        return self.text_encoder(text_ids_masked, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                                 encoder_attention_mask=image_atts, labels=masked_ids, masked_pos=masked_pos).loss

    def predict_bbox(self, image_embeds, text_embeds, text_atts):
        """xvlm.py:910-925."""
        assert image_embeds.size(0) == text_embeds.size(0)
        ones = torch.ones(image_embeds.shape[:2], device=image_embeds.device)
        cls = self.get_cross_embeds(image_embeds, ones, text_embeds=text_embeds, text_atts=text_atts)[:, 0, :]
        return ops.mlp_head(self.bbox_head, cls).sigmoid()

    def get_bbox_loss(self, output_coord, target_bbox, is_image=None):
        """xvlm.py:927-957 ((B,4) elementwise maths; GIoU of matched pairs only)."""
        loss_bbox = (output_coord - target_bbox).abs()
        b1, b2 = box_ops.box_cxcywh_to_xyxy(output_coord), box_ops.box_cxcywh_to_xyxy(target_bbox)
        # The reference's early-out (xvlm.py:943-946: any degenerate box -> every row's GIoU term is 0 and generalized_box_iou is
        # NEVER evaluated), without its host sync: the GIoU runs on boxes made safe under the flag - a unit box for every row
        # when the batch is degenerate - so the branch that is not selected holds no 0/0 whose NaN would poison the backward
        # (torch.where passes a zero gradient into it, and 0 x NaN = NaN through the division).
        degenerate = ((b1[:, 2:] < b1[:, :2]).any() | (b2[:, 2:] < b2[:, :2]).any())
        unit = torch.cat([torch.zeros_like(b1[:, :2]), torch.ones_like(b1[:, 2:])], dim=-1)    # device-side fills only: capturable
        giou = 1 - box_ops.generalized_box_iou_pairs(torch.where(degenerate, unit, b1), torch.where(degenerate, unit, b2))
        loss_giou = torch.where(degenerate, torch.zeros_like(giou), giou)
        if is_image is None:
            num_boxes = target_bbox.size(0)
        else:
            keep = (1 - is_image).to(loss_bbox.dtype)
            num_boxes = keep.sum()
            loss_bbox = loss_bbox * keep.view(-1, 1)
            loss_giou = loss_giou * keep
        return loss_bbox.sum() / num_boxes, loss_giou.sum() / num_boxes
