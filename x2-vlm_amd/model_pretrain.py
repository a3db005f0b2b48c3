"""XVLM pre-training wrapper (models/model_pretrain.py:24-88): forward(...) -> dict of losses.

Same inputs, outputs and state-dict keys as the reference.  MI355X-first execution order: the two
text-encoder passes (clean + masked ids) run as one 2B-row batch and the four fusion passes
(ITM positives, 2 x ITM hard negatives, MLM) as one 4B-row batch whose cross-attention K/V are
projected once per image per layer (rows address their image through kv_idx).  Same arithmetic per
row as the reference's separate passes; 4x fewer, 4x larger GEMMs; F_min FLOPs.
"""
import torch

from . import kernels as K
from . import ops
from .xvlm import XVLMBase


class XVLM(XVLMBase):
    def __init__(self, config, load_vision_params=True, load_text_params=True, pretraining=True):
        super().__init__(config, load_vision_params=load_vision_params, load_text_params=load_text_params,
                         use_contrastive_loss=True, use_matching_loss=True, use_mlm_loss=True, use_bbox_loss=True,
                         config_text=None, pretraining=pretraining)
        self.overlap_towers = True
        self._text_stream = None

    # ---- the multimodal step in four pieces (towers | features | losses), so that a caller can place them on streams /
    # hipGraph segments of its own (graph.SegmentedStep) and cut autograd at the tower outputs; forward_multimodal below
    # composes them for everybody else.
    def tower_text(self, text_ids, text_atts, text_ids_masked):
        """Text layers on [clean ; masked] ids as ONE 2B-row batch -> (2B, L, Hd)."""
        return self.get_text_embeds(torch.cat([text_ids, text_ids_masked]), torch.cat([text_atts, text_atts]))

    def tower_vision(self, image, image_atts=None, idx_to_group_img=None, ret_bbox_loss=False):
        """-> (image_embeds, image_atts, image_embeds_fullatts or None)."""
        if ret_bbox_loss:
            return self.get_vision_embeds(image, image_atts=image_atts, idx_to_group_img=idx_to_group_img)
        return self.get_vision_embeds(image) + (None,)

    def tail_features(self, image_embeds, both):
        """-> (image_feat, text_feat), the (B, 256) ITC features of this rank (clean text rows = both[:B])."""
        return self.get_features(image_embeds, both[:image_embeds.shape[0]])

    def tail_losses(self, image_embeds, image_atts, both, image_feat, text_feat, text_atts, masked_pos, masked_ids,
                    image_embeds_fullatts=None, target_bbox=None, is_image=None, ret_bbox_loss=False, ret_match_loss=True,
                    gathered=None):
        """ITC + hard negatives + the 4B-row fusion pass + ITM + MLM (+ bbox).  gathered: (image_feat_all, text_feat_all)
        when the caller has already exchanged the features between ranks (None: get_contrastive_loss all-gathers)."""
        B = image_embeds.shape[0]
        dev = both.device
        text_embeds = both[:B]
        loss_itc = self.get_contrastive_loss(image_feat, text_feat, gathered=gathered)
        # rows of `both` that enter the fusion pass: pos | neg(text b, image neg) | neg(text neg, image b) | masked, the image each
        # attends to, and the two attention masks per row
        ineg, tneg = self.get_hard_negatives(image_feat, text_feat) if ret_match_loss else (None, None)
        if both.is_cuda:
            t_idx, kv, atts, enc_atts = K.tail_index(ineg, tneg, text_atts.contiguous(), image_atts.contiguous(), with_match=ret_match_loss)
        else:
            ar = torch.arange(B, device=dev, dtype=torch.int32)
            t_idx, kv = (torch.cat([ar, ar, tneg, ar + B]), torch.cat([ar, ineg, ar, ar])) if ret_match_loss else (ar + B, ar)
            atts, enc_atts = torch.cat([text_atts, text_atts])[t_idx.long()], image_atts[kv.long()]
        h0 = ops.gather_rows(both, t_idx)
        fused = self.get_cross_embeds(image_embeds, enc_atts, text_embeds=h0, text_atts=atts, kv_idx=kv)
        if ret_match_loss:
            loss_itm = self._itm_loss(fused[:3 * B, 0, :], B)
        else:
            loss_itm = torch.tensor(0.0)
        loss_mlm, self.last_mlm_lse, self.last_mlm_logits = self.text_encoder.mlm_loss_from_hidden(fused[-B:], masked_pos, masked_ids)
        loss = {"loss_itc": loss_itc, "loss_itm": loss_itm, "loss_mlm": loss_mlm}
        # detached: holding graph tensors here would keep the step's autograd graph (and its AccumulateGrad nodes) alive
        # across iterations
        self.last = dict(image_embeds=image_embeds.detach(), text_embeds=text_embeds.detach(), image_feat=image_feat.detach(),
                         text_feat=text_feat.detach())
        if ret_bbox_loss:
            output_coord = self.predict_bbox(image_embeds_fullatts, text_embeds, text_atts)
            self.last["bbox_coord"] = output_coord.detach()
            loss["loss_bbox"], loss["loss_giou"] = self.get_bbox_loss(output_coord, target_bbox, is_image=is_image)
        return loss

    def forward_multimodal(self, image, text_ids, text_atts, text_ids_masked=None, masked_pos=None, masked_ids=None,
                           image_atts=None, idx_to_group_img=None, target_bbox=None, is_image=None,
                           ret_bbox_loss=False, ret_match_loss=True):
        # text layers on [clean ; masked] ids in one batch, on a second HIP stream: the text tower (15 % of the FLOPs,
        # small GEMMs) is independent of the vision tower until the features meet, so their kernels interleave on the
        # GPU (autograd replays each stage's backward on the stream its forward ran on)
        overlap = self.overlap_towers and image.is_cuda
        if overlap:
            if self._text_stream is None:
                self._text_stream = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            self._text_stream.wait_stream(main)
            with torch.cuda.stream(self._text_stream):
                both = self.tower_text(text_ids, text_atts, text_ids_masked)
        image_embeds, image_atts, image_embeds_fullatts = self.tower_vision(image, image_atts, idx_to_group_img, ret_bbox_loss)
        if overlap:
            main.wait_stream(self._text_stream)
            both.record_stream(main)
        else:
            both = self.tower_text(text_ids, text_atts, text_ids_masked)
        image_feat, text_feat = self.tail_features(image_embeds, both)
        return self.tail_losses(image_embeds, image_atts, both, image_feat, text_feat, text_atts, masked_pos, masked_ids,
                                image_embeds_fullatts=image_embeds_fullatts, target_bbox=target_bbox, is_image=is_image,
                                ret_bbox_loss=ret_bbox_loss, ret_match_loss=ret_match_loss)

    def forward_text(self, text_ids=None, text_atts=None, text_ids_masked=None, masked_pos=None, masked_ids=None):
        """model_pretrain.py:67-72 (Pretrain.run_text_iter): the masked ids through every text layer WITHOUT cross-attention
        (no encoder states: xbert.py:595), MLM head.  = get_mlm_loss(text_ids_masked, text_atts, None, None, ...), spelled out so
        that the inspection copies the parity tests read (last_mlm_lse / last_mlm_logits / last) exist for this branch too."""
        seq = self._bert(text_ids_masked, attention_mask=text_atts, mode="multi_modal").last_hidden_state
        loss, self.last_mlm_lse, self.last_mlm_logits = self.text_encoder.mlm_loss_from_hidden(seq, masked_pos, masked_ids)
        self.last = dict(text_embeds=seq.detach())
        return {"loss_mlm": loss}

    def forward(self, image=None, text_ids=None, text_atts=None, text_ids_masked=None, masked_pos=None, masked_ids=None,
                image_atts=None, idx_to_group_img=None, target_bbox=None, is_image=None, ret_bbox_loss=False,
                ret_match_loss=True):
        if image is None:
            return self.forward_text(text_ids, text_atts, text_ids_masked, masked_pos, masked_ids)
        return self.forward_multimodal(image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, image_atts,
                                       idx_to_group_img, target_bbox, is_image, ret_bbox_loss, ret_match_loss=ret_match_loss)
