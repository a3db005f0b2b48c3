"""Image-text retrieval on the MI355X stages.

  XVLMForRetrieval   models/model_retrieval.py:6-28   fine-tuning forward: (loss_itc, loss_itm) with `idx` soft labels
  rerank_scores      Retrieval.py:113-160             ITM re-ranking of the top-k ITC candidates, both directions

The reference re-ranks one query at a time: a fusion pass over k_test rows, the query's image tokens repeated k_test
times (i2t) or gathered per candidate (t2i).  Here `queries_per_pass` queries share one launch sequence and the image
K/V projections are computed once per DISTINCT image of the pass (kv_idx), never per pair: for i2t that is k_test x
fewer cross-attention K/V GEMM rows, the dominant cost of a fusion layer at L = 30 text tokens vs 197 image tokens.
"""
import torch
import torch.distributed as dist

from . import ops
from .xvlm import XVLMBase


class XVLMForRetrieval(XVLMBase):
    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=True,
                         use_matching_loss=True, use_mlm_loss=False, use_bbox_loss=False)
        self.num_attention_heads = self.text_encoder.config.num_attention_heads
        self.init_params = []

    def forward(self, image, text_ids, text_atts, idx=None):
        image_embeds, image_atts = self.get_vision_embeds(image)
        text_embeds = self.get_text_embeds(text_ids, text_atts)
        with torch.no_grad():
            self.temp.clamp_(0.001, 0.5)
        image_feat, text_feat = self.get_features(image_embeds, text_embeds)
        loss_itc = self.get_contrastive_loss(image_feat, text_feat, idx=idx)
        loss_itm = self.get_matching_loss(image_embeds, image_atts, image_feat, text_embeds, text_atts, text_feat, idx=idx)
        return loss_itc, loss_itm


def _shard(n, rank, world_size):
    step = n // world_size + 1                      # Retrieval.py:118-120
    start = rank * step
    return start, min(n, start + step)


@torch.no_grad()
def rerank_scores(model, image_feats, image_embeds, text_feats, text_atts, text_embeds, k_test, rank=0, world_size=1,
                  queries_per_pass=16, reduce=True):
    """Retrieval.py:113-160.  image_feats [Ni,T,D] / text_feats [Nt,L,Hd]: encoder outputs (the reference's naming);
    image_embeds [Ni,E] / text_embeds [Nt,E]: normalised ITC features.  Returns (score_i2t [Ni,Nt], score_t2i [Nt,Ni]),
    -100 outside each query's top-k; rows of other ranks are filled by the all-reduce when a process group exists."""
    was_training = model.training
    model.eval()
    dev = image_feats.device
    Ni, Nt = image_feats.shape[0], text_feats.shape[0]
    image_atts = torch.ones(image_feats.shape[:2], dtype=torch.long, device=dev)
    sims = ops.linear(image_embeds.float(), text_embeds.float())              # image_embeds @ text_embeds.t()
    score_i2t = torch.full((Ni, Nt), -100.0, device=dev)
    score_t2i = torch.full((Nt, Ni), -100.0, device=dev)

    def itm(images, atts, texts, tatts, t_idx, kv):
        cls = model._fusion_cls(images, atts, texts, tatts, t_idx.to(torch.int32), kv.to(torch.int32))
        return ops.mlp_head(model.itm_head, cls)[:, 1]

    start, end = _shard(Ni, rank, world_size)
    for q0 in range(start, end, queries_per_pass):
        q1 = min(end, q0 + queries_per_pass)
        topk = sims[q0:q1].topk(k=k_test, dim=1).indices                      # [nq, k] text ids
        nq = q1 - q0
        kv = torch.arange(nq, device=dev).repeat_interleave(k_test)           # pair -> image of the pass
        uniq, inv = torch.unique(topk.reshape(-1), return_inverse=True)       # pair -> text of the pass
        score = itm(image_feats[q0:q1], image_atts[q0:q1], text_feats[uniq], text_atts[uniq], inv, kv)
        score_i2t[q0:q1].scatter_(1, topk, score.view(nq, k_test))
    sims_t = sims.t()
    start, end = _shard(Nt, rank, world_size)
    for q0 in range(start, end, queries_per_pass):
        q1 = min(end, q0 + queries_per_pass)
        topk = sims_t[q0:q1].topk(k=k_test, dim=1).indices                    # [nq, k] image ids
        nq = q1 - q0
        uniq, inv = torch.unique(topk.reshape(-1), return_inverse=True)       # distinct images of the pass: K/V once each
        t_idx = torch.arange(nq, device=dev).repeat_interleave(k_test)
        score = itm(image_feats[uniq], image_atts[uniq], text_feats[q0:q1], text_atts[q0:q1], t_idx, inv)
        score_t2i[q0:q1].scatter_(1, topk, score.view(nq, k_test))
    if reduce and world_size > 1 and dist.is_available() and dist.is_initialized():
        # every entry is written by exactly one rank; the others hold -100 there (Retrieval.py:154-157 sums them as is)
        dist.barrier()
        dist.all_reduce(score_i2t, op=dist.ReduceOp.SUM)
        dist.all_reduce(score_t2i, op=dist.ReduceOp.SUM)
    model.train(was_training)
    return score_i2t, score_t2i
