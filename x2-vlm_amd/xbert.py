"""BERT text / fusion encoder with the reference's module tree and state-dict keys
(models/xbert.py), executed by the HIP stages EmbeddingsFn / BertLayersFn / MlmLossFn.
Layers >= config.fusion_layer carry a cross-attention sub-block; `mode` selects the layer range
(xbert.py:674-686).  nn.Modules here are parameter containers only.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import kernels as K
from . import ops
from .engine import BertLayersFn, EmbeddingsFn, MlmLossFn, _mask_pad, bert_layer_param_names


_FIXED_SEEDS = []


def _key_mask(atts, neg):
    """0/1 attention mask [S, L] -> additive fp32 key mask (1 - m) * neg in the attention kernels' padded layout."""
    if atts.is_cuda and atts.dtype == torch.int64:
        return K.additive_mask(atts.contiguous(), neg)
    return _mask_pad((1.0 - atts.float()) * neg, atts.shape[1])


def next_dropout_seed():
    """Per-call dropout seed from the host RNG (torch.manual_seed governs it; no device sync).
    Tests push explicit seeds onto _FIXED_SEEDS."""
    if _FIXED_SEEDS:
        return _FIXED_SEEDS.pop(0)
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


class BertConfig:
    """The handful of fields of HF's BertConfig this path reads (config.json of the text encoder dir)."""
    _defaults = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                     initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0)

    def __init__(self, **kw):
        for k, v in dict(self._defaults, **kw).items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.eps = config.layer_norm_eps
        self.config = config

    def forward(self, input_ids):
        drop = K.NO_DROP
        if self.training and self.config.hidden_dropout_prob > 0:
            drop = K.dropout_spec(self.config.hidden_dropout_prob, next_dropout_seed(), 1000)
        return EmbeddingsFn.apply(input_ids, self.eps, drop, self.word_embeddings.weight, self.position_embeddings.weight,
                                  self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias)


class BertSelfAttention(nn.Module):
    def __init__(self, config, is_cross_attention):
        super().__init__()
        kin = config.encoder_width if is_cross_attention else config.hidden_size
        self.query = nn.Linear(config.hidden_size, config.hidden_size)
        self.key = nn.Linear(kin, config.hidden_size)
        self.value = nn.Linear(kin, config.hidden_size)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertAttention(nn.Module):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = BertSelfAttention(config, is_cross_attention)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLayer(nn.Module):
    def __init__(self, config, layer_num):
        super().__init__()
        self.attention = BertAttention(config)
        self.has_cross_attention = layer_num >= config.fusion_layer
        if self.has_cross_attention:
            self.crossattention = BertAttention(config, is_cross_attention=True)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)


_IDENTITY_CSR = {}      # (rows, device) -> (kv_idx, seq_off, seq_ids) of the identity row -> image map


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])

    def run(self, hidden, text_atts, enc=None, enc_atts=None, mode="multi_modal", kv_idx=None):
        """hidden (S,L,Hd) fp32; enc (Bi,T,Dv) image tokens shared through kv_idx (int (S,), None = identity);
        enc_atts (S,T) per text row."""
        cfg = self.config
        lo, hi = {"text": (0, cfg.fusion_layer), "fusion": (cfg.fusion_layer, cfg.num_hidden_layers),
                  "multi_modal": (0, cfg.num_hidden_layers)}[mode]
        S, L, _ = hidden.shape
        drop = None
        if self.training and (cfg.hidden_dropout_prob > 0 or cfg.attention_probs_dropout_prob > 0):
            drop = dict(seed=next_dropout_seed(), p_hidden=cfg.hidden_dropout_prob, p_attn=cfg.attention_probs_dropout_prob)
        meta = dict(lo=lo, hi=hi, fusion_at=cfg.fusion_layer, heads=cfg.num_attention_heads, eps=cfg.layer_norm_eps,
                    self_mask=_key_mask(text_atts, -10000.0), enc_mask=None, kv_idx=None,
                    seq_off=None, seq_ids=None, drop=drop)
        cross = enc is not None and hi > cfg.fusion_layer
        if cross:
            Bi = enc.shape[0]
            # transformers 4.12.5 invert_attention_mask, fp32 branch: (1 - m) * -1e9
            meta["enc_mask"] = _key_mask(enc_atts, -1e9)
            if kv_idx is None:
                assert Bi == S, "encoder batch %d != text batch %d and no kv_idx given" % (Bi, S)
                if enc.is_cuda and torch.is_grad_enabled() and (hidden.requires_grad or enc.requires_grad or self.training) \
                        and K.attn_bwd_form(L, enc.shape[1], shared_kv=True) == 2:
                    # one text row per image (predict_bbox: xvlm.py:905-915) as the identity sharing: the backward then runs as ONE
                    # kernel per layer (attn_bwd_onepass_grouped_kernel, a workgroup per (image, head)) instead of the per-row dQ and
                    # (only when a backward will follow: inference keeps kv_idx None and with it the forward's walk / resident kernels)
                    # dK/dV kernels - and the tail segment needs no second stream (engine.AUX.auto)
                    ident = _IDENTITY_CSR.get((S, enc.device))
                    if ident is None:
                        ar = torch.arange(S + 1, device=enc.device, dtype=torch.int32)
                        ident = (ar[:S].contiguous(), ar, ar[:S].contiguous())
                        if not torch.cuda.is_current_stream_capturing():      # a tensor born inside a capture belongs to that graph's pool
                            _IDENTITY_CSR[(S, enc.device)] = ident
                    meta.update(kv_idx=ident[0], seq_off=ident[1], seq_ids=ident[2])
            else:
                kv = kv_idx.to(torch.int32).contiguous()
                if kv.is_cuda:
                    off, order = K.kv_csr(kv, Bi)          # one launch: kv changes every step with the sampled negatives
                else:
                    order = torch.argsort(kv, stable=True).to(torch.int32).contiguous()
                    off = torch.zeros(Bi + 1, device=kv.device, dtype=torch.int32)
                    off[1:] = torch.cumsum(torch.bincount(kv, minlength=Bi), 0)
                meta.update(kv_idx=kv, seq_off=off, seq_ids=order)
        cache = self.__dict__.setdefault("_param_lists", {})      # see beit2.VisionTransformer._params
        params = cache.get((lo, hi, cross))
        if params is None:
            sd = dict(self.named_parameters())
            params = cache[(lo, hi, cross)] = [sd[n] for n in bert_layer_param_names(lo, hi, cfg.fusion_layer, cross)]
        return BertLayersFn.apply(hidden, enc if cross else None, meta, *params)


class BertModel(nn.Module):
    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        assert not add_pooling_layer
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.apply(lambda m: _bert_init(m, config.initializer_range))

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, input_ids=None, attention_mask=None, encoder_embeds=None, encoder_hidden_states=None,
                encoder_attention_mask=None, return_dict=True, mode="multi_modal", kv_idx=None, **unused):
        """xbert.py:1075-1220 (encoder path: no decoder cache, no head masks)."""
        for k, v in unused.items():
            if v not in (None, False):
                raise NotImplementedError("BertModel.forward(%s=...) is not supported by the HIP path" % k)
        hidden = self.embeddings(input_ids) if encoder_embeds is None else encoder_embeds
        S, L = hidden.shape[:2]
        if attention_mask is None:
            attention_mask = torch.ones(S, L, device=hidden.device)
        if encoder_hidden_states is not None and encoder_attention_mask is None:
            encoder_attention_mask = torch.ones(S, encoder_hidden_states.shape[1], device=hidden.device)
        out = self.encoder.run(hidden, attention_mask, encoder_hidden_states, encoder_attention_mask, mode, kv_idx)
        return SimpleNamespace(last_hidden_state=out) if return_dict else (out,)


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)


class BertForMaskedLM(nn.Module):
    """xbert.py:1567-1673.  decoder.weight is tied to the word embeddings (HF tie_weights)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = BertModel(config)
        self.cls = BertOnlyMLMHead(config)
        self.apply(lambda m: _bert_init(m, config.initializer_range))
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def mlm_loss_from_hidden(self, sequence_output, masked_pos, labels, keep_logits=False):
        """gather masked positions (xbert.py:1588-1589), head, CE.  Returns (loss, lse [B*M], logits [B*M, Vp] or None: the
        logits only exist when asked for - the loss comes from softmax statistics reduced inside the decoder GEMM)."""
        B, L, Hd = sequence_output.shape
        flat = (torch.arange(B, device=masked_pos.device).unsqueeze(1) * L + masked_pos).reshape(-1)
        rows = ops.gather_rows(sequence_output.reshape(B * L, Hd), flat)
        pr = self.cls.predictions
        return MlmLossFn.apply(rows, labels, self.config.layer_norm_eps, keep_logits, pr.transform.dense.weight, pr.transform.dense.bias,
                               pr.transform.LayerNorm.weight, pr.transform.LayerNorm.bias, pr.bias,
                               self.bert.embeddings.word_embeddings.weight)

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, return_dict=True, mode="multi_modal", masked_pos=None, return_logits=False, **unused):
        if masked_pos is None:
            raise NotImplementedError("need check!")     # same as the reference, xbert.py:1650-1651
        h = self.bert(input_ids, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                      encoder_attention_mask=encoder_attention_mask, mode=mode).last_hidden_state
        lab = labels if labels is not None else torch.full_like(masked_pos, -100)
        loss, _, logits = self.mlm_loss_from_hidden(h, masked_pos, lab, keep_logits=return_logits)
        if logits is not None:
            logits = logits[:, :self.config.vocab_size].view(masked_pos.shape[0], masked_pos.shape[1], -1)
        if return_logits:
            return logits
        return SimpleNamespace(loss=loss if labels is not None else None, logits=logits)


def _bert_init(m, std):
    if isinstance(m, (nn.Linear, nn.Embedding)):
        m.weight.data.normal_(mean=0.0, std=std)
    elif isinstance(m, nn.LayerNorm):
        m.bias.data.zero_()
        m.weight.data.fill_(1.0)
    if isinstance(m, nn.Linear) and m.bias is not None:
        m.bias.data.zero_()


def get_bert_config(encoder_rpath, num_hidden_layers=12, cross_start_at=12):
    """xvlm.py:122-137."""
    config = BertConfig.from_json_file(os.path.join(encoder_rpath, "config.json"))
    config.num_hidden_layers = num_hidden_layers
    config.fusion_layer = cross_start_at
    config.embedding_dim = config.hidden_size
    return config
