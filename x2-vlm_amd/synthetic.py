"""Deterministic synthetic weights and batches for X^2-VLM parity tests, smoke and bench.

Everything here is generated with numpy's PCG64 (platform-stable), keyed by (seed, tensor name),
so a golden fixture only has to carry the seed: the reference run that produced the fixture
(tests/golden/make_golden.py, executed in the build container) and the parity tests on the GPU
box regenerate bit-identical weights and inputs from it.

The input layout follows the reference's collate output (dataset/pretrain_dataset.py:242-287,
612-660): image (B,3,R,R) f32; text_ids/text_atts/text_ids_masked (B,L) i64; masked_pos (B,M)
i64 padded with 0; masked_ids (B,M) i64 padded with -100; region extras idx_to_group_img (B,),
image_atts (B,1+P) i64, target_bbox (B,4) f32 cxcywh in [0,1], is_image (B,) i64.
"""
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode()) & 0x7FFFFFFF])


def _is_norm(name):
    n = name.lower()
    return ("layernorm" in n) or (".norm1." in n) or (".norm2." in n) or ("fc_norm" in n) \
        or n.startswith("itm_head.1.") or n.startswith("bbox_head.1.")


def synth_tensor(name, shape, seed):
    """One parameter tensor (float32) as a function of (seed, name, shape)."""
    shape = tuple(int(s) for s in shape)
    g = _rng(seed, name).standard_normal(shape, dtype=np.float64)
    if name == "temp" or name.endswith(".temp"):
        out = np.full(shape, 0.07)
    elif _is_norm(name):
        out = (1.0 + 0.1 * g) if name.endswith("weight") else 0.1 * g
    elif name.endswith("gamma_1") or name.endswith("gamma_2"):
        out = 0.3 + 0.1 * g
    elif name.endswith("relative_position_bias_table"):
        out = 0.5 * g
    elif name.endswith("cls_token"):
        out = 0.5 * g
    elif name.endswith("absolute_frame_pos_embed"):
        out = 0.1 * g
    elif "embeddings." in name and name.endswith("weight"):
        out = 0.05 * g
    elif name.endswith("bias"):
        out = 0.1 * g
    else:  # linear / conv weights
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else (shape[0] if shape else 1)
        out = g / np.sqrt(max(fan_in, 1))
    return torch.from_numpy(np.ascontiguousarray(out, dtype=np.float32)).reshape(shape)


def synth_state_dict(model, seed):
    """Fill every floating-point parameter of `model` (reference, oracle or HIP module) in place.

    Tied tensors (MLM decoder <-> word embeddings, xbert.py:1578-1581) are filled once under the
    first name `named_parameters()` reports for them.  Integer buffers are left untouched.
    """
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(synth_tensor(name, p.shape, seed).to(p.dtype))
    return model


def subword_flags(vocab):
    """uint8 [V] from a tokenizer vocabulary {token text: id} (tokenizer.get_vocab()): 1 where the token is a WordPiece continuation ('##...').
    What the reference's TextMaskingGenerator asks of every token it expands a whole word over (dataset/pretrain_dataset.py:89-93)."""
    flags = np.zeros(len(vocab), dtype=np.uint8)
    for tok, i in vocab.items():
        flags[i] = 1 if tok.startswith("##") else 0
    return torch.from_numpy(flags)


def synth_subword_flags(vocab_size):
    """Stand-in for subword_flags() when there is no tokenizer (bench, smoke, tests): every fifth id from 1000 on is a '##' piece - about the
    share bert-base-uncased has (5 828 of 30 522)."""
    ids = np.arange(vocab_size)
    return torch.from_numpy(((ids >= min(1000, vocab_size // 2)) & (ids % 5 == 0)).astype(np.uint8))


def masking_config(config, is_subword, seed=0, big_vocab=True):
    """kernels.mask_tokens / graph.SegmentedStep(masking=...) arguments from the reference's config keys (configs/pretrain/x2vlm_base_4m.yaml:53-57)."""
    cls_id, mask_id = (101, 103) if big_vocab else (1, 3)
    return dict(is_subword=is_subword, mask_prob=float(config.get("mask_prob", 0.5)), max_masks=int(config.get("max_masks", 12)),
                skipgram_prb=float(config.get("skipgram_prb", 0.2)), skipgram_size=int(config.get("skipgram_size", 3)),
                mask_whole_word=bool(config.get("mask_whole_word", True)), cls_id=cls_id, mask_id=mask_id, seed=int(seed))


def synth_batch(seed, batch, seq_len, image_res, vocab_size, max_masks=12, ragged=False,
                frames=0):
    """Image-text batch in the reference's collate layout. Returns a dict of CPU tensors."""
    r = _rng(seed, "batch")
    big = vocab_size > 2000
    cls_id, sep_id, mask_id = (101, 102, 103) if big else (1, 2, 3)
    lo = 1000 if big else 5
    if frames:
        image = r.standard_normal((batch, frames, 3, image_res, image_res), dtype=np.float32)
    else:
        image = r.standard_normal((batch, 3, image_res, image_res), dtype=np.float32)
    ids = r.integers(lo, vocab_size, size=(batch, seq_len), dtype=np.int64)
    atts = np.ones((batch, seq_len), dtype=np.int64)
    lens = np.full((batch,), seq_len, dtype=np.int64)
    if ragged:
        lens = r.integers(max(4, seq_len // 3), seq_len + 1, size=(batch,), dtype=np.int64)
        lens[0] = seq_len
    masked_pos = np.zeros((batch, max_masks), dtype=np.int64)
    masked_ids = np.full((batch, max_masks), -100, dtype=np.int64)
    ids_masked = ids.copy()
    for b in range(batch):
        n = int(lens[b])
        ids[b, 0] = cls_id
        ids[b, n - 1] = sep_id
        ids[b, n:] = 0
        atts[b, n:] = 0
        # reference: n_mask = min(max_masks, max(1, round(mask_prob * n_words))), pretrain_dataset.py:60-62
        n_mask = int(min(max_masks, max(1, round(0.5 * (n - 2)))))
        pos = np.sort(r.permutation(n - 2)[:n_mask] + 1)
        masked_pos[b, :n_mask] = pos
        masked_ids[b, :n_mask] = ids[b, pos]
    ids_masked = ids.copy()
    for b in range(batch):
        k = int((masked_ids[b] != -100).sum())
        ids_masked[b, masked_pos[b, :k]] = mask_id
    out = dict(image=image, text_ids=ids, text_atts=atts, text_ids_masked=ids_masked,
               masked_pos=masked_pos, masked_ids=masked_ids)
    return {k: torch.from_numpy(v) for k, v in out.items()}


def synth_region_batch(seed, n_images, batch, seq_len, image_res, patch_size, vocab_size,
                       max_masks=12):
    """Region batch: `n_images` images shared by `batch` (text, box) rows
    (reference layout: dataset/pretrain_dataset.py:612-660)."""
    d = synth_batch(seed, batch, seq_len, image_res, vocab_size, max_masks, ragged=True)
    r = _rng(seed, "region")
    d["image"] = torch.from_numpy(
        r.standard_normal((n_images, 3, image_res, image_res), dtype=np.float32))
    idx = np.sort(r.integers(0, n_images, size=(batch,), dtype=np.int64))
    idx[0] = 0
    grid = image_res // patch_size
    atts = np.zeros((batch, 1 + grid * grid), dtype=np.int64)
    bbox = np.zeros((batch, 4), dtype=np.float32)
    is_image = np.zeros((batch,), dtype=np.int64)
    for b in range(batch):
        if b % 3 == 2:  # whole-image row: full attention, box = whole image
            is_image[b] = 1
            atts[b] = 1
            bbox[b] = (0.5, 0.5, 1.0, 1.0)
            continue
        x0 = int(r.integers(0, grid)); y0 = int(r.integers(0, grid))
        x1 = int(r.integers(x0 + 1, grid + 1)); y1 = int(r.integers(y0 + 1, grid + 1))
        m = np.zeros((grid, grid), dtype=np.int64)
        m[y0:y1, x0:x1] = 1
        atts[b, 0] = 1
        atts[b, 1:] = m.reshape(-1)
        bbox[b] = ((x0 + x1) / 2 / grid, (y0 + y1) / 2 / grid, (x1 - x0) / grid, (y1 - y0) / grid)
    d.update(idx_to_group_img=torch.from_numpy(idx), image_atts=torch.from_numpy(atts),
             target_bbox=torch.from_numpy(bbox), is_image=torch.from_numpy(is_image))
    return d


def synth_negatives(seed, batch):
    """Injected hard-negative indices (image_neg_idx, text_neg_idx), never the diagonal.
    Parity tests inject these instead of torch.multinomial draws (xvlm.py:845-855)."""
    r = _rng(seed, "negatives")
    off_i = r.integers(1, batch, size=(batch,))
    off_t = r.integers(1, batch, size=(batch,))
    ar = np.arange(batch)
    return [int(x) for x in (ar + off_i) % batch], [int(x) for x in (ar + off_t) % batch]
