"""x2_mask_tokens (csrc/masking.hip) through the C ABI: bit-exact against the reference-generated golden vectors (tests/golden/masking.npz) and
against the oracle (oracle/masking_oracle.py) on fresh captions, on injected random words; hashed words checked through the host mirror."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import masking_oracle as mo

pytestmark = pytest.mark.gpu
dev = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "masking.npz"))
GROUPS = [str(g) for g in G["groups"]]


@pytest.fixture(scope="module")
def K():
    return importlib.import_module("x2-vlm_amd.kernels")


def words_tensor(w):          # uint32 numpy -> int32 bit pattern on the device
    return torch.from_numpy(np.ascontiguousarray(w).view(np.int32)).to(dev)


@pytest.mark.parametrize("name", GROUPS)
def test_mask_tokens_matches_reference_goldens(K, name):
    i = GROUPS.index(name)
    V, L, MM, ss, ww = [int(v) for v in G["params"][i]]
    p, sp = [float(v) for v in G["probs"][i]]
    g = lambda k: G["%s/%s" % (name, k)]
    idm, mp, mi = K.mask_tokens(torch.from_numpy(g("text_ids")).to(dev), torch.from_numpy(g("text_atts")).to(dev), torch.from_numpy(g("is_subword")).to(dev),
                                words=words_tensor(g("words")), mask_prob=p, max_masks=MM, skipgram_prb=sp, skipgram_size=ss, mask_whole_word=bool(ww),
                                cls_id=1, mask_id=3)
    assert np.array_equal(idm.cpu().numpy(), g("text_ids_masked"))
    assert np.array_equal(mp.cpu().numpy(), g("masked_pos"))
    assert np.array_equal(mi.cpu().numpy(), g("masked_ids"))


def synth_captions(rng, B, L, V, dens):
    sub = (np.arange(V) % 3 == 0).astype(np.uint8); sub[:8] = 0
    ids = np.zeros((B, L), dtype=np.int64); atts = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(2, L + 1))
        body = rng.integers(8, V, size=n)
        pieces = rng.random(n) < dens
        body = np.where(pieces, body - body % 3, body - body % 3 + 1)          # multiples of 3 are '##' pieces
        body = np.clip(body, 8, V - 1)
        ids[b, :n] = body; ids[b, 0] = 1; ids[b, n - 1] = 2; atts[b, :n] = 1
    return ids, atts, sub


@pytest.mark.parametrize("B,L,MM,p,sp,ss,ww,dens", [(64, 30, 12, 0.5, 0.2, 3, True, 0.3), (257, 40, 12, 0.5, 0.2, 3, True, 0.6), (33, 128, 40, 0.4, 0.5, 5, True, 0.8),
                                                    (16, 512, 64, 0.15, 0.2, 3, True, 0.5), (40, 40, 12, 0.5, 0.3, 3, False, 0.3), (5, 2, 12, 0.5, 0.2, 3, True, 0.0)])
def test_mask_tokens_matches_oracle(K, B, L, MM, p, sp, ss, ww, dens):
    rng = np.random.default_rng(B * 1000 + L)
    V = 30522
    ids, atts, sub = synth_captions(rng, B, L, V, dens)
    words = rng.integers(0, 1 << 32, size=(B, 4 * L + 64), dtype=np.uint64).astype(np.uint32)
    kw = dict(mask_prob=p, max_masks=MM, skipgram_prb=sp, skipgram_size=ss, mask_whole_word=ww, cls_id=1, mask_id=3)
    want = mo.mask_tokens(ids, atts, sub, words, vocab_size=V, **kw)
    got = K.mask_tokens(torch.from_numpy(ids).to(dev), torch.from_numpy(atts).to(dev), torch.from_numpy(sub).to(dev), words=words_tensor(words), **kw)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_.cpu().numpy(), w_)


def test_mask_tokens_hashed_words_follow_the_epoch(K):
    rng = np.random.default_rng(5)
    B, L, V = 48, 30, 30522
    ids, atts, sub = synth_captions(rng, B, L, V, 0.3)
    t = lambda a: torch.from_numpy(a).to(dev)
    epoch = torch.tensor([7], device=dev, dtype=torch.int32)
    got = K.mask_tokens(t(ids), t(atts), t(sub), seed=1234, epoch=epoch, cls_id=1, mask_id=3)
    words = K.mask_words(1234, 7, B, 4 * L + 64).numpy().astype(np.uint32)
    want = mo.mask_tokens(ids, atts, sub, words, vocab_size=V, cls_id=1, mask_id=3)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_.cpu().numpy(), w_)
    epoch += 1
    again = K.mask_tokens(t(ids), t(atts), t(sub), seed=1234, epoch=epoch, cls_id=1, mask_id=3)
    assert not torch.equal(again[1], got[1])                                       # a new step draws a new mask
    # statistics of the reference's rule at mask_prob 0.5, max_masks 12: every caption gets min(12, max(1, round(n / 2))) positions, ~80 % of them [MASK]
    n = atts.sum(1)
    k = (got[2].cpu().numpy() != -100).sum(1)
    npred = np.minimum(12, np.maximum(1, np.rint(n * 0.5).astype(np.int64)))
    assert (k >= 1).all() and (k <= npred).all() and k.sum() >= 0.9 * npred.sum()      # fewer only where a skip-gram / whole word ran into the caption's end
    idm = got[0].cpu().numpy()
    frac = (idm == 3).sum() / k.sum()
    assert 0.7 < frac < 0.9


def test_mask_tokens_argument_checks(K):
    ids = torch.zeros(2, 600, dtype=torch.int64, device=dev)
    with pytest.raises(Exception):
        K.mask_tokens(ids, ids.clone(), torch.zeros(10, dtype=torch.uint8, device=dev))       # L > 512
