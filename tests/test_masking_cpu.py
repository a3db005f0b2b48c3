"""MLM text masking: the oracle's restatement (oracle/masking_oracle.py) against the golden vectors produced by the
reference's own TextMaskingGenerator / preprocess (tests/golden/make_golden_masking.py), bit for bit; the CPython-set restatement against the
interpreter's set.  CPU only."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import masking_oracle as mo  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "masking.npz"))
GROUPS = [str(g) for g in G["groups"]]


def group(name):
    i = GROUPS.index(name)
    V, L, MM, ss, ww = [int(v) for v in G["params"][i]]
    p, sp = [float(v) for v in G["probs"][i]]
    d = {k: G["%s/%s" % (name, k)] for k in ("text_ids", "text_atts", "text_ids_masked", "masked_pos", "masked_ids", "words", "words_used", "is_subword")}
    kw = dict(mask_prob=p, max_masks=MM, skipgram_prb=sp, skipgram_size=ss, mask_whole_word=bool(ww), cls_id=1, mask_id=3, vocab_size=V)
    return d, kw


@pytest.mark.parametrize("name", GROUPS)
def test_oracle_matches_reference_goldens(name):
    d, kw = group(name)
    idm, mp, mi = mo.mask_tokens(d["text_ids"], d["text_atts"], d["is_subword"], d["words"], **kw)
    assert np.array_equal(idm, d["text_ids_masked"])
    assert np.array_equal(mp, d["masked_pos"])
    assert np.array_equal(mi, d["masked_ids"])
    # the oracle consumes exactly the words the reference consumed
    for b in range(d["text_ids"].shape[0]):
        n = int(d["text_atts"][b].sum())
        _, _, k = mo.mask_caption(d["text_ids"][b], n, d["is_subword"], d["words"][b], mask_prob=kw["mask_prob"], mask_max=kw["max_masks"],
                                  skipgram_thr=mo.prob_threshold(kw["skipgram_prb"]) if kw["skipgram_prb"] > 0 else 0, skipgram_size=kw["skipgram_size"],
                                  mask_whole_word=kw["mask_whole_word"], cls_id=1, mask_id=3, vocab_size=kw["vocab_size"])
        assert k == int(d["words_used"][b])


def test_goldens_cover_the_branches():
    d, _ = group("base")
    corrupted = d["text_ids_masked"] != d["text_ids"]
    masked = np.zeros_like(corrupted)
    for b in range(corrupted.shape[0]):
        k = int((d["masked_ids"][b] != -100).sum())
        masked[b, d["masked_pos"][b, :k]] = True
    assert not (corrupted & ~masked).any()                       # only selected positions change
    assert (masked & ~corrupted).any()                           # the 10 % "keep" branch occurs
    assert ((d["text_ids_masked"] != 3) & corrupted).any()       # the 10 % "random word" branch occurs
    assert ((d["text_ids_masked"] == 3) & corrupted).any()
    d2, _ = group("many_masks")
    assert ((d2["masked_ids"] != -100).sum(1) > 19).any()        # a set that grew past 32 slots
    d3, _ = group("short")
    assert (d3["text_atts"].sum(1) == 2).any()                   # [CLS] [SEP] only: the [SEP] is masked


def test_small_int_set_is_cpythons_set():
    r = random.Random(7)
    for trial in range(300):
        s, t = mo.SmallIntSet(), set()
        for _ in range(r.randint(1, 90)):
            k = r.randint(1, r.choice([7, 31, 39, 63, 127, 511]))
            assert (k in s) == (k in t)
            s.add(k); t.add(k)
        assert list(s) == list(t) and len(s) == len(t)


def test_thresholds_and_rounding():
    assert mo.prob_threshold(0.8) == 3435973837 and mo.prob_threshold(0.5) == 2147483648 and mo.prob_threshold(0.2) == 858993460
    for u in (858993459, 858993460):
        assert (u / 4294967296.0 < 0.2) == (u < mo.prob_threshold(0.2))
    for n in range(0, 70):
        assert mo.python_round(n * 0.5) == int(round(n * 0.5)) and mo.python_round(n * 0.15) == int(round(n * 0.15))
