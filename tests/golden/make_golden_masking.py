#!/usr/bin/env python3
"""Golden vectors for the MLM text masking (data half): the REAL reference code -
dataset/pretrain_dataset.py: TextMaskingGenerator.__call__ (:59-130) and ImageTextJsonDataset.preprocess (:242-275), imported from
/root/reference - run on synthetic captions with the three random functions it uses replaced by word-stream versions (the rule is stated in
oracle/masking_oracle.py: shuffle / rand / randint consume one 32-bit word each, in the reference's own draw order).  Build container only.

The module's third-party imports (torchvision, the reference's `dataset` package __init__ with its cv2 / pycocotools / hdfs dependencies) are
replaced by empty stand-ins: none of them is on the masking path.  The tokenizer is a stand-in too (a synthetic vocabulary in which every fourth
word piece starts with '##'; tokenize = split on blanks): the reference only asks it for get_vocab / tokenize / convert_tokens_to_ids and the
special tokens.

writes tests/golden/masking.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("X2VLM_REFERENCE", "/root/reference")
W = 256                      # random words per caption (more than any caption consumes: asserted below)


def install():
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("torchvision"); mod("torchvision.transforms", InterpolationMode=object)
    mod("torchvision.transforms.functional", hflip=None, resize=None)
    pkg = mod("dataset", build_tokenizer=None)
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "dataset")]          # submodules load from the reference, the package __init__ does not run
    mod("dataset.utils", pre_caption=lambda text, max_words: text, sample_frame_ids=None, sample_clip_ids=None)
    mod("dataset.dist_dataset", DistLineReadingDataset=object)
    import importlib
    return importlib.import_module("dataset.pretrain_dataset")


class Vocab:
    """Synthetic WordPiece-like vocabulary: ids 0..V-1, token text 'w<i>' or '##w<i>' (every fourth id from 8 on), specials at fixed ids."""

    def __init__(self, V):
        self.V = V
        self.tok = {}
        for i in range(V):
            self.tok[i] = ("##w%d" % i) if (i >= 8 and i % 4 == 0) else ("w%d" % i)
        self.tok[0], self.tok[1], self.tok[2], self.tok[3] = "[PAD]", "[CLS]", "[SEP]", "[MASK]"
        self.inv = {t: i for i, t in self.tok.items()}
        self.cls_token, self.sep_token, self.mask_token, self.pad_token_id = "[CLS]", "[SEP]", "[MASK]", 0

    def get_vocab(self):
        return dict(self.inv)

    def tokenize(self, text):
        return text.split()

    def convert_tokens_to_ids(self, tokens):
        return [self.inv[t] for t in tokens]

    def is_subword(self):
        return np.array([self.tok[i].startswith("##") for i in range(self.V)], dtype=np.uint8)


class Stream:
    def __init__(self, words):
        self.w, self.k = [int(x) for x in words], 0

    def next(self):
        v = self.w[self.k]
        self.k += 1
        return v


def patch(mod, stream):
    """random.shuffle / random.random / random.randint as imported by name into the reference module (pretrain_dataset.py:16-17)."""
    def shuffle(x):
        for i in reversed(range(1, len(x))):
            j = (stream.next() * (i + 1)) >> 32
            x[i], x[j] = x[j], x[i]
    mod.shuffle = shuffle
    mod.rand = lambda: stream.next() / 4294967296.0
    mod.randint = lambda a, b: a + ((stream.next() * (b - a + 1)) >> 32)


# (name, vocab, max_tokens, max_masks, mask_prob, skipgram_prb, skipgram_size, mask_whole_word, captions, subword density of the captions)
GROUPS = [
    ("base", 2000, 40, 12, 0.5, 0.2, 3, True, 48, 0.25),          # configs/pretrain/x2vlm_base_4m.yaml:51-57
    ("bench30", 2000, 30, 12, 0.5, 0.2, 3, True, 32, 0.25),       # BASELINE's 30-token captions
    ("dense_subwords", 2000, 40, 12, 0.5, 0.2, 3, True, 24, 0.7), # long '##' runs: large whole-word expansions, more than n_pred positions
    ("no_whole_word", 2000, 40, 12, 0.5, 0.2, 3, False, 16, 0.25),
    ("no_skipgram", 2000, 40, 12, 0.5, 0.0, 3, True, 16, 0.25),
    ("low_prob", 2000, 40, 12, 0.15, 0.2, 3, True, 16, 0.25),     # BERT's 15 %
    ("many_masks", 2000, 64, 24, 0.5, 0.3, 4, True, 16, 0.4),     # set grows past 32 slots, skip-grams of 2..4
    ("short", 2000, 8, 12, 0.5, 0.2, 3, True, 16, 0.3),           # 2..8 tokens: n_pred = 1.., set of <= 4 entries stays at 8 slots
]


def main():
    ref = install()
    rng = np.random.default_rng(20261001)
    out = {}
    meta = []
    for name, V, L, MM, p, sp, ss, ww, n, dens in GROUPS:
        tk = Vocab(V)
        sub = tk.is_subword()
        words_ok = np.nonzero(sub == 0)[0]; words_ok = words_ok[words_ok >= 4]
        pieces = np.nonzero(sub == 1)[0]
        gen = ref.TextMaskingGenerator(tk, p, MM, sp, ss, ww)
        ds = types.SimpleNamespace(tokenizer=tk, cls_token=tk.cls_token, eos_token=tk.sep_token, pad_token_id=0, PAD_mask=-100, max_tokens=L,
                                   max_words=L, max_masks=MM, add_eos=True, mask_generator=gen)
        ids = np.zeros((n, L), dtype=np.int64); atts = np.zeros((n, L), dtype=np.int64)
        idm = np.zeros((n, L), dtype=np.int64); mp = np.zeros((n, MM), dtype=np.int64); mi = np.zeros((n, MM), dtype=np.int64)
        words = rng.integers(0, 1 << 32, size=(n, W), dtype=np.uint64).astype(np.uint32)
        used = np.zeros(n, dtype=np.int64)
        for c in range(n):
            nw = int(rng.integers(0 if name == "short" else 2, L + 3))            # body tokens before truncation to max_tokens - 2 (+ [CLS], [SEP])
            body = [int(rng.choice(words_ok))]
            while len(body) < max(nw, 1):
                body.append(int(rng.choice(pieces)) if rng.random() < dens else int(rng.choice(words_ok)))
            if name == "short" and nw == 0:
                body = []
            text = " ".join(tk.tok[i] for i in body)
            st = Stream(words[c]); patch(ref, st)
            t_ids, t_atts, t_idm, t_mp, t_mi = ref.ImageTextJsonDataset.preprocess(ds, text)
            ids[c], atts[c], idm[c], mp[c], mi[c] = t_ids, t_atts, t_idm, t_mp, t_mi
            used[c] = st.k
        assert used.max() < W
        for k, v in (("text_ids", ids), ("text_atts", atts), ("text_ids_masked", idm), ("masked_pos", mp), ("masked_ids", mi), ("words", words),
                     ("words_used", used), ("is_subword", sub)):
            out["%s/%s" % (name, k)] = v
        meta.append((name, V, L, MM, p, sp, ss, int(ww)))
        print("%-16s %3d captions, lengths %d..%d, masked %d..%d, words used <= %d" % (name, n, atts.sum(1).min(), atts.sum(1).max(),
              (mi != -100).sum(1).min(), (mi != -100).sum(1).max(), used.max()))
    out["groups"] = np.array([m[0] for m in meta])
    out["params"] = np.array([[m[1], m[2], m[3], m[6], m[7]] for m in meta], dtype=np.int64)        # vocab, max_tokens, max_masks, skipgram_size, whole_word
    out["probs"] = np.array([[m[4], m[5]] for m in meta], dtype=np.float64)                          # mask_prob, skipgram_prb
    np.savez_compressed(os.path.join(HERE, "masking.npz"), **out)
    print("wrote", os.path.join(HERE, "masking.npz"))


if __name__ == "__main__":
    main()
