"""CPU restatement of the reference's MLM text masking and padding.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; the product path (x2-vlm_amd/, the HIP kernel x2_mask_tokens) never imports it.

Follows /root/reference/dataset/pretrain_dataset.py:
  :36-58    TextMaskingGenerator.__init__  (mask_prob, mask_max, skipgram_prb, skipgram_size, mask_whole_word)
  :59-130   TextMaskingGenerator.__call__  (n_pred rule, shuffled candidates, whole-word expansion over '##' pieces, skip-grams of 2..size,
                                            truncation to n_pred, 80 / 10 / 10 corruption)
  :242-275  ImageTextJsonDataset.preprocess (masked_ids, padding of ids / attention mask / masked ids / positions)

Pinned: tests/golden/masking.npz holds the outputs of the reference's OWN class and preprocess() (imported from /root/reference by
tests/golden/make_golden_masking.py) with the three random functions it uses (random.shuffle / random.random / random.randint) replaced by
the word-stream versions below; tests/test_masking_cpu.py checks this file against every case bit for bit.

Random numbers.  The reference draws from Python's Mersenne Twister; a device kernel cannot replay that stream, so both sides consume a stream
of 32-bit words u[0], u[1], ... per caption, in the reference's draw order:
    rand()        -> u / 2**32                       (compared as  u < ceil(p * 2**32))
    randint(a, b) -> a + ((u * (b - a + 1)) >> 32)
    shuffle(x)    -> for i = len(x) - 1 .. 1: j = (u * (i + 1)) >> 32; swap x[i], x[j]      (random.shuffle's loop, CPython 3.10 Lib/random.py)

Set order.  The reference collects positions in a Python `set` and turns it into a list (`list(masked_pos)`, :117): the order in which positions
are corrupted (one to three draws each) and reported is CPython's set iteration order.  For the small non-negative ints involved that order is a pure
function of the insertion sequence (hash(i) == i; open addressing with 9 linear probes, then i = 5 i + 1 + perturb; growth to the first power of two
above 4 x used once fill * 5 >= mask * 3: Objects/setobject.c, unchanged 3.7 - 3.12); `SmallIntSet` restates it so the result does not depend on the
interpreter that runs the oracle (the test also checks it against the interpreter's own set).
"""
import numpy as np

LINEAR_PROBES = 9
PERTURB_SHIFT = 5
MINSIZE = 8


class SmallIntSet:
    """CPython's set for non-negative ints without deletions: add / contains / iteration order (Objects/setobject.c)."""

    def __init__(self):
        self.mask = MINSIZE - 1
        self.table = [-1] * MINSIZE
        self.used = 0

    @staticmethod
    def _insert_clean(table, mask, key):
        perturb = key
        i = key & mask
        while True:
            probes = LINEAR_PROBES if i + LINEAR_PROBES <= mask else 0
            for e in range(i, i + probes + 1):
                if table[e] < 0:
                    table[e] = key
                    return
            perturb >>= PERTURB_SHIFT
            i = (i * 5 + 1 + perturb) & mask

    def __contains__(self, key):
        mask, table = self.mask, self.table
        perturb = key
        i = key & mask
        while True:
            probes = LINEAR_PROBES if i + LINEAR_PROBES <= mask else 0
            for e in range(i, i + probes + 1):
                if table[e] < 0:
                    return False
                if table[e] == key:
                    return True
            perturb >>= PERTURB_SHIFT
            i = (i * 5 + 1 + perturb) & mask

    def add(self, key):
        if key in self:
            return
        self._insert_clean(self.table, self.mask, key)
        self.used += 1
        if self.used * 5 >= self.mask * 3:                      # fill == used: nothing is ever deleted
            newsize = MINSIZE
            while newsize <= self.used * 4:
                newsize <<= 1
            table = [-1] * newsize
            for k in self.table:                                 # old slot order
                if k >= 0:
                    self._insert_clean(table, newsize - 1, k)
            self.table, self.mask = table, newsize - 1

    def __len__(self):
        return self.used

    def __iter__(self):
        return (k for k in self.table if k >= 0)


class WordStream:
    """The three random functions of the reference on a stream of 32-bit words."""

    def __init__(self, words):
        self.w = [int(x) for x in words]
        self.k = 0

    def next(self):
        v = self.w[self.k]
        self.k += 1
        return v

    def rand_below_thr(self, thr):           # rand() < p  with thr = ceil(p * 2**32)
        return self.next() < thr

    def randint(self, a, b):
        return a + ((self.next() * (b - a + 1)) >> 32)

    def shuffle(self, x):
        for i in reversed(range(1, len(x))):
            j = (self.next() * (i + 1)) >> 32
            x[i], x[j] = x[j], x[i]


def prob_threshold(p):
    """ceil(p * 2**32): u / 2**32 < p  <=>  u < prob_threshold(p) for integer u (p * 2**32 is exact in a double for the p used here)."""
    import math
    return int(math.ceil(float(p) * 4294967296.0))


def python_round(x):
    """Python 3 round(x) for a non-negative double: to nearest, ties to even (pretrain_dataset.py:60-62)."""
    return int(np.rint(np.float64(x)))


def mask_caption(ids, n_tokens, is_subword, words, *, mask_prob, mask_max, skipgram_thr, skipgram_size, mask_whole_word,
                 cls_id, mask_id, vocab_size):
    """One caption (pretrain_dataset.py:59-130).  ids: token ids (the first n_tokens are the caption, [CLS] first); is_subword[v] = token v
    starts with '##'; words: the caption's 32-bit word stream.  Returns (ids_masked list, masked_pos list) and the number of words consumed."""
    rs = WordStream(words)
    tok = [int(v) for v in ids[:n_tokens]]
    n_pred = min(mask_max, max(1, python_round(n_tokens * mask_prob)))
    if tok[0] == cls_id:
        special = 1                                          # positions below `special` are never masked
    else:
        special = 0
    cand = list(range(special, n_tokens))
    rs.shuffle(cand)
    masked = SmallIntSet()
    max_cand = max(cand)

    def sub(i):
        return bool(is_subword[tok[i]])

    def expand(st, end):
        while st >= 0 and sub(st):
            st -= 1
        while end < n_tokens and sub(end):
            end += 1
        return st, end

    for pos in cand:
        if len(masked) >= n_pred:
            break
        if pos in masked:
            continue
        if skipgram_thr > 0 and skipgram_size >= 2 and rs.rand_below_thr(skipgram_thr):
            size = rs.randint(2, skipgram_size)
            st, end = expand(pos, pos + size) if mask_whole_word else (pos, pos + size)
        else:
            st, end = expand(pos, pos + 1) if mask_whole_word else (pos, pos + 1)
        for mp in range(st, end):
            if 0 < mp <= max_cand and mp >= special:
                masked.add(mp)
            else:
                break
    mpos = list(masked)
    if len(mpos) > n_pred:
        rs.shuffle(mpos)
        mpos = mpos[:n_pred]
    out = list(tok)
    thr80, thr50 = prob_threshold(0.8), prob_threshold(0.5)
    for pos in mpos:
        if rs.rand_below_thr(thr80):
            out[pos] = mask_id
        elif rs.rand_below_thr(thr50):
            out[pos] = rs.randint(0, vocab_size - 1)
    return out, mpos, rs.k


def mask_tokens(text_ids, text_atts, is_subword, words, *, mask_prob=0.5, max_masks=12, skipgram_prb=0.2, skipgram_size=3,
                mask_whole_word=True, cls_id=101, mask_id=103, vocab_size=30522, pad_id=0, pad_mask=-100):
    """A padded batch (pretrain_dataset.py:242-275): text_ids / text_atts [B, L] int64 (captions left-aligned, atts = 1 on the caption),
    words [B, W] uint32.  Returns text_ids_masked [B, L], masked_pos [B, max_masks] (pad 0), masked_ids [B, max_masks] (pad -100), int64."""
    text_ids = np.asarray(text_ids, dtype=np.int64)
    text_atts = np.asarray(text_atts, dtype=np.int64)
    B, L = text_ids.shape
    ids_masked = np.full((B, L), pad_id, dtype=np.int64)
    masked_pos = np.zeros((B, max_masks), dtype=np.int64)
    masked_ids = np.full((B, max_masks), pad_mask, dtype=np.int64)
    thr = prob_threshold(skipgram_prb) if skipgram_prb > 0 else 0
    for b in range(B):
        n = int(text_atts[b].sum())
        out, mpos, _ = mask_caption(text_ids[b], n, is_subword, words[b], mask_prob=mask_prob, mask_max=max_masks, skipgram_thr=thr,
                                    skipgram_size=skipgram_size, mask_whole_word=mask_whole_word, cls_id=cls_id, mask_id=mask_id,
                                    vocab_size=vocab_size)
        ids_masked[b, :n] = out
        masked_pos[b, :len(mpos)] = mpos
        masked_ids[b, :len(mpos)] = text_ids[b, mpos]
    return ids_masked, masked_pos, masked_ids
