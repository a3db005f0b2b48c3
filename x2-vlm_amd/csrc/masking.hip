// MLM text masking on the device (gfx950): text_ids / text_atts -> text_ids_masked, masked_pos, masked_ids.
//
// Replaces, per caption, dataset/pretrain_dataset.py:59-130 (TextMaskingGenerator.__call__: n_pred = min(max_masks, max(1, round(n * mask_prob))),
// shuffled candidate positions, whole-word expansion over '##' pieces, skip-grams of 2..size with probability skipgram_prb, truncation to n_pred,
// 80 % [MASK] / 10 % random word / 10 % unchanged) and :242-275 (ImageTextJsonDataset.preprocess: masked_ids, padding with 0 / -100) - work the
// reference does on the host inside its data-loader workers.  On the device it becomes part of the replayed step: the step graph takes the raw
// (text_ids, text_atts) of a batch and draws a fresh mask on every replay (the epoch word that also drives the dropout sites).
//
// Bit-exact restatement, not a look-alike: the reference's draws are taken from a stream of 32-bit words per caption in the reference's own order
//     rand() < p     ->  word < ceil(p * 2^32)         randint(a, b) -> a + ((word * (b - a + 1)) >> 32)
//     shuffle(x)     ->  for i = len - 1 .. 1: j = (word * (i + 1)) >> 32, swap x[i], x[j]
// and the order in which the chosen positions are corrupted and reported is the iteration order of the CPython `set` the reference collects them in
// (`list(masked_pos)`, :117) - restated here as the open-addressing table of Objects/setobject.c for small non-negative ints (hash(i) = i, 9 linear
// probes, then i = 5 i + 1 + perturb, growth to the first power of two above 4 x used at fill * 5 >= mask * 3).  oracle/masking_oracle.py is the same
// restatement on the CPU, pinned to vectors produced by the reference's own class (tests/golden/make_golden_masking.py); tests/test_kernels_gpu.py
// compares this kernel with both, bit for bit, on injected words.
//
// One wave per caption: the algorithm is a few hundred dependent steps on <= 512 positions - lane 0 walks it on LDS arrays, all lanes load the
// caption and store the three outputs.  (B = 64 .. 4096 captions per step: 16 .. 1024 workgroups of 4 waves; ~10 us, off the critical path of a
// step whose text tower starts with it.)
#include "x2_common.h"

#define MK_MAXL 512                 // tokens per caption
#define MK_TABLE 4096               // slots of the largest set table (first power of two above 4 x 512 is 4096: used * 4 < 4096 always holds for used < 1024)

struct MaskArgs {
  const long* ids; const long* atts; long* ids_masked; long* masked_pos; long* masked_ids;
  const unsigned char* is_subword; const uint32_t* words; const uint32_t* epoch;
  int B, L, vocab, words_ld, max_masks, skipgram_size, whole_word;
  uint32_t seed;
  double mask_prob;
  unsigned long long skipgram_thr;   // ceil(skipgram_prb * 2^32), 0 = no skip-grams
  long cls_id, mask_id, pad_id, pad_mask;
};

// word k of caption b when no words are injected: two rounds of the dropout sites' hash over (seed mixed with the step counter, b, k);
// mirrored by kernels.mask_words() on the host
__device__ __forceinline__ uint32_t mk_word(uint32_t seed, uint32_t b, uint32_t k) {
  return x2_hash(x2_hash(seed + 0x9E3779B1u * (b + 1u)) ^ (0x85EBCA6Bu * (k + 1u)));
}

struct MaskState {
  int tok[MK_MAXL];                 // caption token ids (32 bit: vocabularies are < 2^31)
  short cand[MK_MAXL];              // candidate positions, shuffled
  short table[MK_TABLE];            // the set: -1 = empty slot
  short list[MK_MAXL];              // positions in set order / scratch of a table growth
  int n_list;
};

__device__ __forceinline__ void set_insert_clean(short* table, int mask, int key) {
  unsigned perturb = (unsigned)key;
  int i = key & mask;
  while (true) {
    const int probes = (i + 9 <= mask) ? 9 : 0;
    for (int e = i; e <= i + probes; ++e)
      if (table[e] < 0) { table[e] = (short)key; return; }
    perturb >>= 5;
    i = (int)((i * 5u + 1u + perturb) & (unsigned)mask);
  }
}
__device__ __forceinline__ bool set_contains(const short* table, int mask, int key) {
  unsigned perturb = (unsigned)key;
  int i = key & mask;
  while (true) {
    const int probes = (i + 9 <= mask) ? 9 : 0;
    for (int e = i; e <= i + probes; ++e) {
      if (table[e] < 0) return false;
      if (table[e] == key) return true;
    }
    perturb >>= 5;
    i = (int)((i * 5u + 1u + perturb) & (unsigned)mask);
  }
}

__global__ __launch_bounds__(256) void mask_tokens_kernel(MaskArgs a) {
  __shared__ MaskState st_all[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  MaskState& s = st_all[wave];
  const long* ids = a.ids + (long)b * a.L;
  // caption length = number of attended tokens (captions are left-aligned: preprocess() pads on the right)
  int n = 0;
  for (int l = lane; l < a.L; l += 64) { s.tok[l] = (int)ids[l]; n += a.atts[(long)b * a.L + l] != 0 ? 1 : 0; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  for (int l = lane; l < 8; l += 64) s.table[l] = -1;
  __builtin_amdgcn_s_waitcnt(0);                               // one wave: LDS writes above are visible to lane 0 below after the wait
  __builtin_amdgcn_wave_barrier();

  if (lane == 0) {
    uint32_t seed = a.seed;
    if (a.epoch) seed = x2_hash(seed + 0x9E3779B1u * a.epoch[0]);
    uint32_t k = 0;
    const uint32_t* inj = a.words ? a.words + (long)b * a.words_ld : nullptr;
    auto next = [&]() -> unsigned long long { const uint32_t w = inj ? inj[k] : mk_word(seed, (uint32_t)b, k); ++k; return w; };
    const int n_pred = min(a.max_masks, max(1, (int)rint((double)n * a.mask_prob)));
    const int special = s.tok[0] == (int)a.cls_id ? 1 : 0;
    const int ncand = n - special;
    for (int i = 0; i < ncand; ++i) s.cand[i] = (short)(special + i);
    for (int i = ncand - 1; i >= 1; --i) {
      const int j = (int)((next() * (unsigned long long)(i + 1)) >> 32);
      const short t = s.cand[i]; s.cand[i] = s.cand[j]; s.cand[j] = t;
    }
    const int max_cand = n - 1;
    int mask = 7, used = 0;
    auto sub = [&](int i) { return a.is_subword[s.tok[i]] != 0; };
    for (int c = 0; c < ncand; ++c) {
      if (used >= n_pred) break;
      const int pos = s.cand[c];
      if (set_contains(s.table, mask, pos)) continue;
      int st_ = pos, end = pos + 1;
      if (a.skipgram_thr > 0 && a.skipgram_size >= 2 && next() < a.skipgram_thr)
        end = pos + 2 + (int)((next() * (unsigned long long)(a.skipgram_size - 1)) >> 32);
      if (a.whole_word) {
        while (st_ >= 0 && sub(st_)) --st_;
        while (end < n && sub(end)) ++end;
      }
      for (int mp = st_; mp < end; ++mp) {
        if (!(0 < mp && mp <= max_cand && mp >= special)) break;
        if (set_contains(s.table, mask, mp)) continue;
        set_insert_clean(s.table, mask, mp);
        ++used;
        if (used * 5 >= mask * 3) {                             // grow: re-insert in old slot order
          int newsize = 8;
          while (newsize <= used * 4) newsize <<= 1;
          int m = 0;
          for (int e = 0; e <= mask; ++e) if (s.table[e] >= 0) s.list[m++] = s.table[e];
          for (int e = 0; e < newsize; ++e) s.table[e] = -1;
          mask = newsize - 1;
          for (int e = 0; e < m; ++e) set_insert_clean(s.table, mask, s.list[e]);
        }
      }
    }
    int m = 0;
    for (int e = 0; e <= mask; ++e) if (s.table[e] >= 0) s.list[m++] = s.table[e];
    if (m > n_pred) {
      for (int i = m - 1; i >= 1; --i) {
        const int j = (int)((next() * (unsigned long long)(i + 1)) >> 32);
        const short t = s.list[i]; s.list[i] = s.list[j]; s.list[j] = t;
      }
      m = n_pred;
    }
    s.n_list = m;
    // (the reported label is the ORIGINAL id: read from the input row below, tok[] is overwritten here)
    for (int e = 0; e < m; ++e) {
      const int pos = s.list[e];
      if (next() < 3435973837ull) s.tok[pos] = (int)a.mask_id;                                   // rand() < 0.8
      else if (next() < 2147483648ull) s.tok[pos] = (int)((next() * (unsigned long long)a.vocab) >> 32);   // rand() < 0.5: randint(0, vocab - 1)
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int m = s.n_list;
  for (int l = lane; l < a.L; l += 64) a.ids_masked[(long)b * a.L + l] = l < n ? (long)s.tok[l] : a.pad_id;
  for (int e = lane; e < a.max_masks; e += 64) {
    a.masked_pos[(long)b * a.max_masks + e] = e < m ? (long)s.list[e] : 0;
    a.masked_ids[(long)b * a.max_masks + e] = e < m ? ids[s.list[e]] : a.pad_mask;
  }
}

extern "C" int x2_mask_tokens(const long* text_ids, const long* text_atts, int B, int L, const unsigned char* is_subword, int vocab,
                              const unsigned* words, int words_ld, unsigned seed, const unsigned* epoch, double mask_prob, int max_masks,
                              double skipgram_prb, int skipgram_size, int mask_whole_word, long cls_id, long mask_id, long pad_id, long pad_mask,
                              long* text_ids_masked, long* masked_pos, long* masked_ids, void* stream) {
  X2_REQUIRE(text_ids && text_atts && is_subword && text_ids_masked && masked_pos && masked_ids, "x2_mask_tokens: null argument");
  X2_REQUIRE(B > 0 && L >= 2 && L <= MK_MAXL, "x2_mask_tokens: B=%d L=%d (2 <= L <= %d)", B, L, MK_MAXL);
  X2_REQUIRE(vocab > 0 && max_masks > 0 && max_masks <= MK_MAXL, "x2_mask_tokens: vocab=%d max_masks=%d", vocab, max_masks);
  X2_REQUIRE(mask_prob >= 0.0 && mask_prob <= 1.0 && skipgram_prb >= 0.0 && skipgram_prb <= 1.0, "x2_mask_tokens: mask_prob=%g skipgram_prb=%g", mask_prob,
             skipgram_prb);
  X2_REQUIRE(!words || words_ld > 0, "x2_mask_tokens: words_ld=%d", words_ld);
  MaskArgs a;
  a.ids = text_ids; a.atts = text_atts; a.ids_masked = text_ids_masked; a.masked_pos = masked_pos; a.masked_ids = masked_ids;
  a.is_subword = is_subword; a.words = words; a.epoch = epoch;
  a.B = B; a.L = L; a.vocab = vocab; a.words_ld = words_ld; a.max_masks = max_masks; a.skipgram_size = skipgram_size; a.whole_word = mask_whole_word;
  a.seed = seed; a.mask_prob = mask_prob;
  const double t = skipgram_prb * 4294967296.0;
  a.skipgram_thr = skipgram_prb > 0.0 ? (unsigned long long)t + ((double)(unsigned long long)t < t ? 1ull : 0ull) : 0ull;   // ceil
  a.cls_id = cls_id; a.mask_id = mask_id; a.pad_id = pad_id; a.pad_mask = pad_mask;
  hipLaunchKernelGGL(mask_tokens_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return x2_check_launch("x2_mask_tokens");
}
