// read2sdbg stage 1, front: replaces Read2SdbgS1's Lv0CalcBucketSize / Lv1FillOffsets / Lv2ExtractSubString (reference
// src/sorting/read_to_sdbg_s1.cpp:145-366) — the extraction kernels, the digit- and lv1-bucket-histogram passes over the packed
// reads, and the generators that let the first sort pass make its records (s1_gen.h).
#include "s1_gen.h"

namespace mhx {

__global__ void k_s1_item_counts(const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t k, uint32_t *__restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seqs) {
    uint64_t L = start[i + 1] - start[i];
    cnt[i] = L >= k + 1 ? (uint32_t)(L - k + 4) : 0u;  // read_to_sdbg_s1.cpp:228-292
  }
}

template <int KW, int S, bool COMPACT>
__device__ __forceinline__ void s1_make_item(const uint32_t *__restrict__ seq, uint64_t st, uint32_t L, int k, uint32_t j, uint64_t pos_base,
                                             uint32_t pos_bits, uint32_t (&out)[S]) {
  // slot -> ((k-1)-mer offset q, forced strand or -1)
  uint32_t q;
  int forced = -1;
  if (j < 2) { q = 0; forced = (int)j; }
  else if (j >= L - k + 2) { q = L - k + 1; forced = (int)(j - (L - k + 2)); }
  else q = j - 1;
  uint32_t f[KW], rc[KW];
  load_chars<KW>(seq, st + q, k - 1, f);
  rc_chars<KW>(f, k - 1, rc);
  const unsigned head = q >= 1 ? base_at(seq, st + q - 1) : kSentinel;
  const unsigned prev = q >= 2 ? base_at(seq, st + q - 2) : kSentinel;
  const unsigned tail = q + k - 1 < L ? base_at(seq, st + q + k - 1) : kSentinel;
  const unsigned next = q + k < L ? base_at(seq, st + q + k) : kSentinel;
  int strand;
  if (forced >= 0) strand = forced;
  else {
    const int c = cmp_words<KW>(f, rc);
    if (c > 0) strand = 1;
    else if (c < 0) strand = 0;
    else strand = head <= 3 - tail ? 0 : 1;  // palindrome rule, :264-279 (head/tail are bases here)
  }
  const uint64_t full = ((pos_base + st + q) << 1) | (uint64_t)strand;  // pos_base: this rank's offset in the global read set
  uint64_t info;
  if (!strand) {
#pragma unroll
    for (int i = 0; i < KW; ++i) out[i] = f[i];
    out[KW - 1] |= (head << 3) | tail;
    info = (full << 6) | (prev << 3) | next;
  } else {
#pragma unroll
    for (int i = 0; i < KW; ++i) out[i] = rc[i];
    out[KW - 1] |= (comp_or_sentinel(tail) << 3) | comp_or_sentinel(head);
    info = (full << 6) | (comp_or_sentinel(next) << 3) | comp_or_sentinel(prev);
  }
  if constexpr (COMPACT) {
    const uint64_t p = pos_base + st + q;
    out[KW - 1] |= s1_pos_tag(p, pos_bits);
    out[KW] = s1_pos_word(p, pos_bits);
    if constexpr (S > KW + 1) out[KW + 1] = 0;
  } else {
    out[KW] = (uint32_t)(info >> 32);
    out[KW + 1] = (uint32_t)info;
    if constexpr (S > KW + 2) out[KW + 2] = 0;
  }
}

template <int KW, int S, bool COMPACT>
__global__ __launch_bounds__(256) void k_s1_extract(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start,
                                                    const uint64_t *__restrict__ item_start, uint64_t n_seqs, int k,
                                                    uint64_t pos_base, uint32_t pos_bits, uint32_t *__restrict__ items) {
  const int lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const uint64_t n_waves = (uint64_t)gridDim.x * blockDim.x / kWave;
  for (uint64_t r = wave; r < n_seqs; r += n_waves) {
    const uint64_t st = start[r];
    const uint32_t L = (uint32_t)(start[r + 1] - st);
    if (L < (uint32_t)k + 1) continue;
    const uint64_t ibase = item_start[r];
    const uint32_t n_slots = L - k + 4;
    for (uint32_t j = lane; j < n_slots; j += kWave) {
      uint32_t out[S];
      s1_make_item<KW, S, COMPACT>(seq, st, L, k, j, pos_base, pos_bits, out);
      uint32_t *dst = items + (ibase + j) * S;
      if constexpr (S % 2 == 1) {
#pragma unroll
        for (int i = 0; i < S; ++i) dst[i] = out[i];
      } else if constexpr (S % 4 == 0) {
#pragma unroll
        for (int i = 0; i < S / 4; ++i)
          reinterpret_cast<uint4 *>(dst)[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
      } else {
#pragma unroll
        for (int i = 0; i < S / 2; ++i) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(out[2 * i], out[2 * i + 1]);
      }
    }
  }
}

// Reads of one length (the usual case): item g belongs to read g / per, slot g % per, so every lane of every wave has
// work (a wave per read leaves the last of ceil(per/64) rounds nearly empty), and odd-stride records are transposed
// through LDS so that each store instruction writes 256 contiguous bytes.
template <int KW, int S, bool COMPACT>
__global__ __launch_bounds__(256) void k_s1_extract_fixed(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                          uint64_t pos_base, uint32_t pos_bits, uint32_t *__restrict__ items, DigitSpecs specs,
                                                          unsigned long long *__restrict__ ghist) {
  __shared__ uint32_t xpose[S % 2 == 1 ? 256 * S : 1];
  __shared__ uint32_t h[kMaxFusedPasses][256];  // digit histograms of the coming sort passes (specs.n == 0: none)
  for (int i = threadIdx.x; i < specs.n * 256; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint64_t n_blocks = (n_items + 255) / 256;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {  // persistent: one histogram flush per workgroup
    const uint64_t g = blk * 256 + threadIdx.x;
    uint32_t out[S];
    if (g < n_items) {
      const uint64_t r = g / per;
      s1_make_item<KW, S, COMPACT>(seq, r * L, L, k, (uint32_t)(g - r * per), pos_base, pos_bits, out);
      for (int p = 0; p < specs.n; ++p) atomicAdd(&h[p][words_digit2<S>(out, specs.d[p])], 1u);
    }
    if constexpr (S % 2 == 1) {
#pragma unroll
      for (int i = 0; i < S; ++i) xpose[threadIdx.x * S + i] = out[i];
      __syncthreads();
      const uint64_t w0 = blk * 256 * S, n_words = n_items * S;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const uint64_t w = w0 + (uint64_t)i * 256 + threadIdx.x;
        if (w < n_words) items[w] = xpose[i * 256 + threadIdx.x];
      }
      __syncthreads();
    } else if (g < n_items) {
      uint32_t *dst = items + g * S;
      if constexpr (S % 4 == 0) {
#pragma unroll
        for (int i = 0; i < S / 4; ++i)
          reinterpret_cast<uint4 *>(dst)[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
      } else {
#pragma unroll
        for (int i = 0; i < S / 2; ++i) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(out[2 * i], out[2 * i + 1]);
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < specs.n; ++p) {
    const uint32_t v = h[p][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}

// The digit histograms of the coming sort passes without making the records (the pre-pass of the generating first pass):
// every thread takes IT CONSECUTIVE items — one division per trip, the three window words are reloaded only when the
// window moves into the next word (every 16 items), and of the key only its first word is formed (the passes of the
// partial-sort plans take their digits from the top 32 key bits: `hi_only`; head / tail never reach them).
template <int IT>
__global__ __launch_bounds__(256) void k_s1_digit_hist(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                       DigitSpecs specs, unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r) {
  constexpr int B = 256 * IT;
  __shared__ uint32_t h[kFastPasses][4][256];
  for (int i = threadIdx.x; i < kFastPasses * 4 * 256; i += 256) (&h[0][0][0])[i] = 0;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const int km1 = k - 1;
  const uint64_t kmask = ~0ull << (64 - 2 * km1);
  const uint64_t n_blocks = (n_items + B - 1) / B;
  uint64_t q0 = ((uint32_t)blockIdx.x * (uint32_t)B) / per;
  uint32_t rem0 = ((uint32_t)blockIdx.x * (uint32_t)B) % per;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint64_t r = q0 + dq;
    uint32_t j = t - dq * per;
    uint64_t wcur = ~0ull;
    uint32_t x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      if (g0 + u < n_items) {
        uint32_t q;
        int forced = -1;
        if (j < 2) { q = 0; forced = (int)j; }
        else if (j >= L - k + 2) { q = L - k + 1; forced = (int)(j - (L - k + 2)); }
        else q = j - 1;
        const uint64_t a = r * L + q;
        uint32_t hi;
        if (a >= 2) {
          const uint64_t b = a - 2, w = b >> 4;
          if (w != wcur) {
            x0 = seq[w];
            x1 = seq[w + 1];
            x2 = seq[w + 2];
            wcur = w;
          }
          const unsigned sh = (unsigned)(b & 15) * 2;
          const uint64_t win = ((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh);
          const uint64_t f = (win << 4) & kmask;
          const uint64_t rc = rc64(f, km1);
          const bool use_rc = forced >= 0 ? forced == 1 : f > rc;  // (f == rc: the same first word either way)
          hi = (uint32_t)((use_rc ? rc : f) >> 32);
        } else {
          uint32_t out[3];
          s1_make_item<2, 3, true>(seq, r * L, L, k, j, 0, 32u, out);
          hi = out[0];
        }
        const uint32_t o2[2] = {hi, 0u};
        for (int p = 0; p < specs.n; ++p) atomicAdd(&h[p][wv][words_digit2<2>(o2, specs.d[p])], 1u);
      }
      if (++j == per) {
        j = 0;
        ++r;
      }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
  for (int p = 0; p < specs.n; ++p) {
    const uint32_t v = h[p][0][threadIdx.x] + h[p][1][threadIdx.x] + h[p][2][threadIdx.x] + h[p][3][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}

// The same histograms with straight-line code per item (the usual plans: every digit is one bit field of the first key
// word).  k_s1_digit_hist above spends ~100 VALU operations and a dozen branches per item (slot cases, the slow path of the
// store's first bases inlined eight times, the generic two-field digit read from the argument block per pass); here the
// slot -> offset / forced-strand mapping is arithmetic, the read's base offset advances with the slots, the digits are
// shift + mask with both in scalar registers, and the three items whose window would start before the store (read 0,
// slots 0..2) are counted by one thread up front.
// PRE: the window words of a thread's IT consecutive items are requested up front — four words for the run that starts at its
// first item, four for the start of the next read, as in S1GenBlocked — instead of being reloaded (and waited for) inside the
// item loop whenever the window moves into the next word.
template <int IT, int NP, bool PRE = false>  // NP digit histograms
__global__ __launch_bounds__(256) void k_s1_digit_hist_plain(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                             HiDigits hd, unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r,
                                                             const uint32_t *__restrict__ keep) {
  // keep != nullptr: only the items of the kept lv1 buckets are counted (the generating pass drops the others, S1GenT<true>)
  constexpr int B = 256 * IT;
  __shared__ uint32_t h[kFastPasses][4][256];
  for (int i = threadIdx.x; i < kFastPasses * 4 * 256; i += 256) (&h[0][0][0])[i] = 0;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const int km1 = k - 1;
  const uint64_t kmask = ~0ull << (64 - 2 * km1);
  const uint32_t qlast = L - k + 1, jf = L - k + 2;  // last offset of a (k-1)-mer; first slot of the forced pair at the read's end
  auto count = [&](uint32_t hi) {
    if (keep && !s1_bucket_kept(keep, hi)) return;
#pragma unroll
    for (int p = 0; p < NP; ++p) atomicAdd(&h[p][wv][(hi >> hd.sh[p]) & hd.mk[p]], 1u);
  };
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (uint32_t j = 0; j < 3 && j < n_items; ++j) {
      uint32_t out[3];
      s1_make_item<2, 3, true>(seq, 0, L, k, j, 0, 32u, out);
      count(out[0]);
    }
  const uint64_t n_blocks = (n_items + B - 1) / B;
  uint64_t q0 = ((uint32_t)blockIdx.x * (uint32_t)B) / per;
  uint32_t rem0 = ((uint32_t)blockIdx.x * (uint32_t)B) % per;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint32_t j = t - dq * per;
    uint64_t base = (q0 + dq) * L;  // first base of the read
    uint64_t wcur = ~0ull;
    uint32_t x0 = 0, x1 = 0, x2 = 0;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    uint64_t wnext = 0;
    // nothing of this block is this thread's (the last block): its window loads go to the start of the store, nothing is
    // counted — every address a thread asks for lies inside the store and its 32 pad words
    if (g0 >= n_items) {
      j = 0;
      base = 0;
    }
    if constexpr (PRE) {
      static_assert(IT <= 8, "a run of IT windows starts in at most two words");
      const uint32_t qs = min(j > 0 ? j - 1 : 0u, qlast);
      const uint64_t as = base + qs, bs = as >= 2 ? as - 2 : 0;
      wcur = bs >> 4;
      wnext = (base + L - 2) >> 4;
      c0 = seq[wcur]; c1 = seq[wcur + 1]; c2 = seq[wcur + 2]; c3 = seq[wcur + 3];
      n0 = seq[wnext]; n1 = seq[wnext + 1]; n2 = seq[wnext + 2]; n3 = seq[wnext + 3];
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const uint32_t q = min(j > 0 ? j - 1 : 0u, qlast);
      const bool forced = j < 2 || j >= jf;
      const uint32_t fstrand = j < 2 ? j : j - jf;
      const uint64_t a = base + q;
      const uint64_t b = a >= 2 ? a - 2 : 0, w = b >> 4;
      if constexpr (PRE) {
        const bool second = w != wcur;
        x0 = second ? c1 : c0;
        x1 = second ? c2 : c1;
        x2 = second ? c3 : c2;
      } else if (w != wcur) {
        x0 = seq[w];
        x1 = seq[w + 1];
        x2 = seq[w + 2];
        wcur = w;
      }
      const unsigned sh = (unsigned)(b & 15) * 2;
      const uint64_t win = ((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh);
      const uint64_t f = (win << 4) & kmask;
      const uint64_t rc = rc64(f, km1);
      const bool use_rc = forced ? fstrand == 1 : f > rc;  // (f == rc: the same first word either way)
      if (g0 + u < n_items && a >= 2) count((uint32_t)((use_rc ? rc : f) >> 32));
      if (++j == per) {
        j = 0;
        base += L;
        if constexpr (PRE) {
          c0 = n0; c1 = n1; c2 = n2; c3 = n3;
          wcur = wnext;
        }
      }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint32_t v = h[p][0][threadIdx.x] + h[p][1][threadIdx.x] + h[p][2][threadIdx.x] + h[p][3][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}

// The same histograms with the window arithmetic of S1GenRollT: one window and one reverse complement per run of a thread's IT
// consecutive items (and one pair for the start of the next read), two shifts per item (k <= 23, >= IT slots per read).
template <int IT, int NP, bool VAR = false>  // VAR: reads of any length, `per` item slots each, start[] says where they lie (S1GenVarT)
__global__ __launch_bounds__(256) void k_s1_digit_hist_roll(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                            HiDigits hd, unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r,
                                                            const uint32_t *__restrict__ keep, const uint64_t *__restrict__ start, uint64_t n_seqs) {
  static_assert(IT <= 8, "a run of IT items and their flanks inside one 32-base window");
  constexpr int B = 256 * IT;
  __shared__ uint32_t h[kFastPasses][4][256];
  for (int i = threadIdx.x; i < kFastPasses * 4 * 256; i += 256) (&h[0][0][0])[i] = 0;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const int km1 = k - 1;
  const uint64_t kmask = ~0ull << (64 - 2 * km1);
  const unsigned rsh = (unsigned)(2 * (30 - km1));
  auto count = [&](uint32_t hi) {
    if (keep && !s1_bucket_kept(keep, hi)) return;
#pragma unroll
    for (int p = 0; p < NP; ++p) atomicAdd(&h[p][wv][(hi >> hd.sh[p]) & hd.mk[p]], 1u);
  };
  // (the first two bases of the store need no case of their own: the window at base 0 shifted down, s1_window_addr)
  const uint64_t n_blocks = (n_items + B - 1) / B;
  uint64_t q0 = ((uint32_t)blockIdx.x * (uint32_t)B) / per;
  uint32_t rem0 = ((uint32_t)blockIdx.x * (uint32_t)B) % per;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint32_t j = t - dq * per;
    uint64_t r = q0 + dq;  // the read
    if (g0 >= n_items) {   // (nothing of this block is this thread's: loads from the start of the store, nothing counted)
      j = 0;
      r = 0;
    }
    S1ReadGeo rd, rdn;
    if constexpr (VAR) {
      const uint64_t s0 = start[r], s1 = start[r + 1], s2 = start[r + 2 < n_seqs ? r + 2 : n_seqs];
      rd = s1_read_geo(s0, s1, k);
      rdn = s1_read_geo(s1, s2, k);
    } else {
      rd = s1_read_geo(r * L, r * L + L, k);
      rdn = s1_read_geo(r * L + L, r * L + 2 * (uint64_t)L, k);
    }
    uint32_t qrun = min(j > 0 ? j - 1 : 0u, rd.qlast);
    uint64_t wcur, wnext;
    unsigned sh0, down0, shn, downn;
    s1_window_addr(rd.base + qrun, wcur, sh0, down0);
    s1_window_addr(rdn.base, wnext, shn, downn);
    const uint32_t c0 = seq[wcur], c1 = seq[wcur + 1], c2 = seq[wcur + 2];
    const uint32_t n0 = seq[wnext], n1 = seq[wnext + 1], n2 = seq[wnext + 2];
    uint64_t W = (((uint64_t)funnel_l(c0, c1, sh0) << 32) | funnel_l(c1, c2, sh0)) >> down0;
    uint64_t R = rc64(W, 32);
    const uint64_t Wn = (((uint64_t)funnel_l(n0, n1, shn) << 32) | funnel_l(n1, n2, shn)) >> downn;
    const uint64_t Rn = rc64(Wn, 32);
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const uint32_t q = min(j > 0 ? j - 1 : 0u, rd.qlast);
      const bool forced = j < 2 || j >= rd.jf;
      const uint32_t fstrand = j < 2 ? j : j - rd.jf;
      const unsigned d2 = (q - qrun) * 2;
      const uint64_t f = (W << (d2 + 4)) & kmask;
      const uint64_t rc = (R << (rsh - d2)) & kmask;
      const bool use_rc = forced ? fstrand == 1 : f > rc;  // (f == rc: the same first word either way)
      if (g0 + u < n_items && j < rd.cnt) count((uint32_t)((use_rc ? rc : f) >> 32));
      if (++j == per) {
        j = 0;
        rd = rdn;
        W = Wn;
        R = Rn;
        qrun = 0;
      }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint32_t v = h[p][0][threadIdx.x] + h[p][1][threadIdx.x] + h[p][2][threadIdx.x] + h[p][3][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}

// the digit histograms of count's prefix passes (every digit one bit field of the first key word): CountGenT's arithmetic, no records
template <int IT, int NP, bool VAR = false, bool WIDE = false>  // VAR: reads of any length, `per` = max_len - k item slots each (CountGenVarT); WIDE: k = 23..27, a window per item (CountGenWideT)
__global__ __launch_bounds__(256) void k_count_digit_hist_roll(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                               HiDigits hd, unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r,
                                                               const uint64_t *__restrict__ start, uint64_t n_seqs, const uint32_t *__restrict__ keep) {
  static_assert(IT <= 8, "a run of IT edges and their flanks inside one 32-base window");
  constexpr int B = 256 * IT;
  __shared__ uint32_t h[kFastPasses][4][256];
  for (int i = threadIdx.x; i < kFastPasses * 4 * 256; i += 256) (&h[0][0][0])[i] = 0;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const uint64_t emask = ~0ull << (64 - 2 * (k + 1));
  const unsigned rsh = (unsigned)(2 * (30 - k));
  const uint64_t n_blocks = (n_items + B - 1) / B;
  uint64_t q0 = ((uint32_t)blockIdx.x * (uint32_t)B) / per;
  uint32_t rem0 = ((uint32_t)blockIdx.x * (uint32_t)B) % per;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint32_t j = t - dq * per;
    uint64_t r = q0 + dq;
    if (g0 >= n_items) {  // (nothing of this block is this thread's: loads from the start of the store, nothing counted)
      j = 0;
      r = 0;
    }
    uint64_t base, base_n;
    uint32_t cnt, cntn;  // item slots this read / the next one fills
    if constexpr (VAR) {
      const uint64_t s0 = start[r], s1 = start[r + 1], s2 = start[r + 2 < n_seqs ? r + 2 : n_seqs];
      const uint32_t L0 = (uint32_t)(s1 - s0), L1 = (uint32_t)(s2 - s1);
      base = s0;
      base_n = s1;
      cnt = L0 >= (uint32_t)k + 1 ? L0 - k : 0u;
      cntn = L1 >= (uint32_t)k + 1 ? L1 - k : 0u;
    } else {
      base = r * L;
      base_n = base + L;
      cnt = cntn = per;
    }
    uint64_t wcur, wnext;
    unsigned sh0, down0, shn, downn;
    count_window_addr(base + min(j, cnt ? cnt - 1 : 0u), wcur, sh0, down0);
    count_window_addr(base_n, wnext, shn, downn);
    const uint32_t c0 = seq[wcur], c1 = seq[wcur + 1], c2 = seq[wcur + 2];
    const uint32_t n0 = seq[wnext], n1 = seq[wnext + 1], n2 = seq[wnext + 2];
    uint64_t W = (((uint64_t)funnel_l(c0, c1, sh0) << 32) | funnel_l(c1, c2, sh0)) >> down0;
    uint64_t R = rc64(W, 32);
    const uint64_t Wn = (((uint64_t)funnel_l(n0, n1, shn) << 32) | funnel_l(n1, n2, shn)) >> downn;
    const uint64_t Rn = rc64(Wn, 32);
    uint32_t prun = j;
    static_assert(!(VAR && WIDE), "the per-item windows serve fixed-length libraries");
    uint64_t wbase = base, wword = ~0ull;  // WIDE: the read the item is in, the word its window was last loaded from
    uint32_t x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      uint64_t f, rc;
      if constexpr (WIDE) {
        const uint64_t a = wbase + j, b = a >= 1 ? a - 1 : 0, w = b >> 4;
        if (w != wword) {
          x0 = seq[w];
          x1 = seq[w + 1];
          x2 = seq[w + 2];
          wword = w;
        }
        const unsigned sh = (unsigned)(b & 15) * 2;
        const uint64_t win = (((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh)) >> (a >= 1 ? 0u : 2u);
        f = (win << 2) & emask;
        rc = rc64(f, k + 1);
      } else {
        const unsigned d2 = (j - prun) * 2;
        f = (W << (d2 + 2)) & emask;
        rc = (R << (rsh - d2)) & emask;
      }
      const uint32_t hi = (uint32_t)((rc < f ? rc : f) >> 32);
      if (g0 + u < n_items && j < cnt && (!keep || s1_bucket_kept(keep, hi))) {  // (keep: a memory-plan pass counts what its generating pass will keep)
#pragma unroll
        for (int p = 0; p < NP; ++p) atomicAdd(&h[p][wv][(hi >> hd.sh[p]) & hd.mk[p]], 1u);
      }
      if (++j == per) {
        j = 0;
        cnt = cntn;
        W = Wn;
        R = Rn;
        prun = 0;
        wbase = base_n;
      }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint32_t v = h[p][0][threadIdx.x] + h[p][1][threadIdx.x] + h[p][2][threadIdx.x] + h[p][3][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}

// The lv1-bucket histogram of `count` (KmerCounter::Lv0CalcBucketSize, kmer_counter.cpp:114-156) from the packed reads with CountGenT's
// window arithmetic — what a memory plan asks for before it splits a job into bucket ranges; one half of the bucket space per launch,
// as k_s1_bucket_hist_fast.  k <= kCountStreamMaxK, >= IT item slots per read.
template <int IT, bool VAR, bool WIDE = false>
__global__ __launch_bounds__(1024) void k_count_bucket_hist(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                            unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r, uint32_t half,
                                                            const uint64_t *__restrict__ start, uint64_t n_seqs) {
  static_assert(IT <= 8, "a run of IT edges and their flanks inside one 32-base window");
  constexpr int NT = 1024, B = NT * IT, NB = MHX_NUM_BUCKETS / 2;
  __shared__ uint32_t h[NB];
  for (int i = threadIdx.x; i < NB; i += NT) h[i] = 0;
  __syncthreads();
  const uint64_t emask = ~0ull << (64 - 2 * (k + 1));
  const unsigned rsh = (unsigned)(2 * (30 - k));
  const uint64_t n_blocks = (n_items + B - 1) / B;
  uint64_t q0 = ((uint64_t)blockIdx.x * (uint64_t)B) / per;
  uint32_t rem0 = (uint32_t)(((uint64_t)blockIdx.x * (uint64_t)B) % per);
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint32_t j = t - dq * per;
    uint64_t r = q0 + dq;
    if (g0 >= n_items) {  // (nothing of this block is this thread's: loads from the start of the store, nothing counted)
      j = 0;
      r = 0;
    }
    uint64_t base, base_n;
    uint32_t cnt, cntn;  // item slots this read / the next one fills
    if constexpr (VAR) {
      const uint64_t s0 = start[r], s1 = start[r + 1], s2 = start[r + 2 < n_seqs ? r + 2 : n_seqs];
      const uint32_t L0 = (uint32_t)(s1 - s0), L1 = (uint32_t)(s2 - s1);
      base = s0;
      base_n = s1;
      cnt = L0 >= (uint32_t)k + 1 ? L0 - k : 0u;
      cntn = L1 >= (uint32_t)k + 1 ? L1 - k : 0u;
    } else {
      base = r * L;
      base_n = base + L;
      cnt = cntn = per;
    }
    uint64_t wcur, wnext;
    unsigned sh0, down0, shn, downn;
    count_window_addr(base + min(j, cnt ? cnt - 1 : 0u), wcur, sh0, down0);
    count_window_addr(base_n, wnext, shn, downn);
    const uint32_t c0 = seq[wcur], c1 = seq[wcur + 1], c2 = seq[wcur + 2];
    const uint32_t n0 = seq[wnext], n1 = seq[wnext + 1], n2 = seq[wnext + 2];
    uint64_t W = (((uint64_t)funnel_l(c0, c1, sh0) << 32) | funnel_l(c1, c2, sh0)) >> down0;
    uint64_t R = rc64(W, 32);
    const uint64_t Wn = (((uint64_t)funnel_l(n0, n1, shn) << 32) | funnel_l(n1, n2, shn)) >> downn;
    const uint64_t Rn = rc64(Wn, 32);
    uint32_t prun = j;
    static_assert(!(VAR && WIDE), "the per-item windows serve fixed-length libraries");
    uint64_t wbase = base, wword = ~0ull;
    uint32_t x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      uint64_t f, rc;
      if constexpr (WIDE) {
        const uint64_t a = wbase + j, bb = a >= 1 ? a - 1 : 0, w = bb >> 4;
        if (w != wword) {
          x0 = seq[w];
          x1 = seq[w + 1];
          x2 = seq[w + 2];
          wword = w;
        }
        const unsigned sh = (unsigned)(bb & 15) * 2;
        const uint64_t win = (((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh)) >> (a >= 1 ? 0u : 2u);
        f = (win << 2) & emask;
        rc = rc64(f, k + 1);
      } else {
        const unsigned d2 = (j - prun) * 2;
        f = (W << (d2 + 2)) & emask;
        rc = (R << (rsh - d2)) & emask;
      }
      const uint32_t b = (uint32_t)((rc < f ? rc : f) >> 48);
      if (g0 + u < n_items && j < cnt && (b >> 15) == half) atomicAdd(&h[b & (NB - 1)], 1u);
      if (++j == per) {
        j = 0;
        cnt = cntn;
        W = Wn;
        R = Rn;
        prun = 0;
        wbase = base_n;
      }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += NT)
    if (h[i]) atomicAdd(&ghist[half * NB + i], (unsigned long long)h[i]);
}

// The lv1-bucket histogram of stage 1 (the reference's Lv0CalcBucketSize, read_to_sdbg_s1.cpp:145-206) for the fast shape —
// what a memory plan asks for before it splits a 100 M-read job into bucket ranges.  The same window arithmetic as the
// digit-histogram pre-pass; the 65 536 counters do not fit the LDS as 32-bit words, so a launch counts one HALF of the
// bucket space (128 KB, one 1024-thread workgroup per CU) and the host launches twice.  (The general path takes the
// histogram from extracted items with one global atomic per item: seconds at 10^10 items.)
template <int IT, bool ROLL = false>  // ROLL: the window arithmetic of S1GenRollT (k <= 23, >= IT slots per read)
__global__ __launch_bounds__(1024) void k_s1_bucket_hist_fast(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                              unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r, uint32_t half) {
  constexpr int NT = 1024, B = NT * IT, NB = MHX_NUM_BUCKETS / 2;
  __shared__ uint32_t h[NB];
  for (int i = threadIdx.x; i < NB; i += NT) h[i] = 0;
  __syncthreads();
  const int km1 = k - 1;
  const uint64_t kmask = ~0ull << (64 - 2 * km1);
  const uint32_t qlast = L - k + 1, jf = L - k + 2;
  auto count = [&](uint32_t hi) {
    const uint32_t b = hi >> 16;
    if ((b >> 15) == half) atomicAdd(&h[b & (NB - 1)], 1u);
  };
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (uint32_t j = 0; j < 3 && j < n_items; ++j) {  // (the items whose window would start before the store)
      uint32_t out[3];
      s1_make_item<2, 3, true>(seq, 0, L, k, j, 0, 32u, out);
      count(out[0]);
    }
  const uint64_t n_blocks = (n_items + B - 1) / B;
  uint64_t q0 = ((uint64_t)blockIdx.x * (uint64_t)B) / per;
  uint32_t rem0 = (uint32_t)(((uint64_t)blockIdx.x * (uint64_t)B) % per);
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint32_t j = t - dq * per;
    uint64_t base = (q0 + dq) * L;
    if (g0 >= n_items) {  // (nothing of this block is this thread's: loads from the start of the store, nothing counted)
      j = 0;
      base = 0;
    }
    if constexpr (ROLL) {
      static_assert(IT <= 8, "a run of IT items and their flanks inside one 32-base window");
      uint32_t qrun = min(j > 0 ? j - 1 : 0u, qlast);
      const uint64_t a0 = base + qrun, b0 = a0 >= 2 ? a0 - 2 : 0;
      const uint64_t wc0 = b0 >> 4, wnext = (base + L - 2) >> 4;
      const uint32_t c0 = seq[wc0], c1 = seq[wc0 + 1], c2 = seq[wc0 + 2];
      const uint32_t n0 = seq[wnext], n1 = seq[wnext + 1], n2 = seq[wnext + 2];
      const unsigned sh0 = (unsigned)(b0 & 15) * 2, down0 = a0 >= 2 ? 0u : (unsigned)(2 - a0) * 2;
      uint64_t W = (((uint64_t)funnel_l(c0, c1, sh0) << 32) | funnel_l(c1, c2, sh0)) >> down0;
      uint64_t R = rc64(W, 32);
      const unsigned shn = (unsigned)((base + L - 2) & 15) * 2;
      const uint64_t Wn = ((uint64_t)funnel_l(n0, n1, shn) << 32) | funnel_l(n1, n2, shn);
      const uint64_t Rn = rc64(Wn, 32);
      const unsigned rsh = (unsigned)(2 * (30 - km1));
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const uint32_t q = min(j > 0 ? j - 1 : 0u, qlast);
        const bool forced = j < 2 || j >= jf;
        const uint32_t fstrand = j < 2 ? j : j - jf;
        const unsigned d2 = (q - qrun) * 2;
        const uint64_t f = (W << (d2 + 4)) & kmask;
        const uint64_t rc = (R << (rsh - d2)) & kmask;
        const bool use_rc = forced ? fstrand == 1 : f > rc;
        if (g0 + u < n_items && base + q >= 2) count((uint32_t)((use_rc ? rc : f) >> 32));
        if (++j == per) {
          j = 0;
          base += L;
          W = Wn;
          R = Rn;
          qrun = 0;
        }
      }
    } else {
    uint64_t wcur = ~0ull;
    uint32_t x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const uint32_t q = min(j > 0 ? j - 1 : 0u, qlast);
      const bool forced = j < 2 || j >= jf;
      const uint32_t fstrand = j < 2 ? j : j - jf;
      const uint64_t a = base + q;
      const uint64_t b = a >= 2 ? a - 2 : 0, w = b >> 4;
      if (w != wcur) {
        x0 = seq[w];
        x1 = seq[w + 1];
        x2 = seq[w + 2];
        wcur = w;
      }
      const unsigned sh = (unsigned)(b & 15) * 2;
      const uint64_t win = ((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh);
      const uint64_t f = (win << 4) & kmask;
      const uint64_t rc = rc64(f, km1);
      const bool use_rc = forced ? fstrand == 1 : f > rc;  // (f == rc: the same first word either way)
      if (g0 + u < n_items && a >= 2) count((uint32_t)((use_rc ? rc : f) >> 32));
      if (++j == per) {
        j = 0;
        base += L;
      }
    }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += NT)
    if (h[i]) atomicAdd(&ghist[half * NB + i], (unsigned long long)h[i]);
}
// the same histogram for a library of reads of any length: `per` item slots per read, start[] says where the reads lie (S1GenVarT)
template <int IT>
__global__ __launch_bounds__(1024) void k_s1_bucket_hist_var(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint64_t n_seqs,
                                                             uint32_t per, uint64_t n_slots, int k, unsigned long long *__restrict__ ghist, uint32_t step_q,
                                                             uint32_t step_r, uint32_t half) {
  static_assert(IT <= 8, "a run of IT items and their flanks inside one 32-base window");
  constexpr int NT = 1024, B = NT * IT, NB = MHX_NUM_BUCKETS / 2;
  __shared__ uint32_t h[NB];
  for (int i = threadIdx.x; i < NB; i += NT) h[i] = 0;
  __syncthreads();
  const int km1 = k - 1;
  const uint64_t kmask = ~0ull << (64 - 2 * km1);
  const unsigned rsh = (unsigned)(2 * (30 - km1));
  const uint64_t n_blocks = (n_slots + B - 1) / B;
  uint64_t q0 = ((uint64_t)blockIdx.x * (uint64_t)B) / per;
  uint32_t rem0 = (uint32_t)(((uint64_t)blockIdx.x * (uint64_t)B) % per);
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g0 = blk * B + (uint64_t)threadIdx.x * IT;
    const uint32_t t = rem0 + (uint32_t)threadIdx.x * IT, dq = t / per;
    uint32_t j = t - dq * per;
    uint64_t r = q0 + dq;
    if (g0 >= n_slots) {  // (nothing of this block is this thread's: loads from the start of the store, nothing counted)
      j = 0;
      r = 0;
    }
    const uint64_t s0 = start[r], s1 = start[r + 1], s2 = start[r + 2 < n_seqs ? r + 2 : n_seqs];
    S1ReadGeo rd = s1_read_geo(s0, s1, k);
    const S1ReadGeo rdn = s1_read_geo(s1, s2, k);
    uint32_t qrun = min(j > 0 ? j - 1 : 0u, rd.qlast);
    uint64_t wcur, wnext;
    unsigned sh0, down0, shn, downn;
    s1_window_addr(rd.base + qrun, wcur, sh0, down0);
    s1_window_addr(rdn.base, wnext, shn, downn);
    const uint32_t c0 = seq[wcur], c1 = seq[wcur + 1], c2 = seq[wcur + 2];
    const uint32_t n0 = seq[wnext], n1 = seq[wnext + 1], n2 = seq[wnext + 2];
    uint64_t W = (((uint64_t)funnel_l(c0, c1, sh0) << 32) | funnel_l(c1, c2, sh0)) >> down0;
    uint64_t R = rc64(W, 32);
    const uint64_t Wn = (((uint64_t)funnel_l(n0, n1, shn) << 32) | funnel_l(n1, n2, shn)) >> downn;
    const uint64_t Rn = rc64(Wn, 32);
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const uint32_t q = min(j > 0 ? j - 1 : 0u, rd.qlast);
      const bool forced = j < 2 || j >= rd.jf;
      const uint32_t fstrand = j < 2 ? j : j - rd.jf;
      const unsigned d2 = (q - qrun) * 2;
      const uint64_t f = (W << (d2 + 4)) & kmask;
      const uint64_t rc = (R << (rsh - d2)) & kmask;
      const bool use_rc = forced ? fstrand == 1 : f > rc;
      if (g0 + u < n_slots && j < rd.cnt) {
        const uint32_t b = (uint32_t)((use_rc ? rc : f) >> 48);
        if ((b >> 15) == half) atomicAdd(&h[b & (NB - 1)], 1u);
      }
      if (++j == per) {
        j = 0;
        rd = rdn;
        W = Wn;
        R = Rn;
        qrun = 0;
      }
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += NT)
    if (h[i]) atomicAdd(&ghist[half * NB + i], (unsigned long long)h[i]);
}
// -> true when it ran (fixed-length reads, 12-byte compact records); hist: device, 65 536 counters, zeroed by the caller
bool s1_bucket_histogram_fast(mhx_ctx *c, uint32_t k, unsigned long long *hist) {
  SeqSet &s = c->seqs;
  if (c->opt("s1_bucket_hist_fast", 1) && s1_shape_is_var_fast(c, k, s1_compact(c, k, 0))) {  // reads of any length: padded item slots
    constexpr int ITV = 8;
    const uint32_t per = s.max_len - k + 4;
    const uint64_t n_slots = s.n_seqs * (uint64_t)per;
    const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
    const unsigned grid = (unsigned)std::min<uint64_t>(div_ceil(n_slots, 1024 * ITV), cus);
    const uint64_t stride_items = (uint64_t)grid * 1024 * ITV;
    for (uint32_t half = 0; half < 2; ++half)
      MHX_LAUNCH(c, "s1_bucket_hist", (double)s.n_bases / 4 + (double)s.n_seqs * 8,
                 hipLaunchKernelGGL((k_s1_bucket_hist_var<ITV>), dim3(grid), dim3(1024), 0, c->stream, s.words.as<uint32_t>(), s.start.as<uint64_t>(), s.n_seqs,
                                    per, n_slots, (int)k, hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), half));
    return true;
  }
  if (!c->opt("s1_bucket_hist_fast", 1) || !s.n_seqs || s.fixed_len < k + 1 || !s1_compact(c, k, 0) || (2 * (k - 1) + 6 + 31) / 32 != 2 || k > 29) return false;  // (two key words)
  constexpr int IT = 8;
  const uint32_t per = s.fixed_len - k + 4;
  const uint64_t n_items = s.n_seqs * (uint64_t)per;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  const unsigned grid = (unsigned)std::min<uint64_t>(div_ceil(n_items, 1024 * IT), cus);
  const uint64_t stride_items = (uint64_t)grid * 1024 * IT;
  const bool roll = per >= IT && (int)k <= kS1RollMaxK && c->opt("s1_digit_hist_roll", 1) != 0;
  for (uint32_t half = 0; half < 2; ++half) {
    if (roll)
      MHX_LAUNCH(c, "s1_bucket_hist", (double)s.n_bases / 4,
                 hipLaunchKernelGGL((k_s1_bucket_hist_fast<IT, true>), dim3(grid), dim3(1024), 0, c->stream, s.words.as<uint32_t>(), s.fixed_len, per, n_items,
                                    (int)k, hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), half));
    else
      MHX_LAUNCH(c, "s1_bucket_hist", (double)s.n_bases / 4,
                 hipLaunchKernelGGL((k_s1_bucket_hist_fast<IT>), dim3(grid), dim3(1024), 0, c->stream, s.words.as<uint32_t>(), s.fixed_len, per, n_items,
                                    (int)k, hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), half));
  }
  return true;
}

// the shapes CountGenT / CountGenVarT serve: >= 8 item slots per read, k <= kCountStreamMaxK; a library of several read lengths while at
// least s1_var_min_fill per cent of the padded slots are edges
bool count_shape_is_fast(const mhx_ctx *c, uint32_t k) {
  const SeqSet &s = c->seqs;
  if (!s.n_seqs || k < 9) return false;
  if ((int)k > kCountStreamMaxK)  // k = 23..27: a window per item (CountGenWideT), reads of one length, no position tags
    return (int)k <= kCountStreamWideMaxK && c->opt("count_stream_wide", 1) && s.fixed_len >= k + 1 && s.fixed_len - k >= 8 &&
           (c->count_edges_only || ((s.n_bases >> s1_pos_bits(c)) == 0 && (c->global_bases >> s1_pos_bits(c)) == 0));
  if (s.fixed_len) return s.fixed_len >= k + 1 && s.fixed_len - k >= 8;
  if (!c->opt("s1_var_fast", 1) || s.max_len < k + 1 || s.max_len - k < 8 || s.n_bases <= s.n_seqs * (uint64_t)k) return false;
  return (double)s.n_bases * 100.0 >= (double)c->opt("s1_var_min_fill", 50) * (double)s.n_seqs * s.max_len;
}
// -> true when it ran; hist: device, 65 536 counters, zeroed by the caller
bool count_bucket_histogram_fast(mhx_ctx *c, uint32_t k, unsigned long long *hist) {
  SeqSet &s = c->seqs;
  if (!c->opt("s1_bucket_hist_fast", 1) || !c->opt("count_stream", 1) || !count_shape_is_fast(c, k)) return false;
  constexpr int IT = 8;
  const bool var = s.fixed_len == 0;
  const uint32_t per = (var ? s.max_len : s.fixed_len) - k;
  const uint64_t n_slots = s.n_seqs * (uint64_t)per;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  const unsigned grid = (unsigned)std::min<uint64_t>(div_ceil(n_slots, 1024 * IT), cus);
  const uint64_t stride_items = (uint64_t)grid * 1024 * IT;
  for (uint32_t half = 0; half < 2; ++half) {
    if ((int)k > kCountStreamMaxK)
      MHX_LAUNCH(c, "count_bucket_hist", (double)s.n_bases / 4,
                 hipLaunchKernelGGL((k_count_bucket_hist<IT, false, true>), dim3(grid), dim3(1024), 0, c->stream, s.words.as<uint32_t>(), s.fixed_len, per, n_slots, (int)k,
                                    hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), half, s.start.as<uint64_t>(), s.n_seqs));
    else if (var)
      MHX_LAUNCH(c, "count_bucket_hist", (double)s.n_bases / 4 + (double)s.n_seqs * 8,
                 hipLaunchKernelGGL((k_count_bucket_hist<IT, true>), dim3(grid), dim3(1024), 0, c->stream, s.words.as<uint32_t>(), 0u, per, n_slots, (int)k, hist,
                                    (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), half, s.start.as<uint64_t>(), s.n_seqs));
    else
      MHX_LAUNCH(c, "count_bucket_hist", (double)s.n_bases / 4,
                 hipLaunchKernelGGL((k_count_bucket_hist<IT, false>), dim3(grid), dim3(1024), 0, c->stream, s.words.as<uint32_t>(), s.fixed_len, per, n_slots, (int)k,
                                    hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), half, s.start.as<uint64_t>(), s.n_seqs));
  }
  return true;
}

template <int IT, bool WRITE>  // items per thread and trip; WRITE = false: only the digit histograms: their window loads are issued together (one in flight per thread = latency-bound)
__global__ __launch_bounds__(256) void k_s1_extract_fast(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                         uint64_t pos_base, uint32_t pos_bits, uint32_t *__restrict__ items, DigitSpecs specs,
                                                         unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r) {
  constexpr int B = 256 * IT;  // items per workgroup and trip
  __shared__ uint32_t xpose[WRITE ? B * 3 : 1];
  // digit histograms of the coming sort passes (at most kFastPasses of them here), one copy per wavefront: the lanes of
  // different wavefronts never queue up behind each other at a hot digit
  __shared__ uint32_t h[kFastPasses][4][256];
  for (int i = threadIdx.x; i < kFastPasses * 4 * 256; i += 256) (&h[0][0][0])[i] = 0;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const uint64_t n_blocks = (n_items + B - 1) / B;
  // (read, first slot) of this workgroup's current block of B items
  uint64_t q0 = ((uint32_t)blockIdx.x * (uint32_t)B) / per;
  uint32_t rem0 = ((uint32_t)blockIdx.x * (uint32_t)B) % per;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    uint32_t outs[IT][3];
    bool ok[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const uint64_t g = blk * B + (uint64_t)u * 256 + threadIdx.x;
      ok[u] = g < n_items;
      const uint32_t t = rem0 + (uint32_t)u * 256u + threadIdx.x, dq = t / per, j = t - dq * per;
      // (a thread beyond the last item recomputes item 0: unconditional loads, nothing stored)
      s1_fast_item(seq, L, k, ok[u] ? (q0 + dq) * L : 0, ok[u] ? j : 2, pos_base, pos_bits, outs[u]);
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      uint32_t(&out)[3] = outs[u];
      if (ok[u]) {
        for (int p = 0; p < specs.n; ++p) atomicAdd(&h[p][wv][words_digit2<3>(out, specs.d[p])], 1u);
      }
      if constexpr (WRITE) {
#pragma unroll
        for (int i = 0; i < 3; ++i) xpose[(u * 256 + threadIdx.x) * 3 + i] = out[i];
      }
    }
    if constexpr (WRITE) {
      __syncthreads();
      const uint64_t w0 = blk * (uint64_t)(B * 3), n_words = n_items * 3;
#pragma unroll
      for (int i = 0; i < 3 * IT; ++i) {
        const uint64_t w = w0 + (uint64_t)i * 256 + threadIdx.x;
        if (w < n_words) items[w] = xpose[i * 256 + threadIdx.x];
      }
      __syncthreads();
    }
    q0 += step_q;
    rem0 += step_r;
    if (rem0 >= per) {
      rem0 -= per;
      ++q0;
    }
  }
  __syncthreads();
  for (int p = 0; p < specs.n; ++p) {
    const uint32_t v = h[p][0][threadIdx.x] + h[p][1][threadIdx.x] + h[p][2][threadIdx.x] + h[p][3][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}


// items of the local reads -> c->ws("items_a"); returns their number.
// Three ways, fastest first: (1) deferred — only the digit histograms of the coming sort are taken here and the sort's first
// pass makes the records itself (fixed-length reads, 12-byte records; under a bucket filter that pass drops the items of
// the other buckets: c->s1_filter_in_gen); (2) the window-arithmetic extraction; (3) the general kernels.
bool s1_shape_is_fast(const mhx_ctx *c, uint32_t k, bool compact) {
  const SeqSet &s = c->seqs;
  return s.n_seqs && s.fixed_len >= k + 1 && compact && s1_kw(k) == 2 && s1_stride(k, compact) == 3 && k <= 29 && c->opt("s1_extract_fast", 1) != 0;
}
// The same front for a library whose reads are NOT of one length (S1GenVarT): item slots padded to the longest read's count.  Taken
// while at least half of the slots are real records (s1_var_min_fill per cent) — beyond that the extraction kernel + loaded passes
// cost less than generating dropped slots.
bool s1_shape_is_var_fast(const mhx_ctx *c, uint32_t k, bool compact) {
  const SeqSet &s = c->seqs;
  if (!s.n_seqs || s.fixed_len || s.max_len < k + 1 || s.max_len - k + 4 < 8 || !compact || s1_kw(k) != 2 || s1_stride(k, compact) != 3 || (int)k > kS1RollMaxK) return false;
  if (!c->opt("s1_extract_fast", 1) || !c->opt("s1_var_fast", 1)) return false;
  const double fill = (double)s.n_bases / ((double)s.n_seqs * s.max_len);
  return fill * 100.0 >= (double)c->opt("s1_var_min_fill", 50);
}
// can a bucket filter be applied inside the generating first pass (instead of extraction batches + a keep/drop split)?
bool s1_filter_in_gen_applies(const mhx_ctx *c, uint32_t k) {
  const bool compact = s1_compact(c, k, 0);
  const bool var = s1_shape_is_var_fast(c, k, compact);
  if (!c->filter_on || !c->opt("s1_filter_in_gen", 1) || !(s1_shape_is_fast(c, k, compact) || var) || !c->opt("s1_fused_first_pass", 1)) return false;
  if (!c->opt("s1_digit_hist_blocked", 1) || !c->opt("s1_digit_hist_plain", 1) || !c->opt("s1_gen_any_order", 1) || !c->opt("sort_unit_runs", 1)) return false;
  if (var && !(c->opt("s1_gen_blocked", 0) && c->opt("s1_gen_roll", 1) && c->opt("s1_digit_hist_roll", 1))) return false;
  const uint64_t n_slots = (uint64_t)c->seqs.n_seqs * ((var ? c->seqs.max_len : c->seqs.fixed_len) - k + 4);
  const S1Plan plan = s1_plan(c, k, n_slots, compact, 0);
  // (the plans whose digits are bit fields of the first key word: the prefix plans)
  return plan.seg_bits > 0 && (int)plan.passes.size() <= kFastPasses && sort_takes_generated_first_pass(c, std::max<uint64_t>(c->filter_expected, 1), 3, plan.passes);
}

uint64_t s1_extract(mhx_ctx *c, uint32_t k, bool compact) {
  SeqSet &s = c->seqs;
  if (k < 9 || k > MHX_MAX_K) throw Error("read2sdbg: k out of range [9,255]");
  const int KWv = s1_kw(k), S = s1_stride(k, compact);
  const uint64_t ns = s.n_seqs;
  hipStream_t st = c->stream;
  const bool filter_in_gen = c->s1_filter_in_gen;
  c->s1_filter_in_gen = false;
  uint32_t *cnt = c->ws("seq_item_cnt", (ns + 1) * 4).as<uint32_t>();
  uint64_t *item_start = c->ws("seq_item_start", (ns + 2) * 8).as<uint64_t>();
  uint64_t n_items = 0;
  // reads of any length on the generating pass (S1GenVarT): only as deferred items — there is no extraction kernel of that form
  const bool var_fast = s1_shape_is_var_fast(c, k, compact) && (filter_in_gen || c->s1_defer_items) && c->opt("s1_fused_first_pass", 1) &&
                        c->opt("s1_gen_blocked", 0) && c->opt("s1_gen_roll", 1) && c->opt("s1_digit_hist_roll", 1) && c->opt("s1_digit_hist_blocked", 1) &&
                        c->opt("s1_digit_hist_plain", 1) && c->opt("s1_gen_any_order", 1) && c->opt("sort_unit_runs", 1);
  c->s1_var_gen = false;
  const bool shape_fast = s1_shape_is_fast(c, k, compact) || var_fast;
  if (ns && shape_fast && s.fixed_len >= k + 1) {
    n_items = ns * (uint64_t)(s.fixed_len - k + 4);  // (no per-read table for reads of one length)
  } else if (ns) {
    MHX_LAUNCH(c, "item_counts", (double)ns * 12,
               hipLaunchKernelGGL(k_s1_item_counts, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, s.start.as<uint64_t>(), ns, k, cnt));
    exclusive_scan_u32_u64(c, cnt, item_start, ns, item_start + ns + 1);
    MHX_HIP(hipMemcpyAsync(&n_items, item_start + ns + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  if (filter_in_gen && !(shape_fast && n_items)) throw Error("s1_extract: the bucket filter was left to a generating pass that does not apply");
  const size_t item_bytes = (size_t)S * 4;
  // (deferred + filtered: the buffer is sized once the kept items are counted)
  uint32_t *buf_a = filter_in_gen ? nullptr : c->ws("items_a", n_items * item_bytes + 64).as<uint32_t>();
  if (n_items) {
    const unsigned grid = 256 * 8;
    const bool fixed = s.fixed_len >= k + 1 && n_items == (uint64_t)ns * (s.fixed_len - k + 4);
    // item slots the generating pass walks: the records of a fixed-length library, max_len - k + 4 per read otherwise
    const uint32_t per_slots = var_fast ? s.max_len - k + 4 : (fixed ? s.fixed_len - k + 4 : 0u);
    const uint64_t n_slots = var_fast ? ns * (uint64_t)per_slots : n_items;
    const uint64_t pos_base = c->pos_base;
    const uint32_t pos_bits = s1_pos_bits(c);
    // the stage-1 sort's digit histograms come for free while the records are still in registers (fixed-length path)
    DigitSpecs specs;
    specs.n = 0;
    unsigned long long *pre_hist = nullptr;
    c->pre_hist_buf = nullptr;
    std::vector<SortPass> plan_passes;
    if ((fixed || var_fast) && S <= 4) {
      plan_passes = s1_plan(c, k, n_items, compact, compact ? 0 : 1).passes;
      if ((int)plan_passes.size() <= kMaxFusedPasses) {
        c->pre_hist_sig = passes_signature(plan_passes);
        specs.n = (int)plan_passes.size();
        for (int p = 0; p < specs.n; ++p) specs.d[p] = spec_of_pass(plan_passes[p], KWv);
        pre_hist = c->ws("sort_pre_hist", (size_t)kMaxFusedPasses * 256 * 8).as<unsigned long long>();
        MHX_HIP(hipMemsetAsync(pre_hist, 0, (size_t)specs.n * 256 * 8, st));
        c->pre_hist_buf = buf_a;
        c->pre_hist_n = n_items;
        c->pre_hist_passes = specs.n;
      }
    }
    // (a variable-length library whose plan or sort cannot take the generated pass goes the general way below)
    bool first_word_digits = true;
    for (int p = 0; p < specs.n; ++p) first_word_digits = first_word_digits && specs.d[p].wi1 == 0 && specs.d[p].mask2 == 0 && specs.d[p].bit1 < 32;
    const bool var_ok = var_fast && pre_hist && specs.n >= 1 && specs.n <= kFastPasses && first_word_digits &&
                        (filter_in_gen || sort_takes_generated_first_pass(c, n_items, 3, plan_passes));
    if (var_fast && !var_ok) {
      if (filter_in_gen) throw Error("s1_extract: the bucket filter was left to a generating pass that does not apply");
      specs.n = 0;
      pre_hist = nullptr;
      c->pre_hist_buf = nullptr;
    }
    const bool fast = (fixed || var_ok) && shape_fast && specs.n <= kFastPasses;
    if (fast) {
      const int it = (int)c->opt("s1_extract_items", 4);
      const uint32_t per = per_slots;
      // Deferred items: the caller sorts right away (run_s1, the multi-GPU pre-sort), so only the digit histograms are taken
      // here and the first sort pass makes the records itself (S1Gen): "items_a" stays empty until that pass has run.
      const bool defer = filter_in_gen || var_ok || (c->s1_defer_items && pre_hist && c->opt("s1_fused_first_pass", 1) &&
                                                      sort_takes_generated_first_pass(c, n_items, 3, plan_passes));
#define MHX_FAST(ITV, WR, NAME)                                                                                                        \
  do {                                                                                                                                 \
    const unsigned fgrid = (unsigned)std::min<uint64_t>(div_ceil(n_items, 256 * ITV), 256 * 8);                                        \
    const uint64_t stride_items = (uint64_t)fgrid * 256 * ITV;                                                                         \
    MHX_LAUNCH(c, NAME, (WR ? (double)n_items * item_bytes : 0.0) + (double)s.n_bases / 4,                                             \
               hipLaunchKernelGGL((k_s1_extract_fast<ITV, WR>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, per, n_items, \
                                  (int)k, pos_base, pos_bits, buf_a, specs, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per))); \
  } while (0)
      bool hi_only = true;  // every digit of the plan comes from the first key word?
      for (int p = 0; p < specs.n; ++p) hi_only = hi_only && specs.d[p].wi1 == 0 && (!specs.d[p].mask2 || specs.d[p].wi2 == 0);
      if (defer) {
        const uint32_t *keep = filter_in_gen ? c->work["filter_bits"].as<uint32_t>() : nullptr;
        bool plain = hi_only && c->opt("s1_digit_hist_blocked", 1) && c->opt("s1_digit_hist_plain", 1) != 0;  // every digit one bit field of the first key word?
        HiDigits hd;
        hd.n = specs.n;
        for (int p = 0; p < specs.n; ++p) {
          plain = plain && specs.d[p].mask2 == 0 && specs.d[p].wi1 == 0 && specs.d[p].bit1 < 32;
          hd.sh[p] = specs.d[p].bit1;
          hd.mk[p] = specs.d[p].mask1;
        }
        if (filter_in_gen && !plain) throw Error("s1_extract: the bucket filter was left to a generating pass that does not apply");
        if (hi_only && c->opt("s1_digit_hist_blocked", 1)) {
          constexpr int ITH = 8;
          const unsigned fgrid = (unsigned)std::min<uint64_t>(div_ceil(n_slots, 256 * ITH), 256 * 8);
          const uint64_t stride_items = (uint64_t)fgrid * 256 * ITH;
#define MHX_PLAIN2(NPV, PREV)                                                                                                                \
  MHX_LAUNCH(c, "s1_digit_hist", (double)s.n_bases / 4,                                                                                      \
             hipLaunchKernelGGL((k_s1_digit_hist_plain<ITH, NPV, PREV>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, per, \
                                n_items, (int)k, hd, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), keep))
#define MHX_PLAIN(NPV) MHX_PLAIN2(NPV, false)
          // s1_digit_hist_preload: window words requested up front (needs at least 8 slots per read)
#define MHX_ROLL(NPV)                                                                                                                       \
  MHX_LAUNCH(c, "s1_digit_hist", (double)s.n_bases / 4,                                                                                      \
             hipLaunchKernelGGL((k_s1_digit_hist_roll<ITH, NPV>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, per,  \
                                n_items, (int)k, hd, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), keep,         \
                                (const uint64_t *)nullptr, ns))
#define MHX_ROLL_VAR(NPV)                                                                                                                   \
  MHX_LAUNCH(c, "s1_digit_hist", (double)s.n_bases / 4 + (double)ns * 8,                                                                     \
             hipLaunchKernelGGL((k_s1_digit_hist_roll<ITH, NPV, true>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), 0u, per,     \
                                n_slots, (int)k, hd, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per), keep,         \
                                s.start.as<uint64_t>(), ns))
          // s1_digit_hist_roll: one window + one reverse complement per run of a thread's eight items (k <= 23, >= 8 slots per read)
          const bool hroll = plain && per >= 8 && (int)k <= kS1RollMaxK && c->opt("s1_digit_hist_roll", 1) != 0;
          if (var_ok && !(hroll && plain)) throw Error("s1_extract: the variable-length generating pass met a plan it cannot count");
          if (var_ok && specs.n == 1) MHX_ROLL_VAR(1);
          else if (var_ok && specs.n == 2) MHX_ROLL_VAR(2);
          else if (var_ok && specs.n == 3) MHX_ROLL_VAR(3);
          else if (var_ok && specs.n == 4) MHX_ROLL_VAR(4);
#undef MHX_ROLL_VAR
          else if (hroll && specs.n == 1) MHX_ROLL(1);
          else if (hroll && specs.n == 2) MHX_ROLL(2);
          else if (hroll && specs.n == 3) MHX_ROLL(3);
          else if (hroll && specs.n == 4) MHX_ROLL(4);
#undef MHX_ROLL
          else if (plain && specs.n == 2 && per >= 8 && c->opt("s1_digit_hist_preload", 0) != 0) MHX_PLAIN2(2, true);
          else if (plain && specs.n == 1) MHX_PLAIN(1);
          else if (plain && specs.n == 2) MHX_PLAIN(2);
          else if (plain && specs.n == 3) MHX_PLAIN(3);
          else if (plain && specs.n == 4) MHX_PLAIN(4);
#undef MHX_PLAIN
#undef MHX_PLAIN2
          else
            MHX_LAUNCH(c, "s1_digit_hist", (double)s.n_bases / 4,
                       hipLaunchKernelGGL((k_s1_digit_hist<ITH>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, per, n_items,
                                          (int)k, specs, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per)));
        } else {
          if (var_ok) throw Error("s1_extract: the variable-length generating pass needs the blocked digit histogram");
          MHX_FAST(4, false, "s1_digit_hist");
        }
        uint64_t n_records = n_items;  // what the generating pass will leave
        if (filter_in_gen) {  // the kept items = the sum of any one digit histogram
          std::vector<unsigned long long> h0(256);
          MHX_HIP(hipMemcpyAsync(h0.data(), pre_hist, 256 * 8, hipMemcpyDeviceToHost, st));
          MHX_HIP(hipStreamSynchronize(st));
          n_records = 0;
          for (unsigned long long v : h0) n_records += v;
          if (n_records > c->filter_expected) throw Error("bucket filter: more items in the kept buckets than announced");
          buf_a = c->ws("items_a", n_records * item_bytes + 64).as<uint32_t>();
          c->pre_hist_buf = buf_a;
          c->pre_hist_n = n_records;
        }
        // The consumers of this pass (the LDS group-bys behind the remaining passes; compact records, no mercy) count equal
        // keys: they need the records grouped, not in input order — so the first pass may place the records of a digit in
        // any order (the later passes are stable with respect to whatever order it leaves).
        const bool any_order = c->opt("s1_gen_any_order", 1) != 0;
        // s1_gen_blocked: consecutive items per thread (S1GenBlocked) — only where the order inside a digit is free
        const bool blocked = any_order && per >= 8 && c->opt("s1_gen_blocked", 0) != 0;
        const S1GenT<false> g{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, pos_base, pos_bits, nullptr};
        const S1GenT<true> gf{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, pos_base, pos_bits, keep};
        const S1GenBlockedT<false> gb{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                                      (uint32_t)(kSortThreads * 8) % per, nullptr};
        const S1GenBlockedT<true> gbf{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                                      (uint32_t)(kSortThreads * 8) % per, keep};
        // s1_gen_roll: the blocked generator with one window + one reverse complement per run of a thread's items (k <= 23)
        const bool roll = blocked && (int)k <= kS1RollMaxK && c->opt("s1_gen_roll", 1) != 0;
        if (var_ok && !roll) throw Error("s1_extract: the variable-length generating pass needs s1_gen_blocked and s1_gen_roll");
        c->s1_var_gen = var_ok;
        const S1GenVarT<false> gv{s.words.as<uint32_t>(), s.start.as<uint64_t>(), ns, per, (int)k, pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                                  (uint32_t)(kSortThreads * 8) % per, nullptr};
        const S1GenVarT<true> gvf{s.words.as<uint32_t>(), s.start.as<uint64_t>(), ns, per, (int)k, pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                                  (uint32_t)(kSortThreads * 8) % per, keep};
        const S1GenRollT<false> gr{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                                   (uint32_t)(kSortThreads * 8) % per, nullptr};
        const S1GenRollT<true> grf{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                                   (uint32_t)(kSortThreads * 8) % per, keep};
        c->gen_first_pass = [g, gf, gb, gbf, gr, grf, gv, gvf, var_ok, roll, any_order, blocked, filter_in_gen](const OnesweepLaunch &l) {
#define MHX_GEN(KERNEL, SRCT, RANKV, SRCV)                                                                                              \
  hipLaunchKernelGGL((KERNEL<3, 8, 3, SRCT, RANKV>), dim3(l.grid), dim3(kSortThreads), 0, l.stream, SRCV, l.out, l.n, l.ds, l.nbits, l.bin_start, \
                     l.status, l.ticket, l.err, l.tag, l.xcd_units)
#define MHX_GEN_U(SRCT, RANKV, SRCV)                                                                                                    \
  hipLaunchKernelGGL((k_radix_onesweep_u<3, 8, 3, SRCT, RANKV, 0>), dim3(l.grid), dim3(kSortThreads), 0, l.stream, SRCV, l.out, l.n, l.ds, l.nbits, \
                     l.bin_start, l.status, l.ticket, l.err, l.tag, l.xcd_units)
          if (var_ok) {  // reads of any length: item slots padded to the longest read's, the slots a read does not fill declined
            if (!(l.unit_runs && l.wi == 0 && any_order)) throw Error("s1: the variable-length generator needs the unit-wide pass on a first-word digit");
            if (filter_in_gen) MHX_GEN_U(S1GenVarT<true>, 1, gvf);
            else MHX_GEN_U(S1GenVarT<false>, 1, gv);
          } else if (filter_in_gen) {  // (s1_filter_in_gen_applies vouched for unit-wide runs, digits in the first key word, any order)
            if (!(l.unit_runs && l.wi == 0 && any_order)) throw Error("s1: the filtering generator needs the unit-wide pass on a first-word digit");
            if (roll) MHX_GEN_U(S1GenRollT<true>, 1, grf);
            else if (blocked) MHX_GEN_U(S1GenBlockedT<true>, 1, gbf);
            else MHX_GEN_U(S1GenT<true>, 1, gf);
          } else if (l.unit_runs && l.wi == 0 && roll) MHX_GEN_U(S1GenRollT<false>, 1, gr);
          else if (l.unit_runs && l.wi == 0 && blocked) MHX_GEN_U(S1GenBlockedT<false>, 1, gb);
          else if (l.unit_runs && l.wi == 0 && any_order) MHX_GEN_U(S1GenT<false>, 1, g);  // (the digits of this plan lie in the first key word)
          else if (l.unit_runs && l.wi == 0) MHX_GEN_U(S1GenT<false>, 0, g);
          else if (any_order) MHX_GEN(k_radix_onesweep, S1GenT<false>, true, g);
          else MHX_GEN(k_radix_onesweep, S1GenT<false>, false, g);
#undef MHX_GEN
#undef MHX_GEN_U
        };
        c->gen_buf = buf_a;
        c->gen_n = n_records;
        c->gen_slots = n_slots;
        if (n_records == 0) c->gen_first_pass = nullptr;  // (a pass or rank that keeps no record: no sort will come and consume it)
        n_items = n_records;
      } else if (it >= 8) MHX_FAST(8, true, "s1_extract");
      else if (it >= 4) MHX_FAST(4, true, "s1_extract");
      else if (it >= 2) MHX_FAST(2, true, "s1_extract");
      else MHX_FAST(1, true, "s1_extract");
#undef MHX_FAST
      c->s1_defer_items = false;
    } else {
      if (ns && shape_fast) {  // (the per-read table was skipped above: the general kernels want it)
        MHX_LAUNCH(c, "item_counts", (double)ns * 12,
                   hipLaunchKernelGGL(k_s1_item_counts, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, s.start.as<uint64_t>(), ns, k, cnt));
        exclusive_scan_u32_u64(c, cnt, item_start, ns, item_start + ns + 1);
      }
#define MHX_S1X(SV, CP)                                                                                                      \
  do {                                                                                                                       \
    if (fixed) {                                                                                                             \
      MHX_LAUNCH(c, "s1_extract", (double)n_items * item_bytes + (double)s.n_bases / 4,                                      \
                 hipLaunchKernelGGL((k_s1_extract_fixed<KW, SV, CP>), dim3((unsigned)std::min<uint64_t>(div_ceil(n_items, 256), 256 * 16)), \
                                    dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, s.fixed_len - k + 4, n_items, (int)k, \
                                    pos_base, pos_bits, buf_a, specs, pre_hist));                                               \
    } else                                                                                                                   \
      MHX_LAUNCH(c, "s1_extract", (double)n_items * item_bytes + (double)s.n_bases / 4,                                      \
                 hipLaunchKernelGGL((k_s1_extract<KW, SV, CP>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(),        \
                                    s.start.as<uint64_t>(), item_start, ns, (int)k, pos_base, pos_bits, buf_a));              \
  } while (0)
    MHX_DISPATCH_KW(KWv, {
      if (compact) {
        if (S == KW + 1) MHX_S1X(KW + 1, true);
        else MHX_S1X(KW + 2, true);
      } else {
        if (S == KW + 2) MHX_S1X(KW + 2, false);
        else MHX_S1X(KW + 3, false);
      }
    });
#undef MHX_S1X
    }
  }
  c->s1_defer_items = false;
  return n_items;
}

// ---- the front of `count` on the bucket-streaming design (KmerCounter::Lv1FillOffsets + Lv2ExtractSubString, kmer_counter.cpp:158-252) ----
bool count_stream_front(mhx_ctx *c, uint32_t k, const S1Plan &plan, uint32_t **buf_a_out, uint32_t **buf_b_out, uint64_t *n_items_out) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const bool var = s.fixed_len == 0;
  const uint32_t per = (var ? s.max_len : s.fixed_len) - k;  // item slots per read
  const uint64_t n_slots = s.n_seqs * (uint64_t)per;
  const int KWv = 2;
  // a pass of the memory plan: the lv1-bucket filter sits inside the histogram pre-pass and the generating pass (where the reference's
  // OffsetFiller::IsHandling sits, base_engine.h:106-108) — one scan of the reads per pass, only the kept records ever written
  const uint32_t *keep = c->filter_on ? c->work["filter_bits"].as<uint32_t>() : nullptr;
  const bool wide = (int)k > kCountStreamMaxK;  // k = 23..27: a window per item
  if (wide && var) throw Error("count: the wide generator serves reads of one length");
  // digit histograms of the plan's passes (the chained scan wants every pass's bin starts beforehand)
  HiDigits hd;
  hd.n = (int)plan.passes.size();
  for (int p = 0; p < hd.n; ++p) {
    const DigitSpec d = spec_of_pass(plan.passes[p], KWv);
    hd.sh[p] = d.bit1;
    hd.mk[p] = d.mask1;
  }
  unsigned long long *pre_hist = c->ws("sort_pre_hist", (size_t)kMaxFusedPasses * 256 * 8).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(pre_hist, 0, (size_t)hd.n * 256 * 8, st));
  {
    constexpr int ITH = 8;
    const unsigned fgrid = (unsigned)std::min<uint64_t>(div_ceil(n_slots, 256 * ITH), 256 * 8);
    const uint64_t stride_items = (uint64_t)fgrid * 256 * ITH;
#define MHX_CH(NPV, VARV)                                                                                                                   \
  MHX_LAUNCH(c, "count_digit_hist", (double)s.n_bases / 4,                                                                                  \
             hipLaunchKernelGGL((k_count_digit_hist_roll<ITH, NPV, VARV>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, per, \
                                n_slots, (int)k, hd, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per),              \
                                s.start.as<uint64_t>(), s.n_seqs, keep))
#define MHX_CHW(NPV)                                                                                                                        \
  MHX_LAUNCH(c, "count_digit_hist", (double)s.n_bases / 4,                                                                                  \
             hipLaunchKernelGGL((k_count_digit_hist_roll<ITH, NPV, false, true>), dim3(fgrid), dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, per, \
                                n_slots, (int)k, hd, pre_hist, (uint32_t)(stride_items / per), (uint32_t)(stride_items % per),              \
                                s.start.as<uint64_t>(), s.n_seqs, keep))
#define MHX_CH2(NPV)             \
  do {                           \
    if (wide) MHX_CHW(NPV);      \
    else if (var) MHX_CH(NPV, true);  \
    else MHX_CH(NPV, false);     \
  } while (0)
    if (hd.n == 1) MHX_CH2(1);
    else if (hd.n == 2) MHX_CH2(2);
    else if (hd.n == 3) MHX_CH2(3);
    else MHX_CH2(4);
#undef MHX_CH2
#undef MHX_CHW
#undef MHX_CH
  }
  uint64_t n_items = n_slots;  // the records
  if (var || keep) {  // = the sum of any one digit histogram
    std::vector<unsigned long long> h0(256);
    MHX_HIP(hipMemcpyAsync(h0.data(), pre_hist, 256 * 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    n_items = 0;
    for (unsigned long long v : h0) n_items += v;
    if (keep && n_items > c->filter_expected) throw Error("bucket filter: more items in the kept buckets than announced");
    if (n_items == 0) return false;  // (no read holds an edge / no edge in the kept buckets: the general path knows what to publish)
  }
  uint32_t *buf_a = c->ws("items_a", n_items * 12 + 64).as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", n_items * 12 + 64).as<uint32_t>();
  *buf_a_out = buf_a;
  *buf_b_out = buf_b;
  *n_items_out = n_items;
  c->pre_hist_sig = passes_signature(plan.passes);
  c->pre_hist_buf = buf_a;
  c->pre_hist_n = n_items;
  c->pre_hist_passes = hd.n;
  // (edges only — stage 2's aggregated items from a count — nobody reads the positions: no tag bits either, which at k >= 23 would sit
  //  on bits of the (k+1)-mer; count_stream_shape declines tagged read sets there in every other case)
  const uint32_t pos_bits = c->count_edges_only ? 63u : s1_pos_bits(c);
  const uint32_t tq = (uint32_t)(kSortThreads * 8) / per, tr = (uint32_t)(kSortThreads * 8) % per;
  const CountGenT<false> g{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, c->pos_base, pos_bits, tq, tr, nullptr};
  const CountGenT<true> gf{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, c->pos_base, pos_bits, tq, tr, keep};
  const CountGenVarT<false> gv{s.words.as<uint32_t>(), s.start.as<uint64_t>(), s.n_seqs, per, (int)k, c->pos_base, pos_bits, tq, tr, nullptr};
  const CountGenVarT<true> gvf{s.words.as<uint32_t>(), s.start.as<uint64_t>(), s.n_seqs, per, (int)k, c->pos_base, pos_bits, tq, tr, keep};
  const bool filter = keep != nullptr;
  const CountGenWideT<false> gw{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, c->pos_base, pos_bits, tq, tr, nullptr};
  const CountGenWideT<true> gwf{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, c->pos_base, pos_bits, tq, tr, keep};
  c->gen_first_pass = [g, gf, gv, gvf, gw, gwf, var, filter, wide](const OnesweepLaunch &l) {
    if (!(l.unit_runs && l.wi == 0)) throw Error("count: the generating pass needs the unit-wide pass on a first-word digit");
#define MHX_CGEN(SRCT, SRCV)                                                                                                                \
  hipLaunchKernelGGL((k_radix_onesweep_u<3, 8, 3, SRCT, 1, 0>), dim3(l.grid), dim3(kSortThreads), 0, l.stream, SRCV, l.out, l.n, l.ds, l.nbits, \
                     l.bin_start, l.status, l.ticket, l.err, l.tag, l.xcd_units)
    if (wide && filter) MHX_CGEN(CountGenWideT<true>, gwf);
    else if (wide) MHX_CGEN(CountGenWideT<false>, gw);
    else if (var && filter) MHX_CGEN(CountGenVarT<true>, gvf);
    else if (var) MHX_CGEN(CountGenVarT<false>, gv);
    else if (filter) MHX_CGEN(CountGenT<true>, gf);
    else MHX_CGEN(CountGenT<false>, g);
#undef MHX_CGEN
  };
  c->gen_buf = buf_a;
  c->gen_n = n_items;
  c->gen_slots = n_slots;
  return true;
}

}  // namespace mhx
