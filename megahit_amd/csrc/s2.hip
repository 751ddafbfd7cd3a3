// read2sdbg stage 2 and the SdBG emission shared with seq2sdbg.
// Replaces Read2SdbgS2 (reference src/sorting/read_to_sdbg_s2.cpp:271-630) and the
// Lv2Postprocess of SeqToSdbg (src/sorting/seq_to_sdbg.cpp:702-789) + SdbgWriter
// (src/sdbg/sdbg_writer.cpp:25-58) on the GPU.
//
//   s2_count / s2_extract   <= 6 items per solid (k+1)-mer occurrence (Lv1FillOffsets :347-440 +
//                            Lv2ExtractSubString :442-519), item slots from a per-read scan
//   sort                    by k-mer chars, then the "full" flag, then W           (sort.hip)
//   groups                  heads of equal-(k-1)-mer groups                         (scan.hip)
//   sdbg_count / sdbg_emit  per group: W, last, tip, multiplicity -> byte-exact records at offsets
//                           from exclusive scans of (records, tips, large multiplicities)
//   bucket_stats            per-bucket starting offset / item / tip / large counts (sdbg_meta.h:22-34)
#include "dev_prims.h"
#include "mhx_internal.h"
#include "tile_groups.h"

namespace mhx {

// ---------------------------------------------------------------------------
// S2 item generation
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool bit_at(const unsigned long long *__restrict__ bits, uint64_t i) {
  return (bits[i >> 6] >> (i & 63)) & 1ull;
}

// number of items the (k+1)-mer occurrence at read offset p emits (0 if not solid); *mask gets
// bit0 = left-$ pair, bit1 = right-$ pair, bit2 = palindrome
template <int KW>
__device__ __forceinline__ unsigned s2_items_at(const uint32_t *__restrict__ seq, const unsigned long long *__restrict__ solid, bool sure,
                                                uint64_t st, uint32_t L, uint32_t p, int k, unsigned *mask) {
  const uint64_t fo = st + p;
  if (!(sure || bit_at(solid, fo))) return 0;
  uint32_t e[KW], rc[KW];
  load_chars<KW>(seq, fo, k + 1, e);
  rc_chars<KW>(e, k + 1, rc);
  const bool pal = cmp_words<KW>(e, rc) == 0;
  const bool left = p == 0 || !(sure || bit_at(solid, fo - 1));          // :385-386
  const bool right = p + k + 1 == L || !(sure || bit_at(solid, fo + 1)); // :411-412
  *mask = (left ? 1u : 0u) | (right ? 2u : 0u) | (pal ? 4u : 0u);
  return (1u + left + right) * (pal ? 1u : 2u);
}

// lv1 bucket (top 8 chars) of `in` shifted left by c chars
template <int KW>
__device__ __forceinline__ uint32_t shifted_bucket(const uint32_t (&in)[KW], int c) {
  uint64_t hi = (uint64_t)in[0] << 32;
  if constexpr (KW > 1) hi |= in[1];
  return (uint32_t)((hi << (2 * c)) >> 48);
}
// Bucket-range passes (mhx_set_bucket_filter): which of the items of an occurrence fall into kept buckets, as a bit mask in
// the order s2_write_items emits them (left fwd, left rc, solid fwd, solid rc, right fwd, right rc; bits of absent items 0).
// The filter is applied HERE, while the items are counted and written, instead of materialising every item of every read in
// batches and splitting kept from dropped afterwards (the reference's OffsetFiller::IsHandling test, base_engine.h:106-108,
// sits in the same place: Lv1FillOffsets).
template <int KW>
__device__ __forceinline__ unsigned s2_kept_mask(const uint32_t (&e)[KW], const uint32_t (&rc)[KW], unsigned mask, const uint8_t *__restrict__ drop) {
  const bool pal = mask & 4u;
  unsigned keep = 0;
  auto k = [&](const uint32_t (&w)[KW], int c) -> unsigned { return drop[shifted_bucket<KW>(w, c)] ? 0u : 1u; };
  if (mask & 1u) {
    keep |= k(e, 0) << 0;
    if (!pal) keep |= k(rc, 2) << 1;
  }
  keep |= k(e, 1) << 2;
  if (!pal) keep |= k(rc, 1) << 3;
  if (mask & 2u) {
    keep |= k(e, 2) << 4;
    if (!pal) keep |= k(rc, 0) << 5;
  }
  return keep;
}

template <int KW>
__global__ __launch_bounds__(256) void k_s2_count(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint64_t n_seqs,
                                                  int k, const unsigned long long *__restrict__ solid, int sure,
                                                  uint32_t *__restrict__ cnt, const uint8_t *__restrict__ drop) {
  const int lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const uint64_t n_waves = (uint64_t)gridDim.x * blockDim.x / kWave;
  for (uint64_t r = wave; r < n_seqs; r += n_waves) {
    const uint64_t st = start[r];
    const uint32_t L = (uint32_t)(start[r + 1] - st);
    uint32_t total = 0;
    if (L >= (uint32_t)k + 1) {
      for (uint32_t p0 = 0; p0 < L - k; p0 += kWave) {
        const uint32_t p = p0 + lane;
        unsigned mask, c = 0;
        if (p < L - k) c = s2_items_at<KW>(seq, solid, sure != 0, st, L, p, k, &mask);
        if (drop && c) {  // only the items of the kept buckets
          uint32_t e[KW], rc[KW];
          load_chars<KW>(seq, st + p, k + 1, e);
          rc_chars<KW>(e, k + 1, rc);
          c = (unsigned)__builtin_popcount(s2_kept_mask<KW>(e, rc, mask, drop));
        }
        total += wave_sum<uint32_t>(c);
      }
    }
    if (lane == 0) cnt[r] = total;
  }
}

// in << c chars (c = 1 or 2), then keep n chars
template <int KW>
__device__ __forceinline__ void shl_mask_chars(const uint32_t (&in)[KW], int c, int n, uint32_t (&out)[KW]) {
  const unsigned sh = 2u * c;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const uint32_t nxt = i + 1 < KW ? in[i + 1] : 0u;
    out[i] = c ? ((in[i] << sh) | (nxt >> (32 - sh))) : in[i];
  }
  const int full = n >> 4, rem = n & 15;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    if (i > full) out[i] = 0;
    else if (i == full) out[i] = rem ? (out[i] & (0xFFFFFFFFu << (32 - 2 * rem))) : 0u;
  }
}
template <int KW, int S>
__device__ __forceinline__ void s2_store(const uint32_t (&key)[KW], uint32_t low, uint32_t *__restrict__ dst) {
  uint32_t out[S];
#pragma unroll
  for (int i = 0; i < KW; ++i) out[i] = key[i];
  out[KW - 1] |= low;
  if constexpr (S > KW) out[KW] = 0;
  if constexpr (S % 4 == 0) {
#pragma unroll
    for (int i = 0; i < S / 4; ++i)
      reinterpret_cast<uint4 *>(dst)[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < S / 2; ++i) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(out[2 * i], out[2 * i + 1]);
  }
}

// All (<= 6) items of the solid (k+1)-mer occurrence e = r[p..p+k] (Lv2ExtractSubString,
// read_to_sdbg_s2.cpp:442-519; edge types of EncodeOffset :36-41) derived from e and its reverse
// complement rc held in registers: every item is a 1- or 2-char shift of one of the two.
//   left-$  fwd: e[0..k-1] W=$        rc: rc[2..k] (k-1 chars) W=rc[1]
//   solid   fwd: e[1..k]   W=e[0]     rc: rc[1..k]             W=rc[0]
//   right-$ fwd: e[2..k] (k-1) W=e[1] rc: rc[0..k-1]           W=$
template <int KW, int S>
__device__ __forceinline__ void s2_write_items(const uint32_t (&e)[KW], const uint32_t (&rc)[KW], int k, unsigned mask,
                                               uint32_t *__restrict__ dst, unsigned keep = 0x3Fu) {
  const bool pal = mask & 4u;
  const uint32_t e0 = e[0] >> 30, e1 = (e[0] >> 28) & 3u, r0 = rc[0] >> 30, r1 = (rc[0] >> 28) & 3u;
  uint32_t t[KW];
  if (mask & 1u) {
    if (keep & 1u) {
      shl_mask_chars<KW>(e, 0, k, t);
      s2_store<KW, S>(t, 8u | kSentinel, dst); dst += S;
    }
    if (!pal && (keep & 2u)) {
      shl_mask_chars<KW>(rc, 2, k - 1, t);
      s2_store<KW, S>(t, r1, dst); dst += S;
    }
  }
  if (keep & 4u) {
    shl_mask_chars<KW>(e, 1, k, t);
    s2_store<KW, S>(t, 8u | e0, dst); dst += S;
  }
  if (!pal && (keep & 8u)) {
    shl_mask_chars<KW>(rc, 1, k, t);
    s2_store<KW, S>(t, 8u | r0, dst); dst += S;
  }
  if (mask & 2u) {
    if (keep & 16u) {
      shl_mask_chars<KW>(e, 2, k - 1, t);
      s2_store<KW, S>(t, e1, dst); dst += S;
    }
    if (!pal && (keep & 32u)) {
      shl_mask_chars<KW>(rc, 0, k, t);
      s2_store<KW, S>(t, 8u | kSentinel, dst); dst += S;
    }
  }
}

template <int KW, int S>
__global__ __launch_bounds__(256) void k_s2_extract(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start,
                                                    const uint64_t *__restrict__ item_start, uint64_t n_seqs, int k,
                                                    const unsigned long long *__restrict__ solid, int sure,
                                                    uint32_t *__restrict__ items, const uint8_t *__restrict__ drop) {
  const int lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const uint64_t n_waves = (uint64_t)gridDim.x * blockDim.x / kWave;
  for (uint64_t r = wave; r < n_seqs; r += n_waves) {
    const uint64_t st = start[r];
    const uint32_t L = (uint32_t)(start[r + 1] - st);
    if (L < (uint32_t)k + 1) continue;
    uint64_t carry = item_start[r];
    for (uint32_t p0 = 0; p0 < L - k; p0 += kWave) {
      const uint32_t p = p0 + lane;
      unsigned mask = 0, c = 0, keep = 0x3Fu;
      uint32_t e[KW], rc[KW];
      if (p < L - k) {
        const uint64_t fo = st + p;
        if (sure || bit_at(solid, fo)) {
          load_chars<KW>(seq, fo, k + 1, e);
          rc_chars<KW>(e, k + 1, rc);
          const bool pal = cmp_words<KW>(e, rc) == 0;
          const bool left = p == 0 || !(sure || bit_at(solid, fo - 1));          // :385-386
          const bool right = p + k + 1 == L || !(sure || bit_at(solid, fo + 1)); // :411-412
          mask = (left ? 1u : 0u) | (right ? 2u : 0u) | (pal ? 4u : 0u);
          c = (1u + left + right) * (pal ? 1u : 2u);
          if (drop) {
            keep = s2_kept_mask<KW>(e, rc, mask, drop);
            c = (unsigned)__builtin_popcount(keep);
          }
        }
      }
      const uint32_t inc = wave_inclusive_sum<uint32_t>(c);
      const uint32_t tot = __shfl(inc, kWave - 1, kWave);
      if (c) s2_write_items<KW, S>(e, rc, k, mask, items + (carry + inc - c) * S, keep);
      carry += tot;
    }
  }
}

// ---------------------------------------------------------------------------
// Aggregated stage 2 (k <= 22, m >= 2): the "solid" items come from stage 1 with their multiplicity
// (s1.hip, S1Op<AGG>); only the $-dummy items (types 0 and 2 of EncodeOffset) still have to be produced per
// occurrence.  Same seq2sdbg-style layout, count field = 1.  64-bit arithmetic: k+1 <= 23 chars.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rc64_s2(uint64_t x, int n) {
  uint64_t r = __builtin_bitreverse64(x);
  r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
  return (~r) << (64 - 2 * n);
}
// dummies of the occurrence at read offset p: bit0 left pair, bit1 right pair, bit2 palindrome; returns their number
__device__ __forceinline__ unsigned s2d_items_at(const uint32_t *__restrict__ seq, const unsigned long long *__restrict__ solid, uint64_t st,
                                                 uint32_t L, uint32_t p, int k, unsigned *mask, uint64_t *e_out) {
  const uint64_t fo = st + p;
  if (!bit_at(solid, fo)) return 0;
  const bool left = p == 0 || !bit_at(solid, fo - 1);
  const bool right = p + k + 1 == L || !bit_at(solid, fo + 1);
  if (!left && !right) return 0;
  uint32_t w[2];
  load_chars<2>(seq, fo, k + 1, w);
  const uint64_t e = ((uint64_t)w[0] << 32) | w[1];
  const bool pal = e == rc64_s2(e, k + 1);
  *e_out = e;
  *mask = (left ? 1u : 0u) | (right ? 2u : 0u) | (pal ? 4u : 0u);
  return ((left ? 1u : 0u) + (right ? 1u : 0u)) * (pal ? 1u : 2u);
}

// The dummies sit at the ends of the runs of solid positions.  Stage 1 only sets bits where a (k+1)-mer starts, so the
// bit before a read's first position and the bit after its last (k+1)-mer are 0 and the read-boundary cases of
// :385-386 / :411-412 are run ends as well: one thread per 64-bit bitmap word finds them with shifts.
__device__ __forceinline__ void s2d_run_ends(const unsigned long long *__restrict__ solid, uint64_t i, uint64_t n_words,
                                             unsigned long long *l, unsigned long long *r) {
  const unsigned long long w = solid[i];
  const unsigned long long prev = i ? solid[i - 1] >> 63 : 0ull, next = i + 1 < n_words ? solid[i + 1] & 1ull : 0ull;
  *l = w & ~((w << 1) | prev);
  *r = w & ~((w >> 1) | (next << 63));
}
// upper bound of the number of dummy items (exact unless palindromes occur)
// (grid-stride: one atomic on the shared total per workgroup, not per 256 words — 9 x 10^4 same-address atomics were the
// whole millisecond this kernel took for a 190 MB bitmap)
__global__ __launch_bounds__(256) void k_s2d_bound(const unsigned long long *__restrict__ solid, uint64_t n_words,
                                                   unsigned long long *__restrict__ total) {
  __shared__ uint64_t sm[256 / kWave + 1];
  uint64_t c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long l, r;
    s2d_run_ends(solid, i, n_words, &l, &r);
    c += 2ull * (__builtin_popcountll(l) + __builtin_popcountll(r));
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>(c, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(total, (unsigned long long)tot);
}
constexpr int kS2dWords = 8;
// emission in arbitrary order (the items are sorted next): per block one atomicAdd on the output cursor
__global__ __launch_bounds__(256) void k_s2d_emit(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint64_t n_seqs,
                                                  uint32_t fixed_len, int k, const unsigned long long *__restrict__ solid, uint64_t n_words,
                                                  uint2 *__restrict__ items, unsigned long long *__restrict__ cursor, int fsh = 19, int bsh = 16, int csh = 0) {
  // (fsh / bsh / csh: where the "full" flag, the W char and the count sit below the chars: 19 / 16 / 0 in the seq2sdbg layout of the
  //  aggregated items of k <= 22, cb + 3 / cb / 0 with the cb = 60 - 2k count bits of k = 23..28)
  __shared__ uint64_t sm[256 / kWave + 1];
  __shared__ unsigned long long s_base;
  const uint64_t mask_k = ~0ull << (64 - 2 * k), mask_k1 = ~0ull << (64 - 2 * (k - 1));
  // kS2dWords bitmap words per thread and cursor bump: 16 x fewer same-address atomics than one per 256 words
  for (uint64_t blk = blockIdx.x; blk * (256ull * kS2dWords) < n_words; blk += gridDim.x) {
  uint32_t c = 0;
  unsigned long long ls[kS2dWords], rs[kS2dWords];
#pragma unroll
  for (int q = 0; q < kS2dWords; ++q) {
    const uint64_t i = blk * (256ull * kS2dWords) + (uint64_t)q * 256 + threadIdx.x;
    ls[q] = rs[q] = 0;
    if (i < n_words) s2d_run_ends(solid, i, n_words, &ls[q], &rs[q]);
  }
  // pass 1: number of items of these words (palindromes emit the forward item only)
#pragma unroll
  for (int q = 0; q < kS2dWords; ++q) {
    const uint64_t i = blk * (256ull * kS2dWords) + (uint64_t)q * 256 + threadIdx.x;
    const unsigned long long l = ls[q], r = rs[q];
    for (unsigned long long cand = l | r; cand; cand &= cand - 1) {
      const int b = __builtin_ctzll(cand);
      const uint64_t fo = i * 64 + b;
      uint32_t w[2];
      load_chars<2>(seq, fo, k + 1, w);
      const uint64_t e = ((uint64_t)w[0] << 32) | w[1];
      const unsigned ends = (unsigned)((l >> b) & 1ull) + (unsigned)((r >> b) & 1ull);
      c += ends * (e == rc64_s2(e, k + 1) ? 1u : 2u);
    }
  }
  uint64_t tot;
  const uint64_t excl = block_exclusive_sum<uint64_t, 256>((uint64_t)c, sm, &tot);
  if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
  __syncthreads();
  uint2 *dst = items + s_base + excl;
#pragma unroll
  for (int q = 0; q < kS2dWords; ++q) {
    const uint64_t i = blk * (256ull * kS2dWords) + (uint64_t)q * 256 + threadIdx.x;
    const unsigned long long l = ls[q], r = rs[q];
  for (unsigned long long cand = l | r; cand; cand &= cand - 1) {
    const int b = __builtin_ctzll(cand);
    const uint64_t fo = i * 64 + b;
    uint32_t w[2];
    load_chars<2>(seq, fo, k + 1, w);
    const uint64_t e = ((uint64_t)w[0] << 32) | w[1];
    const uint64_t er = rc64_s2(e, k + 1);
    const bool pal = e == er;
    auto put = [&](uint64_t key, uint64_t full, uint64_t wc) {
      const uint64_t v = key | (full << fsh) | (wc << bsh) | (1ull << csh);
      *dst++ = make_uint2((uint32_t)(v >> 32), (uint32_t)v);
    };
    if ((l >> b) & 1ull) {                                               // left-$ (read_to_sdbg_s2.cpp:387-396, :457-512)
      put(e & mask_k, 1, kSentinel);                                     //   fwd: e[0..k-1], W = $
      if (!pal) put((er << 4) & mask_k1, 0, (er >> 60) & 3u);            //   rc : e'[2..k],  W = e'[1]
    }
    if ((r >> b) & 1ull) {                                               // right-$ (:411-425)
      put((e << 4) & mask_k1, 0, (e >> 60) & 3u);                        //   fwd: e[2..k],   W = e[1]
      if (!pal) put(er & mask_k, 1, kSentinel);                          //   rc : e'[0..k-1], W = $
    }
  }
  }  // q
  __syncthreads();
  }  // blk
  (void)start;
  (void)n_seqs;
  (void)fixed_len;
}

// ---------------------------------------------------------------------------
// SdBG emission (shared)
// ---------------------------------------------------------------------------
struct SdbgParams {
  int stride, kw, k;
  int bshift, fshift;  // W char / "full k chars" flag position in the last key word
  int aw, ashift;      // word and shift of the k-th char
  int wpt;             // words per tip label
  int is_seq;          // 0: S2 (multiplicity = run length); 1: seq2sdbg (65535 - key field); 2: aggregated S2 (sum of counts)
  int ref_kw;          // key words of the reference's stage-2 item, ceil((2k+4)/32) (tip-label reconstruction)
  int cshift;          // is_seq 2: the count field of an aggregated item, bits [cshift, cshift + popcount(cmask)) of the last key word: the low 16
  uint32_t cmask;      //           bits up to k = 22, the low 60 - 2k bits at k = 23..28 (s2_agg_compact_bits); flag | W sit right above it
};

template <int S>
struct SdbgTile {
  static constexpr int kRaw = 32768 / (S * 4);
  static constexpr int kT = kRaw >= 2048 ? 2048 : (kRaw >= 256 ? (kRaw / 256) * 256 : 256);
};

__device__ __forceinline__ uint32_t *sdbg_wcnt() {
  __shared__ uint32_t w[10];
  return w;
}

template <int S>
__device__ __forceinline__ uint32_t *sdbg_gmask() {  // per group: which (a,b) pairs occur
  __shared__ uint32_t gm[SdbgTile<S>::kT];
  return gm;
}

constexpr unsigned long long kNoStart = ~0ull;

// Lv2Postprocess of Read2SdbgS2 / SeqToSdbg (read_to_sdbg_s2.cpp:537-611, seq_to_sdbg.cpp:718-786) +
// SdbgWriter::Write (sdbg_writer.cpp:25-58) as a tile operator (tile_groups.h).  A run = the records of
// one (k-1)-mer group with the same (a, b) = (k-th char or $, W char); the reference's per-item loops
// only ever look at (a, b), so they are restated over the <= 25 runs of a group.
template <int S>
struct SdbgOp {
  static constexpr bool kItemPhase = false, kItemFinal = false, kRunPhase = true, kUnitIsRun = true, kAtomicBase = false;
  SdbgParams P;
  uint16_t *out16;
  unsigned long long *w_count;
  unsigned long long *bstart;  // [3][65536]: running (records, tips, large) at the first group of each bucket

  __device__ bool item_phase_enabled() const { return false; }
  __device__ bool item_final_enabled() const { return false; }
  __device__ void item_phase(const TileCtx<S> &, uint32_t, uint32_t) const {}
  __device__ void item_final(const TileCtx<S> &, uint32_t, uint32_t) const {}
  __device__ bool same_run(const uint32_t *cur, const uint32_t *prev) const {
    // same "full" flag + W char, and same k-th char (zero padding when the flag is clear)
    return (((cur[P.kw - 1] ^ prev[P.kw - 1]) >> P.bshift) & 0xFu) == 0 && (((cur[P.aw] ^ prev[P.aw]) >> P.ashift) & 3u) == 0;
  }
  __device__ void begin_block() const {
    if (threadIdx.x < 10) sdbg_wcnt()[threadIdx.x] = 0;
    uint32_t *gm = sdbg_gmask<S>();
    for (int i = threadIdx.x; i < SdbgTile<S>::kT; i += blockDim.x) gm[i] = 0;
    __syncthreads();
  }
  __device__ void end_block() const {
    if (out16 && threadIdx.x < 10 && sdbg_wcnt()[threadIdx.x]) atomicAdd(&w_count[threadIdx.x], (unsigned long long)sdbg_wcnt()[threadIdx.x]);
  }
  __device__ __forceinline__ int ex_a(const TileCtx<S> &c, uint32_t i) const {
    return ((c.acc.word(i, P.kw - 1) >> P.fshift) & 1u) ? (int)((c.acc.word(i, P.aw) >> P.ashift) & 3u) : 4;
  }
  __device__ __forceinline__ int ex_b(const TileCtx<S> &c, uint32_t i) const { return (int)((c.acc.word(i, P.kw - 1) >> P.bshift) & 7u); }

  // Everything the reference's two loops decide for a run (a,b) follows from WHICH (a,b) pairs occur
  // in the group: M bit (a5*5 + b), a5 = 0 for '$', a+1 otherwise.
  //   has_solid_a(a) = any b<4 with (a,b);  has_solid_b(b) = any a<4 with (a,b)
  //   ($,b) is dropped iff has_solid_b(b);  (a,$) is dropped iff has_solid_a(a)        (:568-585)
  //   W = b+5 iff an earlier output run has the same b, i.e. some (a'<a, b) exists      (:587-589)
  //   last = 1 iff no later run with the same a survives: b<4 -> no (a,b'>b,b'<4); (a,$) kept -> 1
  __device__ __forceinline__ bool decide(uint32_t M, int a, int b, int &w, int &last, int &is_dollar) const {
    const int a5 = a == 4 ? 0 : a + 1;
    const uint32_t col = (1u << b) * 0x108420u;  // bits (a'+1)*5 + b for a' = 0..3
    is_dollar = 0;
    if (a == 4) {
      if (M & col) return false;
      is_dollar = 1;
    }
    if (b == 4) {
      if ((M >> (a5 * 5)) & 0xFu) return false;
    }
    if (b == 4) w = 0;
    else {
      const uint32_t earlier = a == 4 ? 0u : (M & col & ((1u << (a5 * 5)) - 1u));
      w = earlier ? b + 5 : b + 1;
    }
    if (a == 4) last = 0;
    else if (b == 4) last = 1;
    else last = (((M >> (a5 * 5)) & 0xFu) >> (b + 1)) == 0;
    return true;
  }
  __device__ __forceinline__ uint32_t run_mul(const TileCtx<S> &c, uint32_t r, uint32_t i) const {
    if (P.is_seq == 2) {  // pre-aggregated stage-2 items: multiplicity = sum of the items' counts
      const uint32_t e = i + c.run_len(r);
      uint64_t sum = 0;
      for (uint32_t j = i; j < e && sum < MHX_MAX_MUL; ++j) sum += (c.acc.word(j, P.kw - 1) >> P.cshift) & P.cmask;
      return sum > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : (uint32_t)sum;
    }
    if (P.is_seq) return MHX_MAX_MUL - (c.acc.word(i, P.kw - 1) & 0xFFFFu);  // seq_to_sdbg.cpp:782-785
    const uint32_t n = c.run_len(r);                                         // read_to_sdbg_s2.cpp:579
    return n > (uint32_t)MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : n;
  }
  __device__ void run_phase(const TileCtx<S> &c, uint32_t r, uint32_t g) const {
    const uint32_t i = c.run_start(r);
    const int a = ex_a(c, i), b = ex_b(c, i);
    atomicOr(&sdbg_gmask<S>()[g], 1u << ((a == 4 ? 0 : a + 1) * 5 + b));
  }
  __device__ GroupCounts unit_count(const TileCtx<S> &c, uint32_t r) const {
    GroupCounts gc;
    const uint32_t i = c.run_start(r);
    int w, last, tip;
    if (!decide(sdbg_gmask<S>()[c.rgid[r]], ex_a(c, i), ex_b(c, i), w, last, tip)) return gc;
    gc.c0 = 1;
    gc.c1 = (uint32_t)tip;
    gc.c2 = run_mul(c, r, i) > 254;
    return gc;
  }
  __device__ void unit_emit(const TileCtx<S> &c, uint32_t r, uint64_t o0, uint64_t o1, uint64_t o2) const {
    const uint32_t i = c.run_start(r);
    if (c.run_is_group_head(r)) {
      // first group of its lv1 bucket?  (the previous record has another bucket prefix)
      const uint32_t bk = c.acc.word(i, 0) >> 16;
      bool first = c.acc.base + i == 0;
      if (!first) {
        const uint32_t prev = c.acc.lds[((int)i - 1) * S];  // record -1 is staged as well
        first = (prev >> 16) != bk;
      }
      if (first) {
        bstart[bk] = o0;
        bstart[MHX_NUM_BUCKETS + bk] = o1;
        bstart[2 * MHX_NUM_BUCKETS + bk] = o2;
      }
    }
    int w, last, tip;
    if (!decide(sdbg_gmask<S>()[c.rgid[r]], ex_a(c, i), ex_b(c, i), w, last, tip)) return;
    const uint32_t mul = run_mul(c, r, i);
    uint64_t o16 = o0 + o2 + 2ull * P.wpt * o1;
    // SdbgItem (sdbg_item.h:14-24): byte0 = w | last<<4 | tip<<5, byte1 = min(mul,255)
    out16[o16++] = (uint16_t)(w | (last << 4) | (tip << 5) | ((mul > 255 ? 255u : mul) << 8));
    if (mul > 254) out16[o16++] = (uint16_t)mul;
    if (tip) {
      for (int x = 0; x < P.wpt; ++x) {
        uint32_t v = c.acc.word(i, x);
        if (P.is_seq == 2 && x == P.wpt - 1) {
          // the reference's stage-2 item carries only (full<<3 | W) below the chars (read_to_sdbg_s2.cpp:480-514) and
          // writes its raw words as the tip label (:603-607): rebuild that word from the aggregated layout
          const int tip_chars = P.k - 1, in_word = tip_chars - 16 * x;  // chars of the (k-1)-char tip in this word
          const uint32_t cm = in_word >= 16 ? 0xFFFFFFFFu : (in_word <= 0 ? 0u : 0xFFFFFFFFu << (32 - 2 * in_word));
          v &= cm;
          if (x == P.ref_kw - 1) v |= (uint32_t)ex_b(c, i);  // full flag is 0 for a tip
        }
        out16[o16++] = (uint16_t)(v & 0xFFFFu);
        out16[o16++] = (uint16_t)(v >> 16);
      }
    }
    atomicAdd(&sdbg_wcnt()[w], 1u);
    if (last) atomicAdd(&sdbg_wcnt()[9], 1u);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The same Lv2Postprocess + SdbgWriter for 8-byte records (S = 2: the aggregated stage-2 items of k <= 22, the per-occurrence
// items of stage 2 up to k = 29, seq2sdbg at small k) without the generic tile machinery.  k_tile_groups builds run and
// group lists of a tile in LDS behind a dozen barriers and serial prefix loops — ~50 us per 2048-record tile whatever the
// operator does with them (sdbg_count + sdbg_emit: 4.3 ms for 0.94 GB of records, 0.05-0.08 of the HBM roofline).  Here
// a record is one 64-bit value, "same group" and "same run" are one xor + and each, and every RUN HEAD does its run's work
// alone: it walks its group in the staged window (tile + a halo either side; groups hold a handful of records), ORs the
// (a, b) bits of what it meets, counts / sums its own run on the way, decides, and the three output counters become ballots.
// A run belongs to the tile its head lies in; outputs are ordered by head position, as in k_tile_groups.
// Groups or runs that reach beyond the window (low-complexity sequence: 10^5 identical '$' items of poly-A reads) take
// the way through memory: the runs of a group are enumerated by galloping searches (records are sorted, so "same run as
// record p" holds on a stretch around p) — O(runs x log length) loads instead of a walk —, and a long run's multiplicity
// sum is formed by the whole wavefront, 64 records per step, until it reaches the cap.
// ---------------------------------------------------------------------------------------------------------------
struct SdbgFastP {
  unsigned long long gmask, rmask;  // bits that tell groups apart / runs apart (rmask includes gmask), record = w0 << 32 | w1
  int fsh, ash, bsh;                // "full" flag, k-th char, W char as shifts of the 64-bit record
  int is_seq, wpt, k, ref_kw;
  int csh;                          // is_seq 2: count field of an aggregated item: (record >> csh) & cmk
  uint32_t cmk;
  int halo;                         // records staged either side of the tile (<= kSdbgFastHalo; tests shrink it)
};
constexpr int kSdbgFastHalo = 128;

__device__ __forceinline__ unsigned long long sdbg_ld64(const uint2 *__restrict__ items, long long i) {
  const uint2 r = items[i];
  return ((unsigned long long)r.x << 32) | r.y;
}
__device__ __forceinline__ uint32_t sdbg_pair_bit(unsigned long long v, const SdbgFastP &P, int &a, int &b) {
  a = ((v >> P.fsh) & 1ull) ? (int)((v >> P.ash) & 3ull) : 4;
  b = (int)((v >> P.bsh) & 7ull);
  return 1u << ((a == 4 ? 0 : a + 1) * 5 + b);
}
// last index + 1 of the stretch of records equal to `u` under `mask` that holds record p0
__device__ __noinline__ long long sdbg_gallop_fwd(const uint2 *__restrict__ items, long long n, long long p0, unsigned long long u, unsigned long long mask) {
  long long good = p0, bad, step = 1;
  for (;;) {
    const long long t = good + step;
    if (t >= n) {
      bad = n;
      break;
    }
    if (((sdbg_ld64(items, t) ^ u) & mask) == 0) {
      good = t;
      step <<= 1;
    } else {
      bad = t;
      break;
    }
  }
  while (bad - good > 1) {
    const long long mid = good + (bad - good) / 2;
    if (((sdbg_ld64(items, mid) ^ u) & mask) == 0) good = mid;
    else bad = mid;
  }
  return bad;
}
// first index of that stretch
__device__ __noinline__ long long sdbg_gallop_bwd(const uint2 *__restrict__ items, long long p0, unsigned long long u, unsigned long long mask) {
  long long good = p0, bad, step = 1;
  for (;;) {
    const long long t = good - step;
    if (t < 0) {
      bad = -1;
      break;
    }
    if (((sdbg_ld64(items, t) ^ u) & mask) == 0) {
      good = t;
      step <<= 1;
    } else {
      bad = t;
      break;
    }
  }
  while (good - bad > 1) {
    const long long mid = bad + (good - bad) / 2;
    if (((sdbg_ld64(items, mid) ^ u) & mask) == 0) good = mid;
    else bad = mid;
  }
  return good;
}
// the (a, b) pairs of the group of the run head at `head`, and where its run ends — from memory, run by run
__device__ __noinline__ void sdbg_group_far(const uint2 *__restrict__ items, long long n, long long head, unsigned long long v, const SdbgFastP &P,
                                            uint32_t &M, long long &run_end) {
  int a, b;
  M = sdbg_pair_bit(v, P, a, b);
  long long p = head;
  while (p > 0) {
    const unsigned long long u = sdbg_ld64(items, p - 1);
    if ((u ^ v) & P.gmask) break;
    M |= sdbg_pair_bit(u, P, a, b);
    p = sdbg_gallop_bwd(items, p - 1, u, P.rmask);
  }
  long long e = sdbg_gallop_fwd(items, n, head, v, P.rmask);
  run_end = e;
  while (e < n) {
    const unsigned long long u = sdbg_ld64(items, e);
    if ((u ^ v) & P.gmask) break;
    M |= sdbg_pair_bit(u, P, a, b);
    e = sdbg_gallop_fwd(items, n, e, u, P.rmask);
  }
}
// SdbgOp::decide on plain values (read_to_sdbg_s2.cpp:568-589)
__device__ __forceinline__ bool sdbg_decide(uint32_t M, int a, int b, int &w, int &last, int &is_dollar) {
  const int a5 = a == 4 ? 0 : a + 1;
  const uint32_t col = (1u << b) * 0x108420u;  // bits (a'+1)*5 + b for a' = 0..3
  is_dollar = 0;
  if (a == 4) {
    if (M & col) return false;
    is_dollar = 1;
  }
  if (b == 4) {
    if ((M >> (a5 * 5)) & 0xFu) return false;
  }
  if (b == 4) w = 0;
  else {
    const uint32_t earlier = a == 4 ? 0u : (M & col & ((1u << (a5 * 5)) - 1u));
    w = earlier ? b + 5 : b + 1;
  }
  if (a == 4) last = 0;
  else if (b == 4) last = 1;
  else last = (((M >> (a5 * 5)) & 0xFu) >> (b + 1)) == 0;
  return true;
}

template <bool EMIT, int T>  // T records per workgroup (256 threads)
__global__ __launch_bounds__(256) void k_sdbg_fast(const uint2 *__restrict__ items, long long n, SdbgFastP P, uint64_t *__restrict__ tile_tot,
                                                   const uint64_t *__restrict__ tile_base, uint64_t n_tiles, uint16_t *__restrict__ out16,
                                                   unsigned long long *__restrict__ w_count, unsigned long long *__restrict__ bstart,
                                                   uint32_t *__restrict__ res_out) {
  // res_out (counting launch): what every run head found out is kept, 4 bytes per record, and the emitting launch is
  // k_sdbg_emit_res — no second walk over the groups
  constexpr int PER = T / 256, H = kSdbgFastHalo, NW = 256 / kWave;
  __shared__ unsigned long long win[T + 2 * H];  // win[H + i] = record base + i
  // what a run head found out, per record of the tile: flags | W << 4 | multiplicity << 12 (0 = not a run head).  Kept in LDS,
  // not in registers: the loops over a thread's records stay rolled and the kernel small (eight waves per SIMD instead of three
  // — every phase here is a chain of dependent LDS reads)
  __shared__ uint32_t res[T];
  __shared__ uint32_t cell[3][PER * NW + 1];
  __shared__ uint32_t wc[10];
  constexpr uint32_t kKept = 1u, kTip = 2u, kLast = 8u, kHead = 0x100u, kBucketFirst = 0x200u, kPending = 0x400u, kFar = 0x800u;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  const long long base = (long long)blockIdx.x * T;
  // the staged window, as indices into win[]: [w_lo, w_hi); at_start / at_end: the window ends where the array ends
  const int halo = P.halo;
  const int w_lo = base >= halo ? H - halo : H - (int)base;
  const long long rest = n - base;
  const int w_hi = rest >= (long long)(T + halo) ? H + T + halo : H + (int)rest;
  const bool at_start = base - (H - w_lo) == 0, at_end = base + (w_hi - H) == n;
  for (int i = w_lo + tid; i < w_hi; i += 256) win[i] = sdbg_ld64(items, base - H + i);
  if (EMIT && tid < 10) wc[tid] = 0;
  __syncthreads();

  // 1. every run head: the (a, b) pairs of its group, its own run's length / count sum, the decision
  auto finish = [&](unsigned long long v, unsigned long long u, bool first, bool gh, uint32_t M, int a, int b, uint32_t mu, uint32_t fl) -> uint32_t {
    if (P.is_seq == 1) mu = (uint32_t)MHX_MAX_MUL - (uint32_t)(v & 0xFFFFull);  // seq_to_sdbg.cpp:782-785
    int w, last, tip;
    if (sdbg_decide(M, a, b, w, last, tip)) fl |= kKept | (tip ? kTip : 0u) | (last ? kLast : 0u) | ((uint32_t)w << 4);
    if (gh && (first || (u >> 48) != (v >> 48))) fl |= kBucketFirst;  // first group of its lv1 bucket
    return fl | kHead | (mu << 12);
  };
  bool any_far = false;
#pragma unroll 1
  for (int j = 0; j < PER; ++j) {
    const int idx = H + j * 256 + tid;
    const long long g = base + j * 256 + tid;
    uint32_t r = 0;
    if (g < n) {
      const unsigned long long v = win[idx];
      const bool first = g == 0;
      const unsigned long long u = first ? 0ull : win[idx - 1];
      const bool rh = first || ((v ^ u) & P.rmask) != 0, gh = first || ((v ^ u) & P.gmask) != 0;
      if (rh) {
        int a, b, ax, bx;
        uint32_t M = sdbg_pair_bit(v, P, a, b);
        bool far = false;
        if (!gh)
          for (int p = idx - 1;; --p) {
            if (p < w_lo) {
              far = !at_start;
              break;
            }
            const unsigned long long x = win[p];
            if ((x ^ v) & P.gmask) break;
            M |= sdbg_pair_bit(x, P, ax, bx);
          }
        uint32_t len = 1, sum = (uint32_t)(v >> P.csh) & P.cmk;
        bool in_run = true;
        if (!far)
          for (int p = idx + 1;; ++p) {
            if (p >= w_hi) {
              far = !at_end;
              break;
            }
            const unsigned long long x = win[p];
            if ((x ^ v) & P.gmask) break;
            in_run = in_run && ((x ^ v) & P.rmask) == 0;
            if (in_run) {
              ++len;
              sum += (uint32_t)(x >> P.csh) & P.cmk;
              if (sum > (uint32_t)MHX_MAX_MUL) sum = MHX_MAX_MUL;  // (a run in the window holds < 2^12 records: no overflow either way)
            } else {
              M |= sdbg_pair_bit(x, P, ax, bx);
            }
          }
        if (far) {
          r = kFar;
          any_far = true;
        } else {
          r = finish(v, u, first, gh, M, a, b, P.is_seq == 2 ? sum : (len > (uint32_t)MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : len), 0u);
        }
      }
    }
    res[j * 256 + tid] = r;
  }
  // 1b. run heads whose group or run reaches beyond the window (rare): the group's runs enumerated in memory; a long run of
  //     aggregated items then has its counts summed by the whole wavefront, 64 records per step, until the cap is reached
  if (__ballot(any_far)) {
#pragma unroll 1
    for (int j = 0; j < PER; ++j) {
      const int idx = H + j * 256 + tid;
      const long long g = base + j * 256 + tid;
      uint32_t r = res[j * 256 + tid];
      uint32_t todo = 0;  // records of a pending sum
      if (r == kFar) {
        const unsigned long long v = win[idx];
        const bool first = g == 0;
        const unsigned long long u = first ? 0ull : win[idx - 1];
        const bool gh = first || ((v ^ u) & P.gmask) != 0;
        int a, b;
        uint32_t M = sdbg_pair_bit(v, P, a, b);
        long long run_end;
        sdbg_group_far(items, n, g, v, P, M, run_end);
        const long long rl = run_end - g;
        uint32_t mu = rl > (long long)MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : (uint32_t)rl;
        uint32_t fl = 0;
        if (P.is_seq == 2) {
          fl = kPending;
          todo = rl > (long long)MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : (uint32_t)rl;  // (every count is >= 1: the cap is reached within that many)
          mu = 0;
        }
        r = finish(v, u, first, gh, M, a, b, mu, fl);
      }
      uint64_t pend = __ballot((r & kPending) != 0);
      while (pend) {
        const int src = __builtin_ctzll(pend);
        pend &= pend - 1;
        const long long head = base + j * 256 + (tid - lane) + src;
        const uint32_t cnt = __shfl(todo, src, kWave);
        uint32_t total = 0;
        constexpr uint32_t kStep = 8;  // loads in flight per lane and step (one load per step: a memory round trip per 64 records)
        for (uint32_t off = 0; off < cnt && total < (uint32_t)MHX_MAX_MUL; off += kStep * kWave) {
          uint32_t x = 0;
#pragma unroll
          for (uint32_t q = 0; q < kStep; ++q) {
            const uint32_t i = off + q * kWave + lane;
            x += i < cnt ? (items[head + i].y >> P.csh) & P.cmk : 0u;
          }
          total += wave_sum(x);
        }
        if (lane == src) r = (r & 0xFFFu & ~kPending) | ((total > (uint32_t)MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : total) << 12);
      }
      res[j * 256 + tid] = r;
    }
  }
  // 2. the three counters of every record -> ballots; cells (round, wave) in record order -> exclusive prefixes
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t r = res[j * 256 + tid];
    const bool kept = r & kKept;
    const uint64_t bk = __ballot(kept), bt = __ballot(kept && (r & kTip)), bl = __ballot(kept && (r >> 12) > 254u);
    if (lane == 0) {
      cell[0][j * NW + wv] = (uint32_t)__builtin_popcountll(bk);
      cell[1][j * NW + wv] = (uint32_t)__builtin_popcountll(bt);
      cell[2][j * NW + wv] = (uint32_t)__builtin_popcountll(bl);
    }
  }
  __syncthreads();
  if (tid < kWave) {
    static_assert(PER * NW <= kWave, "one wavefront scans the cells");
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t x = lane < PER * NW ? cell[c][lane] : 0u;
      const uint32_t incl = wave_inclusive_sum(x);
      if (lane < PER * NW) cell[c][lane] = incl - x;
      if (lane == PER * NW - 1) cell[c][PER * NW] = incl;
    }
  }
  __syncthreads();
  if constexpr (!EMIT) {
    if (tid < 3) tile_tot[(uint64_t)tid * n_tiles + blockIdx.x] = cell[tid][PER * NW];
    if (res_out) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const long long g = base + j * 256 + tid;
        if (g < n) res_out[g] = res[j * 256 + tid];
      }
    }
  } else {
    const uint64_t b0 = tile_base[blockIdx.x], b1 = tile_base[n_tiles + blockIdx.x], b2 = tile_base[2 * n_tiles + blockIdx.x];
#pragma unroll 1
    for (int j = 0; j < PER; ++j) {
      const uint32_t r = res[j * 256 + tid];
      const bool kept = r & kKept;
      const uint32_t m = r >> 12;
      const uint64_t bk = __ballot(kept), bt = __ballot(kept && (r & kTip)), bl = __ballot(kept && m > 254u);
      if (!(r & kHead)) continue;
      const uint64_t o0 = b0 + cell[0][j * NW + wv] + (uint32_t)__builtin_popcountll(bk & lanemask_lt);
      const uint64_t o1 = b1 + cell[1][j * NW + wv] + (uint32_t)__builtin_popcountll(bt & lanemask_lt);
      const uint64_t o2 = b2 + cell[2][j * NW + wv] + (uint32_t)__builtin_popcountll(bl & lanemask_lt);
      const unsigned long long v = win[H + j * 256 + tid];
      if (r & kBucketFirst) {
        const uint32_t bkt = (uint32_t)(v >> 48);
        bstart[bkt] = o0;
        bstart[MHX_NUM_BUCKETS + bkt] = o1;
        bstart[2 * MHX_NUM_BUCKETS + bkt] = o2;
      }
      if (!kept) continue;
      const uint32_t w = (r >> 4) & 0xFu, last = (r & kLast) ? 1u : 0u, tip = (r & kTip) ? 1u : 0u;
      uint64_t o16 = o0 + o2 + 2ull * P.wpt * o1;
      // SdbgItem (sdbg_item.h:14-24): byte0 = w | last<<4 | tip<<5, byte1 = min(mul,255)
      out16[o16++] = (uint16_t)(w | (last << 4) | (tip << 5) | ((m > 255 ? 255u : m) << 8));
      if (m > 254) out16[o16++] = (uint16_t)m;
      if (tip) {
        for (int x = 0; x < P.wpt; ++x) {
          uint32_t t = x == 0 ? (uint32_t)(v >> 32) : (uint32_t)v;
          if (P.is_seq == 2 && x == P.wpt - 1) {  // (the reference's raw tip label: SdbgOp::unit_emit)
            const int tip_chars = P.k - 1, in_word = tip_chars - 16 * x;
            const uint32_t cm = in_word >= 16 ? 0xFFFFFFFFu : (in_word <= 0 ? 0u : 0xFFFFFFFFu << (32 - 2 * in_word));
            t &= cm;
            if (x == P.ref_kw - 1) t |= (uint32_t)((v >> P.bsh) & 7ull);
          }
          out16[o16++] = (uint16_t)(t & 0xFFFFu);
          out16[o16++] = (uint16_t)(t >> 16);
        }
      }
      atomicAdd(&wc[w], 1u);
      if (last) atomicAdd(&wc[9], 1u);
    }
    __syncthreads();
    if (tid < 10 && wc[tid]) atomicAdd(&w_count[tid], (unsigned long long)wc[tid]);
  }
}

// the emitting launch when the counting launch kept its findings (k_sdbg_fast<false>: res_out): prefix sums over the three
// counters of the tile and the SdBG records — the sorted items themselves are read only for tips (their label) and for the
// first group of an lv1 bucket
template <int T>
__global__ __launch_bounds__(256) void k_sdbg_emit_res(const uint2 *__restrict__ items, long long n, SdbgFastP P, const uint32_t *__restrict__ res_g,
                                                       const uint64_t *__restrict__ tile_base, uint64_t n_tiles, uint16_t *__restrict__ out16,
                                                       unsigned long long *__restrict__ w_count, unsigned long long *__restrict__ bstart) {
  constexpr int PER = T / 256, NW = 256 / kWave;
  constexpr uint32_t kKept = 1u, kTip = 2u, kLast = 8u, kHead = 0x100u, kBucketFirst = 0x200u;
  __shared__ uint32_t cell[3][PER * NW + 1];
  __shared__ uint32_t wc[10];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  const long long base = (long long)blockIdx.x * T;
  if (tid < 10) wc[tid] = 0;
  uint32_t r[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const long long g = base + j * 256 + tid;
    r[j] = g < n ? res_g[g] : 0u;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool kept = r[j] & kKept;
    const uint64_t bk = __ballot(kept), bt = __ballot(kept && (r[j] & kTip)), bl = __ballot(kept && (r[j] >> 12) > 254u);
    if (lane == 0) {
      cell[0][j * NW + wv] = (uint32_t)__builtin_popcountll(bk);
      cell[1][j * NW + wv] = (uint32_t)__builtin_popcountll(bt);
      cell[2][j * NW + wv] = (uint32_t)__builtin_popcountll(bl);
    }
  }
  __syncthreads();
  if (tid < kWave) {
    static_assert(PER * NW <= kWave, "one wavefront scans the cells");
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t x = lane < PER * NW ? cell[c][lane] : 0u;
      const uint32_t incl = wave_inclusive_sum(x);
      if (lane < PER * NW) cell[c][lane] = incl - x;
    }
  }
  __syncthreads();
  const uint64_t b0 = tile_base[blockIdx.x], b1 = tile_base[n_tiles + blockIdx.x], b2 = tile_base[2 * n_tiles + blockIdx.x];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t rr = r[j];
    const bool kept = rr & kKept;
    const uint32_t m = rr >> 12;
    const uint64_t bk = __ballot(kept), bt = __ballot(kept && (rr & kTip)), bl = __ballot(kept && m > 254u);
    if (!(rr & kHead)) continue;
    const uint64_t o0 = b0 + cell[0][j * NW + wv] + (uint32_t)__builtin_popcountll(bk & lanemask_lt);
    const uint64_t o1 = b1 + cell[1][j * NW + wv] + (uint32_t)__builtin_popcountll(bt & lanemask_lt);
    const uint64_t o2 = b2 + cell[2][j * NW + wv] + (uint32_t)__builtin_popcountll(bl & lanemask_lt);
    const bool tip = kept && (rr & kTip);
    unsigned long long v = 0;
    if ((rr & kBucketFirst) || tip) v = sdbg_ld64(items, base + j * 256 + tid);
    if (rr & kBucketFirst) {
      const uint32_t bkt = (uint32_t)(v >> 48);
      bstart[bkt] = o0;
      bstart[MHX_NUM_BUCKETS + bkt] = o1;
      bstart[2 * MHX_NUM_BUCKETS + bkt] = o2;
    }
    if (!kept) continue;
    const uint32_t w = (rr >> 4) & 0xFu, last = (rr & kLast) ? 1u : 0u;
    uint64_t o16 = o0 + o2 + 2ull * P.wpt * o1;
    out16[o16++] = (uint16_t)(w | (last << 4) | ((tip ? 1u : 0u) << 5) | ((m > 255 ? 255u : m) << 8));  // SdbgItem, sdbg_item.h:14-24
    if (m > 254) out16[o16++] = (uint16_t)m;
    if (tip) {
      for (int x = 0; x < P.wpt; ++x) {
        uint32_t t = x == 0 ? (uint32_t)(v >> 32) : (uint32_t)v;
        if (P.is_seq == 2 && x == P.wpt - 1) {  // (the reference's raw tip label: SdbgOp::unit_emit)
          const int tip_chars = P.k - 1, in_word = tip_chars - 16 * x;
          const uint32_t cm = in_word >= 16 ? 0xFFFFFFFFu : (in_word <= 0 ? 0u : 0xFFFFFFFFu << (32 - 2 * in_word));
          t &= cm;
          if (x == P.ref_kw - 1) t |= (uint32_t)((v >> P.bsh) & 7ull);
        }
        out16[o16++] = (uint16_t)(t & 0xFFFFu);
        out16[o16++] = (uint16_t)(t >> 16);
      }
    }
    atomicAdd(&wc[w], 1u);
    if (last) atomicAdd(&wc[9], 1u);
  }
  __syncthreads();
  if (tid < 10 && wc[tid]) atomicAdd(&w_count[tid], (unsigned long long)wc[tid]);
}

// bstart[3][65536] (kNoStart = empty bucket) + totals -> per-bucket counts and byte offsets: a bucket ends where the next
// non-empty one starts.  256 workgroups of 256 buckets each (coalesced reads; the single workgroup with 256 buckets per
// thread of rounds 1-2 took 0.31 ms of every step).  k_bucket_first: first non-empty bucket of every workgroup's range.
__global__ __launch_bounds__(256) void k_bucket_first(const unsigned long long *__restrict__ bstart, uint32_t *__restrict__ block_first) {
  __shared__ uint32_t first;
  if (threadIdx.x == 0) first = 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t bk = blockIdx.x * 256 + threadIdx.x;
  if (bstart[bk] != kNoStart) atomicMin(&first, bk);
  __syncthreads();
  if (threadIdx.x == 0) block_first[blockIdx.x] = first;
}
__global__ __launch_bounds__(256) void k_bucket_fix(const unsigned long long *__restrict__ bstart, const uint32_t *__restrict__ block_first,
                                                    const uint64_t *__restrict__ totals, int wpt, unsigned long long *__restrict__ b_items,
                                                    unsigned long long *__restrict__ b_tips, unsigned long long *__restrict__ b_large,
                                                    unsigned long long *__restrict__ b_off) {
  __shared__ unsigned long long s0s[256], s1s[256], s2s[256];
  __shared__ unsigned long long wave_mask[256 / kWave];
  __shared__ unsigned long long after[3];  // start of the first non-empty bucket behind this workgroup's range (or the totals)
  const int t = threadIdx.x, lane = t & (kWave - 1), wv = t / kWave;
  const uint32_t bk = blockIdx.x * 256 + t;
  const unsigned long long s0 = bstart[bk], s1 = bstart[MHX_NUM_BUCKETS + bk], s2 = bstart[2 * MHX_NUM_BUCKETS + bk];
  s0s[t] = s0;
  s1s[t] = s1;
  s2s[t] = s2;
  const unsigned long long mask = __ballot(s0 != kNoStart);
  if (lane == 0) wave_mask[wv] = mask;
  if (t == 0) {
    uint32_t nb = 0xFFFFFFFFu;
    for (uint32_t j = blockIdx.x + 1; j < MHX_NUM_BUCKETS / 256 && nb == 0xFFFFFFFFu; ++j) nb = block_first[j];
    after[0] = nb == 0xFFFFFFFFu ? totals[0] : bstart[nb];
    after[1] = nb == 0xFFFFFFFFu ? totals[1] : bstart[MHX_NUM_BUCKETS + nb];
    after[2] = nb == 0xFFFFFFFFu ? totals[2] : bstart[2 * MHX_NUM_BUCKETS + nb];
  }
  __syncthreads();
  // next non-empty bucket behind mine: in my wavefront, in a later wavefront of the workgroup, or behind the workgroup
  int nx = -1;
  const unsigned long long later = lane == kWave - 1 ? 0ull : (mask >> (lane + 1));
  if (later) nx = wv * kWave + lane + 1 + __builtin_ctzll(later);
  else
    for (int w = wv + 1; w < 256 / kWave && nx < 0; ++w)
      if (wave_mask[w]) nx = w * kWave + __builtin_ctzll(wave_mask[w]);
  const unsigned long long n0 = nx >= 0 ? s0s[nx] : after[0], n1 = nx >= 0 ? s1s[nx] : after[1], n2 = nx >= 0 ? s2s[nx] : after[2];
  if (s0 != kNoStart) {
    b_items[bk] = n0 - s0;
    b_tips[bk] = n1 - s1;
    b_large[bk] = n2 - s2;
    b_off[bk] = 2ull * (s0 + s2) + 4ull * wpt * s1;
  } else {
    b_items[bk] = b_tips[bk] = b_large[bk] = 0;
    b_off[bk] = 2ull * (n0 + n2) + 4ull * wpt * n1;
  }
}

template <int S>
static void emit_sdbg_impl(mhx_ctx *c, const uint32_t *sorted, uint64_t n_items, const SdbgParams &P, int kmer_bits, uint64_t tot[3]) {
  hipStream_t st = c->stream;
  unsigned long long *b_items = c->result(MHX_BUF_BUCKET_COUNT, MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  unsigned long long *b_tips = c->result(MHX_BUF_BUCKET_TIPS, MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  unsigned long long *b_large = c->result(MHX_BUF_BUCKET_LARGE, MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  unsigned long long *b_off = c->result(MHX_BUF_BUCKET_OFFSET, MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  unsigned long long *w_count = c->result(MHX_BUF_W_COUNT, 10 * 8).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(w_count, 0, 80, st));
  tot[0] = tot[1] = tot[2] = 0;
  if (n_items == 0) {
    c->result(MHX_BUF_SDBG_BYTES, 2);
    c->results[MHX_BUF_SDBG_BYTES].used = 0;
    MHX_HIP(hipMemsetAsync(b_items, 0, MHX_NUM_BUCKETS * 8, st));
    MHX_HIP(hipMemsetAsync(b_tips, 0, MHX_NUM_BUCKETS * 8, st));
    MHX_HIP(hipMemsetAsync(b_large, 0, MHX_NUM_BUCKETS * 8, st));
    MHX_HIP(hipMemsetAsync(b_off, 0, MHX_NUM_BUCKETS * 8, st));
    return;
  }
  constexpr int T = SdbgTile<S>::kT;
  const uint64_t n_tiles = div_ceil(n_items, T);
  uint64_t *tt = c->ws("tile_tot", (3 * n_tiles + 4) * 8).as<uint64_t>();
  uint64_t *tb = c->ws("tile_base", (3 * n_tiles + 4) * 8).as<uint64_t>();
  uint64_t *d_tot = c->ws("tile_totals", 64).as<uint64_t>();
  unsigned long long *bstart = c->ws("bucket_start", 3 * MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  const int full_words = kmer_bits / 32, rem = kmer_bits % 32;
  const uint32_t last_mask = rem ? 0xFFFFFFFFu << (32 - rem) : 0;
  SdbgOp<S> op{P, nullptr, w_count, bstart};
  const double bytes = (double)n_items * S * 4;
  if constexpr (S == 2) {
    // 8-byte records: k_sdbg_fast (every run head on its own) where runs are short — the aggregated items of stage 1 and the items of
    // seq2sdbg, one record per distinct (k+1)-mer and strand.  Items per OCCURRENCE (min count 1, k > 22: a run = the occurrences of
    // one edge, ~coverage records) keep the tile kernel, whose work per record does not grow with the run: measured on the configs[4]
    // shard (10 G items, runs of ~15) 79 + 126 ms against 408 + 416 ms for the run-head form.  sdbg_fast = 2 forces it (tests).
    const long long fast_opt = c->opt("sdbg_fast", 1);
    if ((fast_opt == 2 || (fast_opt == 1 && P.is_seq != 0)) && P.kw >= 1 && P.kw <= 2 && P.aw <= 1) {
      SdbgFastP F;
      F.gmask = kmer_bits >= 64 ? ~0ull : (kmer_bits ? ~0ull << (64 - kmer_bits) : 0ull);
      F.bsh = P.bshift + (P.kw - 1 == 0 ? 32 : 0);
      F.fsh = P.fshift + (P.kw - 1 == 0 ? 32 : 0);
      F.ash = P.ashift + (P.aw == 0 ? 32 : 0);
      F.rmask = F.gmask | (0xFull << F.bsh) | (3ull << F.ash);
      F.is_seq = P.is_seq;
      F.csh = P.cshift;
      F.cmk = P.cmask;
      F.wpt = P.wpt;
      F.k = P.k;
      F.ref_kw = P.ref_kw;
      F.halo = (int)std::min<long long>(kSdbgFastHalo, std::max<long long>(1, c->opt("sdbg_fast_halo", kSdbgFastHalo)));
      const int ft = (int)c->opt("sdbg_fast_tile", 2048);  // records per workgroup: 1024 | 2048 | 4096
      const uint64_t ftile = ft == 1024 ? 1024 : (ft == 4096 ? 4096 : 2048);
      const uint64_t n_ft = div_ceil(n_items, ftile);
      uint64_t *ftt = c->ws("tile_tot", (3 * n_ft + 4) * 8).as<uint64_t>();
      uint64_t *ftb = c->ws("tile_base", (3 * n_ft + 4) * 8).as<uint64_t>();
      const uint2 *recs = reinterpret_cast<const uint2 *>(sorted);
      uint16_t *out16 = nullptr;
      uint64_t out_bytes = 0;
      // sdbg_fast_keep: the counting launch keeps what its run heads found (4 bytes per record, up to sdbg_fast_keep_max_mb of them)
      // and the emitting launch only scans and writes
      const bool keep = c->opt("sdbg_fast_keep", 1) != 0 && n_items * 4 <= (uint64_t)c->opt("sdbg_fast_keep_max_mb", 4096) << 20;
      uint32_t *res_g = keep ? c->ws("sdbg_res", n_items * 4 + 64).as<uint32_t>() : nullptr;
#define MHX_SDBG_FAST(EMITV, NAME, BYTES, TT, TB)                                                                                          \
  do {                                                                                                                                     \
    if (ftile == 1024)                                                                                                                     \
      MHX_LAUNCH(c, NAME, BYTES, hipLaunchKernelGGL((k_sdbg_fast<EMITV, 1024>), dim3((unsigned)n_ft), dim3(256), 0, st, recs, (long long)n_items, F, TT, TB, \
                                                    n_ft, out16, w_count, bstart, res_g));                                                 \
    else if (ftile == 4096)                                                                                                                \
      MHX_LAUNCH(c, NAME, BYTES, hipLaunchKernelGGL((k_sdbg_fast<EMITV, 4096>), dim3((unsigned)n_ft), dim3(256), 0, st, recs, (long long)n_items, F, TT, TB, \
                                                    n_ft, out16, w_count, bstart, res_g));                                                 \
    else                                                                                                                                   \
      MHX_LAUNCH(c, NAME, BYTES, hipLaunchKernelGGL((k_sdbg_fast<EMITV, 2048>), dim3((unsigned)n_ft), dim3(256), 0, st, recs, (long long)n_items, F, TT, TB, \
                                                    n_ft, out16, w_count, bstart, res_g));                                                 \
  } while (0)
      MHX_SDBG_FAST(false, "sdbg_count", bytes, ftt, (const uint64_t *)nullptr);
      for (int r = 0; r < 3; ++r) exclusive_scan_u64(c, ftt + r * n_ft, ftb + r * n_ft, n_ft, d_tot + r);
      MHX_HIP(hipMemcpyAsync(tot, d_tot, 24, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      out_bytes = 2 * (tot[0] + tot[2]) + 4ull * P.wpt * tot[1];
      out16 = c->result(MHX_BUF_SDBG_BYTES, out_bytes ? out_bytes : 2).as<uint16_t>();
      c->results[MHX_BUF_SDBG_BYTES].used = out_bytes;
      MHX_HIP(hipMemsetAsync(bstart, 0xFF, 3 * MHX_NUM_BUCKETS * 8, st));
      if (keep) {
        const double ebytes = (double)n_items * 4 + (double)out_bytes;
        if (ftile == 1024)
          MHX_LAUNCH(c, "sdbg_emit", ebytes, hipLaunchKernelGGL((k_sdbg_emit_res<1024>), dim3((unsigned)n_ft), dim3(256), 0, st, recs, (long long)n_items, F, res_g,
                                                                (const uint64_t *)ftb, n_ft, out16, w_count, bstart));
        else if (ftile == 4096)
          MHX_LAUNCH(c, "sdbg_emit", ebytes, hipLaunchKernelGGL((k_sdbg_emit_res<4096>), dim3((unsigned)n_ft), dim3(256), 0, st, recs, (long long)n_items, F, res_g,
                                                                (const uint64_t *)ftb, n_ft, out16, w_count, bstart));
        else
          MHX_LAUNCH(c, "sdbg_emit", ebytes, hipLaunchKernelGGL((k_sdbg_emit_res<2048>), dim3((unsigned)n_ft), dim3(256), 0, st, recs, (long long)n_items, F, res_g,
                                                                (const uint64_t *)ftb, n_ft, out16, w_count, bstart));
      } else {
        res_g = nullptr;
        MHX_SDBG_FAST(true, "sdbg_emit", bytes + (double)out_bytes, (uint64_t *)nullptr, (const uint64_t *)ftb);
      }
#undef MHX_SDBG_FAST
      uint32_t *block_first = c->ws("bucket_block_first", MHX_NUM_BUCKETS / 256 * 4).as<uint32_t>();
      hipLaunchKernelGGL(k_bucket_first, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, bstart, block_first);
      MHX_LAUNCH(c, "bucket_stats", (double)MHX_NUM_BUCKETS * 56,
                 hipLaunchKernelGGL(k_bucket_fix, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, bstart, block_first, d_tot, P.wpt, b_items, b_tips,
                                    b_large, b_off));
      return;
    }
  }
  MHX_LAUNCH(c, "sdbg_count", bytes,
             hipLaunchKernelGGL((k_tile_groups<S, T, SdbgOp<S>, false>), dim3(tile_grid(n_tiles)), dim3(kTileThreads), 0, st, sorted, n_items,
                                full_words, last_mask, op, tt, (const uint64_t *)nullptr, n_tiles, n_tiles));
  for (int r = 0; r < 3; ++r) exclusive_scan_u64(c, tt + r * n_tiles, tb + r * n_tiles, n_tiles, d_tot + r);
  MHX_HIP(hipMemcpyAsync(tot, d_tot, 24, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  const uint64_t out_bytes = 2 * (tot[0] + tot[2]) + 4ull * P.wpt * tot[1];
  uint16_t *out16 = c->result(MHX_BUF_SDBG_BYTES, out_bytes ? out_bytes : 2).as<uint16_t>();
  c->results[MHX_BUF_SDBG_BYTES].used = out_bytes;
  MHX_HIP(hipMemsetAsync(bstart, 0xFF, 3 * MHX_NUM_BUCKETS * 8, st));
  op.out16 = out16;
  MHX_LAUNCH(c, "sdbg_emit", bytes + (double)out_bytes,
             hipLaunchKernelGGL((k_tile_groups<S, T, SdbgOp<S>, true>), dim3(tile_grid(n_tiles)), dim3(kTileThreads), 0, st, sorted, n_items,
                                full_words, last_mask, op, (uint64_t *)nullptr, (const uint64_t *)tb, n_tiles, n_tiles));
  uint32_t *block_first = c->ws("bucket_block_first", MHX_NUM_BUCKETS / 256 * 4).as<uint32_t>();
  hipLaunchKernelGGL(k_bucket_first, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, bstart, block_first);
  MHX_LAUNCH(c, "bucket_stats", (double)MHX_NUM_BUCKETS * 56,
             hipLaunchKernelGGL(k_bucket_fix, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, bstart, block_first, d_tot, P.wpt, b_items, b_tips,
                                b_large, b_off));
}

// sorted: n_items records of stride S with kw key words, sorted; fills the SdBG result buffers.
void emit_sdbg(mhx_ctx *c, const uint32_t *sorted, uint64_t n_items, int S, int kw, uint32_t k, int is_seq, mhx_sdbg_result *out, int compact_bits) {
  hipStream_t st = c->stream;
  SdbgParams P;
  P.stride = S;
  P.kw = kw;
  P.k = (int)k;
  P.ref_kw = (int)div_ceil(k * 2 + 4, 32);
  // compact_bits > 0 (aggregated items of k = 23..28): the seq2sdbg layout with a count field of that many bits instead of 16 — chars,
  // flag | W, count from the top down, so that the whole-key order is still chars, flag, W
  P.bshift = is_seq ? (compact_bits ? compact_bits : 16) : 0;
  P.fshift = P.bshift + 3;
  P.cshift = 0;
  P.cmask = compact_bits ? (1u << compact_bits) - 1u : 0xFFFFu;
  P.aw = (int)(k - 1) / 16;
  P.ashift = (15 - (int)((k - 1) % 16)) * 2;
  P.wpt = (int)div_ceil(k, 16);
  P.is_seq = is_seq;
  const int kmer_bits = (int)(k - 1) * 2;
  uint64_t tot[3] = {0, 0, 0};
  switch (S) {
#define MHX_CASE(SV) \
  case SV: emit_sdbg_impl<SV>(c, sorted, n_items, P, kmer_bits, tot); break;
    MHX_CASE(2) MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
#undef MHX_CASE
    default: throw Error("emit_sdbg: unsupported record stride");
  }
  c->results[MHX_BUF_SORTED_ITEMS].release();
  c->results[MHX_BUF_SORTED_ITEMS].p = const_cast<uint32_t *>(sorted);
  c->results[MHX_BUF_SORTED_ITEMS].cap = 0;
  c->results[MHX_BUF_SORTED_ITEMS].used = n_items * (size_t)S * 4;
  c->sorted_item_words = S;
  MHX_HIP(hipStreamSynchronize(st));
  if (out) {
    out->n_items = n_items;
    out->n_sdbg = tot[0];
    out->n_tips = tot[1];
    out->n_large = tot[2];
    out->sdbg_bytes = 2 * (tot[0] + tot[2]) + 4ull * P.wpt * tot[1];
    out->words_per_tip_label = P.wpt;
    out->item_words = S;
  }
}

// ---- SdBG outputs of several bucket-range passes kept on the device (multi-GPU passes, comm.hip): every bucket is
// produced by exactly one pass, the byte streams concatenate in pass order, the per-bucket tables add up.
__global__ void k_sdbg_acc_tables(const unsigned long long *__restrict__ cnt, const unsigned long long *__restrict__ tips,
                                  const unsigned long long *__restrict__ large, const unsigned long long *__restrict__ off,
                                  unsigned long long prev_bytes, unsigned long long *__restrict__ a_cnt, unsigned long long *__restrict__ a_tips,
                                  unsigned long long *__restrict__ a_large, unsigned long long *__restrict__ a_off) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= MHX_NUM_BUCKETS || !cnt[b]) return;
  a_cnt[b] += cnt[b];
  a_tips[b] += tips[b];
  a_large[b] += large[b];
  a_off[b] = off[b] + prev_bytes;
}
__global__ void k_add_u64(unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
void sdbg_accumulate(mhx_ctx *c, bool first) {
  hipStream_t st = c->stream;
  static const int tabs[4] = {MHX_BUF_BUCKET_COUNT, MHX_BUF_BUCKET_TIPS, MHX_BUF_BUCKET_LARGE, MHX_BUF_BUCKET_OFFSET};
  static const char *names[4] = {"acc_bucket_count", "acc_bucket_tips", "acc_bucket_large", "acc_bucket_offset"};
  unsigned long long *acc[4];
  for (int i = 0; i < 4; ++i) {
    acc[i] = c->ws(names[i], MHX_NUM_BUCKETS * 8).as<unsigned long long>();
    if (first) MHX_HIP(hipMemsetAsync(acc[i], 0, MHX_NUM_BUCKETS * 8, st));
  }
  unsigned long long *acc_w = c->ws("acc_w_count", 80).as<unsigned long long>();
  if (first) {
    MHX_HIP(hipMemsetAsync(acc_w, 0, 80, st));
    c->work["acc_sdbg_bytes"].used = 0;
  }
  DevBuf &src = c->results[MHX_BUF_SDBG_BYTES];
  const uint64_t prev = c->work["acc_sdbg_bytes"].used;
  DevBuf &dst = grow_preserving(c, c->work["acc_sdbg_bytes"], prev + src.used + 64, prev);
  if (src.used) MHX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(dst.p) + prev, src.p, src.used, hipMemcpyDeviceToDevice, st));
  dst.used = prev + src.used;
  hipLaunchKernelGGL(k_sdbg_acc_tables, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, c->results[tabs[0]].as<unsigned long long>(),
                     c->results[tabs[1]].as<unsigned long long>(), c->results[tabs[2]].as<unsigned long long>(),
                     c->results[tabs[3]].as<unsigned long long>(), (unsigned long long)prev, acc[0], acc[1], acc[2], acc[3]);
  hipLaunchKernelGGL(k_add_u64, dim3(1), dim3(64), 0, st, acc_w, c->results[MHX_BUF_W_COUNT].as<unsigned long long>(), 10);
  MHX_HIP(hipGetLastError());
  MHX_HIP(hipStreamSynchronize(st));
}
// the accumulated outputs become the handle's SdBG result buffers
void sdbg_publish_accumulated(mhx_ctx *c) {
  static const int tabs[4] = {MHX_BUF_BUCKET_COUNT, MHX_BUF_BUCKET_TIPS, MHX_BUF_BUCKET_LARGE, MHX_BUF_BUCKET_OFFSET};
  static const char *names[4] = {"acc_bucket_count", "acc_bucket_tips", "acc_bucket_large", "acc_bucket_offset"};
  MHX_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 4; ++i) {
    std::swap(c->results[tabs[i]].p, c->work[names[i]].p);
    std::swap(c->results[tabs[i]].cap, c->work[names[i]].cap);
    c->results[tabs[i]].used = MHX_NUM_BUCKETS * 8;
  }
  std::swap(c->results[MHX_BUF_W_COUNT].p, c->work["acc_w_count"].p);
  std::swap(c->results[MHX_BUF_W_COUNT].cap, c->work["acc_w_count"].cap);
  c->results[MHX_BUF_W_COUNT].used = 80;
  const uint64_t used = c->work["acc_sdbg_bytes"].used;
  std::swap(c->results[MHX_BUF_SDBG_BYTES].p, c->work["acc_sdbg_bytes"].p);
  std::swap(c->results[MHX_BUF_SDBG_BYTES].cap, c->work["acc_sdbg_bytes"].cap);
  c->results[MHX_BUF_SDBG_BYTES].used = used;
}

// fewest 8-bit passes covering the given bit ranges (ascending, disjoint; at most two): the bits of the
// ranges are concatenated into one virtual key, so a digit may consist of the top of one range and the
// bottom of the next (two-field digit) instead of wasting a pass on a partial digit per range.
std::vector<SortPass> make_passes_ranges(int key_words, const std::vector<std::pair<int, int>> &ranges) {
  if (ranges.size() == 1) return make_passes(key_words, ranges[0].first, ranges[0].second);
  if (ranges.size() != 2) throw Error("make_passes_ranges: at most two ranges");
  const int lo0 = ranges[0].first, len0 = ranges[0].second - ranges[0].first;
  const int lo1 = ranges[1].first, len1 = ranges[1].second - ranges[1].first;
  // covering the gap as well costs no extra pass? then a plain contiguous plan is simplest
  if ((int)div_ceil(ranges[1].second - lo0, 8) <= (int)div_ceil(len0 + len1, 8)) return make_passes(key_words, lo0, ranges[1].second);
  std::vector<SortPass> p;
  for (int v = 0; v < len0 + len1; v += 8) {
    const int n = std::min(8, len0 + len1 - v);
    SortPass ps{0, 0, 0, 0};
    if (v + n <= len0) ps = {lo0 + v, n, 0, 0};
    else if (v >= len0) ps = {lo1 + (v - len0), n, 0, 0};
    else ps = {lo0 + v, len0 - v, lo1, n - (len0 - v)};
    p.push_back(ps);
  }
  return p;
}

static int s2_kw(uint32_t k) { return (int)div_ceil(k * 2 + 4, 32); }  // read_to_sdbg_s2.cpp:98-99
int s2_stride(uint32_t k) { return round_up2(s2_kw(k)); }

// items of the local reads -> c->ws("items_a"); returns their number
uint64_t s2_extract(mhx_ctx *c, uint32_t k, uint32_t m) {
  SeqSet &s = c->seqs;
  if (k < 9 || k > MHX_MAX_K) throw Error("read2sdbg: k out of range [9,255]");
  const int KWv = s2_kw(k), S = round_up2(KWv);
  const uint64_t ns = s.n_seqs;
  hipStream_t st = c->stream;
  const int sure = m == 1;  // for_sure_solid, :295
  const unsigned long long *solid = nullptr;
  if (!sure) {
    auto it = c->results.find(c->global_bases ? MHX_BUF_IS_SOLID_LOCAL : MHX_BUF_IS_SOLID);
    if (it == c->results.end() || it->second.used < div_ceil(s.n_bases, 64) * 8)
      throw Error("read2sdbg_s2: no is_solid bitmap (run mhx_read2sdbg_s1 or mhx_set_is_solid first)");
    solid = it->second.as<unsigned long long>();
  }
  uint32_t *cnt = c->ws("seq_item_cnt", (ns + 1) * 4).as<uint32_t>();
  uint64_t *item_start = c->ws("seq_item_start", (ns + 2) * 8).as<uint64_t>();
  uint64_t n_items = 0;
  const unsigned grid = 256 * 8;
  // bucket-range passes: only the items of the kept buckets are counted and written (s2_kept_mask)
  const uint8_t *drop = c->s2_filter_in_extract ? c->work["filter_lut"].as<uint8_t>() : nullptr;
  if (ns) {
    MHX_DISPATCH_KW(KWv, {
      MHX_LAUNCH(c, "s2_count", (double)s.n_bases * 3 / 8 + (double)ns * 20,
                 hipLaunchKernelGGL((k_s2_count<KW>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(), s.start.as<uint64_t>(), ns,
                                    (int)k, solid, sure, cnt, drop));
    });
    exclusive_scan_u32_u64(c, cnt, item_start, ns, item_start + ns + 1);
    MHX_HIP(hipMemcpyAsync(&n_items, item_start + ns + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  const size_t item_bytes = (size_t)S * 4;
  uint32_t *buf_a = c->ws("items_a", n_items * item_bytes + 64).as<uint32_t>();
  if (n_items) {
    MHX_DISPATCH_KW(KWv, {
      if (S == KW)
        MHX_LAUNCH(c, "s2_extract", (double)n_items * item_bytes + (double)s.n_bases * 3 / 8,
                   hipLaunchKernelGGL((k_s2_extract<KW, KW>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(), s.start.as<uint64_t>(),
                                      item_start, ns, (int)k, solid, sure, buf_a, drop));
      else
        MHX_LAUNCH(c, "s2_extract", (double)n_items * item_bytes + (double)s.n_bases * 3 / 8,
                   hipLaunchKernelGGL((k_s2_extract<KW, KW + 1>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(),
                                      s.start.as<uint64_t>(), item_start, ns, (int)k, solid, sure, buf_a, drop));
    });
  }
  return n_items;
}

int s2_process(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_sdbg_result *out) {
  const int KWv = s2_kw(k), S = round_up2(KWv);
  const int char_bits = (int)k * 2;
  uint32_t *sorted = sort_whole_key(c, buf_a, buf_b, n_items, S, KWv, make_passes_ranges(KWv, {{0, 4}, {KWv * 32 - char_bits, KWv * 32}}));
  emit_sdbg(c, sorted, n_items, S, KWv, k, 0, out);
  return 0;
}

// aggregated stage 2 (k <= 22): solid items from stage 1 + dummy items from the local reads -> ws "items_a"
uint64_t s2_agg_extract(mhx_ctx *c, uint32_t k) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint64_t ns = s.n_seqs, n_agg = c->agg_n;
  auto it = c->results.find(c->global_bases ? MHX_BUF_IS_SOLID_LOCAL : MHX_BUF_IS_SOLID);
  if (it == c->results.end() || it->second.used < div_ceil(s.n_bases, 64) * 8) throw Error("read2sdbg_s2: no is_solid bitmap");
  const unsigned long long *solid = it->second.as<unsigned long long>();
  const uint64_t n_words = div_ceil(s.n_bases, 64);
  unsigned long long *cur = c->ws("s2d_cursor", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(cur, 0, 16, st));
  uint64_t bound = 0;
  if (n_words) {
    MHX_LAUNCH(c, "s2_bound", (double)n_words * 8,
               hipLaunchKernelGGL(k_s2d_bound, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words, 256), 2048)), dim3(256), 0, st, solid, n_words, cur + 1));
    MHX_HIP(hipMemcpyAsync(&bound, cur + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  uint32_t *buf_a = c->ws("items_a", (n_agg + bound) * 8 + 64).as<uint32_t>();
  if (n_agg) MHX_HIP(hipMemcpyAsync(buf_a, c->work["s2_agg_items"].p, n_agg * 8, hipMemcpyDeviceToDevice, st));
  uint64_t n_dummy = 0;
  if (bound) {
    MHX_LAUNCH(c, "s2_extract", (double)bound * 8 + (double)n_words * 8,
               hipLaunchKernelGGL(k_s2d_emit, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words, 256 * kS2dWords), 4096)), dim3(256), 0, st, s.words.as<uint32_t>(),
                                  s.start.as<uint64_t>(), ns, s.fixed_len, (int)k, solid, n_words, reinterpret_cast<uint2 *>(buf_a) + n_agg, cur));
    MHX_HIP(hipMemcpyAsync(&n_dummy, cur, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  const uint64_t n_items = n_agg + n_dummy;
  return n_items;
}
// sort by k-mer chars, "full" flag and W (the count bits [0,16) ride along), then emit
int s2_agg_process(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_sdbg_result *out) {
  // (the whole-key order also orders equal keys by their count bits: the emission sums them, any order does)
  uint32_t *sorted = sort_whole_key(c, buf_a, buf_b, n_items, 2, 2, make_passes_ranges(2, {{16, 20}, {64 - 2 * (int)k, 64}}));
  emit_sdbg(c, sorted, n_items, 2, 2, k, 2, out);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Stage 2's solid items from a COUNT of the (k+1)-mers (round 6).  Without mercy edges the solid occurrences are exactly the
// occurrences of the (k+1)-mers that occur at least m times, and per distinct one of them the per-occurrence stage 2 makes two runs of
// equal items (read_to_sdbg_s2.cpp:398-409, counted again at :579) — so `count` on the bucket streaming (k_s1_stream<COUNT>, edges only:
// no positions, no first_0_out / last_0_in) gives every solid item with its multiplicity, and only the '$' items at the ends of the
// solid runs come from the reads.  Serves min count 1 (main_sdbg_build.cpp:139-147: stage 1 skipped, EVERY occurrence is an item: the
// meta presets — 10^10 items at 40 M reads, k = 27) for k <= 27, and min count >= 2 at k = 23..27, where stage 1 has no aggregated items
// of its own (they need 2k + 20 <= 64 bits).  k >= 23: the same 8-byte layout — chars, flag | W, count — with a count field of the 60 - 2k
// bits that are left (an edge with a larger multiplicity becomes several items: the emission sums them).
// ---------------------------------------------------------------------------------------------------------------
int s2_agg_compact_bits(uint32_t k) { return k >= 23 && k <= 28 ? std::min(16, 60 - 2 * (int)k) : 0; }

__global__ __launch_bounds__(256) void k_edges_to_agg(const uint4 *__restrict__ raw, uint32_t cap, const uint32_t *__restrict__ counts, int k, int fsh, int bsh, int csh,
                                                      uint32_t cmax, uint2 *__restrict__ items, unsigned long long *__restrict__ cursor, unsigned long long items_cap,
                                                      uint32_t *__restrict__ err) {
  __shared__ uint64_t sm[256 / kWave + 1];
  __shared__ unsigned long long s_base;
  const uint32_t n = counts[blockIdx.x];  // region blockIdx.x of the count's workgroups: n entries of ((k+1)-mer words, multiplicity, 0)
  const uint4 *src = raw + (size_t)blockIdx.x * cap;
  const uint64_t mask_k = ~0ull << (64 - 2 * k);
  for (uint32_t base = blockIdx.y * 256; base < n; base += gridDim.y * 256) {
    const uint32_t i = base + threadIdx.x;
    uint64_t e = 0, er = 0;
    uint32_t mul = 0, pieces = 0, strands = 0;
    if (i < n) {
      const uint4 en = src[i];
      e = ((uint64_t)en.x << 32) | en.y;
      er = rc64_s2(e, k + 1);
      mul = en.z;
      pieces = (mul + cmax - 1) / cmax;
      strands = e == er ? 1u : 2u;  // (a palindrome makes the forward items only, read_to_sdbg_s2.cpp:385-423)
    }
    uint64_t tot;
    const uint64_t excl = block_exclusive_sum<uint64_t, 256>((uint64_t)pieces * strands, sm, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
    __syncthreads();
    if (s_base + tot > items_cap) {
      if (threadIdx.x == 0) atomicOr(err, 1u);
    } else {
      uint2 *dst = items + s_base + excl;
      uint32_t left = mul;
      for (uint32_t p = 0; p < pieces; ++p) {
        const uint64_t cnt = left > cmax ? cmax : left;
        left -= (uint32_t)cnt;
        const uint64_t f = ((e << 2) & mask_k) | (1ull << fsh) | ((e >> 62) << bsh) | (cnt << csh);  // chars e[1..k], W = e[0]
        *dst++ = make_uint2((uint32_t)(f >> 32), (uint32_t)f);
        if (strands == 2) {
          const uint64_t b = ((er << 2) & mask_k) | (1ull << fsh) | ((er >> 62) << bsh) | (cnt << csh);
          *dst++ = make_uint2((uint32_t)(b >> 32), (uint32_t)b);
        }
      }
    }
    __syncthreads();
  }
}
// every position where a (k+1)-mer starts: the "solid" bitmap of min count 1 (read_to_sdbg_s2.cpp:295,381)
__global__ __launch_bounds__(256) void k_valid_starts(unsigned long long *__restrict__ words, uint64_t n_words, const uint64_t *__restrict__ start, uint64_t n_seqs,
                                                      uint32_t fixed_len, int k) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  unsigned long long v = 0;
  uint64_t r = seq_of_offset(start, n_seqs, fixed_len, w * 64);
  for (int t = 0; t < 64; ++t) {
    const uint64_t p = w * 64 + t;
    while (r + 1 < n_seqs && start[r + 1] <= p) ++r;
    if (p >= start[n_seqs]) break;
    if (p + (uint64_t)k + 1 <= start[r + 1]) v |= 1ull << t;
  }
  words[w] = v;
}

bool s2_agg_from_count_applies(mhx_ctx *c, uint32_t k, uint32_t m) {
  if (!c->opt("s2_agg_from_count", 1) || c->filter_on || c->accumulate || c->n_parts > 1 || c->global_bases) return false;
  if (m < 1 || k < 10 || k > 27) return false;
  if (m > 1 && (k <= 22 || c->solid_plain_k != k || c->solid_plain_m != m)) return false;  // (k <= 22: stage 1 made the aggregated items itself)
  const bool was = c->count_edges_only;
  c->count_edges_only = true;
  const bool ok = count_stream_applies(c, k, m);
  c->count_edges_only = was;
  return ok;
}
// -> false: the count gave up (nothing published: the per-occurrence path runs)
static bool s2_agg_from_count(mhx_ctx *c, uint32_t k, uint32_t m, mhx_sdbg_result *out) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint64_t ns = s.n_seqs;
  const int cb = s2_agg_compact_bits(k);
  const int cbits = cb ? cb : 16;
  const int fsh = cbits + 3, bsh = cbits, csh = 0;  // chars | flag | W | count, as the aggregated items of k <= 22 with their 16 count bits
  const uint32_t cmax = cb ? (1u << cb) - 1u : (uint32_t)MHX_MAX_MUL;
  // 1. the solid (k+1)-mers and their multiplicities: the workgroups' edge regions of the count (scratch first/last/histogram: nobody reads them)
  uint32_t *first = c->ws("s2c_first", (ns ? ns : 1) * 4).as<uint32_t>(), *last = c->ws("s2c_last", (ns ? ns : 1) * 4).as<uint32_t>();
  unsigned long long *hist = c->ws("s2c_hist", (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(hist, 0, (MHX_MAX_MUL + 1) * 8, st));
  CountStreamOut o;
  // Min count 1 is the choice of the meta presets: high-diversity data, where a bucket's records are mostly DISTINCT keys (a quarter of them
  // at the configs[4] shard against a twentieth at 60 x coverage).  Sub-rounds — a second read of every bucket instead of a third sort pass —
  // then still overflow the table and split again (count_groups 198 ms at 40 M reads), a third pass over a finer prefix does not
  // (36 ms + 27 for the pass): the plan is made with s1_stream_sub_max = 0 and 12 000 records per bucket unless the caller set those knobs.
  struct KnobGuard {
    mhx_ctx *c;
    std::vector<std::string> set;
    void put(const char *name, long long v) {
      std::string env = std::string("MHX_") + name;
      for (char &ch : env) ch = (char)toupper((unsigned char)ch);
      if (c->options.count(name) || getenv(env.c_str())) return;
      c->options[name] = v;
      set.push_back(name);
    }
    ~KnobGuard() {
      for (const std::string &n : set) c->options.erase(n);
    }
  } knobs{c, {}};
  if (m == 1) {
    knobs.put("s1_stream_sub_max", 0);
    knobs.put("s1_stream_max3", 12000);
  }
  c->count_edges_only = true;
  c->gen_first_pass = nullptr;
  bool ok = false;
  try {
    ok = count_stream_groups(c, k, m, first, last, hist, &o, nullptr);
  } catch (...) {
    c->count_edges_only = false;
    c->gen_first_pass = nullptr;
    throw;
  }
  c->count_edges_only = false;
  c->gen_first_pass = nullptr;
  if (!ok) return false;
  std::vector<uint32_t> h_counts(o.grid);
  MHX_HIP(hipMemcpyAsync(h_counts.data(), o.counts, (size_t)o.grid * 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  uint64_t n_edges = 0;
  for (uint32_t v : h_counts) n_edges += v;
  // 2. the bitmap the '$' items come from: stage 1's (min count > 1) or every (k+1)-mer start (min count 1)
  const uint64_t n_words = div_ceil(s.n_bases, 64);
  const unsigned long long *solid = nullptr;
  if (m > 1) {
    auto it = c->results.find(MHX_BUF_IS_SOLID);
    if (it == c->results.end() || it->second.used < n_words * 8) throw Error("read2sdbg_s2: no is_solid bitmap");
    solid = it->second.as<unsigned long long>();
  } else {
    unsigned long long *v = c->ws("s2c_valid", (n_words + 1) * 8).as<unsigned long long>();
    if (n_words)
      MHX_LAUNCH(c, "s2_valid_starts", (double)n_words * 8,
                 hipLaunchKernelGGL(k_valid_starts, dim3((unsigned)div_ceil(n_words, 256)), dim3(256), 0, st, v, n_words, s.start.as<uint64_t>(), ns, s.fixed_len, (int)k));
    solid = v;
  }
  unsigned long long *cur = c->ws("s2d_cursor", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(cur, 0, 32, st));
  uint64_t bound = 0;
  if (n_words) {
    MHX_LAUNCH(c, "s2_bound", (double)n_words * 8,
               hipLaunchKernelGGL(k_s2d_bound, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words, 256), 2048)), dim3(256), 0, st, solid, n_words, cur + 1));
    MHX_HIP(hipMemcpyAsync(&bound, cur + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  // 3. items: per edge and strand ceil(multiplicity / cmax) of them, then the dummies.  (The edge regions live in the count's spare sort
  //    buffer "items_a" / "items_b": the items get buffers of their own.)
  const uint64_t items_cap = 2 * (n_edges + o.n_items / cmax + 1);
  uint32_t *buf_a = c->ws("s2c_items_a", (items_cap + bound) * 8 + 64).as<uint32_t>();
  uint32_t *err = reinterpret_cast<uint32_t *>(cur + 3);
  if (n_edges)
    MHX_LAUNCH(c, "s2_edges_to_items", (double)n_edges * 16,
               hipLaunchKernelGGL(k_edges_to_agg, dim3(o.grid, 8), dim3(256), 0, st, reinterpret_cast<const uint4 *>(o.spare), o.cap / 2, o.counts, (int)k, fsh, bsh, csh,
                                  cmax, reinterpret_cast<uint2 *>(buf_a), cur + 2, (unsigned long long)items_cap, err));
  unsigned long long h_cur[4] = {0, 0, 0, 0};
  MHX_HIP(hipMemcpyAsync(h_cur, cur, 32, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h_cur[3] & 0xFFFFFFFFull) throw Error("read2sdbg_s2: more aggregated items than their bound");
  const uint64_t n_agg = h_cur[2];
  uint64_t n_dummy = 0;
  if (bound) {
    MHX_LAUNCH(c, "s2_extract", (double)bound * 8 + (double)n_words * 8,
               hipLaunchKernelGGL(k_s2d_emit, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words, 256 * kS2dWords), 4096)), dim3(256), 0, st, s.words.as<uint32_t>(),
                                  s.start.as<uint64_t>(), ns, s.fixed_len, (int)k, solid, n_words, reinterpret_cast<uint2 *>(buf_a) + n_agg, cur, fsh, bsh, csh));
    MHX_HIP(hipMemcpyAsync(&n_dummy, cur, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  const uint64_t n_items = n_agg + n_dummy;
  uint32_t *buf_b = c->ws("s2c_items_b", n_items * 8 + 64).as<uint32_t>();
  // 4. sort by chars, flag and W (the count bits ride along), emit
  uint32_t *sorted = sort_whole_key(c, buf_a, buf_b, n_items, 2, 2, make_passes_ranges(2, {{cbits, cbits + 4}, {64 - 2 * (int)k, 64}}));
  emit_sdbg(c, sorted, n_items, 2, 2, k, 2, out, cb);
  c->last_s1_plan = "stage 2 from a count of the (k+1)-mers: " + o.plan;
  return true;
}
bool s2_use_aggregated(const mhx_ctx *c, uint32_t k, uint32_t m) { return c->agg_valid && c->agg_k == k && c->agg_m == m && m > 1; }

// Single GPU, everything resident: the aggregated items are sorted where stage 1 left them (ws "s2_agg_items", the dummies
// appended behind them) instead of being copied into "items_a" first — 0.94 GB read + written per step at 10 M reads.  The
// sort consumes them: a second stage 2 without a new stage 1 takes the per-occurrence path (same records).
static int s2_agg_in_place(mhx_ctx *c, uint32_t k, mhx_sdbg_result *out) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint64_t n_agg = c->agg_n;
  auto it = c->results.find(MHX_BUF_IS_SOLID);
  if (it == c->results.end() || it->second.used < div_ceil(s.n_bases, 64) * 8) throw Error("read2sdbg_s2: no is_solid bitmap");
  const unsigned long long *solid = it->second.as<unsigned long long>();
  const uint64_t n_words = div_ceil(s.n_bases, 64);
  unsigned long long *cur = c->ws("s2d_cursor", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(cur, 0, 16, st));
  uint64_t bound = 0;
  if (n_words) {
    MHX_LAUNCH(c, "s2_bound", (double)n_words * 8,
               hipLaunchKernelGGL(k_s2d_bound, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words, 256), 2048)), dim3(256), 0, st, solid, n_words, cur + 1));
    MHX_HIP(hipMemcpyAsync(&bound, cur + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  uint32_t *buf_a = grow_preserving(c, c->work["s2_agg_items"], (n_agg + bound) * 8 + 64, n_agg * 8).as<uint32_t>();
  uint64_t n_dummy = 0;
  if (bound) {
    MHX_LAUNCH(c, "s2_extract", (double)bound * 8 + (double)n_words * 8,
               hipLaunchKernelGGL(k_s2d_emit, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words, 256 * kS2dWords), 4096)), dim3(256), 0, st, s.words.as<uint32_t>(),
                                  s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, (int)k, solid, n_words, reinterpret_cast<uint2 *>(buf_a) + n_agg, cur));
    MHX_HIP(hipMemcpyAsync(&n_dummy, cur, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  const uint64_t n_items = n_agg + n_dummy;
  uint32_t *buf_b = c->ws("items_b", n_items * 8 + 64).as<uint32_t>();
  c->agg_valid = false;
  return s2_agg_process(c, k, buf_a, buf_b, n_items, out);
}

int run_s2(mhx_ctx *c, uint32_t k, uint32_t m, mhx_sdbg_result *out) {
  if (c->global_bases) throw Error("read2sdbg_s2: a global layout is set; use the mhx_dist_* entry points");
  if (!c->filter_on && !c->accumulate && c->n_parts <= 1 && s2_use_aggregated(c, k, m) && c->opt("s2_agg_in_place", 1))
    return s2_agg_in_place(c, k, out);
  if (s2_agg_from_count_applies(c, k, m) && s2_agg_from_count(c, k, m, out)) return 0;
  const StageItems it = extract_stage(c, MHX_STAGE_S2, k, m);
  uint32_t *buf_a = c->work["items_a"].as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
  if (it.agg) return s2_agg_process(c, k, buf_a, buf_b, it.n, out);
  return s2_process(c, k, buf_a, buf_b, it.n, out);
}

}  // namespace mhx
