// read2sdbg stage 1: replaces Read2SdbgS1 (reference src/sorting/read_to_sdbg_s1.cpp) on the GPU.
//
//   extract   one item per canonical (k-1)-mer occurrence (+ both strands at the read ends) with
//             (head,tail) in the low 6 key bits and (prev,next,position) as aux
//             (Lv1FillOffsets :208-296 + Lv2ExtractSubString :298-366 fused)
//   sort      by (k-1)-mer then (head,tail)                                   (sort.hip)
//   groups    heads of equal-(k-1)-mer groups                                 (scan.hip)
//   reduce    per group: (head,tail) run lengths -> is_solid bits (atomicOr), multiplicity histogram,
//             optional mercy candidates                                       (Lv2Postprocess :368-555)
//
// Tie order: the sort is stable and items are emitted in the reference's global order, so the
// "first item" of a group (whose prev/next the reference re-uses for the whole group, :399) is the
// first in read order.  kmlib::kmsort is unstable for buckets > 64 items, so mercy candidates can
// differ from the reference there; is_solid and the histogram never depend on it.
#include <algorithm>

#include "dev_prims.h"
#include <cstdlib>

#include "mhx_internal.h"
#include "sort_digits.h"
#include "sort_kernels.h"
#include "tile_groups.h"

namespace mhx {

__global__ void k_s1_item_counts(const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t k, uint32_t *__restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seqs) {
    uint64_t L = start[i + 1] - start[i];
    cnt[i] = L >= k + 1 ? (uint32_t)(L - k + 4) : 0u;  // read_to_sdbg_s1.cpp:228-292
  }
}

// COMPACT (no mercy requested): the aux part is one word, the absolute position of the (k-1)-mer; that is
// all the group reduction needs to set is_solid, and it makes the record 12 instead of 16 bytes at k <= 29.
// item of slot j (0 .. L-k+3) of the read at base offset st, length L (read_to_sdbg_s1.cpp:228-292, :344-363)
// Positions in compact records: the record's third word holds the low `pos_bits` bits of the (k-1)-mer's global base
// position, the bits above them (the "tag", < 256) ride in key bits that no comparison looks at, between the (k-1)-mer and
// head/tail: bits [6, 14) of the last key word.  pos_bits = 32 unless a test asks for less (s1_pos_bits); read sets below
// 2^pos_bits bases have tag 0 everywhere — the plain 32-bit position.  (Replaces the rank tags of round 2/3: the same bits,
// but a function of the position alone, so one rank may hold more than 2^32 bases: 100 M reads on one GPU.)
__device__ __forceinline__ uint32_t s1_pos_word(uint64_t p, uint32_t pos_bits) { return pos_bits >= 32 ? (uint32_t)p : (uint32_t)p & ((1u << pos_bits) - 1u); }
__device__ __forceinline__ uint32_t s1_pos_tag(uint64_t p, uint32_t pos_bits) { return (uint32_t)(p >> pos_bits) << 6; }

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

__device__ __forceinline__ uint64_t rc64(uint64_t x, int n);
// The same records for the usual shape — fixed-length reads, 12-byte compact records, k <= 29 — with a fraction of the
// instructions (the generic kernel above is bound by instruction issue: ~240 VALU operations per record, among them a
// 64-bit division by the items-per-read count, four separate base look-ups and word-array shuffles):
//   * the k+3 bases prev|head|(k-1)-mer|tail|next of an item are ONE 64-bit window of the packed store (three words, two
//     funnel shifts); head/tail/prev/next are bit fields of it, the reverse complement is a 64-bit bit-reverse;
//   * read index and slot advance incrementally with the persistent loop (the per-iteration stride of the workgroup,
//     divided by the items per read, comes from the host), the only division left is a 32-bit one.
// Same output, bit for bit, as k_s1_extract_fixed<2, 3, true> (read_to_sdbg_s1.cpp:228-292, :344-363).
// the record of the (k-1)-mer at offset q of its read (absolute base a), from the 64 bits of the store that start two bases
// in front of it: prev | head | (k-1)-mer | tail | next ...
__device__ __forceinline__ void s1_item_from_window(uint64_t win, uint32_t q, int forced, uint32_t L, int k, uint64_t a, uint64_t pos_base,
                                                    uint32_t pos_bits, uint32_t (&out)[3]) {
  const int km1 = k - 1;
  const unsigned head_b = (unsigned)(win >> 60) & 3u, tail_b = (unsigned)(win >> (58 - 2 * km1)) & 3u;
  const uint64_t f = (win << 4) & (~0ull << (64 - 2 * km1));
  const uint64_t rc = rc64(f, km1);
  const unsigned head = q >= 1 ? head_b : kSentinel;
  const unsigned tail = q + k - 1 < L ? tail_b : kSentinel;
  int strand;
  if (forced >= 0) strand = forced;
  else strand = f > rc ? 1 : (f < rc ? 0 : (head <= 3 - tail ? 0 : 1));
  const uint64_t key = strand ? (rc | (comp_or_sentinel(tail) << 3) | comp_or_sentinel(head)) : (f | (head << 3) | tail);
  const uint64_t p = pos_base + a;
  out[0] = (uint32_t)(key >> 32);
  out[1] = (uint32_t)key | s1_pos_tag(p, pos_bits);
  out[2] = s1_pos_word(p, pos_bits);
}

// one stage-1 record of a fixed-length read set from the 64-bit window around its (k-1)-mer: read r, slot j (see above)
__device__ __forceinline__ void s1_fast_item(const uint32_t *__restrict__ seq, uint32_t L, int k, uint64_t st, uint32_t j, uint64_t pos_base,
                                             uint32_t pos_bits, uint32_t (&out)[3]) {
  // st = first base of the read (read index x L: the callers advance it with the slots instead of multiplying per item)
  // slot -> offset of the (k-1)-mer; slots 0, 1 and the last two are the forced-strand pairs at the read's ends
  const uint32_t jf = L - k + 2;
  const uint32_t q = min(j > 0 ? j - 1 : 0u, jf - 1);
  const int forced = j < 2 ? (int)j : (j >= jf ? (int)(j - jf) : -1);
  const uint64_t a = st + q;
  // The window starts two bases in front of the (k-1)-mer.  For the first two bases of the store (read 0, offsets 0 and 1) it
  // would start before the store: take the window at base 0 and shift it down instead — what moves in at the top stands for
  // bases that no record uses (offset 0 has no head, and the compact record carries no prev).  Straight-line code: with the
  // general item code behind a branch here, every window load of the generating sort pass was waited for on the spot.
  const uint64_t b = a >= 2 ? a - 2 : 0, w = b >> 4;
  const unsigned sh = (unsigned)(b & 15) * 2, down = a >= 2 ? 0u : (unsigned)(2 - a) * 2;
  const uint32_t x0 = seq[w], x1 = seq[w + 1], x2 = seq[w + 2];
  const uint64_t win = (((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh)) >> down;
  s1_item_from_window(win, q, forced, L, k, a, pos_base, pos_bits, out);
}

// The same records as a SOURCE of the first chained-scan pass (sort_kernels.h): no record array is written by the
// extraction and read back by the sort — 16 GB each way at 10 M reads.  The digit histograms the chained scan needs
// beforehand come from k_s1_extract_fast<IT, false>, the same arithmetic without the stores.
// lv1-bucket filter inside the generators (FILTER): `keep` is a bitmap over the 65 536 lv1 buckets (bit b of word b / 32); an
// item of a dropped bucket becomes a record that is_record() rejects — head/tail bits 63, which no real record carries — and
// the pass leaves it out (Src::kMayDrop, sort_kernels.h).  This is where the reference's OffsetFiller::IsHandling sits
// (base_engine.h:106-108): a bucket-range pass of the memory plan scans the reads once and writes only what it keeps.
__device__ __forceinline__ bool s1_bucket_kept(const uint32_t *__restrict__ keep, uint32_t w0) {
  const uint32_t b = w0 >> 16;
  return (keep[b >> 5] >> (b & 31u)) & 1u;
}
constexpr uint32_t kS1Dropped = 0xFFFFFFFFu;  // second key word of a dropped item

template <bool FILTER>
struct S1GenT {
  const uint32_t *seq;
  uint32_t L, per;
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  const uint32_t *keep;
  static constexpr bool kMayDrop = FILTER;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return !FILTER || (r.w[1] & 63u) != 63u; }
  template <int NI>
  __device__ __forceinline__ void get(uint64_t first, uint64_t n, Rec<3> (&rec)[NI]) const {
    const uint64_t r = first / per;  // one 64-bit division per tile and thread, then read offset and slot advance with the items
    uint32_t j = (uint32_t)(first - r * per);
    uint64_t st = r * L;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // (an item beyond the last one is made from read 0, slot 2 and dropped: unconditional loads inside the store, so that
      // the window loads of a tile are issued together)
      const bool ok = first + (uint64_t)i * kWave < n;
      uint32_t out[3];
      s1_fast_item(seq, L, k, ok ? st : 0, ok ? j : 2u, pos_base, pos_bits, out);
      if constexpr (FILTER)
        if (!s1_bucket_kept(keep, out[0])) out[1] = kS1Dropped;
      if (ok) {
        rec[i].w[0] = out[0];
        rec[i].w[1] = out[1];
        rec[i].w[2] = out[2];
      }
      j += kWave;
      while (j >= per) {
        j -= per;
        st += L;
      }
    }
  }
  // (interface of k_radix_onesweep_u: the item a thread holds in slot j of a tile, and all records of a unit that are one thread's)
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)(w * (kWave * NI) + j * kWave + lane);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
#pragma unroll
    for (int t = 0; t < UT; ++t) get<NI>(unit_base + (uint64_t)t * (kSortThreads * NI) + (uint64_t)(w * (kWave * NI) + lane), n, rec[t]);
  }
};
using S1Gen = S1GenT<false>;

// The same generator with CONSECUTIVE items per thread (a pass whose records may leave in any order does not care which
// thread holds which item of the unit): eight consecutive slots of a read share their window words — four words loaded
// once for the run that starts at the thread's first item and four for the start of the next read, instead of three words
// per item —, the slot and the read's base offset advance by increments, and there is one division per UNIT and thread.
// Needs at least NI slots per read (at most one read boundary inside a thread's items of a tile).
template <bool FILTER>
struct S1GenBlockedT {
  const uint32_t *seq;
  uint32_t L, per;
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;  // items of a tile (256 NI) divided by the slots per read: quotient and remainder
  const uint32_t *keep;
  static constexpr bool kMayDrop = FILTER;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return !FILTER || (r.w[1] & 63u) != 63u; }
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)((w * kWave + lane) * NI + j);
  }
  // all tiles of a unit at once: ONE division, the window words of all UT tiles requested before the first item is made
  // (the striped generator waits for one window load per item: 24 round trips to the store per thread and unit)
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
    constexpr uint32_t kTileItems = kSortThreads * NI;
    const uint64_t g00 = unit_base + (uint64_t)((w * kWave + lane) * NI);
    const uint32_t qlast = L - k + 1, jf = L - k + 2;
    uint32_t jt[UT];   // slot of the thread's first item in tile t
    uint64_t bt[UT];   // first base of that item's read
    {
      const uint64_t r = g00 / per;
      jt[0] = (uint32_t)(g00 - r * per);
      bt[0] = r * L;
#pragma unroll
      for (int t = 1; t < UT; ++t) {
        uint32_t jn = jt[t - 1] + tile_r;
        uint64_t bn = bt[t - 1] + (uint64_t)tile_q * L;
        if (jn >= per) {
          jn -= per;
          bn += L;
        }
        jt[t] = jn;
        bt[t] = bn;
      }
#pragma unroll
      for (int t = 0; t < UT; ++t)
        if (g00 + (uint64_t)t * kTileItems >= n) {  // nothing of this tile is this thread's: loads from the start of the store, nothing kept
          jt[t] = 0;
          bt[t] = 0;
        }
    }
    uint64_t wcur[UT], wnext[UT];
    uint32_t c[UT][4], nx[UT][4];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint32_t q0 = min(jt[t] > 0 ? jt[t] - 1 : 0u, qlast);
      const uint64_t a0 = bt[t] + q0, b0 = a0 >= 2 ? a0 - 2 : 0;
      wcur[t] = b0 >> 4;                    // first word of the windows of the run that starts at the thread's first item
      wnext[t] = (bt[t] + L - 2) >> 4;      // ... of the run at the start of the next read (slot 0: offset 0, window 2 bases in front)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        c[t][x] = seq[wcur[t] + x];
        nx[t][x] = seq[wnext[t] + x];       // (the store is padded: also behind the last read)
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint64_t g0 = g00 + (uint64_t)t * kTileItems;
      uint32_t j = jt[t];
      uint64_t base = bt[t], wc = wcur[t];
      uint32_t c0 = c[t][0], c1 = c[t][1], c2 = c[t][2], c3 = c[t][3];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t q = min(j > 0 ? j - 1 : 0u, qlast);
        const int forced = j < 2 ? (int)j : (j >= jf ? (int)(j - jf) : -1);
        const uint64_t a = base + q;
        const uint64_t b = a >= 2 ? a - 2 : 0;  // (the first two bases of the store: s1_fast_item)
        const unsigned down = a >= 2 ? 0u : (unsigned)(2 - a) * 2;
        const bool second = (b >> 4) != wc;     // a run of NI <= 8 windows starts in at most two different words
        const unsigned sh = (unsigned)(b & 15) * 2;
        const uint32_t x0 = second ? c1 : c0, x1 = second ? c2 : c1, x2 = second ? c3 : c2;
        const uint64_t win = (((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh)) >> down;
        uint32_t out[3];
        s1_item_from_window(win, q, forced, L, k, a, pos_base, pos_bits, out);
        if constexpr (FILTER)
          if (!s1_bucket_kept(keep, out[0])) out[1] = kS1Dropped;
        if (g0 + (uint64_t)i < n) {
          rec[t][i].w[0] = out[0];
          rec[t][i].w[1] = out[1];
          rec[t][i].w[2] = out[2];
        }
        if (++j == per) {
          j = 0;
          base += L;
          c0 = nx[t][0]; c1 = nx[t][1]; c2 = nx[t][2]; c3 = nx[t][3];
          wc = wnext[t];
        }
      }
    }
  }
};
using S1GenBlocked = S1GenBlockedT<false>;

// The blocked generator with the window arithmetic done ONCE per run of a thread's consecutive items (round 5; k <= 23).  A run
// of up to 8 consecutive (k-1)-mers of one read, with the head base in front and the tail base behind each, spans
// 8 + k + 1 <= 32 bases: one 64-bit window W of the store (two funnel shifts) holds them all, and the reverse complement
// of a sub-window is a sub-window of the reverse complement — R = rc(W) is formed once (one 64-bit bit-reverse), and the item
// at offset d inside the run is
//     forward  (W << (2 d + 4)) & mask        reverse complement  (R << 2 (30 - (k-1) - d)) & mask
// two shifts instead of two funnel shifts, three selects and a bit-reverse per item (S1GenBlockedT).  A thread's items cross at
// most one read boundary (>= 8 slots per read): a second pair (W, R) for the start of the next read.  Three words per window
// instead of four.  Same records, bit for bit.
__device__ __forceinline__ void s1_item_from_parts(uint64_t f, uint64_t rc, unsigned head_b, unsigned tail_b, uint32_t q, int forced, uint32_t L, int k,
                                                   uint64_t a, uint64_t pos_base, uint32_t pos_bits, uint32_t (&out)[3]) {
  const unsigned head = q >= 1 ? head_b : kSentinel;
  const unsigned tail = q + k - 1 < L ? tail_b : kSentinel;
  int strand;
  if (forced >= 0) strand = forced;
  else strand = f > rc ? 1 : (f < rc ? 0 : (head <= 3 - tail ? 0 : 1));
  const uint64_t key = strand ? (rc | (comp_or_sentinel(tail) << 3) | comp_or_sentinel(head)) : (f | (head << 3) | tail);
  const uint64_t p = pos_base + a;
  out[0] = (uint32_t)(key >> 32);
  out[1] = (uint32_t)key | s1_pos_tag(p, pos_bits);
  out[2] = s1_pos_word(p, pos_bits);
}
constexpr int kS1RollMaxK = 23;
template <bool FILTER>
struct S1GenRollT {
  const uint32_t *seq;
  uint32_t L, per;
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;  // items of a tile (256 NI) divided by the slots per read: quotient and remainder
  const uint32_t *keep;
  static constexpr bool kMayDrop = FILTER;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return !FILTER || (r.w[1] & 63u) != 63u; }
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)((w * kWave + lane) * NI + j);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
    static_assert(NI <= 8, "a run of NI items and their flanks inside one 32-base window");
    constexpr uint32_t kTileItems = kSortThreads * NI;
    const uint64_t g00 = unit_base + (uint64_t)((w * kWave + lane) * NI);
    const uint32_t qlast = L - k + 1, jf = L - k + 2;
    const int km1 = k - 1;
    const uint64_t kmask = ~0ull << (64 - 2 * km1);
    uint32_t jt[UT];   // slot of the thread's first item in tile t
    uint64_t bt[UT];   // first base of that item's read
    {
      const uint64_t r = g00 / per;
      jt[0] = (uint32_t)(g00 - r * per);
      bt[0] = r * L;
#pragma unroll
      for (int t = 1; t < UT; ++t) {
        uint32_t jn = jt[t - 1] + tile_r;
        uint64_t bn = bt[t - 1] + (uint64_t)tile_q * L;
        if (jn >= per) {
          jn -= per;
          bn += L;
        }
        jt[t] = jn;
        bt[t] = bn;
      }
#pragma unroll
      for (int t = 0; t < UT; ++t)
        if (g00 + (uint64_t)t * kTileItems >= n) {  // nothing of this tile is this thread's: loads from the start of the store, nothing kept
          jt[t] = 0;
          bt[t] = 0;
        }
    }
    uint32_t c[UT][3], nx[UT][3];
    uint32_t q0t[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      q0t[t] = min(jt[t] > 0 ? jt[t] - 1 : 0u, qlast);
      const uint64_t a0 = bt[t] + q0t[t], b0 = a0 >= 2 ? a0 - 2 : 0;
      const uint64_t wcur = b0 >> 4, wnext = (bt[t] + L - 2) >> 4;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        c[t][x] = seq[wcur + x];
        nx[t][x] = seq[wnext + x];  // (the store is padded: also behind the last read)
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint64_t g0 = g00 + (uint64_t)t * kTileItems;
      uint32_t j = jt[t];
      uint64_t base = bt[t];
      // the window of the run that starts at the thread's first item: from two bases in front of its (k-1)-mer (the store's first
      // two bases: the window at base 0 shifted down, s1_fast_item) ...
      const uint64_t a0 = base + q0t[t], b0 = a0 >= 2 ? a0 - 2 : 0;
      const unsigned sh0 = (unsigned)(b0 & 15) * 2, down0 = a0 >= 2 ? 0u : (unsigned)(2 - a0) * 2;
      uint64_t W = (((uint64_t)funnel_l(c[t][0], c[t][1], sh0) << 32) | funnel_l(c[t][1], c[t][2], sh0)) >> down0;
      uint64_t R = rc64(W, 32);
      uint32_t qrun = q0t[t];
      // ... and of the run at the start of the next read (slot 0: offset 0)
      const unsigned shn = (unsigned)((base + L - 2) & 15) * 2;
      const uint64_t Wn = ((uint64_t)funnel_l(nx[t][0], nx[t][1], shn) << 32) | funnel_l(nx[t][1], nx[t][2], shn);
      const uint64_t Rn = rc64(Wn, 32);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t q = min(j > 0 ? j - 1 : 0u, qlast);
        const int forced = j < 2 ? (int)j : (j >= jf ? (int)(j - jf) : -1);
        const unsigned d2 = (q - qrun) * 2;  // offset inside the run's window, in bits
        const uint64_t f = (W << (d2 + 4)) & kmask;
        const uint64_t rc = (R << ((unsigned)(2 * (30 - km1)) - d2)) & kmask;
        const unsigned head_b = (unsigned)(W >> (60 - d2)) & 3u, tail_b = (unsigned)(W >> ((unsigned)(58 - 2 * km1) - d2)) & 3u;
        uint32_t out[3];
        s1_item_from_parts(f, rc, head_b, tail_b, q, forced, L, k, base + q, pos_base, pos_bits, out);
        if constexpr (FILTER)
          if (!s1_bucket_kept(keep, out[0])) out[1] = kS1Dropped;
        if (g0 + (uint64_t)i < n) {
          rec[t][i].w[0] = out[0];
          rec[t][i].w[1] = out[1];
          rec[t][i].w[2] = out[2];
        }
        if (++j == per) {
          j = 0;
          base += L;
          W = Wn;
          R = Rn;
          qrun = 0;
        }
      }
    }
  }
};

// Libraries whose reads are NOT of one length (trimmed reads: every real library) on the same generating pass (round 5).  The item
// index space is padded, not the store: every read gets per = max_len - k + 4 item slots, slot j of read r is item (r, j), and the
// slots a shorter read does not fill are declined (Src::kMayDrop — the mechanism of the bucket filter: the pass compacts what it
// keeps).  The read's place in the store comes from start[] (three 8-byte loads per thread and tile: this read, the next, the
// one after), its slot -> offset mapping from its own length.  Otherwise S1GenRollT: one window + one reverse complement per
// run.  Costs the slots that are dropped ((max_len - mean_len) / per of them) — the host takes this form while at least half of
// the slots are real.  Same records as k_s1_extract, bit for bit (read_to_sdbg_s1.cpp:228-292 serves any mix of lengths,
// sequence_package.h:131-164).
struct S1ReadGeo {
  uint64_t base;   // first base of the read in the store
  uint32_t L;      // its length
  uint32_t qlast;  // last offset of a (k-1)-mer
  uint32_t jf;     // first slot of the forced pair at the read's end
  uint32_t cnt;    // item slots the read fills (0: shorter than k + 1)
};
__device__ __forceinline__ S1ReadGeo s1_read_geo(uint64_t base, uint64_t next_base, int k) {
  S1ReadGeo g;
  g.base = base;
  g.L = (uint32_t)(next_base - base);
  const bool any = g.L >= (uint32_t)k + 1;
  g.qlast = any ? g.L - k + 1 : 0u;
  g.jf = any ? g.L - k + 2 : 0xFFFFFFFFu;
  g.cnt = any ? g.L - k + 4 : 0u;
  return g;
}
// the 32-base window that starts two bases in front of base a (the store's first two bases: the window at base 0 shifted down)
__device__ __forceinline__ void s1_window_addr(uint64_t a, uint64_t &word, unsigned &sh, unsigned &down) {
  const uint64_t b = a >= 2 ? a - 2 : 0;
  word = b >> 4;
  sh = (unsigned)(b & 15) * 2;
  down = a >= 2 ? 0u : (unsigned)(2 - a) * 2;
}
template <bool FILTER>
struct S1GenVarT {
  const uint32_t *seq;
  const uint64_t *start;  // [n_seqs + 1]
  uint64_t n_seqs;
  uint32_t per;           // item slots per read: max_len - k + 4
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;  // items of a tile (256 NI) divided by the slots per read: quotient and remainder
  const uint32_t *keep;
  static constexpr bool kMayDrop = true;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return (r.w[1] & 63u) != 63u; }
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)((w * kWave + lane) * NI + j);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
    static_assert(NI <= 8, "a run of NI items and their flanks inside one 32-base window");
    constexpr uint32_t kTileItems = kSortThreads * NI;
    const uint64_t g00 = unit_base + (uint64_t)((w * kWave + lane) * NI);
    const int km1 = k - 1;
    const uint64_t kmask = ~0ull << (64 - 2 * km1);
    uint32_t jt[UT];   // slot of the thread's first item in tile t
    uint64_t rt[UT];   // its read
    {
      const uint64_t r = g00 / per;
      jt[0] = (uint32_t)(g00 - r * per);
      rt[0] = r;
#pragma unroll
      for (int t = 1; t < UT; ++t) {
        uint32_t jn = jt[t - 1] + tile_r;
        uint64_t rn = rt[t - 1] + tile_q;
        if (jn >= per) {
          jn -= per;
          ++rn;
        }
        jt[t] = jn;
        rt[t] = rn;
      }
#pragma unroll
      for (int t = 0; t < UT; ++t)
        if (g00 + (uint64_t)t * kTileItems >= n) {  // nothing of this tile is this thread's: loads from the start of the store, nothing kept
          jt[t] = 0;
          rt[t] = 0;
        }
    }
    uint64_t s0[UT], s1[UT], s2[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      s0[t] = start[rt[t]];
      s1[t] = start[rt[t] + 1];
      s2[t] = start[rt[t] + 2 < n_seqs ? rt[t] + 2 : n_seqs];
    }
    uint32_t c[UT][3], nx[UT][3];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const S1ReadGeo cur = s1_read_geo(s0[t], s1[t], k);
      const uint32_t q0 = min(jt[t] > 0 ? jt[t] - 1 : 0u, cur.qlast);
      uint64_t wcur, wnext;
      unsigned sh, down;
      s1_window_addr(cur.base + q0, wcur, sh, down);
      s1_window_addr(s1[t], wnext, sh, down);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        c[t][x] = seq[wcur + x];
        nx[t][x] = seq[wnext + x];  // (the store is padded: also behind the last read)
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint64_t g0 = g00 + (uint64_t)t * kTileItems;
      uint32_t j = jt[t];
      S1ReadGeo rd = s1_read_geo(s0[t], s1[t], k);
      const S1ReadGeo rdn = s1_read_geo(s1[t], s2[t], k);
      uint32_t qrun = min(j > 0 ? j - 1 : 0u, rd.qlast);
      uint64_t wd;
      unsigned sh0, down0, shn, downn;
      s1_window_addr(rd.base + qrun, wd, sh0, down0);
      uint64_t W = (((uint64_t)funnel_l(c[t][0], c[t][1], sh0) << 32) | funnel_l(c[t][1], c[t][2], sh0)) >> down0;
      uint64_t R = rc64(W, 32);
      s1_window_addr(rdn.base, wd, shn, downn);
      const uint64_t Wn = (((uint64_t)funnel_l(nx[t][0], nx[t][1], shn) << 32) | funnel_l(nx[t][1], nx[t][2], shn)) >> downn;
      const uint64_t Rn = rc64(Wn, 32);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t q = min(j > 0 ? j - 1 : 0u, rd.qlast);
        const int forced = j < 2 ? (int)j : (j >= rd.jf ? (int)(j - rd.jf) : -1);
        const unsigned d2 = (q - qrun) * 2;  // offset inside the run's window, in bits
        const uint64_t f = (W << (d2 + 4)) & kmask;
        const uint64_t rc = (R << ((unsigned)(2 * (30 - km1)) - d2)) & kmask;
        const unsigned head_b = (unsigned)(W >> (60 - d2)) & 3u, tail_b = (unsigned)(W >> ((unsigned)(58 - 2 * km1) - d2)) & 3u;
        uint32_t out[3];
        s1_item_from_parts(f, rc, head_b, tail_b, q, forced, rd.L, k, rd.base + q, pos_base, pos_bits, out);
        if (j >= rd.cnt) out[1] = kS1Dropped;  // a slot this read does not fill
        if constexpr (FILTER)
          if (!s1_bucket_kept(keep, out[0])) out[1] = kS1Dropped;
        if (g0 + (uint64_t)i < n) {
          rec[t][i].w[0] = out[0];
          rec[t][i].w[1] = out[1];
          rec[t][i].w[2] = out[2];
        }
        if (++j == per) {
          j = 0;
          rd = rdn;
          W = Wn;
          R = Rn;
          qrun = 0;
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// `count` on the design of stage 1 (round 5): KmerCounter's lv2 items (kmer_counter.cpp:208-252) as 12-byte records made by the
// first sort pass — word 0..1: the canonical (k+1)-mer in the top 2(k+1) bits, bits [7, 15) of word 1 the position tag, bit 6
// the strand, bits [0, 6) prev / next as the reference packs them (complemented and swapped on the reverse strand); word 2 the
// low 32 bits of the edge's global offset.  One 64-bit window W per run of a thread's eight consecutive items (prev | edge |
// next = k + 3 bases from one base in front of the edge: k <= 22) and one reverse complement R = rc(W); item d of the run:
// forward (W << (2 d + 2)) & mask, reverse complement (R << 2 (30 - k - d)) & mask.  Fixed-length reads, >= 8 items per read.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCountStreamMaxK = 22;
constexpr uint32_t kCountStrandBit = 64u;
__device__ __forceinline__ uint32_t count_pos_tag(uint64_t p, uint32_t pos_bits) { return (uint32_t)(p >> pos_bits) << 7; }
// the 32-base window that starts ONE base in front of base a (the store's first base: the window at base 0 shifted down)
__device__ __forceinline__ void count_window_addr(uint64_t a, uint64_t &word, unsigned &sh, unsigned &down) {
  const uint64_t b = a >= 1 ? a - 1 : 0;
  word = b >> 4;
  sh = (unsigned)(b & 15) * 2;
  down = a >= 1 ? 0u : 2u;
}
__device__ __forceinline__ void count_item_from_parts(uint64_t f, uint64_t rc, unsigned prev_b, unsigned next_b, uint32_t p, uint32_t L, int k, uint64_t a,
                                                      uint64_t pos_base, uint32_t pos_bits, uint32_t (&out)[3]) {
  const unsigned prev = p > 0 ? prev_b : kSentinel;
  const unsigned next = p + k + 1 < L ? next_b : kSentinel;
  const bool strand = rc < f;  // rev_edge.cmp(edge) < 0, kmer_counter.cpp:179
  const uint64_t key = strand ? (rc | kCountStrandBit | (comp_or_sentinel(next) << 3) | comp_or_sentinel(prev)) : (f | (prev << 3) | next);
  const uint64_t g = pos_base + a;
  out[0] = (uint32_t)(key >> 32);
  out[1] = (uint32_t)key | count_pos_tag(g, pos_bits);
  out[2] = s1_pos_word(g, pos_bits);
}
struct CountGenT {
  const uint32_t *seq;
  uint32_t L, per;  // per = L - k items per read
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;
  static constexpr bool kMayDrop = false;
  __device__ __forceinline__ bool is_record(const Rec<3> &) const { return true; }
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)((w * kWave + lane) * NI + j);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
    static_assert(NI <= 8, "a run of NI edges and their flanks inside one 32-base window");
    constexpr uint32_t kTileItems = kSortThreads * NI;
    const uint64_t g00 = unit_base + (uint64_t)((w * kWave + lane) * NI);
    const uint64_t emask = ~0ull << (64 - 2 * (k + 1));
    const unsigned rsh = (unsigned)(2 * (30 - k));
    uint32_t jt[UT];
    uint64_t bt[UT];
    {
      const uint64_t r = g00 / per;
      jt[0] = (uint32_t)(g00 - r * per);
      bt[0] = r * L;
#pragma unroll
      for (int t = 1; t < UT; ++t) {
        uint32_t jn = jt[t - 1] + tile_r;
        uint64_t bn = bt[t - 1] + (uint64_t)tile_q * L;
        if (jn >= per) {
          jn -= per;
          bn += L;
        }
        jt[t] = jn;
        bt[t] = bn;
      }
#pragma unroll
      for (int t = 0; t < UT; ++t)
        if (g00 + (uint64_t)t * kTileItems >= n) {
          jt[t] = 0;
          bt[t] = 0;
        }
    }
    uint32_t c[UT][3], nx[UT][3];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      uint64_t wcur, wnext;
      unsigned sh, down;
      count_window_addr(bt[t] + jt[t], wcur, sh, down);
      count_window_addr(bt[t] + L, wnext, sh, down);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        c[t][x] = seq[wcur + x];
        nx[t][x] = seq[wnext + x];  // (the store is padded: also behind the last read)
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint64_t g0 = g00 + (uint64_t)t * kTileItems;
      uint32_t j = jt[t];
      uint64_t base = bt[t];
      uint64_t wd;
      unsigned sh0, down0, shn, downn;
      count_window_addr(base + j, wd, sh0, down0);
      uint64_t W = (((uint64_t)funnel_l(c[t][0], c[t][1], sh0) << 32) | funnel_l(c[t][1], c[t][2], sh0)) >> down0;
      uint64_t R = rc64(W, 32);
      uint32_t prun = j;
      count_window_addr(base + L, wd, shn, downn);
      const uint64_t Wn = (((uint64_t)funnel_l(nx[t][0], nx[t][1], shn) << 32) | funnel_l(nx[t][1], nx[t][2], shn)) >> downn;
      const uint64_t Rn = rc64(Wn, 32);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const unsigned d2 = (j - prun) * 2;
        const uint64_t f = (W << (d2 + 2)) & emask;
        const uint64_t rc = (R << (rsh - d2)) & emask;
        const unsigned prev_b = (unsigned)(W >> (62 - d2)) & 3u, next_b = (unsigned)(W >> ((unsigned)(58 - 2 * k) - d2)) & 3u;
        uint32_t out[3];
        count_item_from_parts(f, rc, prev_b, next_b, j, L, k, base + j, pos_base, pos_bits, out);
        if (g0 + (uint64_t)i < n) {
          rec[t][i].w[0] = out[0];
          rec[t][i].w[1] = out[1];
          rec[t][i].w[2] = out[2];
        }
        if (++j == per) {
          j = 0;
          base += L;
          W = Wn;
          R = Rn;
          prun = 0;
        }
      }
    }
  }
};
// the same records from a library whose reads are not of one length: `per` = max_len - k item slots per read, the slots a shorter read
// does not fill declined (S1GenVarT's scheme)
struct CountGenVarT {
  const uint32_t *seq;
  const uint64_t *start;  // [n_seqs + 1]
  uint64_t n_seqs;
  uint32_t per;
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;
  static constexpr bool kMayDrop = true;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return (r.w[1] & 63u) != 63u; }
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)((w * kWave + lane) * NI + j);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
    static_assert(NI <= 8, "a run of NI edges and their flanks inside one 32-base window");
    constexpr uint32_t kTileItems = kSortThreads * NI;
    const uint64_t g00 = unit_base + (uint64_t)((w * kWave + lane) * NI);
    const uint64_t emask = ~0ull << (64 - 2 * (k + 1));
    const unsigned rsh = (unsigned)(2 * (30 - k));
    uint32_t jt[UT];
    uint64_t rt[UT];
    {
      const uint64_t r = g00 / per;
      jt[0] = (uint32_t)(g00 - r * per);
      rt[0] = r;
#pragma unroll
      for (int t = 1; t < UT; ++t) {
        uint32_t jn = jt[t - 1] + tile_r;
        uint64_t rn = rt[t - 1] + tile_q;
        if (jn >= per) {
          jn -= per;
          ++rn;
        }
        jt[t] = jn;
        rt[t] = rn;
      }
#pragma unroll
      for (int t = 0; t < UT; ++t)
        if (g00 + (uint64_t)t * kTileItems >= n) {
          jt[t] = 0;
          rt[t] = 0;
        }
    }
    uint64_t s0[UT], s1[UT], s2[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      s0[t] = start[rt[t]];
      s1[t] = start[rt[t] + 1];
      s2[t] = start[rt[t] + 2 < n_seqs ? rt[t] + 2 : n_seqs];
    }
    uint32_t c[UT][3], nx[UT][3];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint32_t L = (uint32_t)(s1[t] - s0[t]), cnt = L >= (uint32_t)k + 1 ? L - k : 0u;
      const uint32_t j0 = min(jt[t], cnt ? cnt - 1 : 0u);  // (a run of declined slots only: any window of the read will do)
      uint64_t wcur, wnext;
      unsigned sh, down;
      count_window_addr(s0[t] + j0, wcur, sh, down);
      count_window_addr(s1[t], wnext, sh, down);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        c[t][x] = seq[wcur + x];
        nx[t][x] = seq[wnext + x];
      }
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint64_t g0 = g00 + (uint64_t)t * kTileItems;
      uint32_t j = jt[t];
      uint64_t base = s0[t];
      uint32_t L = (uint32_t)(s1[t] - s0[t]), cnt = L >= (uint32_t)k + 1 ? L - k : 0u;
      const uint32_t Ln = (uint32_t)(s2[t] - s1[t]), cntn = Ln >= (uint32_t)k + 1 ? Ln - k : 0u;
      uint64_t wd;
      unsigned sh0, down0, shn, downn;
      count_window_addr(base + min(j, cnt ? cnt - 1 : 0u), wd, sh0, down0);
      uint64_t W = (((uint64_t)funnel_l(c[t][0], c[t][1], sh0) << 32) | funnel_l(c[t][1], c[t][2], sh0)) >> down0;
      uint64_t R = rc64(W, 32);
      uint32_t prun = j;
      count_window_addr(s1[t], wd, shn, downn);
      const uint64_t Wn = (((uint64_t)funnel_l(nx[t][0], nx[t][1], shn) << 32) | funnel_l(nx[t][1], nx[t][2], shn)) >> downn;
      const uint64_t Rn = rc64(Wn, 32);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const unsigned d2 = (j - prun) * 2;
        const uint64_t f = (W << (d2 + 2)) & emask;
        const uint64_t rc = (R << (rsh - d2)) & emask;
        const unsigned prev_b = (unsigned)(W >> (62 - d2)) & 3u, next_b = (unsigned)(W >> ((unsigned)(58 - 2 * k) - d2)) & 3u;
        uint32_t out[3];
        count_item_from_parts(f, rc, prev_b, next_b, j, L, k, base + j, pos_base, pos_bits, out);
        if (j >= cnt) out[1] = kS1Dropped;  // a slot this read does not fill
        if (g0 + (uint64_t)i < n) {
          rec[t][i].w[0] = out[0];
          rec[t][i].w[1] = out[1];
          rec[t][i].w[2] = out[2];
        }
        if (++j == per) {
          j = 0;
          base = s1[t];
          L = Ln;
          cnt = cntn;
          W = Wn;
          R = Rn;
          prun = 0;
        }
      }
    }
  }
};

constexpr int kFastPasses = 4;
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
struct HiDigits {
  unsigned sh[kFastPasses], mk[kFastPasses];
  int n;
};
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
template <int IT, int NP, bool VAR = false>  // VAR: reads of any length, `per` = max_len - k item slots each (CountGenVarT)
__global__ __launch_bounds__(256) void k_count_digit_hist_roll(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                               HiDigits hd, unsigned long long *__restrict__ ghist, uint32_t step_q, uint32_t step_r,
                                                               const uint64_t *__restrict__ start, uint64_t n_seqs) {
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
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const unsigned d2 = (j - prun) * 2;
      const uint64_t f = (W << (d2 + 2)) & emask;
      const uint64_t rc = (R << (rsh - d2)) & emask;
      const uint32_t hi = (uint32_t)((rc < f ? rc : f) >> 32);
      if (g0 + u < n_items && j < cnt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) atomicAdd(&h[p][wv][(hi >> hd.sh[p]) & hd.mk[p]], 1u);
      }
      if (++j == per) {
        j = 0;
        cnt = cntn;
        W = Wn;
        R = Rn;
        prun = 0;
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
static bool s1_shape_is_var_fast(const mhx_ctx *c, uint32_t k, bool compact);
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

constexpr int kS1LocalHist = 1024;

template <int S>
struct S1Tile {
#ifndef MHX_S1_TILE
#define MHX_S1_TILE 2048
#endif
  static constexpr int kRaw = 32768 / (S * 4);
  static constexpr int kT = kRaw >= MHX_S1_TILE ? MHX_S1_TILE : (kRaw >= 256 ? (kRaw / 256) * 256 : 256);
  static constexpr int kRuns = kT + kMaxTailRuns;
};

__device__ __forceinline__ uint32_t *s1_local_hist() {
  __shared__ uint32_t lh[kS1LocalHist];
  return lh;
}
__device__ __forceinline__ unsigned long long *s1_block_solid() {
  __shared__ unsigned long long v;
  return &v;
}
template <int S>
__device__ __forceinline__ uint32_t *s1_run_info() {  // bit0 solid | has_in<<1 | has_out<<5 | l_has_out<<9 | r_has_in<<13
  __shared__ uint32_t ri[S1Tile<S>::kRuns];
  return ri;
}

// Lv2Postprocess of Read2SdbgS1 (read_to_sdbg_s1.cpp:368-555) as a tile operator (tile_groups.h):
// run = records of one (k-1)-mer with the same (head,tail); the per-group logic iterates runs, the
// per-record actions (is_solid bits, mercy candidates) are item-parallel.  No ordered output.
// 64-bit helpers for the aggregated stage-2 items (k <= 22: a (k+1)-mer and the 20 flag/W/count bits fit 64 bits)
__device__ __forceinline__ uint64_t rc64(uint64_t x, int n) {  // reverse complement of the n chars in the top 2n bits
  uint64_t r = __builtin_bitreverse64(x);
  r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
  return (~r) << (64 - 2 * n);
}

// AGG: besides marking, every solid (head,S,tail) run emits the stage-2 items of its (k+1)-mer ONCE, with the
// run length as multiplicity, instead of stage 2 emitting them once per occurrence (read_to_sdbg_s2.cpp:398-409
// emits "solid" items per occurrence and Lv2Postprocess :579 counts them again): same records, ~8x fewer items
// to sort.  Item = seq2sdbg layout (k chars | full<<19 | W<<16 | count).
template <int S, bool COMPACT, bool AGG>
struct S1Op {
  static constexpr bool kItemPhase = false, kItemFinal = true, kRunPhase = false, kUnitIsRun = false, kAtomicBase = AGG;
  __device__ void run_phase(const TileCtx<S> &, uint32_t, uint32_t) const {}
  int k;
  uint2 *agg_items;
  int kw;
  uint32_t m;
  const uint64_t *start;
  uint64_t n_seqs;
  uint32_t fixed_len;
  uint8_t *solid_bytes;  // one byte per base position (plain stores, packed to the bitmap afterwards)
  unsigned long long *solid_bits;  // or: the bitmap itself, set with atomics (mark_atomic)
  int mark_atomic;
  // mark_mode 0: mark solid occurrences; 1: mark the NON-solid ones (fewer scattered stores when most are solid;
  // k_pack_solid_inv turns "valid position and not marked" into is_solid); 2: statistics only (sampled tiles)
  int mark_mode;
  unsigned long long *hist, *n_solid_out;
  int want_mercy;
  long long *mercy;
  unsigned long long *mercy_n;
  uint64_t pos_stride;  // compact records tagged with their source rank: global position = local + rank * pos_stride (else 0)
  // mercy candidates go to a region of the workgroup's own, mercy[mercy_off[blockIdx.x] ...], counted in
  // mercy_counts[blockIdx.x]: 5 x 10^7 candidates at 10 M reads meant ~2 x 10^7 wavefront-level atomics on ONE global word,
  // ~10 ns each = the 190 ms of this kernel in round 2.  A region holds two entries per record of the workgroup's tiles
  // (tile indices blockIdx.x, + gridDim.x, ...).  A workgroup also handles the tail of its last group beyond its tile, so
  // in theory it can meet more candidates than its region holds: then it sets mercy_counts[gridDim.x] and the host runs the
  // kernel again with the shared cursor.  nullptr: the shared cursor mercy_n.
  uint32_t *mercy_counts;
  const uint64_t *mercy_off;

  __device__ bool same_run(const uint32_t *cur, const uint32_t *prev) const { return ((cur[kw - 1] ^ prev[kw - 1]) & 63u) == 0; }
  __device__ bool item_phase_enabled() const { return false; }
  __device__ bool item_final_enabled() const { return mark_mode != 2; }
  __device__ void item_phase(const TileCtx<S> &, uint32_t, uint32_t) const {}
  __device__ void begin_block() const {
    uint32_t *lh = s1_local_hist();
    for (int i = threadIdx.x; i < kS1LocalHist; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x == 0) *s1_block_solid() = 0;
    __syncthreads();
  }
  __device__ void end_block() const {
    if (mark_mode == 2) {  // sampled statistics: [0] solid occurrences, [2] occurrences with head and tail
      if (threadIdx.x == 0 && *s1_block_solid()) atomicAdd(n_solid_out, *s1_block_solid());
      return;
    }
    uint32_t *lh = s1_local_hist();
    for (int i = threadIdx.x; i < kS1LocalHist; i += blockDim.x)
      if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
  }
  // the (k+1)-mer head.S.tail of a run, chars MSB-first in 64 bits
  __device__ __forceinline__ uint64_t edge_of(const TileCtx<S> &c, uint32_t i, unsigned h, unsigned t) const {
    const uint64_t key = ((uint64_t)c.acc.word(i, 0) << 32) | c.acc.word(i, 1);
    const uint64_t smer = key & (~0ull << (64 - 2 * (k - 1)));  // the (k-1)-mer, head/tail bits dropped
    return ((uint64_t)h << 62) | (smer >> 2) | ((uint64_t)t << (62 - 2 * k));
  }
  __device__ void unit_emit(const TileCtx<S> &c, uint32_t g, uint64_t o0, uint64_t, uint64_t) const {
    if constexpr (AGG) {
      const uint32_t r0 = c.gpos[g], r1 = c.gpos[g + 1];
      const uint64_t mask_k = ~0ull << (64 - 2 * k);
      for (uint32_t r = r0; r < r1; ++r) {
        if (!(s1_run_info<S>()[r] & 1u)) continue;
        const uint32_t i = c.run_start(r);
        const unsigned ht = c.acc.word(i, kw - 1) & 63u, h = ht >> 3, t = ht & 7;
        const uint32_t n = c.run_len(r);
        const uint64_t cnt = n > MHX_MAX_MUL ? (uint64_t)MHX_MAX_MUL : n;
        const uint64_t x = edge_of(c, i, h, t), xr = rc64(x, k + 1);
        const uint64_t f = ((x << 2) & mask_k) | (1ull << 19) | ((x >> 62) << 16) | cnt;   // k-mer x[1..k], W = x[0]
        agg_items[o0++] = make_uint2((uint32_t)(f >> 32), (uint32_t)f);
        if (x != xr) {  // palindromic (k+1)-mers emit the forward item only (:385-423)
          const uint64_t b = ((xr << 2) & mask_k) | (1ull << 19) | ((xr >> 62) << 16) | cnt;
          agg_items[o0++] = make_uint2((uint32_t)(b >> 32), (uint32_t)b);
        }
      }
    }
  }
  __device__ GroupCounts unit_count(const TileCtx<S> &c, uint32_t g) const {
    const uint32_t r0 = c.gpos[g], r1 = c.gpos[g + 1];
    // H1: prev/next of the group's FIRST item, :399 (compact records carry none: only mercy needs has_in/has_out)
    const unsigned pn_first = COMPACT ? 0u : (c.acc.word(c.run_start(r0), kw + 1) & 63u);
    uint64_t cnt_head[4] = {0, 0, 0, 0}, cnt_tail[4] = {0, 0, 0, 0};
    unsigned l_has_out = 0, r_has_in = 0;
    for (uint32_t r = r0; r < r1; ++r) {
      const unsigned ht = c.acc.word(c.run_start(r), kw - 1) & 63u, h = ht >> 3, t = ht & 7;
      const uint32_t n = c.run_len(r);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (h == (unsigned)x) cnt_head[x] += n;
        if (t == (unsigned)x) cnt_tail[x] += n;
      }
      if (h < 4 && t < 4 && n >= m) {
        l_has_out |= 1u << h;
        r_has_in |= 1u << t;
      }
    }
    unsigned has_in = 0, has_out = 0;
    if ((pn_first >> 3) < 4) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (cnt_head[x] >= m) has_in |= 1u << x;
    }
    if ((pn_first & 7) < 4) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (cnt_tail[x] >= m) has_out |= 1u << x;
    }
    const uint32_t masks = (has_in << 1) | (has_out << 5) | (l_has_out << 9) | (r_has_in << 13);
    unsigned long long my_solid = 0, my_both = 0;
    uint32_t n_agg = 0;
    for (uint32_t r = r0; r < r1; ++r) {
      const unsigned ht = c.acc.word(c.run_start(r), kw - 1) & 63u, h = ht >> 3, t = ht & 7;
      const uint32_t n = c.run_len(r);
      const bool both = h < 4 && t < 4;
      const bool solid = both && n >= m;
      if (mark_mode == 2) {
        if (solid) my_solid += n;
        if (both) my_both += n;
        continue;
      }
      if (both) {
        const uint32_t hb = n > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : n;
        if (hb < kS1LocalHist) atomicAdd(&s1_local_hist()[hb], 1u);
        else atomicAdd(&hist[hb], 1ull);
      }
      if (solid) my_solid += n;
      s1_run_info<S>()[r] = masks | (solid ? 1u : 0u) | (both ? 1u << 17 : 0u);
      if constexpr (AGG) {
        if (solid && mark_mode != 2) {
          const uint64_t x = edge_of(c, c.run_start(r), h, t);
          n_agg += x == rc64(x, k + 1) ? 1u : 2u;
        }
      }
    }
    if (mark_mode == 2) {
      if (my_solid) atomicAdd(s1_block_solid(), my_solid);
      if (my_both) atomicAdd(n_solid_out + 2, my_both);
    }
    GroupCounts gc;
    gc.c0 = n_agg;
    return gc;
  }
  __device__ void item_final(const TileCtx<S> &c, uint32_t rel, uint32_t run) const {
    const uint32_t ri = s1_run_info<S>()[run];
    const bool solid = ri & 1u;
    const bool mark = mark_mode == 1 ? (!solid && (ri >> 17 & 1u)) : solid;
    if (!mark && !want_mercy) return;
    uint64_t abs;
    int strand = 0;
    if constexpr (COMPACT) abs = c.acc.word(rel, kw) + (uint64_t)((c.acc.word(rel, kw - 1) >> 6) & 0xFFu) * pos_stride;
    else {
      const uint64_t info = (((uint64_t)c.acc.word(rel, kw) << 32) | c.acc.word(rel, kw + 1)) >> 6;
      abs = info >> 1;
      strand = (int)(info & 1);
    }
    if (mark) {  // is_solid.set(pos-1), :464 (or its complement, see mark_mode)
      if (mark_atomic) atomicOr(reinterpret_cast<unsigned int *>(solid_bits) + ((abs - 1) >> 5), 1u << ((abs - 1) & 31));
      else solid_bytes[abs - 1] = 1;
    }
    if (!COMPACT && want_mercy) {
      const unsigned has_in = (ri >> 1) & 15u, has_out = (ri >> 5) & 15u, l_has_out = (ri >> 9) & 15u, r_has_in = (ri >> 13) & 15u;
      const unsigned ht = c.acc.word(rel, kw - 1) & 63u, h = ht >> 3, t = ht & 7;
      // ((pkg_offset + l_offset) << 2 | flag, :466-551) with l_offset/r_offset = the item's offset in its read (+1 on
      // the far side): pkg_offset + offset = abs - 1, so the read itself is never looked up
      const long long base = 0, off = (long long)abs - 1;
      const long long l_off = strand == 0 ? off : off + 1, r_off = strand == 0 ? off + 1 : off;
      long long c0 = -1, c1 = -1;
      if (solid) {  // :466-483
        if (!(has_in & (1u << h))) c0 = ((base + l_off) << 2) | (1 + strand);
        if (!(has_out & (1u << t))) c1 = ((base + r_off) << 2) | (2 - strand);
      } else {      // :485-551 (head/tail may be '$' here: the masks only hold bits 0..3)
        if (l_has_out & (1u << h)) c0 = ((base + l_off) << 2) | ((has_in & (1u << h)) ? 0 : (1 + strand));
        else if (has_in & (1u << h)) c0 = ((base + l_off) << 2) | (2 - strand);
        if (r_has_in & (1u << t)) c1 = ((base + r_off) << 2) | ((has_out & (1u << t)) ? 0 : (2 - strand));
        else if (has_out & (1u << t)) c1 = ((base + r_off) << 2) | (1 + strand);
      }
      // one cursor atomic per wave, not per candidate (same-address global atomics cost ~10 ns each; the list is unordered)
      const unsigned long long m0 = __ballot(c0 >= 0), m1 = __ballot(c1 >= 0);
      if (m0 | m1) {
        const int lane = lane_id(), leader = __builtin_ctzll(m0 | m1);
        const unsigned n0 = (unsigned)__builtin_popcountll(m0);
        const unsigned n_all = n0 + (unsigned)__builtin_popcountll(m1);
        unsigned long long at = 0;
        bool ok = true;
        if (mercy_counts) {
          uint32_t a32 = 0;
          if (lane == leader) a32 = atomicAdd(mercy_counts + blockIdx.x, n_all);
          a32 = __shfl(a32, leader, kWave);
          at = mercy_off[blockIdx.x] + a32;
          ok = at + n_all <= mercy_off[blockIdx.x + 1];
          if (!ok && lane == leader) atomicOr(mercy_counts + gridDim.x, 1u);
        } else {
          if (lane == leader) at = atomicAdd(mercy_n, (unsigned long long)n_all);
          at = __shfl(at, leader, kWave);
        }
        const unsigned long long below = (1ull << lane) - 1;
        if (ok && c0 >= 0) mercy[at + __builtin_popcountll(m0 & below)] = c0;
        if (ok && c1 >= 0) mercy[at + n0 + __builtin_popcountll(m1 & below)] = c1;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Segment group-by: the no-mercy reduction of Read2SdbgS1::Lv2Postprocess (read_to_sdbg_s1.cpp:368-555) WITHOUT a
// full sort.  Without mercy candidates the reduction only needs, per distinct key (k-1)-mer|head|tail, the number of
// records carrying it (:430-464: histogram, count >= m -> is_solid.set per occurrence) — not their order.  So the
// records are radix-sorted on the top `prefix` bits of the (k-1)-mer only (half the LSD passes at k=21), which
// leaves every key inside one contiguous SEGMENT of equal prefix (~100 records on average), and one workgroup
// counts the equal keys of the segments of its tile in an LDS hash table (64-bit compare-and-swap + counter):
//   insert   every record of the tile (and of the look-ahead that completes its last segment) -> slot, count++
//   marks    per record: count of its slot -> solid? -> byte-map store                          (item-parallel)
//   slots    per occupied slot = per distinct key: histogram, aggregated stage-2 items           (key-parallel)
// A segment belongs to the tile that holds its first record: records of the tile that continue the previous tile's
// last segment (prefix == that of the record before the tile) are inserted but neither marked nor emitted, records
// behind the tile with the prefix of its last record are fetched until the prefix changes.  No head flags, no scans,
// three barriers per tile.  A tile whose last segment outgrows the look-ahead or whose keys overflow the table sets
// *err and does nothing; the host then falls back to the full sort + k_tile_groups (same results).
// ---------------------------------------------------------------------------------------------------------------
// Giant buckets of the bucket streaming (round 5).  A workgroup streams a bucket alone, so ONE bucket of millions of records —
// low-complexity sequence: 1 % poly-A reads put 13 M records of one key into lv1 bucket 0 — held the whole stage up for 15 ms.
// Such a bucket (>= min_records) is cut into slices that many workgroups reduce in parallel (k_s1_giant_reduce: an LDS
// table per slice -> "partial entries" = a key's first record + its count in the slice), the streaming kernel skips it
// (flag[bucket]), and a second launch of the streaming kernel (GIANT) inserts the few partial entries with their counts
// and does the per-key work as for any bucket.  A bucket whose slices do not reduce into their region (many distinct keys)
// clears its flag and is streamed as before.  Everything is found and sized on the device: no host round trip.
struct S1Giant {
  uint8_t *flag;            // [n_buckets] 1: taken by the giant path
  uint32_t *ctr;            // [0] giants found (may exceed gcap)  [2..3] partial entries allotted (64-bit)
  uint32_t *bucket, *sl, *ns, *cap, *cur;  // per giant: bucket, slice length, slices, region capacity, entries written
  unsigned long long *off;  // per giant: first entry of its region in `partial`
  uint4 *partial;           // entries: the three words of a key's first record in the slice + its count there
  uint32_t gcap;            // giants the list holds
  uint32_t min_records;     // a bucket at least this large is a giant
  unsigned long long pcap;  // entries `partial` holds
};
constexpr uint32_t kGiantSliceMin = 16384, kGiantEntriesPerSlice = 256;

struct S1SegArgs {
  int k;
  uint32_t m;
  uint32_t pfx_mask;  // bits of key word 0 that form the segment prefix
  uint32_t eq_mask1;  // bits of key word 1 that take part in key equality: (k-1)-mer bits + head/tail (not the rank tag)
  uint8_t *solid_bytes;
  int mark_mode;      // 0: mark solid occurrences, 1: mark the non-solid ones, 2: statistics only
  unsigned long long *hist, *ctr;  // ctr[0] / ctr[2]: solid / head-and-tail occurrences (mark_mode 2)
  // aggregated stage-2 items: every (persistent) workgroup fills a region of its own, agg_raw[blockIdx.x * agg_cap ...],
  // and leaves its item count in agg_counts[blockIdx.x]; k_agg_compact packs the regions afterwards.  (A shared
  // cursor costs one same-address global atomic per wavefront and tile: ~10 ns each, 2.6 M of them at 10 M reads.)
  uint2 *agg_raw;
  uint32_t agg_cap;
  uint32_t *agg_counts;
  // multi-GPU, sparse marks: instead of a store into a byte map of the GLOBAL read set, a mark is the position itself,
  // appended to the workgroup's region marks_raw[blockIdx.x * marks_cap ...] (count in marks_counts[blockIdx.x]); the
  // host packs the regions and routes the positions to the ranks that hold those reads (comm.hip)
  unsigned long long *marks_raw;
  uint32_t marks_cap;
  uint32_t *marks_counts;
  uint64_t pos_stride;
  uint32_t *err;
  int la_chunks;      // look-ahead limit, in chunks of 256 records
  int direct_marks;   // k_s1_stream: the non-solid marks come from the table (one stored position per key), no second read
  S1Giant giant;      // k_s1_stream: buckets handed to the giant path (flag == nullptr: none)
  // k_s1_stream<COUNT>: the reads (first_0_out / last_0_in are per read) and the two arrays (kmer_counter.cpp:307-368)
  const uint64_t *c_start;
  uint64_t c_n_seqs;
  uint32_t c_fixed_len;
  uint32_t *first_0_out, *last_0_in_p1;
};

constexpr unsigned long long kSegEmpty = ~0ull;  // never a key: head/tail bits 63 do not occur (max (4<<3)|4)

constexpr int kSegHist = 512;  // multiplicities counted in LDS

template <int PER, bool AGG>
__global__ __launch_bounds__(256) void k_s1_seg(const uint32_t *__restrict__ items, uint64_t n, S1SegArgs a, uint64_t n_work,
                                                uint32_t tile_stride) {
  constexpr int T = 256 * PER;
  constexpr int NSLOT = 2 * T;
  constexpr int LOGS = PER == 8 ? 12 : (PER == 4 ? 11 : (PER == 16 ? 13 : 10));
  static_assert((1 << LOGS) == NSLOT, "table size");
  constexpr int NR = PER + 1;  // tile records + the first look-ahead chunk, per thread
  constexpr uint32_t kCreated = 0x80000000u;
  __shared__ unsigned long long keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT / 2];   // two 16-bit counters per word (a tile inserts < 65536 records)
  __shared__ uint16_t created[NSLOT];    // slots created by this tile = its distinct keys, in any order
  __shared__ uint32_t lhist[kSegHist];
  __shared__ uint32_t s_bad, s_ncreated, s_agg_cur, s_mark_cur;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  // the workgroup's output region (in the spare sort buffer): marks grow from its front, aggregated items from its back
  uint2 *const agg_end = AGG ? a.agg_raw + (size_t)(blockIdx.x + 1) * a.agg_cap : nullptr;
  for (int i = tid; i < NSLOT; i += 256) keys[i] = kSegEmpty;
  for (int i = tid; i < NSLOT / 2; i += 256) cnts[i] = 0;
  for (int i = tid; i < kSegHist; i += 256) lhist[i] = 0;
  if (tid == 0) {
    s_bad = 0;
    s_ncreated = 0;
    s_agg_cur = 0;
    s_mark_cur = 0;
  }
  __syncthreads();
  unsigned long long *const marks_out = a.marks_raw ? a.marks_raw + (size_t)blockIdx.x * a.marks_cap : nullptr;

  const uint32_t pfx = a.pfx_mask, eqm = a.eq_mask1, m = a.m;
  auto count_of = [&](uint32_t slot) -> uint32_t { return (cnts[slot >> 1] >> ((slot & 1u) * 16)) & 0xFFFFu; };
  auto count_add = [&](uint32_t slot, uint32_t mult) { atomicAdd(&cnts[slot >> 1], mult << ((slot & 1u) * 16)); };
  // probing insert -> slot | kCreated if this call created the slot (exactly one caller per distinct key does)
  auto insert = [&](uint32_t w0, uint32_t w1m, uint32_t mult, uint32_t h) -> uint32_t {
    const unsigned long long key = ((unsigned long long)w0 << 32) | w1m;
    for (int probes = 0; probes < 512; ++probes) {
      const unsigned long long old = atomicCAS(&keys[h], kSegEmpty, key);
      if (old == kSegEmpty || old == key) {
        count_add(h, mult);
        return h | (old == kSegEmpty ? kCreated : 0u);
      }
      h = (h + 1) & (NSLOT - 1);
    }
    s_bad = 1;  // table (nearly) full
    return 0;
  };
  auto lookup = [&](uint32_t w0, uint32_t w1m) -> uint32_t {
    const unsigned long long key = ((unsigned long long)w0 << 32) | w1m;
    uint32_t h = (w0 * 0x9E3779B1u + w1m * 0x85EBCA6Bu) >> (32 - LOGS);
    for (int probes = 0; probes < 512 && keys[h] != key; ++probes) h = (h + 1) & (NSLOT - 1);
    return h;
  };
  // convergent (every lane of the wavefront calls it; `mine` = this lane has a record of ours)
  auto mark = [&](bool mine, uint32_t w1, uint32_t w2, uint32_t cnt) {
    const bool both = (w1 & 0x24u) == 0;  // head < 4 and tail < 4
    const bool solid = both && cnt >= m;
    const bool mk = mine && (a.mark_mode == 1 ? (both && !solid) : solid);
    const uint64_t abs = w2 + (uint64_t)((w1 >> 6) & 0xFFu) * a.pos_stride;
    if (!marks_out) {
      if (mk) a.solid_bytes[abs - 1] = 1;  // is_solid.set(pos - 1), :464 (or its complement)
      return;
    }
    const uint64_t mm = __ballot(mk);
    if (!mm) return;
    uint32_t mbase = 0;
    if (lane == 0) mbase = atomicAdd(&s_mark_cur, (uint32_t)__builtin_popcountll(mm));
    mbase = __shfl(mbase, 0, kWave);
    if (mk) {
      const uint32_t at = mbase + (uint32_t)__builtin_popcountll(mm & lanemask_lt);
      if (at + s_agg_cur < a.marks_cap) marks_out[at] = abs - 1;
      else atomicOr(a.err, 2u);
    }
  };
  // the (k+1)-mer head.S.tail of a key, chars MSB-first in 64 bits
  auto edge_of = [&](unsigned long long key) -> uint64_t {
    const unsigned ht = (uint32_t)key & 63u;
    const uint64_t smer = key & (~0ull << (64 - 2 * (a.k - 1)));
    return ((uint64_t)(ht >> 3) << 62) | (smer >> 2) | ((uint64_t)(ht & 7) << (62 - 2 * a.k));
  };
  unsigned long long st_solid = 0, st_both = 0;

  // records of a tile in registers (striped: thread t holds records j*256 + t), prefetched one tile ahead together
  // with the three uniform words that decide segment ownership
  uint32_t nw0[NR], nw1[NR], nw2[NR];
  uint32_t n_prev = 0, n_last = 0, n_lalast = 0;
  auto prefetch = [&](uint64_t tile_idx) {
    const uint64_t base = tile_idx * tile_stride * T;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const uint64_t gi = base + (uint64_t)j * 256 + tid;
      if (gi < n) {
        const uint32_t *p = items + gi * 3;
        nw0[j] = p[0];
        nw1[j] = p[1];
        nw2[j] = p[2];
      }
    }
    const uint64_t tile_end = n - base < (uint64_t)T ? n : base + T;
    if (base) n_prev = items[(base - 1) * 3];
    n_last = items[(tile_end - 1) * 3];
    if (tile_end + 256 < n) n_lalast = items[(tile_end + 255) * 3];
  };
  if (blockIdx.x < n_work) prefetch(blockIdx.x);

  for (uint64_t tile_idx = blockIdx.x; tile_idx < n_work; tile_idx += gridDim.x) {
    const uint64_t base = tile_idx * tile_stride * T;
    const uint64_t tile_end = n - base < (uint64_t)T ? n : base + T;
    uint32_t w0[NR], w1[NR], w2[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      w0[j] = nw0[j];
      w1[j] = nw1[j];
      w2[j] = nw2[j];
    }
    const bool has_prev = base != 0;
    const uint32_t p_prev = n_prev & pfx, p_last = n_last & pfx;
    // the last segment starts in this tile (else the whole tile continues a segment of an earlier tile)
    const bool la_own = tile_end < n && !(has_prev && p_last == p_prev);
    const bool more = la_own && tile_end + 256 < n && (n_lalast & pfx) == p_last;  // it even outgrows the first look-ahead chunk
    if (tile_idx + gridDim.x < n_work) prefetch(tile_idx + gridDim.x);

    uint32_t slot[NR];
    bool own[NR];
    {
      // Equal keys sit next to each other (a segment holds a handful of distinct keys, the frequent ones dozens of
      // times), and the LDS serialises the lanes of one atomic that hit the same address.  So the lanes of a wavefront
      // first find their equals with a match-any over some hash bits (ballots), confirm against the group's first
      // lane, and only that lane inserts, adding the whole group's size; hash-equal lanes with a different key insert
      // on their own.  NB rounds at a time, phase by phase, so that the LDS round trips of a phase overlap.
      constexpr int NB = NR % 3 == 0 ? 3 : (NR % 5 == 0 ? 5 : 1);
      constexpr int MB = 7;  // match bits
#pragma unroll
      for (int j0 = 0; j0 < NR; j0 += NB) {
        bool ins[NB], eq[NB], doer[NB];
        int leader[NB];
        uint32_t hs[NB], mult[NB], km[NB];
        uint64_t peers[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int j = j0 + q;
          const uint64_t gi = base + (uint64_t)j * 256 + tid;
          if (j < PER) own[j] = gi < tile_end && !(has_prev && (w0[j] & pfx) == p_prev);
          else own[j] = la_own && gi < n && (w0[j] & pfx) == p_last;
          // records of the tile that are not ours are inserted as well (their prefix occurs nowhere else, so they
          // change no count of ours): no divergence on the common path
          ins[q] = j < PER ? gi < tile_end : own[j];
          km[q] = w1[j] & eqm;
          const uint32_t hf = w0[j] * 0x9E3779B1u + km[q] * 0x85EBCA6Bu;
          hs[q] = hf >> (32 - LOGS);
          const uint32_t hm = hf >> (32 - MB);
          uint64_t pm = __ballot(ins[q]);
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            const bool bit = (hm >> b) & 1u;
            const uint64_t mb = __ballot(bit);
            pm &= bit ? mb : ~mb;
          }
          peers[q] = pm;
          leader[q] = ins[q] ? __builtin_ctzll(pm) : lane;
        }
        uint32_t l0[NB], l1[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          l0[q] = __shfl(w0[j0 + q], leader[q], kWave);
          l1[q] = __shfl(km[q], leader[q], kWave);
        }
        unsigned long long old[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          eq[q] = ins[q] && l0[q] == w0[j0 + q] && l1[q] == km[q];
          const uint64_t grp = __ballot(eq[q]) & peers[q];
          doer[q] = ins[q] && (lane == leader[q] || !eq[q]);  // group leaders, and hash-equal lanes with another key
          mult[q] = lane == leader[q] ? (uint32_t)__builtin_popcountll(grp) : 1u;
          old[q] = 0;
          if (doer[q]) old[q] = atomicCAS(&keys[hs[q]], kSegEmpty, ((unsigned long long)w0[j0 + q] << 32) | km[q]);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int j = j0 + q;
          slot[j] = 0;
          if (doer[q]) {
            const unsigned long long key = ((unsigned long long)w0[j] << 32) | km[q];
            if (old[q] == kSegEmpty || old[q] == key) {
              count_add(hs[q], mult[q]);
              slot[j] = hs[q] | (old[q] == kSegEmpty ? kCreated : 0u);
            } else {  // first probe taken by another key: the probing loop
              slot[j] = insert(w0[j], km[q], mult[q], (hs[q] + 1) & (NSLOT - 1));
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int j = j0 + q;
          // the slots this round created go to the tile's list of distinct keys (one LDS cursor bump per wavefront)
          const bool cr = (slot[j] & kCreated) != 0;
          const uint64_t crm = __ballot(cr);
          uint32_t cbase = 0;
          if (lane == 0 && crm) cbase = atomicAdd(&s_ncreated, (uint32_t)__builtin_popcountll(crm));
          cbase = __shfl(cbase, 0, kWave);
          slot[j] &= ~kCreated;
          if (cr) created[cbase + __builtin_popcountll(crm & lanemask_lt)] = (uint16_t)slot[j];
          const uint32_t lslot = __shfl(slot[j], leader[q], kWave);
          if (eq[q] && lane != leader[q]) slot[j] = lslot;
        }
      }
    }
    if (more) {  // rare: further look-ahead chunks straight from HBM
      for (int c = 1;; ++c) {
        const uint64_t cb = tile_end + (uint64_t)c * 256;
        if (c > a.la_chunks) {
          s_bad = 1;
          break;
        }
        const uint64_t gi = cb + tid;
        if (gi < n) {
          const uint32_t *p = items + gi * 3;
          const uint32_t x0 = p[0], x1 = p[1] & eqm;
          if ((x0 & pfx) == p_last) {
            const uint32_t sl = insert(x0, x1, 1u, (x0 * 0x9E3779B1u + x1 * 0x85EBCA6Bu) >> (32 - LOGS));
            if (sl & kCreated) created[atomicAdd(&s_ncreated, 1u)] = (uint16_t)(sl & ~kCreated);
          }
        }
        if (!(cb + 256 < n && (items[(cb + 255) * 3] & pfx) == p_last)) break;
      }
    }
    __syncthreads();
    const bool bad = s_bad != 0;  // workgroup-uniform
    const uint32_t n_created = s_ncreated;
    uint32_t my_agg = 0;
    if (!bad) {
      if (a.mark_mode != 2) {
#pragma unroll
        for (int j = 0; j < NR; ++j) mark(own[j], w1[j], w2[j], count_of(own[j] ? slot[j] : 0u));
        if (more) {
          for (int c = 1; c <= a.la_chunks; ++c) {
            const uint64_t cb = tile_end + (uint64_t)c * 256;
            const uint64_t gi = cb + tid;
            uint32_t x0 = 0, x1 = 0, x2 = 0;
            if (gi < n) {
              const uint32_t *p = items + gi * 3;
              x0 = p[0];
              x1 = p[1];
              x2 = p[2];
            }
            const bool mine = gi < n && (x0 & pfx) == p_last;
            mark(mine, x1, x2, mine ? count_of(lookup(x0, x1 & eqm)) : 0u);
            if (!(cb + 256 < n && (items[(cb + 255) * 3] & pfx) == p_last)) break;
          }
        }
      }
      // per distinct key of ours (dense over the list of created slots): histogram, statistics, aggregated-item count
      for (uint32_t i = tid; i < n_created; i += 256) {
        const uint32_t sl = created[i];
        const unsigned long long key = keys[sl];
        if (has_prev && ((uint32_t)(key >> 32) & pfx) == p_prev) continue;  // a key of the previous tile's last segment
        if (((uint32_t)key & 0x24u) != 0) continue;                         // head or tail is '$'
        const uint32_t cnt = count_of(sl);
        const bool solid = cnt >= m;
        if (a.mark_mode == 2) {
          st_both += cnt;
          if (solid) st_solid += cnt;
          continue;
        }
        const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;  // :430-436
        if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
        else atomicAdd(&a.hist[hb], 1ull);
        if (AGG && solid) {
          const uint64_t x = edge_of(key);
          my_agg += x == rc64(x, a.k + 1) ? 1u : 2u;
        }
      }
    }
    uint32_t agg_at = 0;
    bool agg_ok = true;  // wavefront-uniform
    if constexpr (AGG) {
      // output order is irrelevant (stage 2 sorts): one bump of the workgroup's LDS cursor per wavefront
      const uint32_t incl = wave_inclusive_sum(my_agg);
      const uint32_t tot = __shfl(incl, kWave - 1, kWave);
      uint32_t wbase = 0;
      if (lane == 0 && tot) wbase = atomicAdd(&s_agg_cur, tot);
      wbase = __shfl(wbase, 0, kWave);
      // (marks in front, items at the back: a record yields a mark or a share of an item, never both, so the region —
      // 12 bytes per record of the workgroup — only overflows when the tiles are spread very unevenly; then: classic path)
      agg_ok = wbase + tot + (marks_out ? s_mark_cur : 0u) <= a.agg_cap;
      if (!agg_ok && lane == 0) atomicOr(a.err, 1u);
      agg_at = wbase + incl - my_agg;
    }
    __syncthreads();  // every count has been read: emit, then recycle the slots
    if (tid == 0) {     // (everyone has read these; the barrier below orders the reset before the next tile's inserts)
      s_bad = 0;
      s_ncreated = 0;
    }
    if (!bad) {
      for (uint32_t i = tid; i < n_created; i += 256) {
        const uint32_t sl = created[i];
        if constexpr (AGG) {
          const unsigned long long key = keys[sl];
          const uint32_t cnt = count_of(sl);
          const bool mine = !(has_prev && ((uint32_t)(key >> 32) & pfx) == p_prev);
          if (agg_ok && a.mark_mode != 2 && mine && ((uint32_t)key & 0x24u) == 0 && cnt >= m) {
            const int k = a.k;
            const uint64_t mask_k = ~0ull << (64 - 2 * k);
            const uint64_t x = edge_of(key), xr = rc64(x, k + 1);
            const uint64_t mul = cnt > MHX_MAX_MUL ? (uint64_t)MHX_MAX_MUL : cnt;
            const uint64_t f = ((x << 2) & mask_k) | (1ull << 19) | ((x >> 62) << 16) | mul;  // k-mer x[1..k], W = x[0]
            agg_end[-1 - (long)agg_at++] = make_uint2((uint32_t)(f >> 32), (uint32_t)f);
            if (x != xr) {  // palindromic (k+1)-mers emit the forward item only (read_to_sdbg_s2.cpp:385-423)
              const uint64_t b = ((xr << 2) & mask_k) | (1ull << 19) | ((xr >> 62) << 16) | mul;
              agg_end[-1 - (long)agg_at++] = make_uint2((uint32_t)(b >> 32), (uint32_t)b);
            }
          }
        }
        keys[sl] = kSegEmpty;
        atomicAnd(&cnts[sl >> 1], (sl & 1u) ? 0x0000FFFFu : 0xFFFF0000u);  // its half of the shared counter word
      }
    } else {  // the tile gave up: wipe the table, tell the host
      for (int i = tid; i < NSLOT; i += 256) keys[i] = kSegEmpty;
      for (int i = tid; i < NSLOT / 2; i += 256) cnts[i] = 0;
      if (tid == 0) atomicOr(a.err, 1u);
    }
    __syncthreads();
  }
  if (a.mark_mode == 2) {
    st_solid = wave_sum(st_solid);
    st_both = wave_sum(st_both);
    if (lane == 0 && st_both) {
      atomicAdd(a.ctr, st_solid);
      atomicAdd(a.ctr + 2, st_both);
    }
  } else {
    __syncthreads();
    for (int i = tid; i < kSegHist; i += 256)
      if (lhist[i]) atomicAdd(&a.hist[i], (unsigned long long)lhist[i]);
    if (AGG && tid == 0) a.agg_counts[blockIdx.x] = s_agg_cur < a.agg_cap ? s_agg_cur : a.agg_cap;
    if (marks_out && tid == 0) a.marks_counts[blockIdx.x] = s_mark_cur < a.marks_cap ? s_mark_cur : a.marks_cap;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Bucket-streaming variant of the segment group-by ("two-level bucketed sort" with the second level in LDS): the records
// are sorted on the top `pbits` bits of the (k-1)-mer only — 16 bits = the reference's lv1 bucket and two LSD passes at
// 10 M reads per GPU, up to 24 bits and three passes for larger jobs, so that a streamed bucket stays at ~20-40 K records
// whatever the job size (s1_plan) — and one workgroup takes one whole bucket: it streams the bucket, inserting the keys
// into an LDS table, and then marks every record with its key's count — by streaming the bucket a second time or, when the
// marks wanted are those of the NON-solid occurrences and m <= 2 (direct_marks: the usual case, most occurrences being
// solid), straight from the table: a key that ends with count 1 < m has exactly one record, whose position the insert
// left next to the key, so the second read never happens.  Inside a bucket the prefix is constant, so the table key is the
// remaining 2(k-1)-pbits (k-1)-mer bits + head/tail <= 32 bits at k <= 22 (4-byte compare-and-swap), and nothing of
// k_s1_seg's segment ownership / look-ahead is needed.
//
// SUB-ROUNDS: a bucket whose distinct keys do not fit the table is not the stage's problem but the bucket's: the workgroup
// takes it in 2^s rounds, round j inserting only the records whose top s local-key bits equal j (the bucket is read once
// per round, each round is a complete group-by of a disjoint key set: marks, histogram and aggregated items of a finished
// round stand).  A round that overflows is split in two, recursively; with all local-key bits fixed a round holds one key,
// so the recursion ends.  The host may also ask for 2^sub0 rounds for every bucket up front (a job whose buckets are known
// to hold 2-4 x what the table takes: cheaper than a third sort pass, s1_plan).  Nothing here redoes the stage: *err is
// left for what the host really has to handle (an output region that is too small).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kStreamThreads = 1024;  // one workgroup per CU: 8192 slots of key + count + first position = 98 KB of LDS
constexpr uint32_t kStreamEmpty = 0xFFFFFFFFu;  // never a key: head/tail bits 63 do not occur

__global__ void k_bucket_bounds(const uint32_t *__restrict__ items, uint64_t n, int stride, uint64_t *__restrict__ bstart, int pbits);  // kmsort_emu.hip

struct S1StreamGeom {
  int pbits;          // prefix bits the records are sorted on: 2^pbits buckets, bounds[q * (2^pbits + 1) + b]
  int sub0;           // every bucket starts with 2^sub0 sub-rounds
  uint32_t n_buckets; // 1 << pbits
  uint32_t max_fill;  // a round whose table ends up with more keys than this is redone in two halves
};

// local key of a record inside a bucket of the pbits-bit prefix: the (k-1)-mer bits below the prefix, then head/tail
__device__ __forceinline__ uint32_t s1_stream_local_key(uint32_t w0, uint32_t w1, int k, int pbits) {
  const int rem = 2 * (k - 1) - pbits, mer_sh = 64 - 2 * (k - 1);
  const uint64_t key = ((uint64_t)w0 << 32) | w1;
  const uint32_t lo = (uint32_t)(key >> mer_sh);
  return (rem ? (lo & ((1u << rem) - 1u)) << 6 : 0u) | (w1 & 63u);
}
// which buckets are giants: one thread per bucket; the list, the slices and the regions of partial entries are allotted here
__global__ __launch_bounds__(256) void k_s1_giant_find(const uint64_t *__restrict__ bounds, int n_src, uint32_t n_buckets, S1Giant g) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n_buckets) return;
  const size_t bstride = (size_t)n_buckets + 1;
  uint64_t total = 0;
  for (int q = 0; q < n_src; ++q) total += bounds[q * bstride + b + 1] - bounds[q * bstride + b];
  if (total < g.min_records) return;
  uint64_t sl64 = (total + 255) / 256;
  sl64 = (sl64 + 4095) / 4096 * 4096;
  const uint32_t sl = (uint32_t)(sl64 < kGiantSliceMin ? kGiantSliceMin : (sl64 > (1u << 30) ? (1u << 30) : sl64));
  uint64_t ns = 0;
  for (int q = 0; q < n_src; ++q) ns += (bounds[q * bstride + b + 1] - bounds[q * bstride + b] + sl - 1) / sl;
  const uint32_t gi = atomicAdd(&g.ctr[0], 1u);
  if (gi >= g.gcap) return;
  const unsigned long long cap = ns * kGiantEntriesPerSlice;
  const unsigned long long off = atomicAdd(reinterpret_cast<unsigned long long *>(g.ctr + 2), cap);
  const bool fits = off + cap <= g.pcap && cap < (1ull << 31);
  g.bucket[gi] = b;
  g.sl[gi] = sl;
  g.ns[gi] = fits ? (uint32_t)ns : 0u;
  g.cap[gi] = fits ? (uint32_t)cap : 0u;
  g.off[gi] = off;
  g.cur[gi] = 0;
  if (fits) g.flag[b] = 1;
}
// the slices of the giants, each reduced by one workgroup: LDS table of the slice's keys (count, first record) -> partial entries
__global__ __launch_bounds__(256) void k_s1_giant_reduce(const uint32_t *__restrict__ items0, const uint32_t *const *__restrict__ srcs,
                                                         const uint64_t *__restrict__ bounds, int n_src, uint32_t n_buckets, int pbits, int k, S1Giant g) {
  constexpr int NS = 4096, NT = 256, kFlushAt = NS / 2;
  __shared__ uint32_t keys[NS], cnts[NS], fidx[NS];
  __shared__ uint32_t s_claims, s_out, s_start, s_stop;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const size_t bstride = (size_t)n_buckets + 1;
  const uint32_t n_g = min(g.ctr[0], g.gcap);
  for (int i = tid; i < NS; i += NT) {
    keys[i] = kStreamEmpty;
    cnts[i] = 0;
  }
  if (tid == 0) s_claims = 0;
  __syncthreads();
  for (uint32_t gi = 0; gi < n_g; ++gi) {
    const uint32_t ns = g.ns[gi];
    if (!ns) continue;
    const uint32_t b = g.bucket[gi], sl_len = g.sl[gi], cap = g.cap[gi];
    uint4 *const region = g.partial + g.off[gi];
    for (uint32_t sl = blockIdx.x; sl < ns; sl += gridDim.x) {
      // the slice: `rem`-th slice of the first source that has that many
      uint32_t rem = sl;
      uint64_t lo = 0, hi = 0;
      const uint32_t *src = items0;
      for (int q = 0; q < n_src; ++q) {
        const uint64_t l = bounds[q * bstride + b], h = bounds[q * bstride + b + 1];
        const uint64_t nsq = (h - l + sl_len - 1) / sl_len;
        if (rem < nsq) {
          lo = l + (uint64_t)rem * sl_len;
          hi = lo + sl_len < h ? lo + sl_len : h;
          if (n_src > 1) src = srcs[q];
          break;
        }
        rem -= (uint32_t)nsq;
      }
      // table -> this giant's region (any order; a region that does not hold them gives the bucket back to the streaming kernel)
      auto flush = [&]() {
        __syncthreads();
        uint32_t mine = 0;
        for (int i = tid; i < NS; i += NT) mine += keys[i] != kStreamEmpty;
        if (tid == 0) s_out = 0;
        __syncthreads();
        const uint32_t incl = wave_inclusive_sum(mine);
        uint32_t wbase = 0;
        if (lane == kWave - 1 && incl) wbase = atomicAdd(&s_out, incl);
        wbase = __shfl(wbase, kWave - 1, kWave);
        __syncthreads();
        if (tid == 0) {
          const uint32_t tot = s_out;
          const uint32_t start = tot ? atomicAdd(&g.cur[gi], tot) : 0u;
          s_start = start;
          s_stop = start + tot > cap;
          if (s_stop) g.flag[b] = 0;
          s_claims = 0;
        }
        __syncthreads();
        uint32_t at = s_start + wbase + incl - mine;
        const bool write = !s_stop;
        for (int i = tid; i < NS; i += NT) {
          const uint32_t key = keys[i];
          if (key != kStreamEmpty) {
            if (write) {
              const uint32_t *r = src + (lo + fidx[i]) * 3;
              region[at++] = make_uint4(r[0], r[1], r[2], cnts[i]);
            }
            keys[i] = kStreamEmpty;
            cnts[i] = 0;
          }
        }
        __syncthreads();
      };
      bool stop = false;
      for (uint64_t base = lo; base < hi && !stop; base += NT) {
        const uint64_t idx = base + tid;
        const bool in = idx < hi;
        uint32_t lk = 0;
        if (in) {
          const uint32_t *r = src + idx * 3;
          lk = s1_stream_local_key(r[0], r[1], k, pbits);
        }
        // a wavefront whose records all carry one key (poly-A): one lane inserts for all
        const uint32_t lk0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lk);
        const bool uniform = __ballot(in && lk == lk0) == ~0ull;
        const uint32_t mult = uniform ? (uint32_t)kWave : 1u;
        if (in && (!uniform || lane == 0)) {
          uint32_t h = (lk * 0x9E3779B1u) >> (32 - 12);
          for (;;) {
            const uint32_t old = atomicCAS(&keys[h], kStreamEmpty, lk);
            if (old == kStreamEmpty) {
              fidx[h] = (uint32_t)(idx - lo);
              atomicAdd(&s_claims, 1u);
            }
            if (old == kStreamEmpty || old == lk) {
              atomicAdd(&cnts[h], mult);
              break;
            }
            h = (h + 1) & (NS - 1);  // (the table is flushed at half full: a free slot exists)
          }
        }
        __syncthreads();
        if (s_claims >= (uint32_t)kFlushAt - NT) {  // (uniform; at most NT more keys before the next look)
          flush();
          stop = s_stop != 0;
        }
      }
      if (!stop) flush();
    }
  }
}

constexpr int kStreamBatch = 4;     // buckets per ticket
constexpr int kStreamSrcMax = kWave;  // bucket bounds of up to this many sources are staged in LDS (one lane of wave 0 per source)

// NT / LOGS: 1024 threads and 8192 slots = one workgroup per CU (98 KB of LDS: key, count, first position; + 8 KB of tags
// when the read set has positions past 2^32); 512 threads and 4096 slots = two per CU (s1_stream_half: tables at twice the
// load — the insert phase alone measures 1.9 x slower per record, tools/micro/insert_probe.hip — kept for the tests, whose
// buckets then overflow and split).
//
// What the kernel's time is made of, measured with tools/micro/{lds_probe,insert_probe}.hip on the device before this form
// was written (round 4): an LDS operation of 64 random lanes costs the CU 6.5 cycles (add, read) to 11.8 (compare-and-swap
// with return) — the 1.33 G records of the headline would need 0.6 ms of those; the insert phase took 4.5 ms because every
// record ran its own probe loop (a loop iteration costs its instructions whether 64 lanes or 2 are still looking: ~3.5
// iterations per record and wavefront) and because every new key paid a same-address atomic on a shared counter plus a
// list entry.  Hence: the first probe of the UNR records of a trip is straight-line code for all lanes, the few lanes that
// met another key retry TOGETHER in one loop per trip (whichever of their records is still pending), new keys are counted
// per thread, and the per-key phase is ONE walk over the table (statistics, marks, aggregated items, wipe) instead of
// three phases with a list of occupied slots.  Loads: the records of trip i + 1 — across the end of a round or of a bucket:
// the first trip of what comes next — are requested before the inserts of trip i, and wave 0 fetches the next bucket's
// ticket and bounds while the current bucket is worked on.
// COUNT: the same bucket streaming for `count` (KmerCounter::Lv2Postprocess, kmer_counter.cpp:254-381) on the records of CountGenT:
// the table key is the (k+1)-mer below the prefix, the slot's third word holds, per prev / next char, "seen once" and "seen twice"
// bits (min count <= 2: has_in / has_out need no more), a solid key's packed edge goes to the workgroup's region (AGG's), and the
// records of solid keys without an in- or out-edge — a few per bucket — are found by a second read of the bucket, which brings
// first_0_out / last_0_in of their reads up to date.
template <bool AGG, int UNR, int NT, int LOGS, bool TAGS, bool GIANT = false, bool COUNT = false>
__global__ __launch_bounds__(NT) void k_s1_stream(const uint32_t *__restrict__ items0, const uint64_t *__restrict__ bounds, S1SegArgs a,
                                                  S1StreamGeom geo, uint32_t bucket_stride, uint32_t *__restrict__ ticket,
                                                  const uint32_t *const *__restrict__ srcs, int n_src) {
  // Multi-GPU: the records of a bucket arrive as n_src sub-ranges, one per sending rank, each rank's records sorted by
  // bucket in an array of its own (srcs[q], bounds[q * (n_buckets + 1) + bucket]); single GPU: one source, items0.
  constexpr int NSLOT = 1 << LOGS;
  constexpr int TRIP = NT * UNR;
  static_assert(NSLOT % NT == 0 && UNR <= 8, "table walk / pending mask");
  __shared__ uint32_t keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  __shared__ uint32_t fpos[NSLOT];             // position word of the record that claimed the slot (direct_marks)
  __shared__ uint8_t ftag[TAGS ? NSLOT : 4];   // ... and the position bits above it (s1_pos_tag), when the read set has any
  __shared__ uint32_t lhist[kSegHist];
  __shared__ uint32_t s_bad[2], s_nclaimed[2];  // per round, double-buffered: the next round's are cleared while this round's are read
  constexpr int NLIST = NSLOT / 4;              // solid keys of a round waiting for their aggregated items (more: worked off in place)
  __shared__ uint2 slist[AGG ? NLIST : 1];
  __shared__ uint32_t s_list_n[2];
  __shared__ uint32_t s_agg_cur, s_mark_cur;
  __shared__ uint32_t s_flagged;  // COUNT: the round has a solid key without an in- or out-edge
  // the bucket being worked on and the one after it: ticket and per-source bounds (wave 0 fills [par ^ 1] during bucket [par])
  __shared__ uint32_t s_tk[2];
  __shared__ uint32_t s_bid[2];  // GIANT: the lv1 bucket (of the plan's prefix) the ticket's giant is
  __shared__ uint64_t s_lo[2][kStreamSrcMax], s_hi[2][kStreamSrcMax];
  __shared__ uint64_t s_src[kStreamSrcMax];  // the sources' arrays (multi-GPU)
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const bool wave0 = tid < kWave;
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  uint2 *const agg_end = AGG ? a.agg_raw + (size_t)(blockIdx.x + 1) * a.agg_cap : nullptr;
  unsigned long long *const marks_out = a.marks_raw ? a.marks_raw + (size_t)blockIdx.x * a.marks_cap : nullptr;
  for (int i = tid; i < NSLOT; i += NT) {
    keys[i] = kStreamEmpty;
    cnts[i] = 0;
    if (COUNT) fpos[i] = 0;
  }
  for (int i = tid; i < kSegHist; i += NT) lhist[i] = 0;
  if (tid == 0) {
    s_bad[0] = s_bad[1] = 0;
    s_nclaimed[0] = s_nclaimed[1] = 0;
    s_list_n[0] = s_list_n[1] = 0;
    // GIANT: the second launch over the same grid goes on where this workgroup's regions stand
    s_agg_cur = GIANT && AGG ? a.agg_counts[blockIdx.x] : 0u;
    s_mark_cur = GIANT && marks_out ? a.marks_counts[blockIdx.x] : 0u;
  }
  // GIANT: the "buckets" of this launch are the entries of the giant list, their records the partial entries of k_s1_giant_reduce
  const uint64_t n_lim = GIANT ? (uint64_t)min(a.giant.ctr[0], a.giant.gcap) : (uint64_t)geo.n_buckets;
  const uint32_t m = a.m;
  const int k = a.k;
  const int pbits = geo.pbits;
  const size_t bstride = (size_t)geo.n_buckets + 1;
  // local key: the (k-1)-mer bits below the prefix, then head/tail (the position tag bits in between dropped)
  static_assert(!COUNT || (AGG && !GIANT), "count: edges leave through the regions of the aggregated items; no giant path");
  const int key_chars = COUNT ? k + 1 : k - 1;
  const int rem = 2 * key_chars - pbits;         // 0..26 bits (count: up to 32)
  const int lk_bits = COUNT ? rem : rem + 6;     // <= 32
  const int mer_sh = 64 - 2 * key_chars;
  const uint32_t mer_mask = rem >= 32 ? 0xFFFFFFFFu : (rem ? (1u << rem) - 1u : 0u);
  // (the low 32 bits of (w0:w1) >> mer_sh: one funnel shift while the (k-1)-mer reaches into the second word, k >= 18)
  const bool mer_two_words = mer_sh < 32;
  const uint32_t mer_sh1 = (uint32_t)(mer_two_words ? mer_sh : mer_sh - 32);
  auto local_key = [&](uint32_t w0, uint32_t w1) -> uint32_t {
    const uint32_t lo = mer_two_words ? __builtin_amdgcn_alignbit(w0, w1, mer_sh1) : w0 >> mer_sh1;
    if constexpr (COUNT) return lo & mer_mask;
    else return (lo & mer_mask) << 6 | (w1 & 63u);
  };
  // the (k+1)-mer head.S.tail of a table key of bucket bi, chars MSB-first in 64 bits
  auto edge_of = [&](uint32_t bi, uint32_t lk) -> uint64_t {
    const uint64_t smer = ((uint64_t)bi << (64 - pbits)) | (rem ? (uint64_t)(lk >> 6) << (64 - pbits - rem) : 0ull);
    return ((uint64_t)((lk >> 3) & 7u) << 62) | (smer >> 2) | ((uint64_t)(lk & 7u) << (62 - 2 * k));
  };
  // the aggregated stage-2 items of a solid key (one per strand; one for a palindrome) -> this workgroup's region, from its end.
  // dense: called by whole wavefronts (the place comes from one LDS atomic per wavefront); otherwise by single lanes.
  auto emit_items = [&](uint32_t bi, uint32_t lk, uint32_t cnt, bool dense, bool valid = true) {
    uint64_t x = 0, xr = 0;
    uint32_t n_out = 0;
    if (valid) {
      x = edge_of(bi, lk);
      xr = rc64(x, k + 1);
      n_out = x == xr ? 1u : 2u;
    }
    uint32_t at;
    bool ok;
    if (dense) {
      const uint32_t incl = wave_inclusive_sum(n_out);
      const uint32_t tot = __shfl(incl, kWave - 1, kWave);
      if (!tot) return;
      uint32_t wbase = 0;
      if (lane == 0) wbase = atomicAdd(&s_agg_cur, tot);
      wbase = __shfl(wbase, 0, kWave);
      ok = wbase + tot + (marks_out ? s_mark_cur : 0u) <= a.agg_cap;
      at = wbase + incl - n_out;
    } else {
      at = atomicAdd(&s_agg_cur, n_out);
      ok = at + n_out + (marks_out ? s_mark_cur : 0u) <= a.agg_cap;
    }
    if (!ok) {
      atomicOr(a.err, 1u);
      return;
    }
    if (n_out) {
      const uint64_t mask_k = ~0ull << (64 - 2 * k);
      const uint64_t mul = cnt > MHX_MAX_MUL ? (uint64_t)MHX_MAX_MUL : cnt;
      const uint64_t f = ((x << 2) & mask_k) | (1ull << 19) | ((x >> 62) << 16) | mul;
      agg_end[-1 - (long)at] = make_uint2((uint32_t)(f >> 32), (uint32_t)f);
      if (n_out == 2) {
        const uint64_t b = ((xr << 2) & mask_k) | (1ull << 19) | ((xr >> 62) << 16) | mul;
        agg_end[-2 - (long)at] = make_uint2((uint32_t)(b >> 32), (uint32_t)b);
      }
    }
  };
  auto hash_of = [&](uint32_t lk) -> uint32_t { return (lk * 0x9E3779B1u) >> (32 - LOGS); };
  auto bucket_of = [&](int par) -> uint64_t { return (uint64_t)s_tk[par] * bucket_stride; };
  // (explicit global address space for everything read from memory here: a select between an LDS and a global address would
  //  become a FLAT load, and one FLAT load in flight makes every later wait for a global load a wait for ALL loads)
  typedef const __attribute__((address_space(1))) uint64_t *gptr64;
  const gptr64 gbounds = (gptr64)bounds;
  // (bounds and arrays of the sources live in LDS — at most kStreamSrcMax senders, the host sees to that: a load from memory
  //  inside the trip loop would be waited for together with the record loads in flight)
  auto lo_of = [&](int par, int q) -> uint64_t { return s_lo[par][q]; };
  auto hi_of = [&](int par, int q) -> uint64_t { return s_hi[par][q]; };
  auto src_of = [&](int q) -> uint64_t { return n_src > 1 ? s_src[q] : (uint64_t)items0; };  // the array of source q
  // wave 0 holds the workgroup's place in the bucket sequence.  Tickets come in batches of kStreamBatch consecutive buckets: the
  // answer of the atomic is waited for on the spot (the compiler broadcasts it through a readfirstlane), which stalls wave 0 — an
  // insert worker like the others — for a memory round trip, so it is made rare; neighbouring buckets are also neighbours in memory.
  uint32_t w0_tk = 0, w0_left = 0;
  auto next_ticket = [&]() -> uint32_t {
    if (w0_left == 0) {
      uint32_t r = lane == 0 ? atomicAdd(ticket, 1u) : 0u;
      r = __shfl(r, 0, kWave);
      const uint64_t first = (uint64_t)r * kStreamBatch;
      w0_tk = first > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)first;
      w0_left = kStreamBatch;
    } else if (w0_tk != 0xFFFFFFFFu) {
      ++w0_tk;
    }
    --w0_left;
    return w0_tk;
  };
  // wave 0, lane q: the bounds of source q of bucket nb — requested, and used a bucket's inserts later (publish_desc); with them
  // the host's error word: a workgroup stops taking buckets once the host has to step in anyway
  uint64_t d_lo = 0, d_hi = 0;
  uint32_t d_err = 0, d_gf = 0, d_bid = 0;
  auto request_bounds = [&](uint64_t nb) {
    d_lo = d_hi = 0;
    d_gf = 0;
    d_err = ((const __attribute__((address_space(1))) uint32_t *)a.err)[0];
    if constexpr (GIANT) {
      if (nb < n_lim && lane == 0) {
        d_bid = a.giant.bucket[nb];
        const uint32_t got = min(a.giant.cur[nb], a.giant.cap[nb]);
        d_lo = a.giant.off[nb];
        d_hi = a.giant.ns[nb] && a.giant.flag[d_bid] ? d_lo + got : d_lo;  // (a giant that did not reduce was streamed by the first launch)
      }
    } else if (nb < geo.n_buckets && lane < n_src) {
      d_lo = gbounds[(size_t)lane * bstride + nb];
      d_hi = gbounds[(size_t)lane * bstride + nb + 1];
      if (a.giant.flag && lane == 0) d_gf = ((const __attribute__((address_space(1))) uint8_t *)a.giant.flag)[nb];
    }
  };
  auto publish_desc = [&](int par, uint32_t tk) {
    if (lane == 0) s_tk[par] = d_err ? 0xFFFFFFFFu / (bucket_stride ? bucket_stride : 1u) : tk;
    if (GIANT && lane == 0) s_bid[par] = d_bid;
    if (!GIANT && a.giant.flag && __shfl(d_gf, 0, kWave)) d_hi = d_lo;  // a giant: left to k_s1_giant_reduce and the GIANT launch
    if (lane < n_src) {
      // (statistics on a sample — mark_mode 2 — look at no more than 8 trips of a bucket: one low-complexity bucket, poly-A at
      //  lv1 bucket 0 for one, may hold millions of records, and a workgroup streams a bucket alone)
      const uint64_t cap = (uint64_t)8 * NT * UNR;
      s_lo[par][lane] = d_lo;
      s_hi[par][lane] = a.mark_mode == 2 && d_hi - d_lo > cap ? d_lo + cap : d_hi;
    }
  };
  if (wave0) {  // the first bucket of this workgroup
    if (n_src > 1 && lane < n_src) s_src[lane] = ((gptr64)srcs)[lane];
    const uint32_t t0 = next_ticket();
    request_bounds((uint64_t)t0 * bucket_stride);
    publish_desc(0, t0);
  }
  __syncthreads();

  // a trip = the next TRIP records of one source; the cursor walks the non-empty sources of a bucket in order (uniform values)
  auto first_source = [&](int par, int from) -> int {
    int q = from;
    while (q < n_src && lo_of(par, q) == hi_of(par, q)) ++q;
    return q;
  };
  typedef const __attribute__((address_space(1))) uint32_t *gptr;
  struct TripRef {
    gptr g;      // the trip's first record
    uint32_t n;  // its records (1..TRIP)
  };
  auto uniform64 = [](uint64_t v) -> uint64_t {  // (a value all lanes agree on, moved to scalar registers: addresses become base + 32-bit offset)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  };
  // Explicit global address space: the pointer comes out of a select between a kernel argument and a pointer read from memory,
  // and FLAT loads would count in lgkmcnt as well — every wait for an LDS atomic would then wait for the loads in flight too.
  auto trip_ref = [&](int q, uint64_t base, uint64_t hi) -> TripRef {
    const uint64_t src = src_of(q);
    const uint64_t left = hi - base;
    return TripRef{(gptr)uniform64(src + base * 12), (uint32_t)__builtin_amdgcn_readfirstlane((int)(left < (uint64_t)TRIP ? (uint32_t)left : (uint32_t)TRIP))};
  };
  // A thread's UNR records of a trip are CONSECUTIVE (UNR * 12 contiguous bytes, read as 16-byte loads — the records of a
  // bucket may be inserted in any order, so which thread holds which record is free).  Measured on the device before this
  // form was chosen (tools/micro/read_probe.hip, one 1024-thread workgroup per CU, the next trip requested before the
  // current one is used, compute between the trips): records NT apart as 12-byte loads 2.2 TB/s, this form 3.3 TB/s, both
  // 6.4 TB/s without compute.  Unconditional loads, always: straight-line code, so that all loads are issued before the
  // first wait (a load inside an `if` is waited for at the end of its block); a thread beyond the trip's last record reads
  // the window that starts at that record — up to 36 bytes past the trip's end: every record array here ends in 64 spare
  // bytes (mhx_ctx::ws) — and its mask bits stay clear.  Where no trip follows, the caller passes a one-record stand-in.
  static_assert(UNR == 4, "a thread's window of a trip: four 12-byte records = three 16-byte loads");
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef const __attribute__((address_space(1))) u32x4 *gptr4;
  auto load_trip = [&](const TripRef &t, uint32_t (&w0)[UNR], uint32_t (&w1)[UNR], uint32_t (&w2)[UNR], uint32_t &inm) {
    const uint32_t first = (uint32_t)tid * UNR;  // (a constant of the thread)
    const uint32_t left = t.n > first ? t.n - first : 0u;
    inm = left >= UNR ? (1u << UNR) - 1u : (1u << left) - 1u;
    const gptr4 p = (gptr4)(t.g + (first < t.n ? first : t.n - 1) * 3u);
    const u32x4 a0 = p[0], a1 = p[1], a2 = p[2];
    w0[0] = a0.x, w1[0] = a0.y, w2[0] = a0.z;
    w0[1] = a0.w, w1[1] = a1.x, w2[1] = a1.y;
    w0[2] = a1.z, w1[2] = a1.w, w2[2] = a2.x;
    w0[3] = a2.y, w1[3] = a2.z, w2[3] = a2.w;
  };

  unsigned long long st_solid = 0, st_both = 0;
  // (stream mode: the host passes the probe limit here; tests set it to 0.  Below 7/8 full a chain of 128 slots does not occur
  //  in practice; where it does, the round is redone in two halves)
  const int probe_limit = min(a.la_chunks, 128);
  // register set A: at the top of a round it holds the round's first trip, requested long before (by the round before it, or
  // right here for the first bucket) — one writer on the hot path, so that no copies (= waits for the loads) are needed
  uint32_t nw0[UNR], nw1[UNR], nw2[UNR], n_inm = 0;
  int par = 0, rp = 0;
  TripRef cur{(gptr)bounds, 1u};  // (always a readable address: the stand-in where no trip follows; at first the bounds themselves)
  auto request_first = [&](int bpar) {  // the first trip of bucket [bpar] -> set A (bucket empty or none left: a stand-in, mask cleared)
    bool follows = false;
    if (GIANT) return;  // (the partial entries of a giant are read where they are inserted)
    if (bucket_of(bpar) < n_lim) {
      const int q = first_source(bpar, 0);
      if (q < n_src) {
        cur = trip_ref(q, lo_of(bpar, q), hi_of(bpar, q));
        follows = true;
      }
    }
    load_trip(follows ? cur : TripRef{cur.g, 1u}, nw0, nw1, nw2, n_inm);
    if (!follows) n_inm = 0;
  };
  request_first(0);

  for (;;) {
    MHX_TT_BEGIN
    const uint64_t bi64 = bucket_of(par);
    if (bi64 >= n_lim) break;
    const uint32_t bi = GIANT ? s_bid[par] : (uint32_t)bi64;
    // wave 0: the next bucket — its ticket and the request for its bounds when this bucket's first round starts, handed over
    // when that round's inserts end
    uint32_t next_tk = 0;
    int desc = 0;  // 1: bounds requested, 2: published
    auto desc_step = [&](int upto) {
      if (!wave0) return;
      if (desc == 0) {
        next_tk = next_ticket();
        request_bounds((uint64_t)next_tk * bucket_stride);
        desc = 1;
      }
      if (desc == 1 && upto == 2) {
        publish_desc(par ^ 1, next_tk);
        desc = 2;
      }
    };
    int q0 = first_source(par, 0);
    if (q0 == n_src) {  // an empty bucket
      desc_step(2);
      __syncthreads();
      par ^= 1;
      request_first(par);
      continue;
    }
    MHX_TT(10)
    // the bucket in rounds: round (sub, rj) takes the records whose top `sub` local-key bits are rj
    uint32_t sub = (uint32_t)min(geo.sub0, lk_bits), rj = 0;
    const uint32_t sub_first = sub;
    for (;;) {
      const uint32_t sub_sh = (uint32_t)lk_bits - sub;  // (sub == 0: no test)
      uint32_t claims = 0, seen = 0;
      // A: insert.  Two register sets take turns (A: nw*, B: mw*): while the trip in one is inserted, the loads of the trip after
      // it fill the other — no copies between them (a copy of freshly loaded registers is a wait for the loads).
      // the inserts of one trip
      auto insert_trip = [&](const uint32_t (&rw0)[UNR], const uint32_t (&rw1)[UNR], const uint32_t (&rw2)[UNR], uint32_t inm) {
        // a round that has outgrown its table is redone in two halves anyway: no further inserts (the probe chains of a table that
        // fills up grow without bound long before an insert fails).  `seen` = the round's key count as read behind the trip before.
        if (seen > geo.max_fill) return;
        const uint32_t claims_before = claims;
        uint32_t lk[UNR];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          lk[u] = local_key(rw0[u], rw1[u]);
          const bool mn = ((inm >> u) & 1u) && (sub == 0 || (lk[u] >> sub_sh) == rj);
          mine |= mn ? 1u << u : 0u;
        }
        if (probe_limit <= 0) {
          if (mine) s_bad[rp] = 1;
          mine = 0;
        }
        // low-complexity reads: a whole trip of one wavefront carrying ONE key (a poly-A stretch: tens of thousands of records
        // of one key in a row) is inserted by one lane instead of 64 lanes queueing up at one LDS address UNR times
        bool one_key = mine == (1u << UNR) - 1u;
#pragma unroll
        for (int u = 1; u < UNR; ++u) one_key = one_key && lk[u] == lk[0];
        one_key = __ballot(one_key && lk[0] == (uint32_t)__builtin_amdgcn_readfirstlane((int)lk[0])) == ~0ull;
        uint32_t mult = 1;
        uint32_t wave_add1 = 0, wave_add2 = 0;  // COUNT: the seen-once / seen-twice bits of all records of a one-key trip
        if (one_key) {
          if constexpr (COUNT) {  // (the records' prev / next chars differ even where their keys agree: counted per char over the wavefront)
#pragma unroll
            for (unsigned x = 0; x < 4; ++x) {
              uint32_t cp = 0, cn = 0;
#pragma unroll
              for (int u = 0; u < UNR; ++u) {
                cp += (uint32_t)__builtin_popcountll(__ballot(((rw1[u] >> 3) & 7u) == x));
                cn += (uint32_t)__builtin_popcountll(__ballot((rw1[u] & 7u) == x));
              }
              wave_add1 |= (cp ? 1u << (2 * x) : 0u) | (cn ? 1u << (8 + 2 * x) : 0u);
              wave_add2 |= (cp >= 2 ? 2u << (2 * x) : 0u) | (cn >= 2 ? 2u << (8 + 2 * x) : 0u);
            }
          }
          mine = lane == 0 ? 1u : 0u;
          mult = (uint32_t)(kWave * UNR);
        }
        // First probe of every record, straight-line.  A lane that met another key there keeps the record pending — one per lane;
        // a second one of the same trip (one lane in twenty) is seen to on the spot — and the pending records of all lanes are
        // retried together afterwards: the retries cost their instructions per turn, however few lanes take part.
        // (the slot found — the key's own, or a free one claimed: count it, and remember the record that claimed it)
        auto settle = [&](uint32_t old, uint32_t key, uint32_t hh, uint32_t pos, uint32_t w1v) -> bool {
          if (old != kStreamEmpty && old != key) return false;
          atomicAdd(&cnts[hh], mult);
          if constexpr (COUNT) {
            if (old == kStreamEmpty) ++claims;
            // prev char x: bit 2x = seen once, 2x + 1 = seen twice; next char x: bits 8 + 2x, 9 + 2x ('$' counts for nothing)
            const unsigned pv = (w1v >> 3) & 7u, nx = w1v & 7u;
            const uint32_t add1 = one_key ? wave_add1 : ((pv < 4 ? 1u << (2 * pv) : 0u) | (nx < 4 ? 1u << (8 + 2 * nx) : 0u));
            const uint32_t add2 = one_key ? wave_add2 : 0u;
            if (add1) {
              const uint32_t o = atomicOr(&fpos[hh], add1 | add2);
              const uint32_t again = ((o & add1) << 1) & ~(o | add2);  // a char seen before and now again: seen twice
              if (again) atomicOr(&fpos[hh], again);
            }
          } else if (old == kStreamEmpty) {  // only read back when the count stays 1: then this record is the key's only one
            fpos[hh] = pos;
            if (TAGS) ftag[hh] = (uint8_t)(w1v >> 6);
            ++claims;
          }
          return true;
        };
        auto probe = [&](uint32_t key, uint32_t hh, uint32_t pos, uint32_t w1v) -> bool {
          return settle(atomicCAS(&keys[hh], kStreamEmpty, key), key, hh, pos, w1v);
        };
        // the UNR compare-and-swaps go out back to back: one LDS round trip per trip instead of UNR (with four wavefronts per SIMD
        // the round trips, ~250 cycles each under load, are not hidden)
        uint32_t h1[UNR], old1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          h1[u] = hash_of(lk[u]);
          old1[u] = kStreamEmpty;
          if ((mine >> u) & 1u) old1[u] = atomicCAS(&keys[h1[u]], kStreamEmpty, lk[u]);
        }
        bool has = false;
        uint32_t pk = 0, ph = 0, pw = 0, pt = 0;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if ((mine >> u) & 1u) {
            uint32_t hh = h1[u];
            if (!settle(old1[u], lk[u], hh, rw2[u], rw1[u])) {
              hh = (hh + 1) & (NSLOT - 1);
              if (!has) {
                has = true;
                pk = lk[u], ph = hh, pw = rw2[u], pt = rw1[u];
              } else {
                int n = 0;
                while (!probe(lk[u], hh, rw2[u], rw1[u])) {
                  hh = (hh + 1) & (NSLOT - 1);
                  if (++n >= probe_limit) {
                    s_bad[rp] = 1;
                    break;
                  }
                }
              }
            }
          }
        }
        int turns = 0;
        while (__ballot(has)) {
          if (has) {
            if (probe(pk, ph, pw, pt)) has = false;
            else ph = (ph + 1) & (NSLOT - 1);
          }
          if (++turns > probe_limit) {  // (uniform: every lane counts the same turns)
            if (has) s_bad[rp] = 1;
            break;
          }
        }
        // the keys this wavefront claimed in this trip (0..UNR per lane, counted with three ballots) -> the round's count, which is
        // read back for the next trip's look at it
        {
          const uint32_t d = claims - claims_before;
          const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(d & 1u)) + 2u * (uint32_t)__builtin_popcountll(__ballot(d & 2u)) +
                             4u * (uint32_t)__builtin_popcountll(__ballot(d & 4u));
          if (lane == 0 && c) atomicAdd(&s_nclaimed[rp], c);
          seen = __hip_atomic_load(&s_nclaimed[rp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      if constexpr (GIANT) {
        // the partial entries of the giant (first record of a key in a slice + its count there): few, inserted with their counts
        desc_step(1);
        const uint4 *const part = a.giant.partial;
        const uint64_t lo = lo_of(par, 0), hi = hi_of(par, 0);
        uint32_t my_claims = 0;
        for (uint64_t e = lo + tid; e < hi; e += NT) {
          const uint4 en = part[e];
          const uint32_t lk = local_key(en.x, en.y);
          if (sub != 0 && (lk >> sub_sh) != rj) continue;
          if (probe_limit <= 0) {
            s_bad[rp] = 1;
            continue;
          }
          uint32_t hh = hash_of(lk);
          for (int n = 0;; ++n) {
            const uint32_t old = atomicCAS(&keys[hh], kStreamEmpty, lk);
            if (old == kStreamEmpty || old == lk) {
              atomicAdd(&cnts[hh], en.w);
              if (old == kStreamEmpty) {
                fpos[hh] = en.z;
                if (TAGS) ftag[hh] = (uint8_t)(en.y >> 6);
                ++my_claims;
              }
              break;
            }
            hh = (hh + 1) & (NSLOT - 1);
            if (n >= probe_limit) {
              s_bad[rp] = 1;
              break;
            }
          }
        }
        if (my_claims) atomicAdd(&s_nclaimed[rp], my_claims);
      } else {
        int q = q0;
        uint64_t base = lo_of(par, q), hi = hi_of(par, q);
        // Set A was requested before the per-key walk of the round before this one, whose stores may still be on their way: loads
        // and stores return out of order with respect to each other, so with both pending the compiler waits for ALL of them at the
        // first use of a loaded register — including the loads requested just before.  Waiting here, before anything new is asked
        // for, keeps the waits inside the trip loop at "all but the newest UNR loads".
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        desc_step(1);
        // the trip after the current one (behind the last trip of the round: a one-record stand-in, mask cleared) -> the other set
        bool more = true;
        auto request_next = [&](uint32_t (&w0)[UNR], uint32_t (&w1)[UNR], uint32_t (&w2)[UNR], uint32_t &inm) {
          base += TRIP;
          if (base >= hi) {
            q = first_source(par, q + 1);
            if (q < n_src) {
              base = lo_of(par, q);
              hi = hi_of(par, q);
            } else {
              more = false;
            }
          }
          if (more) cur = trip_ref(q, base, hi);
          load_trip(more ? cur : TripRef{cur.g, 1u}, w0, w1, w2, inm);
          if (!more) inm = 0;
        };
        uint32_t mw0[UNR], mw1[UNR], mw2[UNR], m_inm = 0;
        for (;;) {
          request_next(mw0, mw1, mw2, m_inm);
          insert_trip(nw0, nw1, nw2, n_inm);
          if (!more) break;
          request_next(nw0, nw1, nw2, n_inm);
          insert_trip(mw0, mw1, mw2, m_inm);
          if (!more) break;
        }
      }
      desc_step(2);
      __syncthreads();  // A: the table is complete
      MHX_TT(11)
      const bool bad = s_bad[rp] != 0 || s_nclaimed[rp] > geo.max_fill;
      if (tid == 0) {
        s_bad[rp ^ 1] = 0;
        s_nclaimed[rp ^ 1] = 0;
        s_list_n[rp ^ 1] = 0;  // (read behind barrier B of the round before this one, by threads that have all passed barrier A since)
      }
      // what comes next (uniform: `bad` came out of shared memory behind a barrier)
      uint32_t nsub = sub, nrj = rj;
      bool bucket_done = false, give_up = false;
      if (bad) {
        if ((int)sub >= lk_bits) {  // one key per round and still no room: only a probe limit of 0 (tests) gets here
          give_up = true;
          bucket_done = true;
        } else {
          nsub = sub + 1;
          nrj = rj << 1;
        }
      } else {
        nrj = rj + 1;
        while (nsub > sub_first && (nrj & 1u) == 0) {
          --nsub;
          nrj >>= 1;
        }
        bucket_done = nsub == sub_first && nrj == (1u << sub_first);
      }
      if (give_up && tid == 0) atomicOr(a.err, 1u);
      // ... and its first trip, requested before the per-key work of this round
      if constexpr (!GIANT) {
        if (!bucket_done) {
          cur = trip_ref(q0, lo_of(par, q0), hi_of(par, q0));
          load_trip(cur, nw0, nw1, nw2, n_inm);
        } else {
          request_first(par ^ 1);
        }
      }
      if (!GIANT && !COUNT && !bad) {
        // B: marks by a second read of the bucket (m > 2, or the marks of the solid occurrences are wanted)
        if (a.mark_mode != 2 && !a.direct_marks) {
          for (int q = 0; q < n_src; ++q) {
            const uint64_t lo = lo_of(par, q), hi = hi_of(par, q);
            const gptr items = (gptr)src_of(q);
            for (uint64_t base = lo; base < hi; base += NT) {
              const uint64_t gi = base + tid;
              bool in = gi < hi;
              uint32_t w1 = 0, w2 = 0, cnt = 0;
              if (in) {
                const gptr p = items + gi * 3;
                const uint32_t w0 = p[0];
                w1 = p[1];
                w2 = p[2];
                const uint32_t lk = local_key(w0, w1);
                in = sub == 0 || (lk >> sub_sh) == rj;  // (a key of another round is not in the table)
                if (in) {
                  uint32_t h = hash_of(lk);
                  while (keys[h] != lk) h = (h + 1) & (NSLOT - 1);
                  cnt = cnts[h];
                }
              }
              const bool both = (w1 & 0x24u) == 0;
              const bool solid = both && cnt >= m;
              const bool mk = in && (a.mark_mode == 1 ? (both && !solid) : solid);
              const uint64_t abs = w2 + (uint64_t)((w1 >> 6) & 0xFFu) * a.pos_stride;
              if (!marks_out) {
                if (mk) a.solid_bytes[abs - 1] = 1;  // is_solid.set(pos - 1), :464 (or its complement)
              } else {
                const uint64_t mm = __ballot(mk);
                if (mm) {
                  uint32_t mbase = 0;
                  if (lane == 0) mbase = atomicAdd(&s_mark_cur, (uint32_t)__builtin_popcountll(mm));
                  mbase = __shfl(mbase, 0, kWave);
                  if (mk) {
                    const uint32_t at = mbase + (uint32_t)__builtin_popcountll(mm & lanemask_lt);
                    if (at + s_agg_cur < a.marks_cap) marks_out[at] = abs - 1;
                    else atomicOr(a.err, 2u);
                  }
                }
              }
            }
          }
          __syncthreads();  // (the walk below wipes the table the loop above reads)
        }
      }
      MHX_TT(12)
      // C: one walk over the table — per distinct key: statistics and the mark of a key's only record; the slot is free again.
      // The solid keys (a few per cent of the slots) are only LISTED here: what they need — the (k+1)-mer, its reverse
      // complement, one or two aggregated stage-2 items — is ~100 instructions that every lane of a wavefront would sit
      // through for the one or two lanes that hold a solid key (measured: the walk with that work inline took 29 % of the
      // kernel).  The list is worked off densely behind barrier B, while other wavefronts already insert the next round.
      // (all of a thread's slots are read first and wiped, then looked at: one LDS round trip for the lot instead of three
      //  dependent ones per slot; the places in the list of solid keys — and, on several GPUs, in the region of marks — come
      //  from one wavefront scan and one LDS atomic per wavefront and walk instead of one per slot)
      if constexpr (COUNT) {
        // C (count): per distinct (k+1)-mer — multiplicity histogram, has_in / has_out from the seen-twice (m = 2) or seen-once
        // (m = 1) bits, the packed edge of a solid key -> this workgroup's region; a solid key without an in- or out-edge
        // leaves two flag bits in its slot for the second read below
        constexpr int W = NSLOT / NT;
        if (tid == 0) s_flagged = 0;
        uint32_t wk[W], wc[W], wf[W];
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          wk[it] = keys[sl];
          wc[it] = cnts[sl];
          wf[it] = fpos[sl];
        }
        __syncthreads();  // (s_flagged cleared before anybody sets it)
        const uint32_t lvl = m >= 2 ? 0xAAu : 0x55u;  // which bit of a char's pair says "at least m"
        uint32_t solid_bits = 0, n_dist = 0;
        bool any_flag = false;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const uint32_t lk = wk[it], cnt = wc[it];
          uint32_t fb = 0;
          if (lk != kStreamEmpty && !bad) {
            ++n_dist;
            const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;
            if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
            else atomicAdd(&a.hist[hb], 1ull);
            if (cnt >= m) {
              solid_bits |= 1u << it;
              const bool has_in = (wf[it] & lvl) != 0, has_out = ((wf[it] >> 8) & lvl) != 0;
              fb = (has_in ? 0u : 1u) | (has_out ? 0u : 2u);
              any_flag = any_flag || fb != 0;
            }
            fpos[it * NT + tid] = fb << 30;
          }
        }
        st_solid += n_dist;  // (count: distinct keys)
        if (__ballot(any_flag) && lane == 0) s_flagged = 1;
        {  // the solid keys' packed edges (PackEdge, kmer_counter.cpp:32-52: multiplicity in the low 16 bits) -> the region, from its front
          const uint32_t n_e = (uint32_t)__builtin_popcount(solid_bits);
          const uint32_t incl = wave_inclusive_sum(n_e);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t ebase = 0;
            if (lane == 0) ebase = atomicAdd(&s_agg_cur, tot);
            ebase = __shfl(ebase, 0, kWave);
            if (ebase + tot > a.agg_cap) {
              if (lane == 0) atomicOr(a.err, 1u);
            } else {
              unsigned long long *const eout = reinterpret_cast<unsigned long long *>(a.agg_raw + (size_t)blockIdx.x * a.agg_cap);
              uint32_t at = ebase + incl - n_e;
#pragma unroll
              for (int it = 0; it < W; ++it)
                if ((solid_bits >> it) & 1u) {
                  const uint32_t cnt = wc[it];
                  const unsigned long long edge = ((unsigned long long)bi << (64 - pbits)) | (rem ? (unsigned long long)wk[it] << mer_sh : 0ull);
                  eout[at++] = edge | (cnt > MHX_MAX_MUL ? (unsigned long long)MHX_MAX_MUL : cnt);
                }
            }
          }
        }
        __syncthreads();
        if (s_flagged && !bad) {  // the records of the flagged keys: first_0_out / last_0_in of their reads (kmer_counter.cpp:307-368)
          for (int q = 0; q < n_src; ++q) {
            const uint64_t lo = lo_of(par, q), hi = hi_of(par, q);
            const gptr items = (gptr)src_of(q);
            for (uint64_t base = lo; base < hi; base += NT) {
              const uint64_t gi = base + tid;
              if (gi >= hi) continue;
              const gptr p = items + gi * 3;
              const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
              const uint32_t lk = local_key(w0, w1);
              if (sub != 0 && (lk >> sub_sh) != rj) continue;  // (a key of another round is not in the table)
              uint32_t h = hash_of(lk);
              while (keys[h] != lk) h = (h + 1) & (NSLOT - 1);
              const uint32_t f = fpos[h] >> 30;
              if (!f) continue;
              const uint64_t abs = w2 + (TAGS ? (uint64_t)((w1 >> 7) & 0xFFu) * a.pos_stride : 0ull);
              const bool fwd = (w1 & kCountStrandBit) == 0;
              const uint64_t rid = seq_of_offset(a.c_start, a.c_n_seqs, a.c_fixed_len, abs);
              const uint32_t off = (uint32_t)(abs - a.c_start[rid]);
              if (f & 1u) {  // no in-edge: strand 0 -> last_0_in = max(off), strand 1 -> first_0_out = min(off + 1)
                if (fwd) atomicMax(&a.last_0_in_p1[rid], off + 1);
                else atomicMin(&a.first_0_out[rid], off + 1);
              }
              if (f & 2u) {  // no out-edge: the roles swap
                if (fwd) atomicMin(&a.first_0_out[rid], off + 1);
                else atomicMax(&a.last_0_in_p1[rid], off + 1);
              }
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          keys[sl] = kStreamEmpty;
          cnts[sl] = 0;
          fpos[sl] = 0;
        }
      } else {
        constexpr int W = NSLOT / NT;
        uint32_t wk[W], wc[W], wp[W];
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          wk[it] = keys[sl];
          wc[it] = cnts[sl];
          wp[it] = fpos[sl];
        }
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          keys[sl] = kStreamEmpty;
          cnts[sl] = 0;
        }
        uint32_t want_bits = 0, mark_bits = 0;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const uint32_t lk = wk[it], cnt = wc[it];
          if (lk != kStreamEmpty && !bad && (lk & 0x24u) == 0) {
            const bool solid = cnt >= m;
            if (a.mark_mode == 2) {
              st_both += cnt;
              if (solid) st_solid += cnt;
            } else {
              const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;  // :430-436
              if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
              else atomicAdd(&a.hist[hb], 1ull);
              if (a.direct_marks && !solid) mark_bits |= 1u << it;  // count 1 < m <= 2: the key's only record (mark_mode 1)
              if (AGG && solid) want_bits |= 1u << it;
            }
          }
        }
        if (a.direct_marks) {  // (uniform)
          if (!marks_out) {
#pragma unroll
            for (int it = 0; it < W; ++it)
              if ((mark_bits >> it) & 1u) a.solid_bytes[wp[it] + (TAGS ? (uint64_t)ftag[it * NT + tid] * a.pos_stride : 0ull) - 1] = 1;
          } else {  // multi-GPU: the mark is the global position itself, appended to this workgroup's region
            const uint32_t n_mk = (uint32_t)__builtin_popcount(mark_bits);
            const uint32_t incl = wave_inclusive_sum(n_mk);
            const uint32_t tot = __shfl(incl, kWave - 1, kWave);
            if (tot) {
              uint32_t mbase = 0;
              if (lane == 0) mbase = atomicAdd(&s_mark_cur, tot);
              mbase = __shfl(mbase, 0, kWave);
              uint32_t at = mbase + incl - n_mk;
#pragma unroll
              for (int it = 0; it < W; ++it)
                if ((mark_bits >> it) & 1u) {
                  if (at + s_agg_cur < a.marks_cap) marks_out[at] = wp[it] + (TAGS ? (uint64_t)ftag[it * NT + tid] * a.pos_stride : 0ull) - 1;
                  else atomicOr(a.err, 2u);
                  ++at;
                }
            }
          }
        }
        if constexpr (AGG) {
          const uint32_t n_w = (uint32_t)__builtin_popcount(want_bits);
          const uint32_t incl = wave_inclusive_sum(n_w);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t lbase = 0;
            if (lane == 0) lbase = atomicAdd(&s_list_n[rp], tot);
            lbase = __shfl(lbase, 0, kWave);
            uint32_t at = lbase + incl - n_w;
#pragma unroll
            for (int it = 0; it < W; ++it)
              if ((want_bits >> it) & 1u) {
                if (at < (uint32_t)NLIST) slist[at] = make_uint2(wk[it], wc[it]);
                else emit_items(bi, wk[it], wc[it], false);  // (more solid keys in one round than the list holds: in place)
                ++at;
              }
          }
        }
      }
      MHX_TT(13)
      __syncthreads();  // B: the table is empty
      MHX_TT(14)
      if constexpr (AGG && !COUNT) {  // the listed solid keys -> aggregated items
        const uint32_t n_list = min(s_list_n[rp], (uint32_t)NLIST);
        for (uint32_t base = 0; base < n_list; base += NT) {
          const uint32_t i = base + tid;
          const uint2 e = i < n_list ? slist[i] : make_uint2(0u, 0u);
          emit_items(bi, e.x, e.y, true, i < n_list);
        }
      }
      rp ^= 1;
      sub = nsub;
      rj = nrj;
      if (bucket_done) break;
    }
    par ^= 1;
  }
  if constexpr (COUNT) {
    st_solid = wave_sum(st_solid);
    if (lane == 0 && st_solid) atomicAdd(a.ctr + 4, st_solid);
    st_solid = 0;
  }
  if (a.mark_mode == 2) {
    st_solid = wave_sum(st_solid);
    st_both = wave_sum(st_both);
    if (lane == 0 && st_both) {
      atomicAdd(a.ctr, st_solid);
      atomicAdd(a.ctr + 2, st_both);
    }
  } else {
    __syncthreads();
    for (int i = tid; i < kSegHist; i += NT)
      if (lhist[i]) atomicAdd(&a.hist[i], (unsigned long long)lhist[i]);
    if (AGG && tid == 0) a.agg_counts[blockIdx.x] = s_agg_cur < a.agg_cap ? s_agg_cur : a.agg_cap;
    if (marks_out && tid == 0) a.marks_counts[blockIdx.x] = s_mark_cur < a.marks_cap ? s_mark_cur : a.marks_cap;
  }
}

// regions of k_s1_seg -> one dense array: block (r, j) copies slice j of region r behind the items of the regions before it
__global__ __launch_bounds__(256) void k_agg_compact(const uint2 *__restrict__ raw, uint32_t cap, const uint32_t *__restrict__ counts,
                                                    uint2 *__restrict__ dense, int from_back) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint32_t r = blockIdx.x;
  uint64_t part = 0;
  for (uint32_t i = threadIdx.x; i < r; i += 256) part += counts[i];
  uint64_t off;
  block_exclusive_sum<uint64_t, 256>(part, sm, &off);
  const uint32_t n = counts[r];
  const uint2 *src = raw + (size_t)r * cap;
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) dense[off + i] = from_back ? src[cap - 1 - i] : src[i];
}

// regions of different sizes (start offsets in off[]) -> one dense array, region order kept
__global__ __launch_bounds__(256) void k_regions_compact(const uint2 *__restrict__ raw, const uint64_t *__restrict__ off, const uint32_t *__restrict__ counts,
                                                        uint2 *__restrict__ dense) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint32_t r = blockIdx.x;
  uint64_t part = 0;
  for (uint32_t i = threadIdx.x; i < r; i += 256) part += counts[i];
  uint64_t at;
  block_exclusive_sum<uint64_t, 256>(part, sm, &at);
  const uint32_t n = counts[r];
  const uint2 *src = raw + off[r];
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) dense[at + i] = src[i];
}

// multi-GPU, sparse marks, classic path: positions of the set bytes of the (global) byte map, appended in any order
__global__ __launch_bounds__(256) void k_collect_marks(const uint8_t *__restrict__ bytes, uint64_t n_bytes, unsigned long long *__restrict__ out,
                                                      unsigned long long *__restrict__ cursor) {
  __shared__ uint32_t s_n;
  __shared__ unsigned long long s_base;
  const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  uint32_t mask = 0;
  if (p0 < n_bytes) {
    const uint4 v = *reinterpret_cast<const uint4 *>(bytes + p0);  // the map is padded to a multiple of 64 bytes
    const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (p0 + t < n_bytes && ((xs[t >> 2] >> ((t & 3) * 8)) & 1u)) mask |= 1u << t;
  }
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t cnt = __builtin_popcount(mask);
  uint32_t at = cnt ? atomicAdd(&s_n, cnt) : 0u;
  __syncthreads();
  if (threadIdx.x == 0 && s_n) s_base = atomicAdd(cursor, (unsigned long long)s_n);
  __syncthreads();
  for (uint32_t mm = mask; mm; mm &= mm - 1) out[s_base + at++] = p0 + (uint64_t)__builtin_ctz(mm);
}
// routed marks (global positions of non-solid occurrences in the local reads) -> local byte map
// the marks the bucket streaming left in its workgroups' regions (positions, 8 bytes each) -> the byte map (one GPU, s1_marks_list = 1:
// an experiment of round 5, see launch_partial)
__global__ __launch_bounds__(256) void k_apply_mark_regions(const unsigned long long *__restrict__ raw, uint32_t cap, const uint32_t *__restrict__ counts,
                                                            uint8_t *__restrict__ bytes, uint64_t n_bytes) {
  const uint32_t n = counts[blockIdx.x];
  const unsigned long long *src = raw + (size_t)blockIdx.x * cap;
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
    const unsigned long long p = src[i];
    if (p < n_bytes) bytes[p] = 1;
  }
}
__global__ void k_apply_marks(const unsigned long long *__restrict__ pos, uint64_t n, uint64_t pos_base, uint64_t n_local, uint8_t *__restrict__ bytes,
                              uint32_t *__restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = pos[i] - pos_base;
  if (p < n_local) bytes[p] = 1;
  else atomicOr(bad, 1u);
}

// byte map -> AtomicBitVector layout (bit i = word i/64, bit i%64; kmbitvector.h:67-88) + popcount.
// A thread takes 16 bytes (one coalesced 16-byte load per lane: a wavefront reads 1 KB in one instruction), four
// neighbouring lanes put their 16 bits together.  (One thread per 64 bytes made every load instruction touch 64 lines.)
__device__ __forceinline__ uint32_t low_bits_of_16_bytes(const uint4 x) {
  const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
  uint32_t m = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t b = xs[t] & 0x01010101u;  // bytes are 0/1: gather the low bit of each of the 4 bytes
    m |= ((b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xFu) << (4 * t);
  }
  return m;
}
// m16 of lanes 4j..4j+3 -> bits of word j (returned on lane 4j)
__device__ __forceinline__ unsigned long long join_4_lanes(uint32_t m16) {
  return (unsigned long long)m16 | ((unsigned long long)__shfl_down(m16, 1, kWave) << 16) | ((unsigned long long)__shfl_down(m16, 2, kWave) << 32) |
         ((unsigned long long)__shfl_down(m16, 3, kWave) << 48);
}
// Persistent: a thread walks over chunks g, g + T, g + 2T, ... (T = all threads, a multiple of 4) and adds up its popcounts;
// one block reduction and one atomic per workgroup at the end.
__global__ __launch_bounds__(256) void k_pack_solid(const uint8_t *__restrict__ bytes, uint64_t n_bits, unsigned long long *__restrict__ words,
                                                    uint64_t n_words, unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x, n_chunks = n_words * 4;  // the byte map is padded to whole words
  uint32_t pop = 0;
  for (uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x; g0 < n_chunks; g0 += T) {  // workgroup-uniform trip count (shuffles inside)
    const uint64_t g = g0 + threadIdx.x, p0 = g * 16;
    uint32_t m16 = 0;
    if (g < n_chunks && p0 < n_bits) {
      m16 = low_bits_of_16_bytes(reinterpret_cast<const uint4 *>(bytes)[g]);
      if (p0 + 16 > n_bits) m16 &= (1u << (n_bits - p0)) - 1u;
    }
    const unsigned long long v = join_4_lanes(m16);
    if ((threadIdx.x & 3) == 0 && g < n_chunks) words[g >> 2] = v;
    pop += (uint32_t)__builtin_popcount(m16);
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>((uint64_t)pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

__global__ __launch_bounds__(256) void k_count_solid(const unsigned long long *__restrict__ words, uint64_t n_words,
                                                     unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  uint64_t pop = 0;  // one atomic per workgroup of a bounded grid: 10^5 atomics on one word cost ~1 ms by themselves
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x)
    pop += (uint64_t)__builtin_popcountll(words[w]);
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>(pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

// mark_mode 1: bytes mark the NON-solid occurrences; a position is solid iff a (k+1)-mer starts there
// (offset + k + 1 <= read length) and it is not marked.  One thread per 64 positions.
__global__ __launch_bounds__(256) void k_pack_solid_inv(const uint8_t *__restrict__ bytes, uint64_t n_bits, const uint64_t *__restrict__ start,
                                                        uint64_t n_seqs, uint32_t fixed_len, int k, unsigned long long *__restrict__ words,
                                                        uint64_t n_words, unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  uint64_t pop = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long v = 0;
    const uint64_t p0 = w * 64;
    uint64_t rid = p0 < n_bits ? seq_of_offset(start, n_seqs, fixed_len, p0) : 0;
    uint64_t re = start[rid + 1];
    const uint4 *pb = reinterpret_cast<const uint4 *>(bytes + p0);
    for (int q = 0; q < 4; ++q) {
      const uint4 x = pb[q];
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
      for (int t = 0; t < 16; ++t) {
        const uint64_t p = p0 + q * 16 + t;
        if (p >= n_bits) break;
        while (p >= re) re = start[++rid + 1];
        const bool valid = p + (uint64_t)k + 1 <= re;  // a (k+1)-mer of this read starts at p
        const bool marked = (xs[t >> 2] >> ((t & 3) * 8)) & 1u;
        if (valid && !marked) v |= 1ull << (q * 16 + t);
      }
    }
    words[w] = v;
    pop += (uint64_t)__builtin_popcountll(v);
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>(pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

// The same for reads of one length L with L - k >= 16: the valid positions of 16 consecutive ones follow from the offset
// of the first in its read (valid: offset <= L - k - 1), so a thread needs one 16-byte load, one remainder and a few masks
// instead of a 64-step walk over the read boundaries.
__global__ __launch_bounds__(256) void k_pack_solid_inv_fixed(const uint8_t *__restrict__ bytes, uint64_t n_bits, uint32_t L, int k,
                                                              unsigned long long *__restrict__ words, uint64_t n_words,
                                                              unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x, n_chunks = n_words * 4;
  const uint32_t step = (uint32_t)((T * 16) % L);  // the offset in the read advances by this much (mod L) per trip: one
  uint32_t off = (uint32_t)((((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16) % L);  // 64-bit remainder per thread, not per chunk
  uint32_t pop = 0;
  for (uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x; g0 < n_chunks; g0 += T) {
    const uint64_t g = g0 + threadIdx.x, p0 = g * 16;
    uint32_t m16 = 0;
    if (g < n_chunks && p0 < n_bits) {
      const uint32_t marked = low_bits_of_16_bytes(reinterpret_cast<const uint4 *>(bytes)[g]);
      const int t1 = min(max((int)L - k - (int)off, 0), 16);  // positions [0, t1): a (k+1)-mer of this read starts there
      const int t2 = min((int)L - (int)off, 16);              // positions [t2, 16): the next read, offsets < 16 <= L - k
      uint32_t valid = ((1u << t1) - 1u) | (0xFFFFu & ~((1u << t2) - 1u));
      if (p0 + 16 > n_bits) valid &= (1u << (n_bits - p0)) - 1u;
      m16 = valid & ~marked & 0xFFFFu;
    }
    const unsigned long long v = join_4_lanes(m16);
    if ((threadIdx.x & 3) == 0 && g < n_chunks) words[g >> 2] = v;
    pop += (uint32_t)__builtin_popcount(m16);
    off += step;
    if (off >= L) off -= L;
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>((uint64_t)pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

// multi-GPU: the adopted slice holds the summed NON-solid marks of the local reads -> is_solid = valid & ~marked
__global__ __launch_bounds__(256) void k_invert_marks(unsigned long long *__restrict__ words, uint64_t n_words, uint64_t n_bits,
                                                      const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t fixed_len, int k) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const uint64_t p0 = w * 64;
  unsigned long long valid = 0;
  if (p0 < n_bits) {
    uint64_t rid = seq_of_offset(start, n_seqs, fixed_len, p0);
    uint64_t re = start[rid + 1];
    for (int t = 0; t < 64; ++t) {
      const uint64_t p = p0 + t;
      if (p >= n_bits) break;
      while (p >= re) re = start[++rid + 1];
      if (p + (uint64_t)k + 1 <= re) valid |= 1ull << t;
    }
  }
  words[w] = valid & ~words[w];
}
void invert_local_marks(mhx_ctx *c, unsigned long long *words, uint64_t n_words) {
  SeqSet &s = c->seqs;
  if (n_words)
    MHX_LAUNCH(c, "invert_marks", (double)n_words * 16,
               hipLaunchKernelGGL(k_invert_marks, dim3((unsigned)div_ceil(n_words, 256)), dim3(256), 0, c->stream, words, n_words, s.n_bases,
                                  s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, (int)c->s1_acc_k));
}

template <int S, bool COMPACT, bool AGG>
static void s1_groups_launch(mhx_ctx *c, const uint32_t *sorted, uint64_t n_items, int KWv, int kmer_bits, uint32_t m,
                             uint8_t *is_solid, unsigned long long *solid_bits, int mark_atomic, unsigned long long *hist, unsigned long long *ctr, int want_mercy,
                             long long *&mercy, int k, uint2 *agg_items, uint64_t *agg_cursor, int mark_mode) {
  SeqSet &s = c->seqs;
  constexpr int T = S1Tile<S>::kT;
  const uint64_t n_tiles = div_ceil(n_items, T);
  const int full_words = kmer_bits / 32, rem = kmer_bits % 32;
  const uint32_t last_mask = rem ? 0xFFFFFFFFu << (32 - rem) : 0;
  const uint64_t pos_stride = COMPACT ? s1_pos_stride(c, (uint32_t)k) : 0;
  S1Op<S, COMPACT, AGG> op{k, agg_items, KWv, m, s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, is_solid, solid_bits, mark_atomic, mark_mode, hist, ctr, want_mercy, mercy, ctr + 1, pos_stride,
                           nullptr, nullptr};
  if (mark_mode == 2) {  // statistics on every 64th tile (no output): solid fraction -> marking polarity
    const uint32_t stride = 64;
    const uint64_t nt = div_ceil(n_tiles, stride);
    MHX_LAUNCH(c, "s1_sample", (double)nt * T * S * 4,
               hipLaunchKernelGGL((k_tile_groups<S, T, S1Op<S, COMPACT, false>, false>), dim3(tile_grid(nt)), dim3(kTileThreads), 0, c->stream, sorted,
                                  n_items, full_words, last_mask, S1Op<S, COMPACT, false>{k, nullptr, KWv, m, s.start.as<uint64_t>(), s.n_seqs,
                                  s.fixed_len, is_solid, solid_bits, mark_atomic, 2, hist, ctr, 0, mercy, ctr + 1, pos_stride, nullptr, nullptr},
                                  (uint64_t *)nullptr, (const uint64_t *)nullptr, n_tiles, nt, stride));
    return;
  }
  hipStream_t st = c->stream;
  const unsigned grid = tile_grid(n_tiles);
  // mercy candidates in per-workgroup regions of the spare sort buffer (2 entries of 8 bytes per 16-byte record): region b
  // starts at twice the number of records in the tiles of the workgroups before b — 2 * n_items entries in all
  const bool regions = !COMPACT && want_mercy && n_items && c->opt("s1_mercy_regions", 1);
  uint32_t *counts = nullptr;
  uint64_t *d_off = nullptr;
  if (regions) {
    counts = c->ws("s1_mercy_counts", (size_t)grid * 4 + 64).as<uint32_t>();
    MHX_HIP(hipMemsetAsync(counts, 0, (size_t)grid * 4 + 4, st));
    std::vector<uint64_t> off(grid + 1);
    const uint64_t q = n_tiles / grid, r = n_tiles % grid, b_last = (n_tiles - 1) % grid, short_by = n_tiles * (uint64_t)T - n_items;
    for (uint64_t b = 0; b <= grid; ++b) off[b] = 2 * ((uint64_t)T * (b * q + std::min<uint64_t>(b, r)) - (b > b_last ? short_by : 0));
    d_off = c->ws("s1_mercy_off", (size_t)(grid + 1) * 8).as<uint64_t>();
    MHX_HIP(hipMemcpyAsync(d_off, off.data(), (size_t)(grid + 1) * 8, hipMemcpyHostToDevice, st));
    MHX_HIP(hipStreamSynchronize(st));  // `off` is a local
    op.mercy_counts = counts;
    op.mercy_off = d_off;
  }
  // state to go back to should a region overflow: the histogram and (AGG) the cursor of the aggregated items
  unsigned long long *hist_save = nullptr;
  uint64_t agg_before[3] = {0, 0, 0};
  if (regions) {
    hist_save = c->ws("s1_hist_save2", (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
    MHX_HIP(hipMemcpyAsync(hist_save, hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
    if (AGG && agg_cursor) MHX_HIP(hipMemcpyAsync(agg_before, agg_cursor, 24, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  auto launch = [&]() {
    if constexpr (AGG)
      MHX_LAUNCH(c, "s1_groups", (double)n_items * S * 4,
                 hipLaunchKernelGGL((k_tile_groups<S, T, S1Op<S, COMPACT, true>, true>), dim3(grid), dim3(kTileThreads), 0, st, sorted, n_items,
                                    full_words, last_mask, op, agg_cursor, (const uint64_t *)nullptr, n_tiles, n_tiles));
    else
      MHX_LAUNCH(c, "s1_groups", (double)n_items * S * 4,
                 hipLaunchKernelGGL((k_tile_groups<S, T, S1Op<S, COMPACT, false>, false>), dim3(grid), dim3(kTileThreads), 0, st, sorted, n_items,
                                    full_words, last_mask, op, (uint64_t *)nullptr, (const uint64_t *)nullptr, n_tiles, n_tiles));
  };
  launch();
  if (!regions) return;
  std::vector<uint32_t> h_counts(grid + 1);
  MHX_HIP(hipMemcpyAsync(h_counts.data(), counts, (size_t)(grid + 1) * 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h_counts[grid] || c->opt("s1_mercy_regions", 1) == 2) {  // (2: tests force the way back)
    MHX_HIP(hipMemcpyAsync(hist, hist_save, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
    MHX_HIP(hipMemsetAsync(ctr, 0, 16, st));
    if (AGG && agg_cursor) MHX_HIP(hipMemcpyAsync(agg_cursor, agg_before, 24, hipMemcpyHostToDevice, st));
    op.mercy_counts = nullptr;
    launch();
    MHX_HIP(hipStreamSynchronize(st));
    return;
  }
  h_counts.resize(grid);
  uint64_t total = 0;
  for (unsigned i = 0; i < grid; ++i) total += h_counts[i];
  long long *dense = c->ws("s1_mercy_dense", total * 8 + 64).as<long long>();
  if (total)
    MHX_LAUNCH(c, "mercy_compact", (double)total * 16,
               hipLaunchKernelGGL(k_regions_compact, dim3(grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(mercy), d_off, counts,
                                  reinterpret_cast<uint2 *>(dense)));
  MHX_HIP(hipMemcpyAsync(ctr + 1, &total, 8, hipMemcpyHostToDevice, st));
  MHX_HIP(hipStreamSynchronize(st));  // `total` is a stack variable
  mercy = dense;
}

// int64 <-> (hi,lo) word pairs so that the big-endian record sort orders them numerically
__global__ void k_swap_words(uint32_t *__restrict__ v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint2 x = reinterpret_cast<uint2 *>(v)[i];
    reinterpret_cast<uint2 *>(v)[i] = make_uint2(x.y, x.x);
  }
}

// make `b` at least `bytes` large, preserving its first `keep` bytes
DevBuf &grow_preserving(mhx_ctx *c, DevBuf &b, size_t bytes, size_t keep) {
  if (b.cap >= bytes) return b;
  DevBuf nb;
  nb.reserve(bytes + bytes / 4);
  if (keep && b.p) MHX_HIP(hipMemcpyAsync(nb.p, b.p, keep, hipMemcpyDeviceToDevice, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  b.release();
  b = nb;
  return b;
}

// numeric sort of n 64-bit records on the device: swap to (hi,lo) words, record sort with 2 key words, swap back.
// Returns a pointer into the workspace ("u64_sort_a" / "u64_sort_b") holding the sorted copy.
const uint64_t *sort_u64(mhx_ctx *c, const void *src, uint64_t n, int hi_bit) {
  hipStream_t st = c->stream;
  uint32_t *ma = c->ws("u64_sort_a", n * 8 + 64).as<uint32_t>();
  uint32_t *mb = c->ws("u64_sort_b", n * 8 + 64).as<uint32_t>();
  if (!n) return reinterpret_cast<const uint64_t *>(ma);
  MHX_HIP(hipMemcpyAsync(ma, src, n * 8, hipMemcpyDeviceToDevice, st));
  const unsigned g2 = (unsigned)div_ceil(n, 256);
  hipLaunchKernelGGL(k_swap_words, dim3(g2), dim3(256), 0, st, ma, n);
  uint32_t *ms = radix_sort(c, ma, mb, n, 2, 2, make_passes(2, 0, hi_bit));
  hipLaunchKernelGGL(k_swap_words, dim3(g2), dim3(256), 0, st, ms, n);
  return reinterpret_cast<const uint64_t *>(ms);
}
// multi-GPU: position-keyed records (count events, mercy candidates) waiting to be routed to the ranks that hold the reads
void stash_route_records(mhx_ctx *c, const void *src, uint64_t n, int hi_bit) {
  const uint64_t *sorted = sort_u64(c, src, n, hi_bit);
  DevBuf &r = c->ws("route_records", n * 8 + 64);
  if (n) MHX_HIP(hipMemcpyAsync(r.p, sorted, n * 8, hipMemcpyDeviceToDevice, c->stream));
  c->n_route = n;
}

// ---- host driver, in two halves so that the multi-GPU path can exchange items in between ----
static int s1_kw(uint32_t k) { return (int)div_ceil((k - 1) * 2 + 6, 32); }  // read_to_sdbg_s1.cpp:107-108
// LSD passes of the stage-1 sort: the 6 head/tail bits, then the (k-1)-mer
static std::vector<SortPass> s1_sort_passes(uint32_t k) {
  const int KWv = s1_kw(k), kmer_bits = (int)(k - 1) * 2;
  return make_passes_ranges(KWv, {{0, 6}, {KWv * 32 - kmer_bits, KWv * 32}});
}
// Stage-1 sort plan.  seg_bits > 0: only the top seg_bits of the (k-1)-mer are sorted and an LDS group-by counts the equal
// keys of each segment (no-mercy 12-byte records).
//   stream: one workgroup streams one bucket of the seg_bits-bit prefix (k_s1_stream).  The width follows the DENSITY of the
//   job — records per lv1 bucket where the group-by runs — so that a streamed bucket holds at most s1_stream_max records
//   (a third of which are distinct keys at 60x coverage: what the 8192-slot table takes with room to spare) whatever the job
//   size: 16 bits and two LSD passes up to s1_stream_max records per lv1 bucket (10 M reads per GPU), the same two passes
//   with 2^sub0 sub-rounds per bucket up to 2^s1_stream_sub_max times that (a second read of the bucket is cheaper than a
//   third pass over all records), beyond that 17..24 bits in three passes.  A bucket that overflows anyway splits itself
//   (k_s1_stream): no job size and no single bucket sends the stage anywhere else.
//   otherwise: k_s1_seg on tiles, the width chosen so that a segment holds ~100 records.
struct S1Plan {
  std::vector<SortPass> passes;
  int seg_bits;
  bool stream;
  int sub0 = 0;            // stream: sub-rounds every bucket starts with (log2)
  double per_bucket = 0;   // the density the plan was made for
};
// records per lv1 bucket at the group-by: the items of the whole job over the buckets in play.  A rank of a multi-GPU run
// owns 1/n_parts of the key space; under a bucket filter (memory plan) the announced item count of the kept buckets stands
// for the call's own (the plan is made before the kept items are counted, and every later look must give the same plan);
// comm.hip sets the figure the ranks agreed on (s1_density).
static double s1_density(const mhx_ctx *c, uint64_t n_items) {
  if (c->s1_density > 0) return c->s1_density;
  const double n = c->filter_on ? (double)c->filter_expected : (double)n_items;
  const double buckets = c->filter_on && c->filter_kept ? (double)c->filter_kept : (double)MHX_NUM_BUCKETS;
  return n * (double)(c->n_parts > 1 ? c->n_parts : 1) / buckets;
}
// LSD passes over the top `pbits` key bits in digits of about equal width (<= 8 bits); every pass after the first declares
// the bits sorted before it: the consumers of these plans count equal keys, the order of records equal in all sorted bits is
// free (SortPass::prev_lo)
static std::vector<SortPass> s1_prefix_passes(int pbits) {
  const int np = (pbits + 7) / 8, lo = 64 - pbits;
  std::vector<SortPass> p;
  int at = lo;
  for (int i = 0; i < np; ++i) {
    const int w = pbits / np + (i < pbits % np ? 1 : 0);
    SortPass sp{at, w, 0, 0};
    sp.prev_lo = i ? lo : -1;
    p.push_back(sp);
    at += w;
  }
  return p;
}
static S1Plan s1_plan(const mhx_ctx *c, uint32_t k, uint64_t n_items, bool compact, int want_mercy, bool allow_stream = true) {
  const int force_bits = (int)c->opt("s1_seg_bits", 0);
  const int kmer_bits = (int)(k - 1) * 2;
  S1Plan p{s1_sort_passes(k), 0, false};
  if (!c->opt("s1_seg", 1) || !compact || want_mercy || s1_kw(k) != 2 || s1_stride(k, compact) != 3 || !n_items) return p;
  const double per_bucket = s1_density(c, n_items);
  p.per_bucket = per_bucket;
  if (allow_stream && c->opt("s1_stream", 1) && !force_bits && k >= 10 && k <= 22) {
    const double cap = (double)std::max<long long>(1, c->opt("s1_stream_max", 40000));
    const int sub_max = (int)std::min<long long>(std::max<long long>(c->opt("s1_stream_sub_max", 1), 0), 6);
    int need = 0;  // the buckets have to be 2^need times finer than the lv1 buckets
    while (need < 30 && per_bucket > cap * (double)(1ull << need)) ++need;
    int pbits = 16, sub0 = 0;
    if (need <= sub_max) sub0 = need;
    else {
      // a third pass costs the same for 17 or 24 prefix bits: aim at buckets that fill a third of the table — the inserts of a
      // table at two thirds take twice as long (tools/micro/insert_probe.hip), and the canonical (k-1)-mers make the low lv1
      // buckets twice as full as the average — but not at so many buckets that their fixed cost (a walk over 8192 slots) shows
      const double cap3 = (double)std::max<long long>(1, c->opt("s1_stream_max3", std::max<long long>(1, (long long)cap / 2)));
      need = 0;
      while (need < 30 && per_bucket > cap3 * (double)(1ull << need)) ++need;
      pbits = std::min(16 + std::max(need, 1), 24);
      sub0 = std::min(16 + need - pbits, 6);  // (past 24 bits: the rest as sub-rounds; the kernel splits further if it has to)
    }
    if (const long long f = c->opt("s1_stream_bits", 0)) pbits = (int)std::min<long long>(std::max<long long>(f, 9), 24);
    const long long fs = c->opt("s1_stream_sub0", -1);
    if (fs >= 0) sub0 = (int)std::min<long long>(fs, 6);
    pbits = std::min(pbits, kmer_bits);
    p.seg_bits = pbits;
    p.stream = true;
    p.sub0 = sub0;
    p.passes = s1_prefix_passes(pbits);
    return p;
  }
  const double n_eff = per_bucket * (double)MHX_NUM_BUCKETS;
  int bits = 8;
  while (bits < 32 && n_eff / 96.0 > (double)(1ull << bits)) bits += 8;
  if (force_bits) bits = force_bits;
  bits = std::max(1, std::min(bits, std::min(32, kmer_bits)));
  p.seg_bits = bits;
  p.passes = make_passes(2, 64 - bits, 64);
  return p;
}
// what a caller may print: "stream p16 s0 2 passes" / "seg 24" / "full sort"
std::string s1_plan_text(const mhx_ctx *c, uint32_t k, uint64_t n_items) {
  const S1Plan p = s1_plan(c, k, n_items, s1_compact(c, k, 0), 0);
  char buf[128];
  if (p.stream) snprintf(buf, sizeof buf, "stream p%d sub%d %zu passes (%.0f records per lv1 bucket)", p.seg_bits, p.sub0, p.passes.size(), p.per_bucket);
  else if (p.seg_bits) snprintf(buf, sizeof buf, "seg p%d %zu passes", p.seg_bits, p.passes.size());
  else snprintf(buf, sizeof buf, "full sort %zu passes", p.passes.size());
  return buf;
}
// Compact records (12 bytes at k <= 29: key + one position word) whenever no mercy candidates are wanted and the positions
// fit: below 2^pos_bits bases as they are, beyond that with the upper position bits as a tag inside the key words
// (s1_pos_tag: 8 spare bits between the (k-1)-mer and head/tail, i.e. up to 2^(pos_bits + 8) bases: 7 G reads of 150 bp).
uint32_t s1_pos_bits(const mhx_ctx *c) {
  const bool force = getenv("MHX_S1_FORCE_TAGGED") != nullptr;  // tests: tags at small sizes too
  const long long f = c->opt("s1_pos_bits", 0);
  if (f > 0) return (uint32_t)std::min<long long>(std::max<long long>(f, 4), 32);
  if (force) {  // the narrowest position word that keeps the tags below 128
    const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
    uint32_t b = 4;
    while (b < 32 && (n_bits >> b) >= 128) ++b;
    return b;
  }
  return 32;
}
bool s1_rank_tagged(const mhx_ctx *c, uint32_t k) {
  const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
  const uint32_t pb = s1_pos_bits(c);
  if ((n_bits >> pb) == 0) return false;  // every tag is 0
  const int spare = 32 * s1_kw(k) - (int)(k - 1) * 2 - 6;
  return spare >= 8 && (n_bits >> pb) < 256;
}
// global position = position word + tag * s1_pos_stride (0: no tags in this read set)
uint64_t s1_pos_stride(const mhx_ctx *c, uint32_t k) { return s1_rank_tagged(c, k) ? 1ull << s1_pos_bits(c) : 0ull; }
bool s1_compact(const mhx_ctx *c, uint32_t k, int want_mercy) {
  if (want_mercy) return false;
  if (s1_kw(k) < 2) return false;  // k <= 14: a one-word key; the 16-byte records serve (8-byte compact records have no kernels)
  const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
  return (n_bits >> s1_pos_bits(c)) == 0 || s1_rank_tagged(c, k);
}
int s1_stride(uint32_t k, bool compact) {
  const int kw = s1_kw(k);
  if (!compact) return round_up2(kw + 2);
  return kw + 1 == 3 ? 3 : round_up2(kw + 1);  // 12-byte records are supported natively, other odd widths are padded
}

// items of the local reads -> c->ws("items_a"); returns their number.
// Three ways, fastest first: (1) deferred — only the digit histograms of the coming sort are taken here and the sort's first
// pass makes the records itself (fixed-length reads, 12-byte records; under a bucket filter that pass drops the items of
// the other buckets: c->s1_filter_in_gen); (2) the window-arithmetic extraction; (3) the general kernels.
static bool s1_shape_is_fast(const mhx_ctx *c, uint32_t k, bool compact) {
  const SeqSet &s = c->seqs;
  return s.n_seqs && s.fixed_len >= k + 1 && compact && s1_kw(k) == 2 && s1_stride(k, compact) == 3 && k <= 29 && c->opt("s1_extract_fast", 1) != 0;
}
// The same front for a library whose reads are NOT of one length (S1GenVarT): item slots padded to the longest read's count.  Taken
// while at least half of the slots are real records (s1_var_min_fill per cent) — beyond that the extraction kernel + loaded passes
// cost less than generating dropped slots.
static bool s1_shape_is_var_fast(const mhx_ctx *c, uint32_t k, bool compact) {
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

// sort + group reduction of n_items items held in buf_a (buf_b = ping-pong space of the same size)
bool s1_presort_applies(const mhx_ctx *c, uint32_t k, uint64_t n_local_items) {
  // (k_s1_stream keeps the bounds and the arrays of up to kStreamSrcMax senders in LDS; more ranks take the classic exchange)
  return c->n_parts <= kStreamSrcMax && s1_plan(c, k, n_local_items, s1_compact(c, k, 0), 0).stream;
}
// the LSD passes that order this rank's stage-1 records by the plan's prefix (the first half of s1_process on the stream
// plan; the ranks agreed on the density the plan follows: mhx_ctx::s1_density)
uint32_t *s1_presort(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, int *pbits) {
  const S1Plan plan = s1_plan(c, k, std::max<uint64_t>(n_items, 1), true, 0);
  if (!plan.stream) throw Error("s1_presort: the bucket-streaming plan does not apply");
  *pbits = plan.seg_bits;
  uint32_t *sorted = n_items ? radix_sort(c, buf_a, buf_b, n_items, 3, s1_kw(k), plan.passes) : buf_a;
  c->pre_hist_buf = nullptr;
  return sorted;
}

// ---- stage 1 behind the extraction, in steps (Read2SdbgS1::Lv2Postprocess and what it needs, read_to_sdbg_s1.cpp:368-555) ----
//   sort_records   the plan's passes (or the reference-exact order for want_mercy == 2)
//   open_outputs   is_solid / histogram / marks / aggregated stage-2 items, continued or fresh (bucket-range passes accumulate)
//   polarity       mark the solid or the non-solid occurrences (a 1/64 sample decides on a single GPU)
//   group_partial  the LDS group-bys on partially sorted records (bucket streaming, segments), with their ways back
//   group_classic  full sort + k_tile_groups (mercy, wide keys, every give-up)
//   publish        bitmap, counters, mercy candidates, result buffers
namespace {
struct S1Stage {
  mhx_ctx *c;
  uint32_t k, m;
  int want_mercy;
  uint32_t *buf_a, *buf_b;
  uint64_t n_items;
  const S1Sources *pre;
  SeqSet &s;
  hipStream_t st;
  bool compact, global, tagged_keys;
  int KWv, S, kmer_bits;
  size_t item_bytes;
  uint64_t pos_stride;
  S1Plan plan;
  uint32_t *sorted = nullptr, *spare = nullptr;
  // outputs
  uint64_t n_bits = 0, n_words64 = 0;
  int mark_atomic = 0;
  const char *mark_env = nullptr;
  bool sparse = false, acc = false;
  bool list_local = false;  // the last launch_partial left its marks in the workgroups' regions (one GPU: applied by run_partial)
  unsigned long long *is_solid = nullptr, *hist = nullptr, *ctr = nullptr;
  uint8_t *solid_bytes = nullptr;
  uint64_t marks_prev = 0, seg_marks = 0;
  const uint32_t *giant_ctr = nullptr;  // device counters of the giant path of the last launch_partial (nullptr: not taken)
  bool classic_ran = false;
  long long *mercy = nullptr;
  bool agg = false;
  uint2 *agg_items = nullptr;
  uint64_t *agg_cursor = nullptr;
  uint64_t agg_prev = 0, agg_bound = 0;
  uint32_t seg_grid = 0, seg_cap = 0, seg_mcap = 0;
  uint32_t *seg_err = nullptr;
  int mark_mode = 0, mark_mode_used = 0;

  S1Stage(mhx_ctx *c_, uint32_t k_, uint32_t m_, int wm, uint32_t *a, uint32_t *b, uint64_t n, const S1Sources *pre_)
      : c(c_), k(k_), m(m_), want_mercy(wm), buf_a(a), buf_b(b), n_items(n), pre(pre_), s(c_->seqs), st(c_->stream) {
    compact = s1_compact(c, k, want_mercy);
    KWv = s1_kw(k);
    S = s1_stride(k, compact);
    item_bytes = (size_t)S * 4;
    global = c->global_bases != 0;  // multi-GPU: positions index the global read set
    kmer_bits = (int)(k - 1) * 2;
    // (records that carry position bits between the (k-1)-mer and head/tail must not be ordered by whole key words)
    tagged_keys = compact && s1_rank_tagged(c, k);
    pos_stride = compact ? s1_pos_stride(c, k) : 0;
    plan = s1_plan(c, k, n_items, compact, want_mercy);
    // pre: the records come pre-sorted by the plan's prefix in several arrays (multi-GPU: one per sending rank, comm.hip);
    // n_items is their total.  Only the bucket-streaming group-by reads them in place; if it gives up, they are gathered and
    // the stage continues as if they had arrived unsorted.
    if (pre && (!plan.stream || want_mercy || !compact)) throw Error("s1_process: pre-sorted sources need the bucket-streaming plan");
    if (pre && pre->pbits != plan.seg_bits) throw Error("s1_process: the sources were sorted for another plan than the one this rank makes");
  }

  void sort_records() {
    // want_mercy == 2: records with equal keys in exactly the order the reference's kmsort leaves them (H1)
    sorted = pre ? nullptr
             : want_mercy == 2
                 ? kmsort_exact(c, buf_a, buf_b, n_items, S, KWv)
                 : (plan.seg_bits || tagged_keys ? radix_sort(c, buf_a, buf_b, n_items, S, KWv, plan.passes)
                                                 : sort_whole_key(c, buf_a, buf_b, n_items, S, KWv, plan.passes));
    c->pre_hist_buf = nullptr;
    set_spare(pre ? pre->spare : (sorted == buf_a ? buf_b : buf_a));
  }
  void set_spare(uint32_t *sp) {
    spare = sp;
    mercy = reinterpret_cast<long long *>(spare);  // mercy candidates (<= 2 per item, 8 B each): S*4 >= 16 bytes per item
  }
  void ensure_byte_map() {  // (sparse: only when the classic kernel runs; always zeroed, its marks are collected afterwards)
    if (solid_bytes) return;
    const uint64_t nw = div_ceil(n_bits, 64);
    solid_bytes = c->ws("solid_bytes", (nw + 1) * 64).as<uint8_t>();
    if (!acc || sparse) MHX_HIP(hipMemsetAsync(solid_bytes, 0, (nw + 1) * 64, st));
  }
  void open_outputs() {
    n_bits = global ? c->global_bases : s.n_bases;
    // MHX_S1_MARK: atomic (atomicOr into the bitmap) | solid | nonsolid (force the byte-map polarity) | unset = auto
    mark_env = getenv("MHX_S1_MARK");
    mark_atomic = mark_env && !strcmp(mark_env, "atomic") ? 1 : 0;
    // multi-GPU with sparse marks (comm.hip): the marks of the non-solid occurrences leave this stage as a list of
    // global positions (ws "s1_marks", c->n_marks) to be routed to the read owners; no bitmap / byte map of the GLOBAL read
    // set exists unless the classic tile kernel has to run (then its byte map is converted to the list)
    sparse = global && !mark_atomic && c->opt("dist_sparse_marks", 0) != 0;
    n_words64 = sparse ? 0 : div_ceil(n_bits, 64);
    is_solid = c->result(MHX_BUF_IS_SOLID, (n_words64 + 1) * 8).as<unsigned long long>();
    c->results[MHX_BUF_IS_SOLID].used = n_words64 * 8;
    hist = c->result(MHX_BUF_MUL_HIST, (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
    // accumulate (bucket-range passes after the first, passes.hip): marks, histogram, aggregated stage-2 items and
    // mercy candidates of the earlier passes are kept and the published results are cumulative
    acc = c->accumulate && c->s1_acc_bits == n_bits && c->s1_acc_k == k && c->s1_acc_m == m;
    marks_prev = sparse && acc ? c->n_marks : 0;  // marks of the earlier bucket-range passes stay in front
    if (!sparse || !acc) c->n_marks = 0;
    if (mark_atomic) {
      if (!acc) MHX_HIP(hipMemsetAsync(is_solid, 0, (n_words64 + 1) * 8, st));
    } else {
      if (!sparse) ensure_byte_map();
      MHX_HIP(hipMemsetAsync(is_solid + n_words64, 0, 8, st));
    }
    if (!acc) MHX_HIP(hipMemsetAsync(hist, 0, (MHX_MAX_MUL + 1) * 8, st));
    c->s1_acc_bits = n_bits;
    c->s1_acc_k = k;
    c->s1_acc_m = m;
    ctr = c->ws("s1_counters", 64).as<unsigned long long>();
    MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
    // aggregated stage-2 items (k <= 22, m >= 2): at most 2 per solid run, a solid run has >= m records
    const bool agg_off = getenv("MHX_S2_PER_OCCURRENCE") != nullptr;  // (read per call: a resident server answers requests with different environments)
    agg = !agg_off && k <= 22 && m >= 2 && KWv == 2;
    const bool agg_continues = acc && c->agg_valid && c->agg_k == k && c->agg_m == m;
    c->agg_valid = false;
    agg_cursor = c->ws("s2_agg_cursor", 64).as<uint64_t>();
    agg_prev = agg_continues ? c->agg_n : 0;  // items of the earlier passes stay in front
    agg_bound = (n_items / m + 16) * 2;
    seg_err = c->ws("s1_seg_err", 64).as<uint32_t>();
  }
  void agg_prepare_classic() {  // k_tile_groups appends to the dense array through a global cursor
    if (!agg) return;
    agg_items = grow_preserving(c, c->work["s2_agg_items"], (agg_prev + agg_bound) * 8, agg_prev * 8).as<uint2>();
    MHX_HIP(hipMemsetAsync(agg_cursor, 0, 24, st));
    if (agg_prev) MHX_HIP(hipMemcpyAsync(agg_cursor, &agg_prev, 8, hipMemcpyHostToDevice, st));
  }

  // the LDS group-bys on the partially sorted records: k_s1_stream (one bucket of the plan's prefix per workgroup at a time) or
  // k_s1_seg (tiles).  mode = mark_mode (2: statistics on a 1/64 sample)
  void launch_partial(int mode) {
    const int per = (int)c->opt("s1_seg_per", 8);
    const int la = (int)std::min<long long>(std::max<long long>(c->opt("s1_seg_la", 3), 0), 200);  // 16-bit tile counters
    const int T = 256 * (per == 4 ? 4 : 8);
    const uint64_t n_tiles = div_ceil(n_items, (uint64_t)T);
    const uint32_t stride = mode == 2 ? 64u : 1u;
    const uint64_t n_buckets = plan.stream ? 1ull << plan.seg_bits : 0;
    const uint64_t n_work = plan.stream ? div_ceil(n_buckets, stride) : div_ceil(n_tiles, stride);
    const uint32_t pfx_mask = plan.seg_bits >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> plan.seg_bits);
    const uint32_t eq_mask1 = (kmer_bits > 32 ? ~(0xFFFFFFFFu >> (kmer_bits - 32)) : 0u) | 63u;
    const bool agg_on = agg && mode != 2;
    // s1_stream_half: two 512-thread workgroups with 4096-slot tables per CU instead of one with 1024 threads and 8192 slots
    const bool half = plan.stream && c->opt("s1_stream_half", 0) != 0;
    const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_work, plan.stream ? (half ? 2 * cus : cus) : (per == 4 ? 256 * 6 : 256 * 3));
    // per-workgroup output regions in the spare sort buffer (S*4 >= 12 bytes per record, outputs are 8-byte entries)
    const uint32_t region = (uint32_t)std::min<uint64_t>(n_items * (uint64_t)S * 4 / 8 / grid, 0xFFFFFFF0u);
    uint2 *raw = nullptr;
    uint32_t *counts = nullptr;
    if (agg_on) {
      seg_grid = grid;
      seg_cap = region;
      raw = reinterpret_cast<uint2 *>(spare);
      counts = c->ws("s2_agg_counts", (size_t)grid * 4).as<uint32_t>();
    }
    // sparse marks go to the spare sort buffer (>= 12 bytes per record, a record yields at most one 8-byte mark): a
    // workgroup's region holds every record it can meet
    unsigned long long *mraw = nullptr;
    uint32_t *mcounts = nullptr;
    uint32_t mcap = 0;
    // s1_marks_list (one GPU; off by default): the marks taken from the table leave the streaming kernel the same way and are applied
    // to the byte map by a kernel of their own (k_apply_mark_regions) instead of being stored byte by byte from inside the inserts.
    // Measured (profiles/r05_ab_marks_list.jsonl): k_s1_stream 7.89 -> 5.90 ms, but the 1.3 x 10^8 random byte stores then cost 2.94 ms
    // on their own (8.3 GB of partial lines; the multi-GPU path pays 1.2 ms because its list went through an 8-bit pass over the top
    // position bits first, which a single GPU would have to add): 35.0 -> 36.0 ms per step.  Kept as a knob, not as the default.
    list_local = !sparse && mode == 1 && plan.stream && m <= 2 && solid_bytes != nullptr && c->opt("s1_stream_direct", 1) != 0 &&
                 c->opt("s1_stream_half", 0) == 0 && c->opt("s1_marks_list", 0) != 0;
    if ((sparse || list_local) && mode != 2) {
      mraw = reinterpret_cast<unsigned long long *>(spare);
      mcap = region;
      mcounts = c->ws("s1_mark_counts", (size_t)grid * 4).as<uint32_t>();
      seg_grid = grid;
      seg_mcap = mcap;
    }
    // the stream kernel marks the non-solid occurrences from its table when each of them is its key's only record
    const int direct = plan.stream && mode == 1 && m <= 2 && (mraw || solid_bytes) && c->opt("s1_stream_direct", 1) ? 1 : 0;
    S1SegArgs a{(int)k, m, pfx_mask, eq_mask1, solid_bytes, mode, hist, ctr, raw, seg_cap, counts, mraw, mcap, mcounts, pos_stride, seg_err,
                plan.stream ? (int)c->opt("s1_stream_probes", 1024) : la, direct};
    MHX_HIP(hipMemsetAsync(seg_err, 0, 4, st));
    const char *nm = mode == 2 ? "s1_sample" : "s1_groups";
    const double bytes = plan.stream ? (double)n_items * 12 / stride * (double)(1u << plan.sub0) : (double)n_work * T * 12;
    if (plan.stream) {
      const int n_src = pre && sorted == nullptr ? pre->n : 1;
      uint64_t *bounds = c->ws("s1_bucket_bounds", (size_t)n_src * (n_buckets + 1) * 8 + 64).as<uint64_t>();
      uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
      MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
      const unsigned bgrid = (unsigned)((n_buckets + 1 + 255) / 256);
      const uint32_t *const *srcs = nullptr;
      if (n_src > kStreamSrcMax) throw Error("s1: more pre-sorted sources than the bucket streaming takes");
      if (n_src > 1 || (pre && sorted == nullptr)) {
        DevBuf &sp = c->ws("s1_src_ptrs", (size_t)n_src * 8 + 64);
        MHX_HIP(hipMemcpyAsync(sp.p, pre->ptr.data(), (size_t)n_src * 8, hipMemcpyHostToDevice, st));
        srcs = sp.as<const uint32_t *>();
        for (int q = 0; q < n_src; ++q)
          hipLaunchKernelGGL(k_bucket_bounds, dim3(bgrid), dim3(256), 0, st, pre->ptr[q], pre->count[q], 3, bounds + (size_t)q * (n_buckets + 1),
                             plan.seg_bits);
      } else {
        MHX_LAUNCH(c, "bucket_bounds", (double)n_buckets * 8 * 30,
                   hipLaunchKernelGGL(k_bucket_bounds, dim3(bgrid), dim3(256), 0, st, sorted, n_items, 3, bounds, plan.seg_bits));
      }
      const uint32_t *items0 = pre && sorted == nullptr ? pre->ptr[0] : sorted;
      const uint32_t nslot = half ? 4096u : 8192u;
      const S1StreamGeom geo{plan.seg_bits, plan.sub0, (uint32_t)n_buckets,
                             (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", nslot * 7 / 8), 1), nslot)};
      const bool tags = pos_stride != 0;
      // giant buckets (S1Giant): found, cut into slices and reduced on the device before the streaming launch, finished by a second
      // launch of the streaming kernel over the same grid (direct marks only: a key's first record stands for its slice)
      const bool giant_on = direct && !half && mode == 1 && c->opt("s1_giant", 1) != 0;
      uint32_t *ticket2 = nullptr;
      if (giant_on) {
        S1Giant &g = a.giant;
        g.gcap = 4096;
        g.min_records = (uint32_t)std::max<long long>(1, c->opt("s1_giant_min", 262144));
        g.pcap = std::max<uint64_t>(2u << 20, n_items / 64);
        g.flag = c->ws("s1_giant_flag", n_buckets + 64).as<uint8_t>();
        uint32_t *lists = c->ws("s1_giant_lists", 64 + (size_t)g.gcap * (5 * 4 + 8)).as<uint32_t>();
        g.ctr = lists;
        ticket2 = lists + 8;
        g.off = reinterpret_cast<unsigned long long *>(lists + 16);
        g.bucket = lists + 16 + 2 * g.gcap;
        g.sl = g.bucket + g.gcap;
        g.ns = g.sl + g.gcap;
        g.cap = g.ns + g.gcap;
        g.cur = g.cap + g.gcap;
        g.partial = c->ws("s1_giant_partial", g.pcap * 16 + 64).as<uint4>();
        giant_ctr = lists;
        MHX_HIP(hipMemsetAsync(g.flag, 0, n_buckets, st));
        MHX_HIP(hipMemsetAsync(lists, 0, 64, st));
        MHX_LAUNCH(c, "s1_giant_find", (double)n_src * n_buckets * 8,
                   hipLaunchKernelGGL(k_s1_giant_find, dim3((unsigned)div_ceil(n_buckets, 256)), dim3(256), 0, st, bounds, n_src, (uint32_t)n_buckets, g));
        MHX_LAUNCH(c, "s1_giant_reduce", 0.0,
                   hipLaunchKernelGGL(k_s1_giant_reduce, dim3((unsigned)(3 * cus)), dim3(256), 0, st, items0, srcs, bounds, n_src, (uint32_t)n_buckets,
                                      plan.seg_bits, (int)k, g));
      }
#define MHX_STREAM(AGGV, NTV, LOGV, TAGV)                                                                                                 \
  MHX_LAUNCH(c, nm, bytes, hipLaunchKernelGGL((k_s1_stream<AGGV, 4, NTV, LOGV, TAGV>), dim3(grid), dim3(NTV), 0, st, items0, bounds, a, geo, \
                                              stride, ticket, srcs, n_src))
#define MHX_STREAM_T(AGGV, NTV, LOGV)        \
  do {                                       \
    if (tags) MHX_STREAM(AGGV, NTV, LOGV, true); \
    else MHX_STREAM(AGGV, NTV, LOGV, false);     \
  } while (0)
      if (half) {
        if (agg_on) MHX_STREAM_T(true, 512, 12);
        else MHX_STREAM_T(false, 512, 12);
      } else {
        if (agg_on) MHX_STREAM_T(true, kStreamThreads, 13);
        else MHX_STREAM_T(false, kStreamThreads, 13);
      }
#undef MHX_STREAM_T
#undef MHX_STREAM
      if (giant_on) {  // the giants' partial entries -> the same per-key work, the same per-workgroup output regions
#define MHX_GIANT(AGGV, TAGV)                                                                                                                  \
  MHX_LAUNCH(c, "s1_giant_groups", 0.0, hipLaunchKernelGGL((k_s1_stream<AGGV, 4, kStreamThreads, 13, TAGV, true>), dim3(grid), dim3(kStreamThreads), 0, \
                                                           st, items0, bounds, a, geo, 1u, ticket2, (const uint32_t *const *)nullptr, 1))
        if (agg_on && tags) MHX_GIANT(true, true);
        else if (agg_on) MHX_GIANT(true, false);
        else if (tags) MHX_GIANT(false, true);
        else MHX_GIANT(false, false);
#undef MHX_GIANT
      }
      return;
    }
#define MHX_SEG(PERV, AGGV) \
  MHX_LAUNCH(c, nm, bytes, hipLaunchKernelGGL((k_s1_seg<PERV, AGGV>), dim3(grid), dim3(256), 0, st, sorted, n_items, a, n_work, stride))
    if (per == 4) {
      if (agg_on) MHX_SEG(4, true);
      else MHX_SEG(4, false);
    } else {
      if (agg_on) MHX_SEG(8, true);
      else MHX_SEG(8, false);
    }
#undef MHX_SEG
  }

  template <int SV, bool CP, bool AGGV>
  void classic_case(int wm) {
    s1_groups_launch<SV, CP, AGGV>(c, sorted, n_items, KWv, kmer_bits, m, solid_bytes, is_solid, mark_atomic, hist, ctr, wm, mercy, (int)k, agg_items,
                                   agg_cursor, mark_mode);
  }
  void launch_classic_plain() {  // k_tile_groups without aggregated items
#define MHX_CASE(SV)                                  \
  case SV:                                            \
    if (compact) classic_case<SV, true, false>(0);    \
    else classic_case<SV, false, false>(want_mercy);  \
    break;
    switch (S) {
      MHX_CASE(3) MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
      default: throw Error("read2sdbg_s1: unsupported record stride");
    }
#undef MHX_CASE
  }

  // marking polarity from a 1/64 sample: when most occurrences are solid it is cheaper to mark the non-solid ones (each
  // mark is a partial HBM write).  Single GPU only: ranks must agree on the meaning.
  void polarity() {
    const bool can_invert = !global && !mark_atomic && !c->filter_on && !c->accumulate;  // passes must agree on the meaning
    // multi-GPU: every rank marks the NON-solid occurrences of the buckets it owns (a fixed convention, so that the
    // summed bitmaps mean the same on all ranks; typical inputs are mostly solid);
    // mhx_adopt_is_solid_slice turns "valid position and not marked" into the local is_solid
    if (global && !mark_atomic) mark_mode = 1;
    // bucket-range passes (memory plan): the same fixed convention, so that the passes' marks add up — and the bucket
    // streaming takes them straight from its table (direct_marks)
    else if ((c->filter_on || c->accumulate) && !mark_atomic && !mark_env) mark_mode = 1;
    else if (can_invert && mark_env && !strcmp(mark_env, "nonsolid")) mark_mode = 1;
    else if (can_invert && !mark_env && n_items > (1u << 16)) {
      mark_mode = 2;
      if (plan.seg_bits) launch_partial(2);
      else launch_classic_plain();
      unsigned long long hs[3] = {0, 0, 0};
      MHX_HIP(hipMemcpyAsync(hs, ctr, 24, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      mark_mode = hs[2] > 0 && hs[0] * 2 > hs[2] ? 1 : 0;
      MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
    }
  }

  // one run of the partial group-by: launch, read its error word and region counts back, pack the regions.  -> error word
  uint32_t run_partial() {
    giant_ctr = nullptr;
    launch_partial(mark_mode);
    uint32_t e = 0;
    std::vector<uint32_t> h_counts(agg ? seg_grid : 0), h_mcounts(sparse ? seg_grid : 0);
    uint32_t h_giant[4] = {0, 0, 0, 0};  // giants found, -, partial entries allotted (64 bits)
    if (giant_ctr) MHX_HIP(hipMemcpyAsync(h_giant, giant_ctr, 16, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipMemcpyAsync(&e, seg_err, 4, hipMemcpyDeviceToHost, st));
    if (agg) MHX_HIP(hipMemcpyAsync(h_counts.data(), c->work["s2_agg_counts"].p, (size_t)seg_grid * 4, hipMemcpyDeviceToHost, st));
    if (sparse) MHX_HIP(hipMemcpyAsync(h_mcounts.data(), c->work["s1_mark_counts"].p, (size_t)seg_grid * 4, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    if (e) return e;
    if (h_giant[0]) c->last_s1_plan += " [" + std::to_string(h_giant[0]) + " giant buckets in slices]";
    if (list_local)
      MHX_LAUNCH(c, "apply_marks", (double)n_items * 0.1 * 9,
                 hipLaunchKernelGGL(k_apply_mark_regions, dim3(seg_grid, 16), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(spare), seg_mcap,
                                    c->work["s1_mark_counts"].as<uint32_t>(), solid_bytes, (uint64_t)((n_words64 + 1) * 64)));
    if (sparse) {  // pack the workgroups' mark regions (they live in the spare sort buffer) behind the earlier passes' marks
      for (uint32_t v : h_mcounts) seg_marks += v;
      unsigned long long *dense = grow_preserving(c, c->work["s1_marks"], (marks_prev + seg_marks) * 8 + 64, marks_prev * 8).as<unsigned long long>();
      if (seg_marks)
        MHX_LAUNCH(c, "marks_compact", (double)seg_marks * 16,
                   hipLaunchKernelGGL(k_agg_compact, dim3(seg_grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(spare), seg_mcap,
                                      c->work["s1_mark_counts"].as<uint32_t>(), reinterpret_cast<uint2 *>(dense + marks_prev), 0));
      c->n_marks = marks_prev + seg_marks;
    }
    if (agg) {  // pack the workgroups' regions behind the items of the earlier passes
      uint64_t total = 0;
      for (uint32_t v : h_counts) total += v;
      uint2 *dense = grow_preserving(c, c->work["s2_agg_items"], (agg_prev + total) * 8 + 64, agg_prev * 8).as<uint2>();
      if (total)
        MHX_LAUNCH(c, "agg_compact", (double)total * 16,
                   hipLaunchKernelGGL(k_agg_compact, dim3(seg_grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(spare), seg_cap,
                                      c->work["s2_agg_counts"].as<uint32_t>(), dense + agg_prev, 1));
      const uint64_t agg_n = agg_prev + total;
      MHX_HIP(hipMemcpyAsync(agg_cursor, &agg_n, 8, hipMemcpyHostToDevice, st));
      MHX_HIP(hipStreamSynchronize(st));  // agg_n is a stack variable
    }
    return 0;
  }
  // the sources of a pre-sorted stage become one unsorted array (a way back only)
  void gather_sources() {
    buf_a = c->ws("s1_gather_a", n_items * item_bytes + 64).as<uint32_t>();
    buf_b = c->ws("s1_gather_b", n_items * item_bytes + 64).as<uint32_t>();
    uint64_t at = 0;
    for (int q = 0; q < pre->n; ++q) {
      if (pre->count[q]) MHX_HIP(hipMemcpyAsync(buf_a + at * S, pre->ptr[q], pre->count[q] * item_bytes, hipMemcpyDeviceToDevice, st));
      at += pre->count[q];
    }
    sorted = buf_a;
  }
  // -> true when the partial group-by did the job; false: the records are fully sorted now and the classic kernel has to run.
  // The ways back (marks are idempotent, the histogram is restored): streaming -> segments when an output REGION of the
  // streaming overflowed (a bucket whose keys overflow the table is split inside the kernel, it never comes here);
  // segments -> full sort + classic when a segment outgrows the look-ahead or a tile's table.
  bool group_partial() {
    unsigned long long *hist_save = c->ws("s1_hist_save", (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
    MHX_HIP(hipMemcpyAsync(hist_save, hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
    for (;;) {
      const uint32_t e = run_partial();
      if (!e) return true;
      MHX_HIP(hipMemcpyAsync(hist, hist_save, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
      seg_marks = 0;
      uint32_t *other;
      if (plan.stream) {
        if (pre && sorted == nullptr) gather_sources();
        plan = s1_plan(c, k, n_items, compact, want_mercy, false);
        other = sorted == buf_a ? buf_b : buf_a;
        sorted = radix_sort(c, sorted, other, n_items, S, KWv, plan.passes);
        set_spare(sorted == buf_a ? buf_b : buf_a);
        continue;
      }
      other = sorted == buf_a ? buf_b : buf_a;
      sorted = tagged_keys ? radix_sort(c, sorted, other, n_items, S, KWv, s1_sort_passes(k)) : sort_whole_key(c, sorted, other, n_items, S, KWv, s1_sort_passes(k));
      set_spare(sorted == buf_a ? buf_b : buf_a);
      return false;
    }
  }
  void group_classic() {
    agg_prepare_classic();
    ensure_byte_map();
    classic_ran = true;
    if (agg && S == 3) classic_case<3, true, true>(0);
    else if (agg && S == 4 && !compact) classic_case<4, false, true>(want_mercy);
    else launch_classic_plain();
  }

  void publish(mhx_s1_result *out) {
    if (agg && (S == 3 || (S == 4 && !compact))) {  // also with zero local items: every rank takes the same stage-2 path
      c->agg_n = 0;
      MHX_HIP(hipMemcpyAsync(&c->agg_n, agg_cursor, 8, hipMemcpyDeviceToHost, st));
      c->agg_valid = true;
      c->agg_k = k;
      c->agg_m = m;
    }
    if (sparse && classic_ran) {  // the classic kernel marked a byte map of the global read set: turn it into the list
      const uint64_t n_bytes = div_ceil(n_bits, 64) * 64;
      unsigned long long *cur = c->ws("s1_mark_cursor", 64).as<unsigned long long>();
      MHX_HIP(hipMemsetAsync(cur, 0, 8, st));
      // upper bound of the marks: every record
      unsigned long long *dense = grow_preserving(c, c->work["s1_marks"], (marks_prev + n_items) * 8 + 64, marks_prev * 8).as<unsigned long long>();
      MHX_LAUNCH(c, "collect_marks", (double)n_bytes,
                 hipLaunchKernelGGL(k_collect_marks, dim3((unsigned)div_ceil(n_bytes, 256 * 16)), dim3(256), 0, st, solid_bytes, n_bytes,
                                    dense + marks_prev, cur));
      uint64_t got = 0;
      MHX_HIP(hipMemcpyAsync(&got, cur, 8, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      c->n_marks = marks_prev + got;
    }
    if (n_words64 && mark_atomic)
      MHX_LAUNCH(c, "count_solid", (double)n_words64 * 8,
                 hipLaunchKernelGGL(k_count_solid, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64, 256), 4096)), dim3(256), 0, st, is_solid, n_words64, ctr));
    c->global_marks_inverted = global && !mark_atomic;
    if (n_words64 && !mark_atomic && mark_mode_used == 1 && !global) {
      if (s.fixed_len >= k + 16 && c->opt("s1_pack_fixed", 1))
        MHX_LAUNCH(c, "pack_solid", (double)n_words64 * 72,
                   hipLaunchKernelGGL(k_pack_solid_inv_fixed, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64 * 4, 256), 4096)), dim3(256), 0, st, solid_bytes, n_bits,
                                      s.fixed_len, (int)k, is_solid, n_words64, ctr));
      else
        MHX_LAUNCH(c, "pack_solid", (double)n_words64 * 72,
                   hipLaunchKernelGGL(k_pack_solid_inv, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64, 256), 4096)), dim3(256), 0, st, solid_bytes, n_bits,
                                      s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, (int)k, is_solid, n_words64, ctr));
    }
    if (n_words64 && !mark_atomic && (mark_mode_used != 1 || global))  // global: the marks themselves (see polarity)
      MHX_LAUNCH(c, "pack_solid", (double)n_words64 * 72,
                 hipLaunchKernelGGL(k_pack_solid, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64 * 4, 256), 4096)), dim3(256), 0, st, solid_bytes, n_bits, is_solid,
                                    n_words64, ctr));
    uint64_t n_solid = 0, n_mercy = 0;
    {
      unsigned long long h[2];
      MHX_HIP(hipMemcpyAsync(h, ctr, 16, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      n_solid = h[0];
      n_mercy = h[1];
      if (want_mercy && (c->accumulate || c->filter_on)) {  // keep the candidates of every pass; publish their sorted union
        const uint64_t prev = acc ? c->mercy_acc_n : 0;
        DevBuf &ma = grow_preserving(c, c->work["mercy_acc"], (prev + n_mercy) * 8 + 8, prev * 8);
        if (n_mercy) MHX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(ma.p) + prev * 8, mercy, n_mercy * 8, hipMemcpyDeviceToDevice, st));
        c->mercy_acc_n = n_mercy = prev + n_mercy;
        mercy = ma.as<long long>();
      }
      if (want_mercy && n_mercy) {
        int hi_bit = 3;
        while (hi_bit < 64 && ((n_bits << 2) >> hi_bit)) ++hi_bit;
        const uint64_t *ms = sort_u64(c, mercy, n_mercy, hi_bit);
        DevBuf &res = c->result(MHX_BUF_MERCY_CAND, n_mercy * 8);
        MHX_HIP(hipMemcpyAsync(res.p, ms, n_mercy * 8, hipMemcpyDeviceToDevice, st));
      }
    }
    if (!want_mercy || !n_mercy) {
      c->result(MHX_BUF_MERCY_CAND, 8);
      c->results[MHX_BUF_MERCY_CAND].used = 0;
    }
    c->results[MHX_BUF_SORTED_ITEMS].release();
    c->results[MHX_BUF_SORTED_ITEMS].p = sorted;
    c->results[MHX_BUF_SORTED_ITEMS].cap = 0;
    c->results[MHX_BUF_SORTED_ITEMS].used = sorted ? n_items * item_bytes : 0;  // (pre-sorted sources stay where they are)
    c->sorted_item_words = S;
    MHX_HIP(hipStreamSynchronize(st));
    if (out) {
      out->n_items = n_items;
      out->n_solid = n_solid;
      out->n_mercy_cand = want_mercy ? n_mercy : 0;
      out->item_words = S;
    }
  }
};
}  // namespace

int s1_process(mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items,
               mhx_s1_result *out, const S1Sources *pre) {
  S1Stage stage(c, k, m, want_mercy, buf_a, buf_b, n_items, pre);
  c->last_s1_plan = s1_plan_text(c, k, n_items);
  if (c->s1_var_gen) c->last_s1_plan += " [reads of several lengths: " + std::to_string(c->seqs.max_len - k + 4) + " item slots per read on the generating pass]";
  c->s1_var_gen = false;
  stage.sort_records();
  stage.open_outputs();
  if (n_items) {
    stage.polarity();
    const bool done = stage.plan.seg_bits && stage.group_partial();
    if (!done) stage.group_classic();
    stage.mark_mode_used = stage.mark_mode;
  } else {
    stage.agg_prepare_classic();  // no launch at all: the cursor still has to hold the earlier passes' count
  }
  stage.publish(out);
  return 0;
}

// multi-GPU, sparse marks: the routed marks of the local reads (global positions of their non-solid (k+1)-mer occurrences)
// -> local is_solid = "a (k+1)-mer starts here and it is not marked" (MHX_BUF_IS_SOLID_LOCAL, read by stage 2)
void s1_apply_marks(mhx_ctx *c, const unsigned long long *recv, uint64_t n) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint64_t need = div_ceil(s.n_bases, 64);
  uint8_t *bytes = c->ws("solid_bytes_local", (need + 1) * 64).as<uint8_t>();
  MHX_HIP(hipMemsetAsync(bytes, 0, (need + 1) * 64, st));
  unsigned long long *ctr = c->ws("s1_counters", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  uint32_t *bad = reinterpret_cast<uint32_t *>(ctr + 4);
  if (n)
    MHX_LAUNCH(c, "apply_marks", (double)n * 40,
               hipLaunchKernelGGL(k_apply_marks, dim3((unsigned)div_ceil(n, 256)), dim3(256), 0, st, recv, n, c->pos_base, s.n_bases, bytes, bad));
  DevBuf &b = c->result(MHX_BUF_IS_SOLID_LOCAL, (need + 1) * 8);
  b.used = need * 8;
  if (need && s.fixed_len >= c->s1_acc_k + 16 && c->opt("s1_pack_fixed", 1))
    MHX_LAUNCH(c, "pack_solid", (double)need * 72,
               hipLaunchKernelGGL(k_pack_solid_inv_fixed, dim3((unsigned)std::min<uint64_t>(div_ceil(need * 4, 256), 4096)), dim3(256), 0, st, bytes, s.n_bases, s.fixed_len,
                                  (int)c->s1_acc_k, b.as<unsigned long long>(), need, ctr));
  else if (need)
    MHX_LAUNCH(c, "pack_solid", (double)need * 72,
               hipLaunchKernelGGL(k_pack_solid_inv, dim3((unsigned)std::min<uint64_t>(div_ceil(need, 256), 4096)), dim3(256), 0, st, bytes, s.n_bases, s.start.as<uint64_t>(),
                                  s.n_seqs, s.fixed_len, (int)c->s1_acc_k, b.as<unsigned long long>(), need, ctr));
  unsigned long long h[5] = {0, 0, 0, 0, 0};
  MHX_HIP(hipMemcpyAsync(h, ctr, 40, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h[4] & 0xFFFFFFFFull) throw Error("dist_apply_routed: a mark outside this rank's reads (ranks disagree on the global layout)");
  c->dist_local_solid = h[0];
  c->global_marks_inverted = false;
}

// ---- `count` on the bucket streaming (k_s1_stream<COUNT>): fixed-length reads on one GPU, k <= 22, min count <= 2 ----
bool count_stream_applies(const mhx_ctx *c, uint32_t k, uint32_t m) {
  const SeqSet &s = c->seqs;
  if (!c->opt("count_stream", 1) || c->global_bases || c->filter_on || c->accumulate || c->n_parts > 1) return false;
  // (a caller that asks for a particular form of the tile path gets the tile path)
  if (!c->opt("count_seg", 1) || c->opt("count_seg_bits", 0) || !c->opt("count_extract_fixed", 1)) return false;
  if (!s.n_seqs || k < 9 || (int)k > kCountStreamMaxK || m < 1 || m > 2) return false;
  const bool var = s.fixed_len == 0;  // reads of several lengths: item slots padded to the longest read's (CountGenVarT)
  if (var) {
    if (!c->opt("s1_var_fast", 1) || s.max_len < k + 1 || s.max_len - k < 8 || s.n_bases <= s.n_seqs * (uint64_t)k) return false;
    if ((double)s.n_bases * 100.0 < (double)c->opt("s1_var_min_fill", 50) * (double)s.n_seqs * s.max_len) return false;
  } else if (s.fixed_len < k + 1 || s.fixed_len - k < 8) {
    return false;
  }
  if (!c->opt("s1_fused_first_pass", 1) || !c->opt("sort_unit_runs", 1) || !c->opt("s1_gen_any_order", 1)) return false;
  const uint64_t n_items = var ? s.n_bases - s.n_seqs * (uint64_t)k : s.n_seqs * (uint64_t)(s.fixed_len - k);  // (var: the estimate the plan is made for)
  const uint64_t n_bits = s.n_bases;
  if ((n_bits >> s1_pos_bits(c)) >= 256) return false;  // (positions beyond the tags)
  const S1Plan plan = s1_plan(c, k, n_items, true, 0);
  if (!plan.stream || plan.passes.empty() || (int)plan.passes.size() > kFastPasses) return false;
  for (const SortPass &ps : plan.passes)
    if (ps.bits2 || ps.shift < 32) return false;  // (digits: bit fields of the first key word)
  return sort_takes_generated_first_pass(c, n_items, 3, plan.passes);
}
// records made by the first sort pass, prefix passes, bucket streaming.  -> false: gave up (an output region too small): nothing
// published, the caller runs the extraction + tile path; true: the solid edges lie in the per-workgroup regions of *spare
bool count_stream_groups(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *first_0_out, uint32_t *last_0_in_p1, unsigned long long *hist,
                         CountStreamOut *o) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const bool var = s.fixed_len == 0;
  const uint32_t per = (var ? s.max_len : s.fixed_len) - k;  // item slots per read
  const uint64_t n_slots = s.n_seqs * (uint64_t)per;
  const uint64_t n_est = var ? s.n_bases - s.n_seqs * (uint64_t)k : n_slots;
  const S1Plan plan = s1_plan(c, k, n_est, true, 0);
  const int KWv = 2;
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
                                s.start.as<uint64_t>(), s.n_seqs))
#define MHX_CH2(NPV)             \
  do {                           \
    if (var) MHX_CH(NPV, true);  \
    else MHX_CH(NPV, false);     \
  } while (0)
    if (hd.n == 1) MHX_CH2(1);
    else if (hd.n == 2) MHX_CH2(2);
    else if (hd.n == 3) MHX_CH2(3);
    else MHX_CH2(4);
#undef MHX_CH2
#undef MHX_CH
  }
  uint64_t n_items = n_slots;  // the records
  if (var) {  // = the sum of any one digit histogram
    std::vector<unsigned long long> h0(256);
    MHX_HIP(hipMemcpyAsync(h0.data(), pre_hist, 256 * 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    n_items = 0;
    for (unsigned long long v : h0) n_items += v;
    if (n_items == 0) return false;  // (no read holds an edge: the general path knows what to publish)
  }
  uint32_t *buf_a = c->ws("items_a", n_items * 12 + 64).as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", n_items * 12 + 64).as<uint32_t>();
  c->pre_hist_sig = passes_signature(plan.passes);
  c->pre_hist_buf = buf_a;
  c->pre_hist_n = n_items;
  c->pre_hist_passes = hd.n;
  const uint32_t pos_bits = s1_pos_bits(c);
  const uint64_t pos_stride = (s.n_bases >> pos_bits) ? 1ull << pos_bits : 0ull;
  const CountGenT g{s.words.as<uint32_t>(), s.fixed_len, per, (int)k, c->pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per, (uint32_t)(kSortThreads * 8) % per};
  const CountGenVarT gv{s.words.as<uint32_t>(), s.start.as<uint64_t>(), s.n_seqs, per, (int)k, c->pos_base, pos_bits, (uint32_t)(kSortThreads * 8) / per,
                        (uint32_t)(kSortThreads * 8) % per};
  c->gen_first_pass = [g, gv, var](const OnesweepLaunch &l) {
    if (!(l.unit_runs && l.wi == 0)) throw Error("count: the generating pass needs the unit-wide pass on a first-word digit");
    if (var)
      hipLaunchKernelGGL((k_radix_onesweep_u<3, 8, 3, CountGenVarT, 1, 0>), dim3(l.grid), dim3(kSortThreads), 0, l.stream, gv, l.out, l.n, l.ds, l.nbits,
                         l.bin_start, l.status, l.ticket, l.err, l.tag, l.xcd_units);
    else
      hipLaunchKernelGGL((k_radix_onesweep_u<3, 8, 3, CountGenT, 1, 0>), dim3(l.grid), dim3(kSortThreads), 0, l.stream, g, l.out, l.n, l.ds, l.nbits, l.bin_start,
                         l.status, l.ticket, l.err, l.tag, l.xcd_units);
  };
  c->gen_buf = buf_a;
  c->gen_n = n_items;
  c->gen_slots = n_slots;
  uint32_t *sorted = radix_sort(c, buf_a, buf_b, n_items, 3, KWv, plan.passes);
  c->pre_hist_buf = nullptr;
  uint32_t *spare = sorted == buf_a ? buf_b : buf_a;
  // bucket streaming
  const uint64_t n_buckets = 1ull << plan.seg_bits;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  const unsigned grid = (unsigned)std::min<uint64_t>(n_buckets, cus);
  const uint32_t region = (uint32_t)std::min<uint64_t>(n_items * 12 / 8 / grid, 0xFFFFFFF0u);
  uint32_t *counts = c->ws("cs_edge_counts", (size_t)grid * 4).as<uint32_t>();
  unsigned long long *ctr = c->ws("s1_counters", 64).as<unsigned long long>();
  uint32_t *seg_err = c->ws("s1_seg_err", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  MHX_HIP(hipMemsetAsync(seg_err, 0, 4, st));
  uint64_t *bounds = c->ws("s1_bucket_bounds", (n_buckets + 1) * 8 + 64).as<uint64_t>();
  uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
  MHX_LAUNCH(c, "bucket_bounds", (double)n_buckets * 8 * 30,
             hipLaunchKernelGGL(k_bucket_bounds, dim3((unsigned)((n_buckets + 1 + 255) / 256)), dim3(256), 0, st, sorted, n_items, 3, bounds, plan.seg_bits));
  S1SegArgs a{};
  a.k = (int)k;
  a.m = m;
  a.mark_mode = 1;
  a.hist = hist;
  a.ctr = ctr;
  a.agg_raw = reinterpret_cast<uint2 *>(spare);
  a.agg_cap = region;
  a.agg_counts = counts;
  a.pos_stride = pos_stride;
  a.err = seg_err;
  a.la_chunks = (int)c->opt("s1_stream_probes", 1024);
  a.c_start = s.start.as<uint64_t>();
  a.c_n_seqs = s.n_seqs;
  a.c_fixed_len = s.fixed_len;
  a.first_0_out = first_0_out;
  a.last_0_in_p1 = last_0_in_p1;
  const S1StreamGeom geo{plan.seg_bits, plan.sub0, (uint32_t)n_buckets,
                         (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", 8192 * 7 / 8), 1), 8192)};
  const double bytes = (double)n_items * 12 * (double)(1u << plan.sub0);
  if (pos_stride)
    MHX_LAUNCH(c, "count_groups", bytes,
               hipLaunchKernelGGL((k_s1_stream<true, 4, kStreamThreads, 13, true, false, true>), dim3(grid), dim3(kStreamThreads), 0, st, sorted, bounds, a, geo, 1u,
                                  ticket, (const uint32_t *const *)nullptr, 1));
  else
    MHX_LAUNCH(c, "count_groups", bytes,
               hipLaunchKernelGGL((k_s1_stream<true, 4, kStreamThreads, 13, false, false, true>), dim3(grid), dim3(kStreamThreads), 0, st, sorted, bounds, a, geo, 1u,
                                  ticket, (const uint32_t *const *)nullptr, 1));
  uint32_t e = 0;
  unsigned long long h_ctr[8] = {0};
  MHX_HIP(hipMemcpyAsync(&e, seg_err, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_ctr, ctr, 64, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  o->grid = grid;
  o->cap = region;
  o->counts = counts;
  o->spare = spare;
  o->sorted = sorted;
  o->n_items = n_items;
  o->n_distinct = h_ctr[4];
  o->plan = s1_plan_text(c, k, n_items);
  return e == 0;
}

int run_s1(mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy, mhx_s1_result *out) {
  if (c->global_bases) throw Error("read2sdbg_s1: a global layout is set; use the mhx_dist_* entry points");
  c->s1_defer_items = !want_mercy;  // s1_process sorts "items_a" first thing: its first pass may make the records (and apply a bucket filter)
  c->gen_first_pass = nullptr;
  const StageItems it = extract_stage(c, want_mercy ? MHX_STAGE_S1_MERCY : MHX_STAGE_S1, k, m);
  c->s1_defer_items = false;
  uint32_t *buf_a = c->work["items_a"].as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
  return s1_process(c, k, m, want_mercy, buf_a, buf_b, it.n, out);
}

}  // namespace mhx

#ifdef MHX_TILE_TIMING
// debug build only: phase clocks of the stage-1 tile kernels (this translation unit's copy of g_tile_phase)
extern "C" int mhx_debug_tile_phases(unsigned long long *out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mhx::g_tile_phase), 16 * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mhx::g_tile_phase), z, 16 * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif
