// The stage-1 / count records made from the packed reads by window arithmetic: the sources of the generating first sort pass
// (sort_kernels.h k_radix_onesweep_u<..., Src, ...>) and the arithmetic its histogram pre-passes repeat (s1_front.hip).
#pragma once
#include "s1_shared.h"

namespace mhx {

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
// FILTER: the lv1-bucket filter of a memory-plan pass inside the generator, as in S1GenT (`keep`: a bitmap over the 65 536 lv1 buckets; a
// dropped item becomes a record that is_record() rejects: prev / next bits 63, which PackEdge's fields never hold)
template <bool FILTER>
struct CountGenT {
  const uint32_t *seq;
  uint32_t L, per;  // per = L - k items per read
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;
  const uint32_t *keep;
  static constexpr bool kMayDrop = FILTER;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return !FILTER || (r.w[1] & 63u) != 63u; }
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
          prun = 0;
        }
      }
    }
  }
};
// The same records for k = 23..27 (round 6): prev | edge | next no longer fit a window SHARED by a run of eight items (8 + k + 2 > 32
// bases), so every item takes its own 64-bit window — k + 3 <= 30 bases — out of the four words its thread's run of eight touches
// (S1GenBlockedT's scheme).  The (k+1)-mer, the strand bit and prev / next fill the two key words (2 (k + 1) + 7 <= 63 bits): no room
// for a position tag — read sets below 2^32 bases.  Fixed-length reads, >= 8 items per read.
constexpr int kCountStreamWideMaxK = 27;
template <bool FILTER>
struct CountGenWideT {
  const uint32_t *seq;
  uint32_t L, per;  // per = L - k items per read
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;
  const uint32_t *keep;
  static constexpr bool kMayDrop = FILTER;
  __device__ __forceinline__ bool is_record(const Rec<3> &r) const { return !FILTER || (r.w[1] & 63u) != 63u; }
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)((w * kWave + lane) * NI + j);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<3> (&rec)[UT][NI]) const {
    static_assert(NI <= 8, "a run of NI windows starts in at most two different words");
    constexpr uint32_t kTileItems = kSortThreads * NI;
    const uint64_t g00 = unit_base + (uint64_t)((w * kWave + lane) * NI);
    const uint64_t emask = ~0ull << (64 - 2 * (k + 1));
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
    uint64_t wcur[UT], wnext[UT];
    uint32_t c[UT][4], nx[UT][4];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const uint64_t a0 = bt[t] + jt[t];
      wcur[t] = (a0 >= 1 ? a0 - 1 : 0) >> 4;
      wnext[t] = (bt[t] + L - 1) >> 4;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        c[t][x] = seq[wcur[t] + x];
        nx[t][x] = seq[wnext[t] + x];  // (the store is padded: also behind the last read)
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
        const uint64_t a = base + j;
        const uint64_t b = a >= 1 ? a - 1 : 0;  // (the store's first base: the window at base 0 shifted down, count_window_addr)
        const unsigned down = a >= 1 ? 0u : 2u;
        const bool second = (b >> 4) != wc;
        const unsigned sh = (unsigned)(b & 15) * 2;
        const uint32_t x0 = second ? c1 : c0, x1 = second ? c2 : c1, x2 = second ? c3 : c2;
        const uint64_t win = (((uint64_t)funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh)) >> down;
        const uint64_t f = (win << 2) & emask;
        const uint64_t rc = rc64(f, k + 1);
        const unsigned prev_b = (unsigned)(win >> 62) & 3u, next_b = (unsigned)(win >> (58 - 2 * k)) & 3u;
        uint32_t out[3];
        count_item_from_parts(f, rc, prev_b, next_b, j, L, k, a, pos_base, pos_bits, out);
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
// the same records from a library whose reads are not of one length: `per` = max_len - k item slots per read, the slots a shorter read
// does not fill declined (S1GenVarT's scheme)
template <bool FILTER>
struct CountGenVarT {
  const uint32_t *seq;
  const uint64_t *start;  // [n_seqs + 1]
  uint64_t n_seqs;
  uint32_t per;
  int k;
  uint64_t pos_base;
  uint32_t pos_bits;
  uint32_t tile_q, tile_r;
  const uint32_t *keep;
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
        if constexpr (FILTER)
          if (!s1_bucket_kept(keep, out[0])) out[1] = kS1Dropped;
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


}  // namespace mhx
