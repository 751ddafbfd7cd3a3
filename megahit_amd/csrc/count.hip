// `count` engine: replaces KmerCounter (reference src/sorting/kmer_counter.cpp) on the GPU.
//
//   extract     one lv2 item per (k+1)-mer occurrence, straight from the packed reads
//               (Lv1FillOffsets :158-206 + Lv2ExtractSubString :208-252 fused; no lv1 offsets)
//   radix sort  by the 2(k+1) key bits (sort.hip)  -> bucket order falls out of the key order
//   runs        heads of equal-key runs (scan.hip)
//   reduce      per run: multiplicity, prev/next counts, first_0_out / last_0_in atomics,
//               multiplicity histogram  (Lv2Postprocess :254-381)
//   emit        packed solid edges + per-bucket counts (PackEdge :32-52, EdgeWriter::Write)
#include "dev_prims.h"
#include "mhx_internal.h"
#include "sort_digits.h"
#include "tile_groups.h"

namespace mhx {

// ---------------------------------------------------------------------------
__global__ void k_seq_item_counts(const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t sub, uint32_t min_len,
                                  uint32_t *__restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seqs) {
    uint64_t L = start[i + 1] - start[i];
    cnt[i] = L >= min_len ? (uint32_t)(L - sub) : 0u;
  }
}

template <int KW, int S>
__global__ __launch_bounds__(256) void k_count_extract(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start,
                                                       const uint64_t *__restrict__ item_start, uint64_t n_seqs, int k,
                                                       uint64_t pos_base, uint32_t *__restrict__ items) {
  const int lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const uint64_t n_waves = (uint64_t)gridDim.x * blockDim.x / kWave;
  for (uint64_t r = wave; r < n_seqs; r += n_waves) {
    const uint64_t st = start[r];
    const uint32_t L = (uint32_t)(start[r + 1] - st);
    if (L < (uint32_t)k + 1) continue;
    const uint64_t ibase = item_start[r];
    for (uint32_t p = lane; p + k < L; p += kWave) {
      uint32_t e[KW], rc[KW];
      load_chars<KW>(seq, st + p, k + 1, e);
      rc_chars<KW>(e, k + 1, rc);
      const int strand = cmp_words<KW>(rc, e) < 0;  // rev_edge.cmp(edge) < 0, kmer_counter.cpp:179
      unsigned prev = p > 0 ? base_at(seq, st + p - 1) : kSentinel;
      unsigned next = p + k + 1 < L ? base_at(seq, st + p + k + 1) : kSentinel;
      const uint64_t full = ((pos_base + st + p) << 1) | (uint64_t)strand;
      uint64_t info;
      uint32_t out[S];
      if (!strand) {
#pragma unroll
        for (int i = 0; i < KW; ++i) out[i] = e[i];
        info = (full << 6) | (prev << 3) | next;
      } else {
#pragma unroll
        for (int i = 0; i < KW; ++i) out[i] = rc[i];
        info = (full << 6) | (comp_or_sentinel(next) << 3) | comp_or_sentinel(prev);
      }
      out[KW] = (uint32_t)(info >> 32);
      out[KW + 1] = (uint32_t)info;
      if constexpr (S > KW + 2) out[KW + 2] = 0;
      uint32_t *dst = items + (ibase + p) * S;
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
}

// Reads of one length (the usual case): item g belongs to read g / per, position g % per, so every lane has work (a wave
// per read leaves its last round of 64 nearly empty: 129 items at 150 bp), and the digit histograms of the coming sort
// are taken while the record is in registers (no separate read of the 16-byte records).
template <int KW, int S>
__global__ __launch_bounds__(256) void k_count_extract_fixed(const uint32_t *__restrict__ seq, uint32_t L, uint32_t per, uint64_t n_items, int k,
                                                             uint64_t pos_base, uint32_t *__restrict__ items, DigitSpecs specs,
                                                             unsigned long long *__restrict__ ghist) {
  __shared__ uint32_t h[kMaxFusedPasses][256];
  for (int i = threadIdx.x; i < specs.n * 256; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint64_t n_blocks = (n_items + 255) / 256;
  for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const uint64_t g = blk * 256 + threadIdx.x;
    if (g >= n_items) continue;
    const uint64_t r = g / per;
    const uint32_t p = (uint32_t)(g - r * per);
    const uint64_t st = r * L;
    uint32_t e[KW], rc[KW];
    load_chars<KW>(seq, st + p, k + 1, e);
    rc_chars<KW>(e, k + 1, rc);
    const int strand = cmp_words<KW>(rc, e) < 0;  // rev_edge.cmp(edge) < 0, kmer_counter.cpp:179
    const unsigned prev = p > 0 ? base_at(seq, st + p - 1) : kSentinel;
    const unsigned next = p + k + 1 < L ? base_at(seq, st + p + k + 1) : kSentinel;
    const uint64_t full = ((pos_base + st + p) << 1) | (uint64_t)strand;
    uint64_t info;
    uint32_t out[S];
    if (!strand) {
#pragma unroll
      for (int i = 0; i < KW; ++i) out[i] = e[i];
      info = (full << 6) | (prev << 3) | next;
    } else {
#pragma unroll
      for (int i = 0; i < KW; ++i) out[i] = rc[i];
      info = (full << 6) | (comp_or_sentinel(next) << 3) | comp_or_sentinel(prev);
    }
    out[KW] = (uint32_t)(info >> 32);
    out[KW + 1] = (uint32_t)info;
    if constexpr (S > KW + 2) out[KW + 2] = 0;
    for (int q = 0; q < specs.n; ++q) atomicAdd(&h[q][words_digit2<S>(out, specs.d[q])], 1u);
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
  __syncthreads();
  for (int q = 0; q < specs.n; ++q) {
    const uint32_t v = h[q][threadIdx.x];
    if (v) atomicAdd(&ghist[q * 256 + threadIdx.x], (unsigned long long)v);
  }
}

// ---------------------------------------------------------------------------
constexpr int kLocalHist = 1024;

template <int S>
struct CountTile {
  static constexpr int kRaw = 16384 / (S * 4);
  static constexpr int kT = kRaw >= 1024 ? 1024 : (kRaw >= 256 ? (kRaw / 256) * 256 : 256);
  static constexpr int kRuns = kT + kMaxTailRuns;
};

__device__ __forceinline__ uint32_t *count_local_hist() {
  __shared__ uint32_t lh[kLocalHist];
  return lh;
}
template <int S>
__device__ __forceinline__ uint32_t *count_run_ctr() {  // [run][8]: prev A,C,G,T / next A,C,G,T
  __shared__ uint32_t rc[CountTile<S>::kRuns * 8];
  return rc;
}
template <int S>
__device__ __forceinline__ uint8_t *count_run_mark() {
  __shared__ uint8_t mk[CountTile<S>::kRuns];
  return mk;
}

// Lv2Postprocess of KmerCounter (kmer_counter.cpp:254-381) as a tile operator (tile_groups.h).
// A run is a whole group here (all records of a (k+1)-mer).
template <int S>
struct CountOp {
  static constexpr bool kItemPhase = true, kItemFinal = true, kRunPhase = false, kUnitIsRun = false, kAtomicBase = false;
  __device__ void run_phase(const TileCtx<S> &, uint32_t, uint32_t) const {}
  int kw, wpe;
  uint32_t m;
  int side_effects;  // first launch: histogram + first_0_out/last_0_in atomics; second launch: emit only
  const uint64_t *start;
  uint64_t n_seqs;
  uint32_t fixed_len;
  uint32_t *first_0_out, *last_0_in_p1;
  unsigned long long *hist, *bucket_count;
  uint32_t *edges;
  // multi-GPU: the reads of an item may live on another rank, so the first_0_out / last_0_in updates are recorded as
  // events ((global position << 1) | which; 0: last_0_in = max(offset), 1: first_0_out = min(offset + 1)) and
  // routed to the rank holding the read (mhx_dist_route_records / mhx_dist_apply_routed)
  unsigned long long *events, *n_events;

  __device__ bool same_run(const uint32_t *, const uint32_t *) const { return true; }
  __device__ bool item_phase_enabled() const { return side_effects != 0; }
  __device__ bool item_final_enabled() const { return side_effects != 0; }
  __device__ void begin_block() const {
    if (!side_effects) return;
    uint32_t *lh = count_local_hist();
    for (int i = threadIdx.x; i < kLocalHist; i += blockDim.x) lh[i] = 0;
    uint32_t *rc = count_run_ctr<S>();
    for (int i = threadIdx.x; i < CountTile<S>::kRuns * 8; i += blockDim.x) rc[i] = 0;
    __syncthreads();
  }
  __device__ void end_block() const {
    if (!side_effects) return;
    uint32_t *lh = count_local_hist();
    for (int i = threadIdx.x; i < kLocalHist; i += blockDim.x)
      if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
  }
  // per record: count_prev / count_next of its run (:283-292)
  __device__ void item_phase(const TileCtx<S> &c, uint32_t rel, uint32_t run) const {
    const unsigned pn = c.acc.word(rel, kw + 1) & 63u, pv = pn >> 3, nx = pn & 7;
    uint32_t *rc = count_run_ctr<S>() + run * 8;
    if (pv < 4) atomicAdd(&rc[pv], 1u);
    if (nx < 4) atomicAdd(&rc[4 + nx], 1u);
  }
  __device__ GroupCounts unit_count(const TileCtx<S> &c, uint32_t g) const {
    GroupCounts gc;
    const uint32_t r = c.gpos[g];
    const uint32_t count = c.run_len(r);
    const bool solid = count >= m;
    gc.c0 = solid ? 1u : 0u;
    gc.c1 = 1u;
    if (!side_effects) return gc;
    const uint32_t *rc = count_run_ctr<S>() + r * 8;
    bool has_in = false, has_out = false;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      has_in |= rc[x] >= m;
      has_out |= rc[4 + x] >= m;
    }
    count_run_mark<S>()[r] = (uint8_t)((solid && !has_in ? 1 : 0) | (solid && !has_out ? 2 : 0));
    const uint32_t hb = count > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : count;
    if (hb < kLocalHist) atomicAdd(&count_local_hist()[hb], 1u);
    else atomicAdd(&hist[hb], 1ull);
    return gc;
  }
  // per record of a solid run without in/out: first_0_out / last_0_in (:307-368)
  __device__ void item_final(const TileCtx<S> &c, uint32_t rel, uint32_t run) const {
    const unsigned f = count_run_mark<S>()[run];
    if (!f) return;
    const uint64_t info = (((uint64_t)c.acc.word(rel, kw) << 32) | c.acc.word(rel, kw + 1)) >> 6;
    const uint64_t abs = info >> 1;
    const unsigned strand = (unsigned)(info & 1);
    if (events) {
      if (f & 1u) events[atomicAdd(n_events, 1ull)] = (abs << 1) | (strand == 0 ? 0u : 1u);
      if (f & 2u) events[atomicAdd(n_events, 1ull)] = (abs << 1) | (strand == 0 ? 1u : 0u);
      return;
    }
    const uint64_t rid = seq_of_offset(start, n_seqs, fixed_len, abs);
    const uint32_t off = (uint32_t)(abs - start[rid]);
    if (f & 1u) {  // !has_in: strand 0 -> last_0_in = max(off), strand 1 -> first_0_out = min(off+1)
      if (strand == 0) atomicMax(&last_0_in_p1[rid], off + 1);
      else atomicMin(&first_0_out[rid], off + 1);
    }
    if (f & 2u) {  // !has_out: the roles swap
      if (strand == 0) atomicMin(&first_0_out[rid], off + 1);
      else atomicMax(&last_0_in_p1[rid], off + 1);
    }
  }
  // PackEdge (kmer_counter.cpp:32-52) + EdgeWriter::Write bucket accounting
  __device__ void unit_emit(const TileCtx<S> &c, uint32_t g, uint64_t o0, uint64_t, uint64_t) const {
    const uint32_t r = c.gpos[g];
    const uint32_t count = c.run_len(r), b = c.run_start(r);
    if (count < m) return;
    uint32_t *ed = edges + o0 * wpe;
    for (int x = 0; x < wpe; ++x) ed[x] = x < kw ? c.acc.word(b, x) : 0u;
    ed[wpe - 1] |= count > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : count;
    atomicAdd(&bucket_count[c.acc.word(b, 0) >> 16], 1ull);
  }
};

// routed events -> first_0_out / last_0_in of the local reads
__global__ void k_apply_count_events(const unsigned long long *__restrict__ ev, uint64_t n, uint64_t pos_base,
                                     const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t fixed_len,
                                     uint32_t *__restrict__ first_0_out, uint32_t *__restrict__ last_0_in_p1) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t abs = (ev[i] >> 1) - pos_base;
  const uint64_t rid = seq_of_offset(start, n_seqs, fixed_len, abs);
  const uint32_t off = (uint32_t)(abs - start[rid]);
  if (ev[i] & 1ull) atomicMin(&first_0_out[rid], off + 1);
  else atomicMax(&last_0_in_p1[rid], off + 1);
}

__global__ void k_fix_last(const uint32_t *__restrict__ last_p1, uint32_t *__restrict__ last_out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) last_out[i] = last_p1[i] - 1u;  // 0 (unset) -> 0xFFFFFFFF sentinel, v+1 -> v
}

// ---------------------------------------------------------------------------------------------------------------
// Segment group-by for count (k <= 23: 16-byte records, one-word-pair keys, 2-word edges) — the counterpart of k_s1_seg
// (s1.hip).  KmerCounter::Lv2Postprocess (kmer_counter.cpp:254-381) needs, per distinct (k+1)-mer, its count and the
// counts of the bases before / after its occurrences; none of it depends on the order of the records.  So the records
// are sorted on the top `prefix` bits of the key only (half the LSD passes at k=21), every key then lies inside one
// SEGMENT of equal prefix, and a workgroup counts the keys of its tile's segments in an LDS hash table:
//   A  insert   per record (match-any over hash bits per wavefront: the first lane of a group of equal keys inserts
//               for the group): slot.next5 += counts of the 5 possible next chars of the group (their sum is the
//               multiplicity), slot.prev5 likewise
//   B  per key  (dense over the slots this tile created) has_in / has_out, histogram, solid edge -> the workgroup's
//               output region (unordered; the ~2 % of the records that are solid edges are sorted afterwards); the
//               "no in-edge / no out-edge" flags go into the zero low bits of the key
//   C  per record of a flagged key: first_0_out / last_0_in atomics (or routed events)          (:307-368)
// Ownership of segments across tiles, look-ahead, give-up -> classic fallback: exactly as k_s1_seg.
// ---------------------------------------------------------------------------------------------------------------
struct CountSegArgs {
  uint32_t m;
  uint32_t pfx_mask;   // bits of key word 0 that form the segment prefix
  const uint64_t *start;
  uint64_t n_seqs;
  uint32_t fixed_len;
  uint32_t *first_0_out, *last_0_in_p1;
  unsigned long long *hist;
  unsigned long long *edges_raw;  // per-workgroup regions of edges_cap 8-byte edges (in the spare sort buffer)
  uint32_t edges_cap;
  uint32_t *edge_counts;          // per workgroup
  unsigned long long *n_distinct;
  unsigned long long *events, *n_events;
  uint32_t *err;
  int la_chunks;
};
constexpr int kCountSegMaxDistinct = 1024;  // distinct keys a tile may hold (NSLOT = 2 x this); more -> the tile gives up

static_assert(kCountSegMaxDistinct == 4 * 256, "dense pass: 4 keys per thread");
template <int PER>
__global__ __launch_bounds__(256) void k_count_seg(const uint32_t *__restrict__ items, uint64_t n, CountSegArgs a, uint64_t n_work) {
  constexpr int T = 256 * PER;
  constexpr int NSLOT = 2 * kCountSegMaxDistinct;
  constexpr int LOGS = 11;
  static_assert((1 << LOGS) == NSLOT, "table size");
  constexpr int NR = PER + 1;
  constexpr uint32_t kCreated = 0x80000000u;
  constexpr unsigned long long kEmpty = ~0ull;  // not a key: its low (zero-padding) bits are all ones
  __shared__ unsigned long long keys[NSLOT];
  __shared__ unsigned long long prev5[NSLOT], next5[NSLOT];  // 5 x 12-bit counters each ('$' = 4 included)
  __shared__ uint16_t created[kCountSegMaxDistinct];
  __shared__ uint32_t lhist[512];
  __shared__ uint32_t s_bad, s_ncreated, s_edge_cur;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  unsigned long long *const edges_out = a.edges_raw + (size_t)blockIdx.x * a.edges_cap;
  for (int i = tid; i < NSLOT; i += 256) {
    keys[i] = kEmpty;
    prev5[i] = 0;
    next5[i] = 0;
  }
  for (int i = tid; i < 512; i += 256) lhist[i] = 0;
  if (tid == 0) {
    s_bad = 0;
    s_ncreated = 0;
    s_edge_cur = 0;
  }
  __syncthreads();
  const uint32_t pfx = a.pfx_mask, m = a.m;
  unsigned long long my_distinct = 0;
  auto hash_of = [&](uint32_t w0, uint32_t w1) -> uint32_t { return w0 * 0x9E3779B1u + w1 * 0x85EBCA6Bu; };
  auto probe_insert = [&](unsigned long long key, uint32_t h) -> uint32_t {
    for (int probes = 0; probes < 256; ++probes) {
      const unsigned long long old = atomicCAS(&keys[h], kEmpty, key);
      if (old == kEmpty || old == key) return h | (old == kEmpty ? kCreated : 0u);
      h = (h + 1) & (NSLOT - 1);
    }
    s_bad = 1;
    return 0;
  };
  auto lookup = [&](unsigned long long key) -> uint32_t {
    uint32_t h = hash_of((uint32_t)(key >> 32), (uint32_t)key) >> (32 - LOGS);
    for (int probes = 0; probes < 256 && (keys[h] & ~7ull) != key; ++probes) h = (h + 1) & (NSLOT - 1);
    return h;
  };
  // per record of a key that lacks an in- or an out-edge (flags f): kmer_counter.cpp:307-368
  auto record_final = [&](unsigned f, uint32_t w2, uint32_t w3) {
    const uint64_t info = (((uint64_t)w2 << 32) | w3) >> 6;
    const uint64_t abs = info >> 1;
    const unsigned strand = (unsigned)(info & 1);
    if (a.events) {
      if (f & 1u) a.events[atomicAdd(a.n_events, 1ull)] = (abs << 1) | (strand == 0 ? 0u : 1u);
      if (f & 2u) a.events[atomicAdd(a.n_events, 1ull)] = (abs << 1) | (strand == 0 ? 1u : 0u);
      return;
    }
    const uint64_t rid = seq_of_offset(a.start, a.n_seqs, a.fixed_len, abs);
    const uint32_t off = (uint32_t)(abs - a.start[rid]);
    if (f & 1u) {
      if (strand == 0) atomicMax(&a.last_0_in_p1[rid], off + 1);
      else atomicMin(&a.first_0_out[rid], off + 1);
    }
    if (f & 2u) {
      if (strand == 0) atomicMin(&a.first_0_out[rid], off + 1);
      else atomicMax(&a.last_0_in_p1[rid], off + 1);
    }
  };

  for (uint64_t tile_idx = blockIdx.x; tile_idx < n_work; tile_idx += gridDim.x) {
    const uint64_t base = tile_idx * T;
    const uint64_t tile_end = n - base < (uint64_t)T ? n : base + T;
    uint32_t w0[NR], w1[NR], w2[NR], w3[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const uint64_t gi = base + (uint64_t)j * 256 + tid;
      if (gi < n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(items + gi * 4);
        w0[j] = v.x;
        w1[j] = v.y;
        w2[j] = v.z;
        w3[j] = v.w;
      }
    }
    const bool has_prev = base != 0;
    const uint32_t p_prev = has_prev ? items[(base - 1) * 4] & pfx : 0u, p_last = items[(tile_end - 1) * 4] & pfx;
    const bool la_own = tile_end < n && !(has_prev && p_last == p_prev);
    const bool more = la_own && tile_end + 256 < n && (items[(tile_end + 255) * 4] & pfx) == p_last;

    uint32_t slot[NR];
    bool own[NR];
    constexpr int NB = NR % 3 == 0 ? 3 : (NR % 5 == 0 ? 5 : 1);
    constexpr int MB = 7;
#pragma unroll
    for (int j0 = 0; j0 < NR; j0 += NB) {
      bool ins[NB], eq[NB], doer[NB];
      int leader[NB];
      uint32_t hs[NB];
      uint64_t peers[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int j = j0 + q;
        const uint64_t gi = base + (uint64_t)j * 256 + tid;
        if (j < PER) own[j] = gi < tile_end && !(has_prev && (w0[j] & pfx) == p_prev);
        else own[j] = la_own && gi < n && (w0[j] & pfx) == p_last;
        ins[q] = j < PER ? gi < tile_end : own[j];
        const uint32_t hf = hash_of(w0[j], w1[j]);
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
        l1[q] = __shfl(w1[j0 + q], leader[q], kWave);
      }
      unsigned long long addp[NB], addn[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int j = j0 + q;
        eq[q] = ins[q] && l0[q] == w0[j] && l1[q] == w1[j];
        const uint64_t grp = __ballot(eq[q]) & peers[q];
        doer[q] = ins[q] && (lane == leader[q] || !eq[q]);
        // what this lane's insert adds to the slot's five prev / next counters: the whole group's chars for a leader,
        // its own for a hash-equal lane with another key
        const unsigned pv = (w3[j] >> 3) & 7u, nx = w3[j] & 7u;
        const uint64_t mine = lane == leader[q] ? grp : (1ull << lane);
        unsigned long long ap = 0, an = 0;
#pragma unroll
        for (unsigned x = 0; x < 5; ++x) {
          ap |= (unsigned long long)__builtin_popcountll(__ballot(ins[q] && pv == x) & mine) << (12 * x);
          an |= (unsigned long long)__builtin_popcountll(__ballot(ins[q] && nx == x) & mine) << (12 * x);
        }
        addp[q] = ap;
        addn[q] = an;
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int j = j0 + q;
        slot[j] = 0;
        if (doer[q]) {
          const uint32_t sl = probe_insert(((unsigned long long)w0[j] << 32) | w1[j], hs[q]);
          atomicAdd(&prev5[sl & ~kCreated], addp[q]);
          atomicAdd(&next5[sl & ~kCreated], addn[q]);
          slot[j] = sl;
        }
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int j = j0 + q;
        const bool cr = (slot[j] & kCreated) != 0;
        const uint64_t crm = __ballot(cr);
        uint32_t cbase = 0;
        if (lane == 0 && crm) cbase = atomicAdd(&s_ncreated, (uint32_t)__builtin_popcountll(crm));
        cbase = __shfl(cbase, 0, kWave);
        slot[j] &= ~kCreated;
        if (cr) {
          const uint32_t at = cbase + (uint32_t)__builtin_popcountll(crm & lanemask_lt);
          if (at < (uint32_t)kCountSegMaxDistinct) created[at] = (uint16_t)slot[j];
          else s_bad = 1;
        }
        const uint32_t lslot = __shfl(slot[j], leader[q], kWave);
        if (eq[q] && lane != leader[q]) slot[j] = lslot;
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
          const uint4 v = *reinterpret_cast<const uint4 *>(items + gi * 4);
          if ((v.x & pfx) == p_last) {
            const uint32_t sl = probe_insert(((unsigned long long)v.x << 32) | v.y, hash_of(v.x, v.y) >> (32 - LOGS));
            atomicAdd(&prev5[sl & ~kCreated], 1ull << (12 * ((v.w >> 3) & 7u)));
            atomicAdd(&next5[sl & ~kCreated], 1ull << (12 * (v.w & 7u)));
            if (sl & kCreated) {
              const uint32_t at = atomicAdd(&s_ncreated, 1u);
              if (at < (uint32_t)kCountSegMaxDistinct) created[at] = (uint16_t)(sl & ~kCreated);
              else s_bad = 1;
            }
          }
        }
        if (!(cb + 256 < n && (items[(cb + 255) * 4] & pfx) == p_last)) break;
      }
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    const uint32_t n_created = s_ncreated < (uint32_t)kCountSegMaxDistinct ? s_ncreated : (uint32_t)kCountSegMaxDistinct;
    // B: per distinct key of ours
    uint32_t my_edges = 0;
    unsigned long long ebuf[4];  // (n_created <= 1024 = 4 per thread: the thread's edges stay in registers)
#pragma unroll
    for (int q = 0; q < 4; ++q) ebuf[q] = 0;
    if (!bad) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t i = tid + 256 * q;
        if (i >= n_created) continue;
        const uint32_t sl = created[i];
        const unsigned long long key = keys[sl];
        if (has_prev && ((uint32_t)(key >> 32) & pfx) == p_prev) continue;
        const unsigned long long pp = prev5[sl], nn = next5[sl];
        uint32_t count = 0;
        bool has_in = false, has_out = false;
#pragma unroll
        for (unsigned x = 0; x < 5; ++x) {
          const uint32_t cn = (uint32_t)(nn >> (12 * x)) & 0xFFFu, cp = (uint32_t)(pp >> (12 * x)) & 0xFFFu;
          count += cn;
          if (x < 4) {
            has_in |= cp >= m;
            has_out |= cn >= m;
          }
        }
        ++my_distinct;
        const bool solid = count >= m;
        const uint32_t hb = count > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : count;
        if (hb < 512) atomicAdd(&lhist[hb], 1u);
        else atomicAdd(&a.hist[hb], 1ull);
        if (solid) {
          keys[sl] = key | (has_in ? 0u : 1u) | (has_out ? 0u : 2u);  // flags ride in the key's zero padding
          ebuf[q] = key | hb;  // PackEdge (kmer_counter.cpp:32-52): multiplicity in the low 16 bits of the last word
          ++my_edges;
        }
      }
    }
    const uint32_t incl = wave_inclusive_sum(my_edges);
    const uint32_t tot = __shfl(incl, kWave - 1, kWave);
    uint32_t wbase = 0;
    if (lane == 0 && tot) wbase = atomicAdd(&s_edge_cur, tot);
    wbase = __shfl(wbase, 0, kWave);
    if (wbase + tot > a.edges_cap) {
      if (lane == 0) atomicOr(a.err, 1u);
    } else if (!bad) {
      uint32_t at = wbase + incl - my_edges;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ebuf[q]) edges_out[at++] = ebuf[q];
    }
    __syncthreads();
    // C: per record of a key without an in- or out-edge
    if (!bad) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if (!own[j]) continue;
        const unsigned f = (unsigned)keys[slot[j]] & 3u;
        if (f) record_final(f, w2[j], w3[j]);
      }
      if (more) {
        for (int c = 1; c <= a.la_chunks; ++c) {
          const uint64_t cb = tile_end + (uint64_t)c * 256;
          const uint64_t gi = cb + tid;
          if (gi < n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(items + gi * 4);
            if ((v.x & pfx) == p_last) {
              const unsigned f = (unsigned)keys[lookup(((unsigned long long)v.x << 32) | v.y)] & 3u;
              if (f) record_final(f, v.z, v.w);
            }
          }
          if (!(cb + 256 < n && (items[(cb + 255) * 4] & pfx) == p_last)) break;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      s_bad = 0;
      s_ncreated = 0;
    }
    if (!bad) {
      for (uint32_t i = tid; i < n_created; i += 256) {
        const uint32_t sl = created[i];
        keys[sl] = kEmpty;
        prev5[sl] = 0;
        next5[sl] = 0;
      }
    } else {
      for (int i = tid; i < NSLOT; i += 256) {
        keys[i] = kEmpty;
        prev5[i] = 0;
        next5[i] = 0;
      }
      if (tid == 0) atomicOr(a.err, 1u);
    }
    __syncthreads();
  }
  for (int i = tid; i < 512; i += 256)
    if (lhist[i]) atomicAdd(&a.hist[i], (unsigned long long)lhist[i]);
  my_distinct = wave_sum(my_distinct);
  if (lane == 0 && my_distinct) atomicAdd(a.n_distinct, my_distinct);
  if (tid == 0) a.edge_counts[blockIdx.x] = s_edge_cur < a.edges_cap ? s_edge_cur : a.edges_cap;
}
// regions -> dense (the same job as k_agg_compact in s1.hip, for 8-byte edges)
__global__ __launch_bounds__(256) void k_edges_compact(const unsigned long long *__restrict__ raw, uint32_t cap, const uint32_t *__restrict__ counts,
                                                      unsigned long long *__restrict__ dense) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint32_t r = blockIdx.x;
  uint64_t part = 0;
  for (uint32_t i = threadIdx.x; i < r; i += 256) part += counts[i];
  uint64_t off;
  block_exclusive_sum<uint64_t, 256>(part, sm, &off);
  const uint32_t nn = counts[r];
  const unsigned long long *src = raw + (size_t)r * cap;
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < nn; i += gridDim.y * 256) dense[off + i] = src[i];
}
// sorted 8-byte edges (hi word first in memory after the sort) -> bucket counts
// edges per lv1 bucket.  The edges are sorted, so the lanes of a wavefront hold a few runs of equal buckets: the first lane of a run
// adds the run's length (one atomic per lane on ~450 equal addresses in a row took 2.6 ms for 29.7 M edges: rocprofv3, round 5)
__global__ __launch_bounds__(256) void k_edge_buckets(const uint32_t *__restrict__ edges, uint64_t n, unsigned long long *__restrict__ bcount, int wpe = 2) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const bool valid = i < n;
  const uint32_t b = valid ? edges[(uint64_t)wpe * i] >> 16 : 0xFFFFFFFFu;
  const uint32_t pb = __shfl_up(b, 1, kWave);
  const bool head = valid && (lane == 0 || pb != b);
  const uint64_t heads = __ballot(head), valids = __ballot(valid);
  if (head) {
    const uint64_t later = lane == kWave - 1 ? 0ull : (heads >> (lane + 1)) << (lane + 1);  // run heads behind this lane
    const int end = later ? __builtin_ctzll(later) : (valids == ~0ull ? kWave : __builtin_ctzll(~valids));
    atomicAdd(&bcount[b], (unsigned long long)(end - lane));
  }
}
// the same for the 16-byte entries of k >= 24 ((k+1)-mer words, multiplicity, 0), and their three-word form (kmer_counter.cpp:32-52)
__global__ __launch_bounds__(256) void k_edges_compact16(const uint4 *__restrict__ raw, uint32_t cap, const uint32_t *__restrict__ counts, uint4 *__restrict__ dense) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint32_t r = blockIdx.x;
  uint64_t part = 0;
  for (uint32_t i = threadIdx.x; i < r; i += 256) part += counts[i];
  uint64_t off;
  block_exclusive_sum<uint64_t, 256>(part, sm, &off);
  const uint32_t nn = counts[r];
  const uint4 *src = raw + (size_t)r * cap;
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < nn; i += gridDim.y * 256) dense[off + i] = src[i];
}
__global__ void k_edges_pack3(const uint4 *__restrict__ sorted, uint64_t n, uint32_t *__restrict__ edges) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint4 e = sorted[i];
    edges[3 * i] = e.x;
    edges[3 * i + 1] = e.y;
    edges[3 * i + 2] = e.z;
  }
}
__global__ void k_swap_pairs(uint32_t *__restrict__ v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint2 x = reinterpret_cast<uint2 *>(v)[i];
    reinterpret_cast<uint2 *>(v)[i] = make_uint2(x.y, x.x);
  }
}

template <int S>
static void count_postprocess(mhx_ctx *c, const uint32_t *sorted, uint64_t n_items, int KWv, int key_bits, uint32_t m, int wpe,
                              uint32_t *first, uint32_t *last, unsigned long long *hist, unsigned long long *bcount, uint64_t *n_runs,
                              uint64_t *n_edges, unsigned long long *events, unsigned long long *n_events) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  *n_runs = *n_edges = 0;
  if (n_items == 0) {
    c->result(MHX_BUF_EDGES, 4);
    c->results[MHX_BUF_EDGES].used = 0;
    return;
  }
  constexpr int T = CountTile<S>::kT;
  const uint64_t n_tiles = div_ceil(n_items, T);
  uint64_t *tot = c->ws("tile_tot", (3 * n_tiles + 4) * 8).as<uint64_t>();
  uint64_t *tb = c->ws("tile_base", (3 * n_tiles + 4) * 8).as<uint64_t>();
  const int full_words = key_bits / 32, rem = key_bits % 32;
  const uint32_t last_mask = rem ? 0xFFFFFFFFu << (32 - rem) : 0;
  CountOp<S> op{KWv, wpe, m, 1, s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, first, last, hist, bcount, nullptr, events, n_events};
  const double bytes = (double)n_items * S * 4;
  MHX_LAUNCH(c, "count_runs", bytes,
             hipLaunchKernelGGL((k_tile_groups<S, T, CountOp<S>, false>), dim3(tile_grid(n_tiles)), dim3(kTileThreads), 0, st, sorted, n_items,
                                full_words, last_mask, op, tot, (const uint64_t *)nullptr, n_tiles, n_tiles));
  uint64_t *d_tot = c->ws("tile_totals", 64).as<uint64_t>();
  exclusive_scan_u64(c, tot, tb, n_tiles, d_tot);
  exclusive_scan_u64(c, tot + n_tiles, tb + n_tiles, n_tiles, d_tot + 1);
  MHX_HIP(hipMemsetAsync(tb + 2 * n_tiles, 0, n_tiles * 8, st));
  uint64_t h[2];
  MHX_HIP(hipMemcpyAsync(h, d_tot, 16, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  *n_edges = h[0];
  *n_runs = h[1];
  uint32_t *edges = c->result(MHX_BUF_EDGES, (h[0] ? h[0] : 1) * wpe * 4).as<uint32_t>();
  c->results[MHX_BUF_EDGES].used = h[0] * wpe * 4;
  op.side_effects = 0;
  op.edges = edges;
  MHX_LAUNCH(c, "count_emit", bytes + (double)h[0] * wpe * 4,
             hipLaunchKernelGGL((k_tile_groups<S, T, CountOp<S>, true>), dim3(tile_grid(n_tiles)), dim3(kTileThreads), 0, st, sorted, n_items,
                                full_words, last_mask, op, (uint64_t *)nullptr, (const uint64_t *)tb, n_tiles, n_tiles));
}

// ---------------------------------------------------------------------------
static int count_kw(uint32_t k) { return (int)div_ceil((k + 1) * 2, 32); }
int count_stride(uint32_t k) { return round_up2(count_kw(k) + 2); }

// the record sort of count: only the top seg_bits of the key when the segment group-by (k_count_seg) follows, else all of it
static std::vector<SortPass> count_sort_passes(mhx_ctx *c, uint32_t k, uint32_t m, uint64_t n_items, int *seg_bits_out) {
  const int KWv = count_kw(k), S = count_stride(k);
  const int wpe = (int)div_ceil((k + 1) * 2 + 16, 32);
  const int key_bits = (int)(k + 1) * 2;
  // Segment group-by (k_count_seg): sort only the top seg_bits of the key (a segment then holds ~100 records), count the
  // equal keys of every segment in LDS, sort the few solid edges afterwards.  16-byte records / 2-word edges (k <= 23).
  int seg_bits = 0;
  if (c->opt("count_seg", 1) && S == 4 && KWv == 2 && wpe == 2 && m < 4096 && n_items) {
    const double n_eff = (double)n_items * (double)(c->n_parts > 1 ? c->n_parts : 1);
    seg_bits = 8;
    while (seg_bits < 32 && n_eff / 96.0 > (double)(1ull << seg_bits)) seg_bits += 8;
    if (c->opt("count_seg_bits", 0)) seg_bits = (int)c->opt("count_seg_bits", 0);
    seg_bits = std::max(1, std::min(seg_bits, 32));
  }
  if (seg_bits_out) *seg_bits_out = seg_bits;
  if (!seg_bits) return make_passes(KWv, KWv * 32 - key_bits, KWv * 32);
  // (the segment group-by counts the equal keys of a segment whatever their order: every pass after the first may say which
  //  bits were sorted before it, and ranks with LDS atomics where a wavefront's records agree on those — SortPass::prev_lo)
  std::vector<SortPass> passes = make_passes(KWv, 64 - seg_bits, 64);
  for (size_t i = 1; i < passes.size(); ++i) passes[i].prev_lo = 64 - seg_bits;
  return passes;
}

// items of the local reads -> c->ws("items_a"); returns their number.  m > 0: the caller will sort them for mhx_count with
// this minimum count, so the digit histograms of that sort can be taken on the way (fixed-length reads)
uint64_t count_extract(mhx_ctx *c, uint32_t k, uint32_t m) {
  SeqSet &s = c->seqs;
  if (k < 9 || k > MHX_MAX_K) throw Error("count: k out of range [9,255]");
  const int KWv = count_kw(k), S = count_stride(k);
  const uint64_t ns = s.n_seqs;
  hipStream_t st = c->stream;
  // per-read item counts -> item_start
  uint32_t *cnt = c->ws("seq_item_cnt", (ns + 1) * 4).as<uint32_t>();
  uint64_t *item_start = c->ws("seq_item_start", (ns + 2) * 8).as<uint64_t>();
  uint64_t *d_total = item_start + ns + 1;
  uint64_t n_items = 0;
  if (ns) {
    MHX_LAUNCH(c, "item_counts", (double)ns * 12,
               hipLaunchKernelGGL(k_seq_item_counts, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, s.start.as<uint64_t>(), ns, k,
                                  k + 1, cnt));
    exclusive_scan_u32_u64(c, cnt, item_start, ns, d_total);
    MHX_HIP(hipMemcpyAsync(&n_items, d_total, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  const size_t item_bytes = (size_t)S * 4;
  uint32_t *buf_a = c->ws("items_a", n_items * item_bytes + 64).as<uint32_t>();
  c->pre_hist_buf = nullptr;
  if (n_items) {
    const unsigned grid = 256 * 8;
    const bool fixed = s.fixed_len >= k + 1 && n_items == (uint64_t)ns * (s.fixed_len - k) && c->opt("count_extract_fixed", 1) != 0;
    DigitSpecs specs;
    specs.n = 0;
    unsigned long long *pre_hist = nullptr;
    if (fixed && m > 0) {
      const std::vector<SortPass> passes = count_sort_passes(c, k, m, n_items, nullptr);
      if ((int)passes.size() <= kMaxFusedPasses) {
        c->pre_hist_sig = passes_signature(passes);
        specs.n = (int)passes.size();
        for (int p = 0; p < specs.n; ++p) specs.d[p] = spec_of_pass(passes[p], KWv);
        pre_hist = c->ws("sort_pre_hist", (size_t)kMaxFusedPasses * 256 * 8).as<unsigned long long>();
        MHX_HIP(hipMemsetAsync(pre_hist, 0, (size_t)specs.n * 256 * 8, st));
        c->pre_hist_buf = buf_a;
        c->pre_hist_n = n_items;
        c->pre_hist_passes = specs.n;
      }
    }
#define MHX_CX(SV)                                                                                                               \
  do {                                                                                                                           \
    if (fixed)                                                                                                                   \
      MHX_LAUNCH(c, "count_extract", (double)n_items * item_bytes + (double)s.n_bases / 4,                                       \
                 hipLaunchKernelGGL((k_count_extract_fixed<KW, SV>), dim3((unsigned)std::min<uint64_t>(div_ceil(n_items, 256), 256 * 16)), \
                                    dim3(256), 0, st, s.words.as<uint32_t>(), s.fixed_len, s.fixed_len - k, n_items, (int)k, c->pos_base, \
                                    buf_a, specs, pre_hist));                                                                     \
    else                                                                                                                         \
      MHX_LAUNCH(c, "count_extract", (double)n_items * item_bytes + (double)s.n_bases / 4,                                       \
                 hipLaunchKernelGGL((k_count_extract<KW, SV>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(),              \
                                    s.start.as<uint64_t>(), item_start, ns, (int)k, c->pos_base, buf_a));                         \
  } while (0)
    MHX_DISPATCH_KW(KWv, {
      if (S == KW + 2) MHX_CX(KW + 2);
      else MHX_CX(KW + 3);
    });
#undef MHX_CX
  }
  return n_items;
}

// sort + run reduction of n_items items held in buf_a (buf_b = ping-pong space of the same size)
int count_process(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_count_result *out) {
  SeqSet &s = c->seqs;
  const int KWv = count_kw(k), S = count_stride(k);
  const int wpe = (int)div_ceil((k + 1) * 2 + 16, 32);
  const uint64_t ns = s.n_seqs;
  const size_t item_bytes = (size_t)S * 4;
  hipStream_t st = c->stream;
  const bool global = c->global_bases != 0;
  const int key_bits = (int)(k + 1) * 2;
  int seg_bits = 0;
  const std::vector<SortPass> sort_passes = count_sort_passes(c, k, m, n_items, &seg_bits);
  // (no segment group-by: the whole zero-padded (k+1)-mer is the key; equal keys keep their input order)
  uint32_t *sorted = seg_bits ? radix_sort(c, buf_a, buf_b, n_items, S, KWv, sort_passes) : sort_whole_key(c, buf_a, buf_b, n_items, S, KWv, sort_passes);
  uint32_t *spare = sorted == buf_a ? buf_b : buf_a;

  c->last_s1_plan = seg_bits ? "count: tile path, segment group-by on " + std::to_string(seg_bits) + " prefix bits" : "count: tile path, full sort";
  // results
  // accumulate (bucket-range passes after the first): first_0_out, the raw last_0_in (+1) values and the histogram of
  // the earlier passes are kept; the published last_0_in is re-derived from the raw values after every pass
  const bool acc = c->accumulate && c->results.count(MHX_BUF_FIRST_0_OUT) && c->results[MHX_BUF_FIRST_0_OUT].used == ns * 4 &&
                   c->work.count("last_p1") && c->results.count(MHX_BUF_MUL_HIST) && c->count_acc_k == k && c->count_acc_m == m;
  c->count_acc_k = k;  // a pass with another (k, m) starts from scratch instead of merging into stale state
  c->count_acc_m = m;
  uint32_t *first = c->result(MHX_BUF_FIRST_0_OUT, (ns ? ns : 1) * 4).as<uint32_t>();
  uint32_t *last_out = c->result(MHX_BUF_LAST_0_IN, (ns ? ns : 1) * 4).as<uint32_t>();
  uint32_t *last = c->ws("last_p1", (ns ? ns : 1) * 4).as<uint32_t>();
  c->results[MHX_BUF_FIRST_0_OUT].used = ns * 4;
  c->results[MHX_BUF_LAST_0_IN].used = ns * 4;
  unsigned long long *hist = c->result(MHX_BUF_MUL_HIST, (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
  unsigned long long *bcount = c->result(MHX_BUF_BUCKET_COUNT, MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  if (!acc) {
    MHX_HIP(hipMemsetAsync(first, 0xFF, (ns ? ns : 1) * 4, st));
    MHX_HIP(hipMemsetAsync(last, 0x00, (ns ? ns : 1) * 4, st));
    MHX_HIP(hipMemsetAsync(hist, 0, (MHX_MAX_MUL + 1) * 8, st));
  }
  MHX_HIP(hipMemsetAsync(bcount, 0, MHX_NUM_BUCKETS * 8, st));
  // multi-GPU: at most 2 events of 8 bytes per item fit the spare sort buffer (records are >= 16 bytes)
  unsigned long long *events = global ? reinterpret_cast<unsigned long long *>(spare) : nullptr;
  unsigned long long *ev_n = c->ws("count_ev_n", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(ev_n, 0, 8, st));

  uint64_t n_runs = 0, n_edges = 0;
  bool seg_done = false;
  if (seg_bits) {
    // state as it is now, in case a tile gives up and the classic path has to redo the job (the atomics of the tiles
    // that did finish cannot be taken back otherwise)
    uint32_t *sv_first = c->ws("cs_save_first", (ns ? ns : 1) * 4).as<uint32_t>(), *sv_last = c->ws("cs_save_last", (ns ? ns : 1) * 4).as<uint32_t>();
    unsigned long long *sv_hist = c->ws("cs_save_hist", (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
    MHX_HIP(hipMemcpyAsync(sv_first, first, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
    MHX_HIP(hipMemcpyAsync(sv_last, last, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
    MHX_HIP(hipMemcpyAsync(sv_hist, hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
    constexpr int PER = 8, T = 256 * PER;
    const uint64_t n_work = div_ceil(n_items, (uint64_t)T);
    const unsigned grid = (unsigned)std::min<uint64_t>(n_work, 256 * 3);
    const uint32_t cap = (uint32_t)std::min<uint64_t>(n_items * (uint64_t)S * 4 / 8 / grid, 0xFFFFFFF0u);
    uint32_t *counts = c->ws("cs_edge_counts", (size_t)grid * 4).as<uint32_t>();
    unsigned long long *ctrs = c->ws("cs_counters", 64).as<unsigned long long>();  // [0] distinct, [1] err
    MHX_HIP(hipMemsetAsync(ctrs, 0, 64, st));
    const uint32_t pfx_mask = seg_bits >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> seg_bits);
    const int la = (int)std::min<long long>(std::max<long long>(c->opt("count_seg_la", 3), 0), 6);  // 12-bit tile counters
    CountSegArgs a{m, pfx_mask, s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, first, last, hist, reinterpret_cast<unsigned long long *>(spare),
                   cap, counts, ctrs, events, ev_n, reinterpret_cast<uint32_t *>(ctrs + 1), la};
    // (multi-GPU: the events go to a buffer of their own — the spare sort buffer holds the edge regions here)
    if (global) a.events = c->ws("cs_events", n_items * 2 * 8 + 64).as<unsigned long long>();
    MHX_LAUNCH(c, "count_groups", (double)n_items * S * 4,
               hipLaunchKernelGGL((k_count_seg<PER>), dim3(grid), dim3(256), 0, st, sorted, n_items, a, n_work));
    std::vector<uint32_t> h_counts(grid);
    unsigned long long h_ctr[2] = {0, 0};
    MHX_HIP(hipMemcpyAsync(h_counts.data(), counts, (size_t)grid * 4, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipMemcpyAsync(h_ctr, ctrs, 16, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    if (!(h_ctr[1] & 0xFFFFFFFFull)) {
      for (uint32_t v : h_counts) n_edges += v;
      n_runs = h_ctr[0];
      uint32_t *ea = c->ws("cs_edges_a", (n_edges + 1) * 8).as<uint32_t>(), *eb = c->ws("cs_edges_b", (n_edges + 1) * 8).as<uint32_t>();
      uint32_t *edges = c->result(MHX_BUF_EDGES, (n_edges ? n_edges : 1) * wpe * 4).as<uint32_t>();
      c->results[MHX_BUF_EDGES].used = n_edges * wpe * 4;
      if (n_edges) {
        MHX_LAUNCH(c, "edges_compact", (double)n_edges * 16,
                   hipLaunchKernelGGL(k_edges_compact, dim3(grid, 4), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(spare), cap, counts,
                                      reinterpret_cast<unsigned long long *>(ea)));
        // uint64 (lo word first in memory) -> (hi, lo) word pairs = the edge's word order; sort by the (k+1)-mer bits
        hipLaunchKernelGGL(k_swap_pairs, dim3((unsigned)div_ceil(n_edges, 256)), dim3(256), 0, st, ea, n_edges);
        uint32_t *es = sort_whole_key(c, ea, eb, n_edges, 2, 2, make_passes(2, 64 - key_bits, 64));  // distinct keys: the count bits never decide
        MHX_HIP(hipMemcpyAsync(edges, es, n_edges * 8, hipMemcpyDeviceToDevice, st));
        MHX_LAUNCH(c, "edge_buckets", (double)n_edges * 8,
                   hipLaunchKernelGGL(k_edge_buckets, dim3((unsigned)div_ceil(n_edges, 256)), dim3(256), 0, st, edges, n_edges, bcount));
        MHX_HIP(hipGetLastError());
      }
      if (global) events = a.events;
      seg_done = true;
    } else {  // a tile gave up: restore, sort fully, run the classic tile kernel
      MHX_HIP(hipMemcpyAsync(first, sv_first, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemcpyAsync(last, sv_last, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemcpyAsync(hist, sv_hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemsetAsync(ev_n, 0, 8, st));
      uint32_t *other = sorted == buf_a ? buf_b : buf_a;
      sorted = sort_whole_key(c, sorted, other, n_items, S, KWv, make_passes(KWv, KWv * 32 - key_bits, KWv * 32));
      spare = sorted == buf_a ? buf_b : buf_a;
      events = global ? reinterpret_cast<unsigned long long *>(spare) : nullptr;
    }
  }
  if (!seg_done) switch (S) {
#define MHX_CASE(SV) \
  case SV: count_postprocess<SV>(c, sorted, n_items, KWv, key_bits, m, wpe, first, last, hist, bcount, &n_runs, &n_edges, events, ev_n); break;
    MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
#undef MHX_CASE
    default: throw Error("count: unsupported record stride");
  }
  if (global) {  // events -> sorted by position in ws("route_records"); first/last are finished by mhx_dist_apply_routed
    unsigned long long h = 0;
    MHX_HIP(hipMemcpyAsync(&h, ev_n, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    int hi_bit = 2;
    while (hi_bit < 64 && ((c->global_bases << 1) >> hi_bit)) ++hi_bit;
    stash_route_records(c, events, h, hi_bit);
  } else if (ns) {
    MHX_LAUNCH(c, "fix_last", (double)ns * 8,
               hipLaunchKernelGGL(k_fix_last, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, last, last_out, ns));
  }

  // expose the sorted items for tests (no copy: alias the workspace)
  mhx::DevBuf &si = c->results[MHX_BUF_SORTED_ITEMS];
  si.release();
  c->sorted_item_words = S;
  c->results[MHX_BUF_SORTED_ITEMS].p = sorted;
  c->results[MHX_BUF_SORTED_ITEMS].cap = 0;  // cap 0 = not owned
  c->results[MHX_BUF_SORTED_ITEMS].used = n_items * item_bytes;

  MHX_HIP(hipStreamSynchronize(st));
  if (out) {
    out->n_items = n_items;
    out->n_distinct = n_runs;
    out->n_edges = n_edges;
    out->words_per_edge = wpe;
    out->item_words = S;
  }
  return 0;
}

// received events (device, n of them) -> first_0_out / last_0_in of the local reads
void count_apply_events(mhx_ctx *c, const unsigned long long *ev, uint64_t n) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  auto itf = c->results.find(MHX_BUF_FIRST_0_OUT), itl = c->results.find(MHX_BUF_LAST_0_IN);
  if (itf == c->results.end() || itl == c->results.end() || itf->second.used != s.n_seqs * 4 || !c->work.count("last_p1"))
    throw Error("dist_apply_routed: run mhx_dist_process_count first");
  uint32_t *last_p1 = c->work["last_p1"].as<uint32_t>();
  if (n)
    MHX_LAUNCH(c, "count_apply_events", (double)n * 24,
               hipLaunchKernelGGL(k_apply_count_events, dim3((unsigned)div_ceil(n, 256)), dim3(256), 0, st, ev, n, c->pos_base,
                                  s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, itf->second.as<uint32_t>(), last_p1));
  if (s.n_seqs)
    MHX_LAUNCH(c, "fix_last", (double)s.n_seqs * 8,
               hipLaunchKernelGGL(k_fix_last, dim3((unsigned)div_ceil(s.n_seqs, 256)), dim3(256), 0, st, last_p1, itl->second.as<uint32_t>(), s.n_seqs));
  MHX_HIP(hipStreamSynchronize(st));
}

// count on the design of stage 1 (round 5; s1.hip: the first sort pass makes 12-byte records, two prefix passes, LDS group-by per
// bucket with the has_in / has_out evidence in the table, first_0_out / last_0_in from a second look at the few buckets that
// hold a solid key without an in- or out-edge).  -> false: that form gave up; nothing is published, the caller runs the
// extraction + tile path.  Reference: KmerCounter::Lv2ExtractSubString + Lv2Postprocess (kmer_counter.cpp:208-381).
static bool count_run_stream(mhx_ctx *c, uint32_t k, uint32_t m, mhx_count_result *out, const S1Sources *pre = nullptr) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint64_t ns = s.n_seqs;
  const int wpe = (int)div_ceil((k + 1) * 2 + 16, 32);
  const int key_bits = (int)(k + 1) * 2;
  // accumulate (bucket-range passes after the first): first_0_out, the raw last_0_in (+1) values and the histogram of the earlier
  // passes are kept (count_process does the same on the tile path: the two may take turns pass by pass)
  const bool acc = c->accumulate && c->results.count(MHX_BUF_FIRST_0_OUT) && c->results[MHX_BUF_FIRST_0_OUT].used == ns * 4 &&
                   c->work.count("last_p1") && c->results.count(MHX_BUF_MUL_HIST) && c->count_acc_k == k && c->count_acc_m == m;
  uint32_t *first = c->result(MHX_BUF_FIRST_0_OUT, (ns ? ns : 1) * 4).as<uint32_t>();
  uint32_t *last_out = c->result(MHX_BUF_LAST_0_IN, (ns ? ns : 1) * 4).as<uint32_t>();
  uint32_t *last = c->ws("last_p1", (ns ? ns : 1) * 4).as<uint32_t>();
  c->results[MHX_BUF_FIRST_0_OUT].used = ns * 4;
  c->results[MHX_BUF_LAST_0_IN].used = ns * 4;
  unsigned long long *hist = c->result(MHX_BUF_MUL_HIST, (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
  unsigned long long *bcount = c->result(MHX_BUF_BUCKET_COUNT, MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  if (!acc) {
    MHX_HIP(hipMemsetAsync(first, 0xFF, (ns ? ns : 1) * 4, st));
    MHX_HIP(hipMemsetAsync(last, 0x00, (ns ? ns : 1) * 4, st));
    MHX_HIP(hipMemsetAsync(hist, 0, (MHX_MAX_MUL + 1) * 8, st));
  }
  MHX_HIP(hipMemsetAsync(bcount, 0, MHX_NUM_BUCKETS * 8, st));
  // state as it is now: should the streaming form give up half-way (an edge region too small), the tile path redoes this pass on top
  // of the earlier passes' results, not on top of what the finished buckets of this attempt left (atomics cannot be taken back)
  uint32_t *sv_first = c->ws("cs_save_first", (ns ? ns : 1) * 4).as<uint32_t>(), *sv_last = c->ws("cs_save_last", (ns ? ns : 1) * 4).as<uint32_t>();
  unsigned long long *sv_hist = c->ws("cs_save_hist", (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
  if (acc) {
    MHX_HIP(hipMemcpyAsync(sv_first, first, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
    MHX_HIP(hipMemcpyAsync(sv_last, last, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
    MHX_HIP(hipMemcpyAsync(sv_hist, hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
  }
  CountStreamOut o;
  // super-k-mer records first where they serve (one GPU, no memory plan, k <= 21: s1_skm.hip); a job they give up on — low-complexity reads —
  // goes on below from clean arrays
  bool skm_done = false;
  uint64_t skm_edges = 0;  // the passes' edges, packed behind each other in ws "cs_edges_a"
  if (!pre && !acc && count_skm_applies(c, k, m)) {
    bool touched = false;
    const int n_passes = s1_skm_passes(c, k);
    uint64_t n_dist = 0, skm_records = 0;
    uint32_t skm_max_bin = 0;
    skm_done = true;
    for (int p = 0; p < n_passes && skm_done; ++p) {
      skm_done = count_skm_groups(c, k, m, first, last, hist, &o, &touched, p, n_passes);
      if (!skm_done) break;
      std::vector<uint32_t> hc(o.grid);
      MHX_HIP(hipMemcpyAsync(hc.data(), o.counts, (size_t)o.grid * 4, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      uint64_t tot = 0;
      for (uint32_t v : hc) tot += v;
      // (sized for all passes after the first: the bins fill evenly)
      const uint64_t expect = (uint64_t)((double)(skm_edges + tot) * (double)n_passes / (double)(p + 1) * 1.05);
      unsigned long long *dense = grow_preserving(c, c->work["cs_edges_a"], (std::max(skm_edges + tot, expect) + 1) * 8, skm_edges * 8).as<unsigned long long>();
      if (tot)
        MHX_LAUNCH(c, "edges_compact", (double)tot * 16,
                   hipLaunchKernelGGL(k_edges_compact, dim3(o.grid, 4), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(o.spare), o.cap, o.counts,
                                      dense + skm_edges));
      skm_edges += tot;
      n_dist += o.n_distinct;
      skm_records += o.skm_records;
      skm_max_bin = std::max(skm_max_bin, o.skm_max_bin);
    }
    if (skm_done && o.skm_hp) {  // the one or two keys of the homopolymer windows, behind the passes' edges
      unsigned long long *dense = grow_preserving(c, c->work["cs_edges_a"], (skm_edges + 3) * 8, skm_edges * 8).as<unsigned long long>();
      uint64_t ne = 0, nk = 0;
      bool flagged = false;
      count_skm_hp_publish(c, o.skm_hp, k, m, hist, dense + skm_edges, &ne, &nk, &flagged);
      skm_edges += ne;
      n_dist += nk;
      if (flagged) skm_done = false;  // (a solid homopolymer key without an in- or out-edge: its windows would move first_0_out / last_0_in)
    }
    o.n_distinct = n_dist;
    if (skm_done) {
      char txt[320];
      snprintf(txt, sizeof txt, "super-k-mers m%u, 2^%d bins (%llu records for %llu windows: %.2f per record; largest bin %u)%s", k + 1 - 9, o.skm_bin_bits,
               (unsigned long long)skm_records, (unsigned long long)o.skm_windows, skm_records ? (double)o.skm_windows / (double)skm_records : 0.0, skm_max_bin,
               s.fixed_len ? "" : " [reads of several lengths]");
      o.plan = txt;
      if (n_passes > 1) o.plan += " [" + std::to_string(n_passes) + " passes over ranges of bins]";
    }
    if (!skm_done) {
      // count_skm = 3: a caller that left the memory plan to this path (mhx_count_self_planned) hears that it did not serve
      if (c->opt("count_skm", 1) == 3) throw Error("count: super-k-mer records given up (low-complexity reads, or more records than the arrays hold)");
      skm_edges = 0;
      if (touched) {
        MHX_HIP(hipMemsetAsync(first, 0xFF, (ns ? ns : 1) * 4, st));
        MHX_HIP(hipMemsetAsync(last, 0x00, (ns ? ns : 1) * 4, st));
        MHX_HIP(hipMemsetAsync(hist, 0, (MHX_MAX_MUL + 1) * 8, st));
      }
    }
  }
  if (!skm_done && !count_stream_groups(c, k, m, first, last, hist, &o, pre)) {
    if (acc) {
      MHX_HIP(hipMemcpyAsync(first, sv_first, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemcpyAsync(last, sv_last, (ns ? ns : 1) * 4, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemcpyAsync(hist, sv_hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
    }
    return false;
  }
  uint64_t n_edges = skm_edges;
  if (!skm_done) {
    std::vector<uint32_t> h_counts(o.grid);
    MHX_HIP(hipMemcpyAsync(h_counts.data(), o.counts, (size_t)o.grid * 4, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    for (uint32_t v : h_counts) n_edges += v;
  }
  const size_t eb_bytes = wpe == 3 ? 16 : 8;  // a region entry
  uint32_t *ea = c->ws("cs_edges_a", (n_edges + 1) * eb_bytes).as<uint32_t>(), *eb = c->ws("cs_edges_b", (n_edges + 1) * eb_bytes).as<uint32_t>();
  uint32_t *edges = c->result(MHX_BUF_EDGES, (n_edges ? n_edges : 1) * wpe * 4).as<uint32_t>();
  c->results[MHX_BUF_EDGES].used = n_edges * wpe * 4;
  if (n_edges && wpe == 3) {  // k >= 24: 16-byte entries with the words in edge order -> sorted by the (k+1)-mer -> three-word edges
    MHX_LAUNCH(c, "edges_compact", (double)n_edges * 32,
               hipLaunchKernelGGL(k_edges_compact16, dim3(o.grid, 4), dim3(256), 0, st, reinterpret_cast<const uint4 *>(o.spare), o.cap / 2, o.counts,
                                  reinterpret_cast<uint4 *>(ea)));
    uint32_t *es = sort_whole_key(c, ea, eb, n_edges, 4, 2, make_passes(2, 64 - key_bits, 64));  // distinct keys
    hipLaunchKernelGGL(k_edges_pack3, dim3((unsigned)div_ceil(n_edges, 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(es), n_edges, edges);
    MHX_LAUNCH(c, "edge_buckets", (double)n_edges * 12,
               hipLaunchKernelGGL(k_edge_buckets, dim3((unsigned)div_ceil(n_edges, 256)), dim3(256), 0, st, edges, n_edges, bcount, 3));
    MHX_HIP(hipGetLastError());
  } else if (n_edges) {
    if (!skm_done)  // (the passes over super-k-mer records packed theirs already)
      MHX_LAUNCH(c, "edges_compact", (double)n_edges * 16,
                 hipLaunchKernelGGL(k_edges_compact, dim3(o.grid, 4), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(o.spare), o.cap, o.counts,
                                    reinterpret_cast<unsigned long long *>(ea)));
    // uint64 (lo word first in memory) -> (hi, lo) word pairs = the edge's word order; sort by the (k+1)-mer bits
    hipLaunchKernelGGL(k_swap_pairs, dim3((unsigned)div_ceil(n_edges, 256)), dim3(256), 0, st, ea, n_edges);
    uint32_t *es = sort_whole_key(c, ea, eb, n_edges, 2, 2, make_passes(2, 64 - key_bits, 64));  // distinct keys: the count bits never decide
    MHX_HIP(hipMemcpyAsync(edges, es, n_edges * 8, hipMemcpyDeviceToDevice, st));
    MHX_LAUNCH(c, "edge_buckets", (double)n_edges * 8,
               hipLaunchKernelGGL(k_edge_buckets, dim3((unsigned)div_ceil(n_edges, 256)), dim3(256), 0, st, edges, n_edges, bcount, 2));
    MHX_HIP(hipGetLastError());
  }
  if (pre) {  // several GPUs: the events -> sorted by position in ws("route_records"); first / last are finished by mhx_dist_apply_routed
    int hi_bit = 2;
    while (hi_bit < 64 && ((c->global_bases << 1) >> hi_bit)) ++hi_bit;
    stash_route_records(c, o.events, o.n_events, hi_bit);
  } else if (ns) {
    MHX_LAUNCH(c, "fix_last", (double)ns * 8, hipLaunchKernelGGL(k_fix_last, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, last, last_out, ns));
  }
  mhx::DevBuf &si = c->results[MHX_BUF_SORTED_ITEMS];
  si.release();
  c->sorted_item_words = 3;
  si.p = o.sorted;
  si.cap = 0;  // cap 0 = not owned
  si.used = o.sorted ? o.n_items * 12 : 0;  // (pre-sorted sources stay where they are)
  c->count_acc_k = k;
  c->count_acc_m = m;
  c->last_s1_plan = "count: " + o.plan;
  MHX_HIP(hipStreamSynchronize(st));
  if (out) {
    out->n_items = o.n_items;
    out->n_distinct = o.n_distinct;
    out->n_edges = n_edges;
    out->words_per_edge = wpe;
    out->item_words = 3;
  }
  return true;
}

// several GPUs: the records this rank owns, pre-sorted by the plan's prefix in one array per sending rank (comm.hip).  -> 0, or -1 when
// the streaming form gave up (an output region too small): nothing published, the caller gathers the records for the tile path
int count_process_presorted(mhx_ctx *c, uint32_t k, uint32_t m, const S1Sources &src, mhx_count_result *out) {
  if (!c->global_bases) throw Error("count_process_presorted: call mhx_set_global_layout first");
  if (count_run_stream(c, k, m, out, &src)) {
    c->last_s1_plan += " [pre-sorted exchange]";
    return 0;
  }
  return -1;
}

int run_count(mhx_ctx *c, uint32_t k, uint32_t m, mhx_count_result *out) {
  if (c->global_bases) throw Error("count: the global layout is set; use the mhx_dist_* entry points (or mhx_set_global_layout(0, 0))");
  if (count_stream_applies(c, k, m) && (int)div_ceil((k + 1) * 2 + 16, 32) <= 3) {
    c->gen_first_pass = nullptr;
    if (count_run_stream(c, k, m, out)) return 0;
    c->gen_first_pass = nullptr;  // (gave up: the extraction + tile path redoes the job from the reads)
  }
  const StageItems it = extract_stage(c, MHX_STAGE_COUNT, k, m);
  uint32_t *buf_a = c->work["items_a"].as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
  return count_process(c, k, m, buf_a, buf_b, it.n, out);
}

}  // namespace mhx
