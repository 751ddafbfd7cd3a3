// read2sdbg stage 1, tile group-bys: Read2SdbgS1::Lv2Postprocess (reference src/sorting/read_to_sdbg_s1.cpp:368-555) as a tile
// operator on fully sorted records (k_tile_groups<S1Op>: mercy candidates, wide keys, every give-up) and as the segment group-by
// on partially sorted 12-byte records (k_s1_seg).
#include "s1_shared.h"
#include "tile_groups.h"

namespace mhx {

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


// ---- launchers (the only way into this unit's kernels) ----
void s1_seg_launch(mhx_ctx *c, const char *name, double bytes, int per, bool agg, unsigned grid, const uint32_t *sorted, uint64_t n_items, const S1SegArgs &a,
                   uint64_t n_work, uint32_t stride) {
  hipStream_t st = c->stream;
#define MHX_SEG(PERV, AGGV) \
  MHX_LAUNCH(c, name, bytes, hipLaunchKernelGGL((k_s1_seg<PERV, AGGV>), dim3(grid), dim3(256), 0, st, sorted, n_items, a, n_work, stride))
  if (per == 4) {
    if (agg) MHX_SEG(4, true);
    else MHX_SEG(4, false);
  } else {
    if (agg) MHX_SEG(8, true);
    else MHX_SEG(8, false);
  }
#undef MHX_SEG
}

void s1_classic_launch(mhx_ctx *c, int S, bool compact, bool agg, const uint32_t *sorted, uint64_t n_items, int KWv, int kmer_bits, uint32_t m, uint8_t *solid_bytes,
                       unsigned long long *is_solid, int mark_atomic, unsigned long long *hist, unsigned long long *ctr, int want_mercy, long long *&mercy, int k,
                       uint2 *agg_items, uint64_t *agg_cursor, int mark_mode) {
#define MHX_ARGS c, sorted, n_items, KWv, kmer_bits, m, solid_bytes, is_solid, mark_atomic, hist, ctr
  if (agg) {
    if (S == 3 && compact) s1_groups_launch<3, true, true>(MHX_ARGS, 0, mercy, k, agg_items, agg_cursor, mark_mode);
    else if (S == 4 && !compact) s1_groups_launch<4, false, true>(MHX_ARGS, want_mercy, mercy, k, agg_items, agg_cursor, mark_mode);
    else throw Error("read2sdbg_s1: aggregated items on an unsupported record shape");
    return;
  }
#define MHX_CASE(SV)                                                                                               \
  case SV:                                                                                                         \
    if (compact) s1_groups_launch<SV, true, false>(MHX_ARGS, 0, mercy, k, agg_items, agg_cursor, mark_mode);       \
    else s1_groups_launch<SV, false, false>(MHX_ARGS, want_mercy, mercy, k, agg_items, agg_cursor, mark_mode);     \
    break;
  switch (S) {
    MHX_CASE(3) MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
    default: throw Error("read2sdbg_s1: unsupported record stride");
  }
#undef MHX_CASE
#undef MHX_ARGS
}

#ifdef MHX_TILE_TIMING
int s1_stream_phases(unsigned long long *out16, int reset);  // s1_stream.hip
#endif

}  // namespace mhx

#ifdef MHX_TILE_TIMING
// debug build only: phase clocks of the stage-1 kernels (this unit's copy of g_tile_phase: phases 0..9; + the bucket streaming's: 10..14)
extern "C" int mhx_debug_tile_phases(unsigned long long *out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mhx::g_tile_phase), 16 * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mhx::g_tile_phase), z, 16 * 8) != hipSuccess) return -1;
  }
  return mhx::s1_stream_phases(out16, reset);
}
#endif
