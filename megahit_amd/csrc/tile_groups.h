// Tile-wise group processing of a sorted record array (the Lv2Postprocess step of every engine),
// without serial per-item walks.
//
// A workgroup stages a tile of T consecutive records in LDS (coalesced 16-byte loads) and derives, with
// item-parallel flag evaluation + one block scan,
//   runs    maximal stretches of records with the same key AND the same engine-defined run key
//   groups  maximal stretches with the same key prefix (cmp_bits); a group is a list of runs
// A group belongs to the tile that holds its head; the part of its last group that lies beyond the tile
// ("tail") is discovered by wavefront 0 in 64-record coalesced chunks and appended to the run list.
// Engines then work thread-per-GROUP over the short run list (<= a few dozen runs per group), and
// item-parallel where a per-record action is needed (bit sets, atomics).
//
// Ordered output = two launches of the same kernel:
//   EMIT=false  Op::unit_count -> three counters, summed per tile -> tile_tot[3][n_tiles]
//   (exclusive scans of the three rows)
//   EMIT=true   the counters again -> block scan in LDS -> Op::unit_emit(offsets)
// Nothing per-group or per-run is ever written to HBM; each launch reads the records once.
#pragma once
#include "dev_prims.h"

namespace mhx {

constexpr int kTileThreads = 256;
constexpr int kMaxTailRuns = 64;

struct GroupCounts {
  uint32_t c0 = 0, c1 = 0, c2 = 0;
};

// record accessor: LDS inside the tile, global memory beyond it (relative index)
template <int S>
struct TileAcc {
  const uint32_t *lds;
  const uint32_t *glob;  // records of the whole array
  uint64_t base;         // index of the tile's first record
  uint64_t n;            // total records
  int t_n;               // records staged in LDS (tile + look-ahead); record -1 (the predecessor) is staged too
  // Keep the LDS path and the (rare) HBM path as separate code: a select between an LDS and a global
  // pointer would turn every access into a slow FLAT load.
  __device__ __attribute__((noinline)) uint32_t far_word(uint32_t rel, int w) const { return glob[(base + rel) * S + w]; }
  __device__ __forceinline__ uint32_t word(uint32_t rel, int w) const {
    if (__builtin_expect((int)rel < t_n, 1)) return lds[(int)rel * S + w];
    return far_word(rel, w);
  }
};

// what the engines see of a tile
template <int S>
struct TileCtx {
  TileAcc<S> acc;
  const uint32_t *rpos;   // run starts, relative to the tile; bit 31 = first run of a group; rpos[n_runs] = end
  const uint16_t *gpos;   // first run of each group; gpos[n_groups] = n_runs
  const uint16_t *rgid;   // group of each run
  uint32_t n_runs, n_groups;
  __device__ __forceinline__ bool run_is_group_head(uint32_t r) const { return rpos[r] >> 31; }
  __device__ __forceinline__ uint32_t run_start(uint32_t r) const { return rpos[r] & 0x7FFFFFFFu; }
  __device__ __forceinline__ uint32_t run_len(uint32_t r) const { return (rpos[r + 1] & 0x7FFFFFFFu) - (rpos[r] & 0x7FFFFFFFu); }
};

template <int S>
__device__ __forceinline__ bool key_differs(const uint32_t *a, const uint32_t *b, int full_words, uint32_t last_mask) {
  for (int w = 0; w < full_words; ++w)
    if (a[w] != b[w]) return true;
  return last_mask && ((a[full_words] ^ b[full_words]) & last_mask);
}

// Op interface (all const; LDS state through function-local __shared__ accessors):
//   static constexpr bool kItemPhase, kItemFinal;
//   bool same_run(const uint32_t *cur, const uint32_t *prev)     records of the same group: same run?
//   void begin_block() / end_block()
//   void item_phase(const TileCtx<S>&, uint32_t rel, uint32_t run)   (if kItemPhase; before the group phase)
//   static constexpr bool kRunPhase, kUnitIsRun, kAtomicBase (EMIT only: unordered output, single launch);
//   void run_phase(const TileCtx<S>&, uint32_t r, uint32_t g)       (if kRunPhase: e.g. LDS atomics into group aggregates)
//   GroupCounts unit_count(const TileCtx<S>&, uint32_t u)           u = run (kUnitIsRun) or group
//   void unit_emit(const TileCtx<S>&, uint32_t u, uint64_t o0, uint64_t o1, uint64_t o2)
//   void item_final(const TileCtx<S>&, uint32_t rel, uint32_t run)   (if kItemFinal; after the group phase)
constexpr int kLookAhead = 64;  // records staged beyond the tile so that short tails never touch HBM again
constexpr int kTailWalk = 1024;  // a tail longer than this is searched (64-ary, the records are sorted), not walked

// Debug build only (make timing -> libmhx_timing.so, tools/probe_phases.py): shader-clock ticks per phase of the
// tile kernel, summed over workgroups.
#ifdef MHX_TILE_TIMING
static __device__ unsigned long long g_tile_phase[16];
#define MHX_TT(i)                                      \
  if (threadIdx.x == 0) {                              \
    const unsigned long long now_ = clock64();         \
    atomicAdd(&g_tile_phase[i], now_ - tt_prev_);      \
    tt_prev_ = now_;                                   \
  }
#define MHX_TT_BEGIN unsigned long long tt_prev_ = clock64();
#else
#define MHX_TT(i)
#define MHX_TT_BEGIN
#endif

template <int S, int T, class Op, bool EMIT>
__global__ __launch_bounds__(kTileThreads) void k_tile_groups(const uint32_t *__restrict__ items, uint64_t n, int full_words,
                                                             uint32_t last_mask, Op op, uint64_t *__restrict__ tile_tot,
                                                             const uint64_t *__restrict__ tile_base, uint64_t n_tiles,
                                                             uint64_t n_work, uint32_t tile_stride = 1) {
  constexpr int PER = T / kTileThreads;
  constexpr int NW = kTileThreads / kWave;
  static_assert(T % kTileThreads == 0 && PER <= 16, "tile shape");
  __shared__ __attribute__((aligned(16))) uint32_t tile0[(T + kLookAhead + 4) * S];
  uint32_t *const tile = tile0 + 4 * S;  // tile[-1] holds the record before the tile (16-byte alignment kept)
  __shared__ uint32_t rpos[T + kMaxTailRuns + 1];
  __shared__ uint16_t gpos[T + 1];
  __shared__ uint16_t rgid[T + kMaxTailRuns];
  __shared__ uint64_t gcnt[EMIT ? T + kMaxTailRuns : 1];
  __shared__ uint64_t sm64[NW + 1];
  __shared__ uint32_t cnt_g[PER * NW + 1], cnt_r[PER * NW + 1];  // per (round, wave) head counts -> exclusive prefixes
  __shared__ uint32_t s_first_head, s_tail_end, s_nruns;

  const int tid = threadIdx.x, wv = tid / kWave, lane = tid & (kWave - 1);
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  // Persistent workgroups: tile indices blockIdx.x, blockIdx.x + gridDim.x, ... < n_work.  The records of the NEXT tile
  // are fetched into registers (16-byte loads of the flat word array) while the current tile is processed, so the HBM
  // latency of staging overlaps with the LDS-only phases instead of stalling every tile.
  constexpr int kStageVec = (T + kLookAhead) * S / 4;  // T is a multiple of 256: whole 16-byte vectors, 16-byte aligned
  constexpr int NV = (kStageVec + kTileThreads - 1) / kTileThreads;
  static_assert(NV <= 9, "prefetch registers");
  uint4 pf0, pf1, pf2, pf3, pf4, pf5, pf6, pf7, pf8;  // named (not an array) so that they stay in VGPRs
  uint32_t pf_prev = 0;
  bool pf_valid = false;  // workgroup-uniform: the next tile is a full one (tile + look-ahead inside the array)
#define MHX_PF_EACH(X) X(0, pf0) X(1, pf1) X(2, pf2) X(3, pf3) X(4, pf4) X(5, pf5) X(6, pf6) X(7, pf7) X(8, pf8)
#define MHX_PF_LOAD(I, R)                                                          \
  if constexpr (I < NV) {                                                          \
    if (I + 1 < NV || I * kTileThreads + tid < kStageVec) R = src_[I * kTileThreads + tid]; \
  }
#define MHX_PF_STORE(I, R)                                                         \
  if constexpr (I < NV) {                                                          \
    if (I + 1 < NV || I * kTileThreads + tid < kStageVec) reinterpret_cast<uint4 *>(tile)[I * kTileThreads + tid] = R; \
  }
#define MHX_TILE_PREFETCH(TILE_IDX)                                                     \
  {                                                                                     \
    const uint64_t b_ = (TILE_IDX) * tile_stride * T;                                   \
    pf_valid = n - b_ >= (uint64_t)(T + kLookAhead);                                    \
    if (pf_valid) {                                                                     \
      const uint4 *src_ = reinterpret_cast<const uint4 *>(items + b_ * S);              \
      MHX_PF_EACH(MHX_PF_LOAD)                                                          \
      if (tid < S && b_ != 0) pf_prev = items[b_ * S - S + tid];                        \
    }                                                                                   \
  }
  if (blockIdx.x < n_work) MHX_TILE_PREFETCH((uint64_t)blockIdx.x)
  for (uint64_t tile_idx = blockIdx.x; tile_idx < n_work; tile_idx += gridDim.x) {
  const uint64_t base = tile_idx * tile_stride * T;  // tile_stride > 1: a sample of the tiles (statistics only)
  const uint64_t rem = n - base;
  const int t_n = rem < (uint64_t)T ? (int)rem : T;
  const int staged = rem < (uint64_t)(T + kLookAhead) ? (int)rem : T + kLookAhead;  // tile + look-ahead
  if (tid == 0) s_first_head = 0xFFFFFFFFu;
  MHX_TT_BEGIN
  op.begin_block();
  MHX_TT(0)
  // 1. stage the tile, the look-ahead and the predecessor record
  if (pf_valid) {
    MHX_PF_EACH(MHX_PF_STORE)
    if (tid < S && base != 0) tile[-S + tid] = pf_prev;
  } else {  // the last, partial tile
    const uint32_t *src = items + base * S;
    const int n_words = staged * S, n_vec = n_words / 4;
    for (int v = tid; v < n_vec; v += kTileThreads) reinterpret_cast<uint4 *>(tile)[v] = reinterpret_cast<const uint4 *>(src)[v];
    for (int w = n_vec * 4 + tid; w < n_words; w += kTileThreads) tile[w] = src[w];
    if (tid < S && base != 0) tile[-S + tid] = src[-S + tid];
  }
  __syncthreads();
  if (tile_idx + gridDim.x < n_work) MHX_TILE_PREFETCH(tile_idx + gridDim.x)
  MHX_TT(1)
  // 2. group-head / run-head flags, striped: round j, thread t looks at record j*256 + t (conflict-free LDS reads)
  uint32_t gflags = 0, rflags = 0;
  uint32_t my_first = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = j * kTileThreads + tid;
    if (i < t_n) {
      const uint32_t *cur = tile + i * S;
      const uint32_t *prev = tile + (i - 1) * S;  // LDS also for i == 0 (staged predecessor)
      const bool gh = (i == 0 && base == 0) || key_differs<S>(cur, prev, full_words, last_mask);
      const bool rh = gh || !op.same_run(cur, prev);
      if (gh) {
        gflags |= 1u << j;
        if (my_first == 0xFFFFFFFFu) my_first = (uint32_t)i;
      }
      if (rh) rflags |= 1u << j;
    }
  }
  if (my_first != 0xFFFFFFFFu) atomicMin(&s_first_head, my_first);
  __syncthreads();
  // records before the tile's first group head belong to a group owned by an earlier tile: not ours
  const uint32_t first_head = s_first_head;
  uint64_t gball[PER], rball[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t i = (uint32_t)(j * kTileThreads + tid);
    const bool rv = ((rflags >> j) & 1u) && i >= first_head;
    if (!rv) rflags &= ~(1u << j);
    gball[j] = __ballot((gflags >> j) & 1u);
    rball[j] = __ballot(rv);
    if (lane == 0) {
      cnt_g[j * NW + wv] = (uint32_t)__builtin_popcountll(gball[j]);
      cnt_r[j * NW + wv] = (uint32_t)__builtin_popcountll(rball[j]);
    }
  }
  __syncthreads();
  if (tid == 0) {  // exclusive prefix over the PER*NW (round, wave) cells, in record order
    uint32_t ag = 0, ar = 0;
    for (int x = 0; x < PER * NW; ++x) {
      const uint32_t g = cnt_g[x], r = cnt_r[x];
      cnt_g[x] = ag;
      cnt_r[x] = ar;
      ag += g;
      ar += r;
    }
    cnt_g[PER * NW] = ag;
    cnt_r[PER * NW] = ar;
    s_tail_end = (uint32_t)t_n;
    s_nruns = ar;
  }
  __syncthreads();
  const uint32_t n_groups = cnt_g[PER * NW], n_runs_tile = cnt_r[PER * NW];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if ((rflags >> j) & 1u) {
      const uint32_t r = cnt_r[j * NW + wv] + (uint32_t)__builtin_popcountll(rball[j] & lanemask_lt);
      const bool gh = (gflags >> j) & 1u;
      // groups at or before this record (this record's own head included)
      const uint32_t g_incl = cnt_g[j * NW + wv] + (uint32_t)__builtin_popcountll(gball[j] & lanemask_lt) + (gh ? 1u : 0u);
      rpos[r] = (uint32_t)(j * kTileThreads + tid) | (gh ? 0x80000000u : 0u);
      if (gh) gpos[g_incl - 1] = (uint16_t)r;
      rgid[r] = (uint16_t)(g_incl - 1);
    }
  }
  __syncthreads();
  MHX_TT(2)
  // 3. tail of the last group, beyond the tile: wavefront 0, 64 records per step; the first step reads the
  //    look-ahead records already in LDS.  A tail that is still going after kTailWalk records (low-complexity input: 10^7 records of
  //    one key in a row) is not walked any further — 64 records per memory round trip took seconds per group — but SEARCHED: the
  //    records are sorted, so "same group as the head" and "same run as record q" each hold on a stretch whose end a 64-ary search
  //    finds in four round trips (round 6)
  if (n_groups > 0 && base + t_n < n && tid < kWave) {
    const uint32_t *gkey = tile + (size_t)(rpos[gpos[n_groups - 1]] & 0x7FFFFFFFu) * S;
    uint32_t nr = n_runs_tile, e = (uint32_t)t_n;
    // first position in [lo, hi) where pred turns true (pred: false ... false true ... true); hi if it never does
    auto wave_first = [&](uint64_t lo, uint64_t hi, auto pred) -> uint64_t {
      while (lo < hi) {
        const uint64_t span = hi - lo, step = (span + kWave - 1) / kWave;
        const uint64_t p = lo + (uint64_t)lane * step;
        const uint64_t m = __ballot(p < hi && pred(p));
        if (!m) {  // every probe false: the answer lies behind the last one
          if (step == 1) return hi;  // (every position probed: the narrowed hi — where pred holds — or the end of the range)
          lo += (span - 1) / step * step + 1;
          continue;
        }
        const int f = __builtin_ctzll(m);
        if (f == 0) return lo;
        hi = lo + (uint64_t)f * step;            // pred(hi) is true ...
        lo = hi - step + 1;                      // ... pred(lo - 1) is false
      }
      return hi;
    };
    for (;;) {
      if (e - (uint32_t)t_n >= (uint32_t)kTailWalk) {
        // the group's end, then the run heads between here and there, one search each
        const uint64_t g_end = wave_first(base + e, n, [&](uint64_t p) { return key_differs<S>(items + p * S, gkey, full_words, last_mask); });
        uint64_t q = base + e;  // the first record the walk has not looked at (record q - 1 belongs to the group)
        auto head_at = [&](uint64_t p) {
          if (lane == 0 && nr < (uint32_t)(T + kMaxTailRuns)) {
            rpos[nr] = (uint32_t)(p - base);
            rgid[nr] = (uint16_t)(n_groups - 1);
          }
          ++nr;
        };
        if (q < g_end && !op.same_run(items + q * S, items + (q - 1) * S)) head_at(q);
        while (q < g_end) {
          const uint32_t *rq = items + q * S;
          const uint64_t r_end = wave_first(q + 1, g_end, [&](uint64_t p) { return !op.same_run(items + p * S, rq); });
          if (r_end < g_end) head_at(r_end);
          q = r_end;
        }
        e = (uint32_t)(g_end - base);
        break;
      }
      const uint32_t rel = e + lane;
      const uint64_t p = base + rel;
      bool in_group = false, rh = false;
      if (e + kWave <= (uint32_t)staged) {  // whole step inside the staged look-ahead: LDS only
        const uint32_t *cur = tile + (size_t)rel * S;
        in_group = !key_differs<S>(cur, gkey, full_words, last_mask);
        if (in_group) rh = !op.same_run(cur, cur - S);
      } else if (p < n) {                   // beyond it: HBM
        const uint32_t *cur = items + p * S;
        in_group = !key_differs<S>(cur, gkey, full_words, last_mask);
        if (in_group) rh = !op.same_run(cur, cur - S);
      }
      const uint64_t m_in = __ballot(in_group);
      const int cnt = m_in == ~0ull ? 64 : __builtin_ctzll(~m_in);  // leading in-group lanes
      const uint64_t m_rh = __ballot(rh && lane < cnt);
      if (rh && lane < cnt) {
        const uint32_t slot = nr + __builtin_popcountll(m_rh & lanemask_lt);
        if (slot < (uint32_t)(T + kMaxTailRuns)) {
          rpos[slot] = rel;
          rgid[slot] = (uint16_t)(n_groups - 1);
        }
      }
      nr += __builtin_popcountll(m_rh);
      e += cnt;
      if (cnt < 64) break;
    }
    if (lane == 0) {
      s_tail_end = e;
      s_nruns = nr < (uint32_t)(T + kMaxTailRuns) ? nr : (uint32_t)(T + kMaxTailRuns);
    }
  }
  __syncthreads();
  const uint32_t n_runs = s_nruns, tail_end = s_tail_end;
  if (tid == 0) {
    rpos[n_runs] = tail_end;
    gpos[n_groups] = (uint16_t)n_runs;
  }
  __syncthreads();

  TileCtx<S> ctx{{tile, items, base, n, staged}, rpos, gpos, rgid, n_runs, n_groups};
  MHX_TT(3)

  // run index of a tail record (few tail runs: linear search)
  auto tail_run = [&](uint32_t rel) -> uint32_t {
    uint32_t r = n_runs_tile ? n_runs_tile - 1 : 0;
    while (r + 1 < n_runs && (rpos[r + 1] & 0x7FFFFFFFu) <= rel) ++r;
    return r;
  };
  // run of this thread's record in round j (records before the first head have none)
  auto own_run = [&](int j) -> int {
    return (int)(cnt_r[j * NW + wv] + (uint32_t)__builtin_popcountll(rball[j] & (lanemask_lt | (1ull << lane)))) - 1;
  };

  // 4. optional item-parallel pass before the unit phase
  if constexpr (Op::kItemPhase) {
    if (op.item_phase_enabled() && n_groups > 0) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const uint32_t i = (uint32_t)(j * kTileThreads + tid);
        if (i < (uint32_t)t_n && i >= first_head) op.item_phase(ctx, i, (uint32_t)own_run(j));
      }
      // (the tail: the whole workgroup — one wavefront with one load in flight per lane took seconds for a tail of 10^7 records)
      for (uint32_t rel = (uint32_t)t_n + tid; rel < tail_end; rel += kTileThreads) op.item_phase(ctx, rel, tail_run(rel));
    }
    __syncthreads();
  }

  MHX_TT(4)
  // 5. unit phase: a unit is a run (Op::kUnitIsRun) or a group
  if constexpr (Op::kRunPhase) {
    for (uint32_t r = tid; r < n_runs; r += kTileThreads) op.run_phase(ctx, r, rgid[r]);
    __syncthreads();
  }
  const uint32_t n_units = Op::kUnitIsRun ? n_runs : n_groups;
  if constexpr (!EMIT) {
    uint64_t t0 = 0, t1 = 0, t2 = 0;
    for (uint32_t u = tid; u < n_units; u += kTileThreads) {
      GroupCounts c = op.unit_count(ctx, u);
      t0 += c.c0;
      t1 += c.c1;
      t2 += c.c2;
    }
    if (tile_tot) {
      uint64_t s0, s1, s2;
      block_exclusive_sum<uint64_t, kTileThreads>(t0, sm64, &s0);
      block_exclusive_sum<uint64_t, kTileThreads>(t1, sm64, &s1);
      block_exclusive_sum<uint64_t, kTileThreads>(t2, sm64, &s2);
      if (tid == 0) {
        tile_tot[tile_idx] = s0;
        tile_tot[n_tiles + tile_idx] = s1;
        tile_tot[2 * n_tiles + tile_idx] = s2;
      }
    }
  } else {
    // per-unit counters, packed 21 bits each (a tile never yields 2^21 outputs)
    for (uint32_t u = tid; u < n_units; u += kTileThreads) {
      GroupCounts c = op.unit_count(ctx, u);
      gcnt[u] = (uint64_t)c.c0 | ((uint64_t)c.c1 << 21) | ((uint64_t)c.c2 << 42);
    }
    __syncthreads();
    MHX_TT(5)
    const uint32_t chunk = (n_units + kTileThreads - 1) / kTileThreads;
    const uint32_t lo = min(n_units, tid * chunk), hi = min(n_units, lo + chunk);
    uint64_t s = 0;
    for (uint32_t u = lo; u < hi; ++u) s += gcnt[u];
    uint64_t tot;
    uint64_t run = block_exclusive_sum<uint64_t, kTileThreads>(s, sm64, &tot);
    for (uint32_t u = lo; u < hi; ++u) {
      const uint64_t v = gcnt[u];
      gcnt[u] = run;
      run += v;
    }
    __syncthreads();
    uint64_t b0, b1, b2;
    if constexpr (Op::kAtomicBase) {
      // output order across tiles does not matter: reserve this tile's slice with one atomic per counter
      // (tile_tot = the three global cursors) instead of a scan over per-tile totals in a previous launch
      __shared__ uint64_t s_base[3];
      if (tid == 0) {
        const uint64_t t0 = tot & 0x1FFFFF, t1 = (tot >> 21) & 0x1FFFFF, t2 = tot >> 42;
        s_base[0] = t0 ? atomicAdd(reinterpret_cast<unsigned long long *>(tile_tot), (unsigned long long)t0) : 0;
        s_base[1] = t1 ? atomicAdd(reinterpret_cast<unsigned long long *>(tile_tot) + 1, (unsigned long long)t1) : 0;
        s_base[2] = t2 ? atomicAdd(reinterpret_cast<unsigned long long *>(tile_tot) + 2, (unsigned long long)t2) : 0;
      }
      __syncthreads();
      b0 = s_base[0];
      b1 = s_base[1];
      b2 = s_base[2];
    } else {
      b0 = tile_base[tile_idx];
      b1 = tile_base[n_tiles + tile_idx];
      b2 = tile_base[2 * n_tiles + tile_idx];
    }
    MHX_TT(6)
    for (uint32_t u = tid; u < n_units; u += kTileThreads) {
      const uint64_t p = gcnt[u];
      op.unit_emit(ctx, u, b0 + (p & 0x1FFFFF), b1 + ((p >> 21) & 0x1FFFFF), b2 + (p >> 42));
    }
  }
  // 6. optional item-parallel pass after the unit phase
  if constexpr (Op::kItemFinal) {
    __syncthreads();
    MHX_TT(7)
    if (op.item_final_enabled() && n_groups > 0) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const uint32_t i = (uint32_t)(j * kTileThreads + tid);
        if (i < (uint32_t)t_n && i >= first_head) op.item_final(ctx, i, (uint32_t)own_run(j));
      }
      for (uint32_t rel = (uint32_t)t_n + tid; rel < tail_end; rel += kTileThreads) op.item_final(ctx, rel, tail_run(rel));
    }
  }
  __syncthreads();
  MHX_TT(8)
  op.end_block();
  MHX_TT(9)
  __syncthreads();
  }  // tiles of this workgroup
#undef MHX_TILE_PREFETCH
#undef MHX_PF_EACH
#undef MHX_PF_LOAD
#undef MHX_PF_STORE
}

// grid of the persistent launch: a few workgroups per CU (LDS allows 2-4), each looping over its tiles
inline unsigned tile_grid(uint64_t n_work) { return (unsigned)(n_work < 2048 ? (n_work ? n_work : 1) : 2048); }

}  // namespace mhx
