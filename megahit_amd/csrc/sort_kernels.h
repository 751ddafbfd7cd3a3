// Device-side pieces of the radix sort that other translation units instantiate too (sort.hip is their home; s1.hip
// builds the chained-scan pass whose records are GENERATED from the packed reads instead of loaded).
#pragma once
#include "dev_prims.h"
#include "sort_digits.h"

namespace mhx {

constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / kWave;

// default records per thread per tile, chosen so the LDS stage stays <= 64 KiB
template <int S>
constexpr int default_items() {
  return (S <= 4) ? 4 : (S <= 8 ? 8 : (S <= 12 ? 5 : (S <= 16 ? 4 : 3)));
}
template <int S, int ITEMS>
struct SortCfg {
  static constexpr int kItems = ITEMS;
  static constexpr int kTile = kSortThreads * kItems;
  static constexpr int kTilesPerChunk = (16384 / kTile) > 4 ? (16384 / kTile) : 4;  // ~16 K records per chunk
  static constexpr int kChunk = kTile * kTilesPerChunk;
};

template <int S>
struct Rec {
  uint32_t w[S];
};

template <int S>
__device__ __forceinline__ void load_rec(const uint32_t *__restrict__ p, Rec<S> &r) {
  if constexpr (S % 4 == 0) {
#pragma unroll
    for (int i = 0; i < S / 4; ++i) {
      uint4 v = reinterpret_cast<const uint4 *>(p)[i];
      r.w[4 * i] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
    }
  } else if constexpr (S % 2 == 0) {
#pragma unroll
    for (int i = 0; i < S / 2; ++i) {
      uint2 v = reinterpret_cast<const uint2 *>(p)[i];
      r.w[2 * i] = v.x; r.w[2 * i + 1] = v.y;
    }
  } else {  // odd strides (12-byte records): dword accesses, merged by the compiler where alignment allows
#pragma unroll
    for (int i = 0; i < S; ++i) r.w[i] = p[i];
  }
}
template <int S>
__device__ __forceinline__ void store_rec(uint32_t *__restrict__ p, const Rec<S> &r) {
  if constexpr (S % 4 == 0) {
#pragma unroll
    for (int i = 0; i < S / 4; ++i)
      reinterpret_cast<uint4 *>(p)[i] = make_uint4(r.w[4 * i], r.w[4 * i + 1], r.w[4 * i + 2], r.w[4 * i + 3]);
  } else if constexpr (S % 2 == 0) {
#pragma unroll
    for (int i = 0; i < S / 2; ++i) reinterpret_cast<uint2 *>(p)[i] = make_uint2(r.w[2 * i], r.w[2 * i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) p[i] = r.w[i];
  }
}

// digit of a record held in registers: bits [bit, bit+nbits) of word wi (and wi-1 when straddling)
template <int S>
__device__ __forceinline__ unsigned rec_digit(const Rec<S> &r, int wi, unsigned bit, unsigned mask) {
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    if (i == wi) lo = r.w[i];
    if (i == wi - 1) hi = r.w[i];
  }
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (unsigned)(v >> bit) & mask;
}
template <int S>
__device__ __forceinline__ unsigned rec_digit2(const Rec<S> &r, const DigitSpec &ds) {
  unsigned d = rec_digit<S>(r, ds.wi1, ds.bit1, ds.mask1);
  if (ds.mask2) d |= rec_digit<S>(r, ds.wi2, ds.bit2, ds.mask2) << ds.sh2;
  return d;
}
__device__ __forceinline__ unsigned mem_digit(const uint32_t *p, int wi, unsigned bit, unsigned mask) {
  uint64_t v = p[wi];
  if (bit + (32 - __builtin_clz(mask)) > 32 && wi > 0) v |= (uint64_t)p[wi - 1] << 32;
  return (unsigned)(v >> bit) & mask;
}


// where a pass takes its records from: the array of the previous pass ...
template <int S>
struct SrcArray {
  const uint32_t *in;
  // kMayDrop: the source may decline item slots (a generator that filters by lv1 bucket): a unit then holds fewer records
  // than item slots and the pass compacts them (k_radix_onesweep_u counts what is_record() accepts)
  static constexpr bool kMayDrop = false;
  __device__ __forceinline__ bool is_record(const Rec<S> &) const { return true; }
  // records first, first + 64, ... (NI of them, those below n) of one thread
  template <int NI>
  __device__ __forceinline__ void get(uint64_t first, uint64_t n, Rec<S> (&rec)[NI]) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint64_t gi = first + (uint64_t)j * kWave;
      if (gi < n) load_rec<S>(in + gi * S, rec[j]);
    }
  }
  // (interface of k_radix_onesweep_u: the item a thread holds in slot j of a tile, and all records of a unit that are one thread's)
  template <int NI>
  __device__ __forceinline__ uint64_t index(uint64_t tile_base, int w, int lane, int j) const {
    return tile_base + (uint64_t)(w * (kWave * NI) + j * kWave + lane);
  }
  template <int NI, int UT>
  __device__ __forceinline__ void get_unit(uint64_t unit_base, int w, int lane, uint64_t n, Rec<S> (&rec)[UT][NI]) const {
#pragma unroll
    for (int t = 0; t < UT; ++t) get<NI>(unit_base + (uint64_t)t * (kSortThreads * NI) + (uint64_t)(w * (kWave * NI) + lane), n, rec[t]);
  }
};
// ... or a generator (s1.hip: S1Gen makes the stage-1 records straight from the packed reads in the first pass)

constexpr unsigned long long kStValMask = (1ull << 56) - 1;

// ANY_ORDER: the records of one digit may leave in any order (the pass is a partition, not a stable sort): a record's rank
// inside its wavefront is then ONE returning LDS atomic instead of the 8-ballot match-any (~45 VALU operations) — for a
// first pass whose consumer only needs the records grouped (s1.hip: the generating pass, which is instruction-bound).
template <int S, int NI, int UT, class Src, bool ANY_ORDER = false>
__global__ __launch_bounds__(kSortThreads) void k_radix_onesweep(Src src, uint32_t *__restrict__ out, uint64_t n,
                                                                 DigitSpec ds, int nbits, const unsigned long long *__restrict__ bin_start,
                                                                 unsigned long long *__restrict__ status, uint32_t *__restrict__ ticket,
                                                                 uint32_t *__restrict__ err, unsigned long long tag, int xcd_units) {
  constexpr int kTile = kSortThreads * NI;
  __shared__ __attribute__((aligned(16))) uint32_t stage[kTile * S];
  __shared__ uint32_t wave_cnt[kSortWaves][256];
  __shared__ long long g_off[256];
  __shared__ uint64_t g_base[256];
  __shared__ uint32_t sm_scan[kSortThreads / kWave + 1];
  __shared__ uint32_t s_unit;

  const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  if (tid == 0) {
    if (xcd_units) {
      // Neighbouring units write neighbouring runs of every bin.  Block b is placed on XCD b % 8 and each XCD has an L2 of
      // its own, so with one ticket counter the two halves of nearly every boundary cache line are dirtied in two
      // different L2s and leave them as two partial-line writes (measured: 1.5x the algorithmic bytes written).  With one
      // counter per b % 8 and runs of 16 consecutive units per counter, 15 of 16 boundaries stay inside one L2.
      // A unit only ever waits for lower-numbered units = tickets of lower or equal index = lower block ids: blocks that
      // were dispatched before it, on whichever XCD (the grid is a multiple of 128, so the map is a bijection).
      const uint32_t cls = blockIdx.x & 7u, tk = atomicAdd(ticket + cls, 1u);
      s_unit = ((tk >> 4) * 8 + cls) * 16 + (tk & 15u);
    } else {
      s_unit = atomicAdd(ticket, 1u);
    }
  }
#pragma unroll
  for (int i = 0; i < kSortWaves; ++i) wave_cnt[i][tid] = 0;
  __syncthreads();
  const uint64_t unit = s_unit;
  const uint64_t unit_base = unit * (uint64_t)(kTile * UT);

  // 1. the unit's records -> registers (wave-blocked striped arrangement inside each tile, as in k_radix_scatter)
  Rec<S> rec[UT][NI];
#pragma unroll
  for (int t = 0; t < UT; ++t) src.template get<NI>(unit_base + (uint64_t)t * kTile + (uint64_t)(w * (kWave * NI) + lane), n, rec[t]);
  // 2. digit counts of the unit (per-wave LDS histograms), published before anything depends on other units
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint64_t gi = unit_base + (uint64_t)t * kTile + (uint64_t)(w * (kWave * NI) + j * kWave + lane);
      if (gi < n) atomicAdd(&wave_cnt[w][rec_digit2<S>(rec[t][j], ds)], 1u);
    }
  __syncthreads();
  unsigned long long unit_tot = 0;
#pragma unroll
  for (int i = 0; i < kSortWaves; ++i) unit_tot += wave_cnt[i][tid];
  unsigned long long *const st = status + unit * 256 + tid;
  const unsigned long long tagbits = tag << 58;
  __hip_atomic_store(st, tagbits | ((unit == 0 ? 2ull : 1ull) << 56) | unit_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();

  // 4. tile by tile: rank, stage in LDS by digit, write the per-digit runs (same as k_radix_scatter)
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    const uint64_t tile_base = unit_base + (uint64_t)t * kTile;
    if (tile_base >= n) break;
    const uint64_t rem = n - tile_base;
    const int tile_n = rem < (uint64_t)kTile ? (int)rem : kTile;
#pragma unroll
    for (int i = 0; i < kSortWaves; ++i) wave_cnt[i][tid] = 0;
    __syncthreads();
    uint32_t rank[NI];
    unsigned dig[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int li = w * (kWave * NI) + j * kWave + lane;
      const bool valid = li < tile_n;
      const unsigned d = valid ? rec_digit2<S>(rec[t][j], ds) : 0u;
      dig[j] = d;
      if constexpr (ANY_ORDER) {
        rank[j] = valid ? atomicAdd(&wave_cnt[w][d], 1u) : 0u;
      } else {
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
          const bool bitset = (d >> b) & 1u;
          const uint64_t m = __ballot(bitset);
          peers &= bitset ? m : ~m;
        }
        const uint32_t before = wave_cnt[w][d];
        rank[j] = before + __builtin_popcountll(peers & lanemask_lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lanemask_lt) == 0) wave_cnt[w][d] = before + __builtin_popcountll(peers);
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    {
      uint32_t c[kSortWaves], tot = 0;
#pragma unroll
      for (int i = 0; i < kSortWaves; ++i) {
        c[i] = wave_cnt[i][tid];
        tot += c[i];
      }
      const uint32_t start = block_exclusive_sum<uint32_t, kSortThreads>(tot, sm_scan, nullptr);
      uint32_t run = start;
#pragma unroll
      for (int i = 0; i < kSortWaves; ++i) {
        wave_cnt[i][tid] = run;
        run += c[i];
      }
      if (t == 0) {
        // 3. decoupled look-back (after the first tile's ranking, so that the predecessors had time to publish): sum the
        //    counts of the predecessors until one of them knows its inclusive prefix
        unsigned long long excl = 0;
        if (unit > 0) {
          for (uint64_t p = unit; p-- > 0;) {
            unsigned long long v;
            uint32_t polls = 0;  // per predecessor: a unit only ever waits for units that already run (ticket order)
            for (;;) {
              v = __hip_atomic_load(status + p * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((v >> 58) == tag && ((v >> 56) & 3ull)) break;
              if (++polls > (1u << 27)) {  // seconds of polling one predecessor (a wedged GPU): terminate, the host raises on *err
                atomicOr(err, 1u);
                v = 2ull << 56;
                break;
              }
              if (polls < 64) __builtin_amdgcn_s_sleep(1);
              else __builtin_amdgcn_s_sleep(8);
            }
            excl += v & kStValMask;
            if (((v >> 56) & 3ull) == 2ull) break;
          }
          __hip_atomic_store(st, tagbits | (2ull << 56) | (excl + unit_tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        g_base[tid] = bin_start[tid] + excl;
      }
      g_off[tid] = (long long)g_base[tid] - (long long)start;
      g_base[tid] += tot;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int li = w * (kWave * NI) + j * kWave + lane;
      if (li < tile_n) store_rec<S>(stage + (size_t)(wave_cnt[w][dig[j]] + rank[j]) * S, rec[t][j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int li = j * kSortThreads + tid;
      if (li < tile_n) {
        Rec<S> r;
        load_rec<S>(stage + (size_t)li * S, r);
        const unsigned d = rec_digit2<S>(r, ds);
        store_rec<S>(out + (uint64_t)(g_off[d] + li) * S, r);
      }
    }
    __syncthreads();
  }
}


// digit of a record when the pass's digit is ONE bit field inside word WI (no second field, no straddle): a shift and an
// and — the generic rec_digit picks the word with 2 S selects per call (~40 v_cndmask per record over the three uses)
template <int S, int WI>
__device__ __forceinline__ unsigned rec_digit_w(const Rec<S> &r, const DigitSpec &ds) {
  if constexpr (WI >= 0) return (r.w[WI] >> ds.bit1) & ds.mask1;
  else return rec_digit2<S>(r, ds);
}
// host side of the same: the word a pass's digit lies in, -1 when the generic form is needed
inline int digit_word_of(const DigitSpec &ds, int nbits, int max_word) {
  return (ds.mask2 == 0 && ds.wi1 >= 0 && ds.wi1 <= max_word && (int)ds.bit1 + nbits <= 32) ? ds.wi1 : -1;
}

// The chained-scan pass with UNIT-WIDE runs.  k_radix_onesweep above ranks, stages and writes its unit tile by tile: a
// unit of UT tiles leaves UT separate runs per digit (8 records = 96 bytes each at 12-byte records), written tens of
// microseconds apart — longer than a dirty line survives in the XCD's L2 under this kernel's own traffic, so the two
// halves of most boundary lines reach the memory as two partial writes (PMC: 1.5 x the algorithmic bytes written).
// Here all UT tiles are ranked first (the per-(tile, wave) counters of one digit laid out in input order: the same
// stable order), every record gets its position p inside the unit's digit-sorted sequence, and the LDS stage is a WINDOW
// sliding over that sequence: round r stages the records with p in [r T, (r+1) T) and writes them, so every digit's run
// of the unit leaves once, in one piece (two pieces for the at most UT - 1 digits a window edge cuts).  Same LDS stage,
// the counters UT times as large, no separate counting phase before the ranking (the ranking's counters ARE the counts
// the look-back publishes), fewer barriers.  Measured at 1.33 G 12-byte records (round 3, A/B inside one run, PMC passes):
// bytes written per pass 24.7 -> 17.0 GB (1.55 -> 1.06 x algorithmic), 10.6 -> 8.9 ms per pass; the generating first pass
// 19.6 -> 17.0 GB, 10.0 -> 9.3 ms; 8-byte records 0.73 -> 0.65 ms per pass; 16-byte records (count) 11.5 -> 10.9 ms.
// (12-byte records, 8x3 units: 138 registers = 3 workgroups per CU.  Asking the compiler for 128 = 4 workgroups costs 13
// spilled registers and measured slower: 10.7 instead of 9.2 ms per pass.)
// RANK: how a record finds its rank among the records of its wavefront instruction with the same digit.
//   0  match-any over the digit bits (8 ballots): stable by construction — every pass whose input order matters;
//   1  ONE returning LDS atomic on the digit's counter: the lanes of a digit are ranked in whatever order the LDS applies
//      them — for a pass whose records may leave in any order inside a digit (the generating first pass of stage 1);
//   2  the atomic where it provably cannot matter, the ballots elsewhere: an LSD pass has to keep the order of two records
//      only if they differ in the bits sorted so far (ds.prev_mask, bits of word WI) — when all records of the instruction
//      agree on those, any order among them is a correct outcome for a consumer that does not look at the order of records
//      equal in ALL sorted bits (the LDS group-by of stage 1).  Behind a pass over the neighbouring digit nearly every
//      instruction qualifies (its 64 records are consecutive in the input).  No assumption about the LDS is involved.
template <int S, int NI, int UT, class Src, int RANK, int WI>
__global__ __launch_bounds__(kSortThreads) void k_radix_onesweep_u(Src src, uint32_t *__restrict__ out, uint64_t n, DigitSpec ds, int nbits,
                                                                   const unsigned long long *__restrict__ bin_start,
                                                                   unsigned long long *__restrict__ status, uint32_t *__restrict__ ticket,
                                                                   uint32_t *__restrict__ err, unsigned long long tag, int xcd_units) {
  constexpr int kTile = kSortThreads * NI;
  static_assert(kTile * (UT + 1) < 0xFFFF, "16-bit ranks and positions");
  __shared__ __attribute__((aligned(16))) uint32_t stage[kTile * S];
  __shared__ uint32_t cnt[UT][kSortWaves][256];  // counts of (tile, wave, digit), then their starts inside the unit
  __shared__ long long g_off[256];               // global position minus position inside the unit, per digit
  __shared__ uint32_t sm_scan[kSortThreads / kWave + 1];
  __shared__ uint32_t s_unit;

  const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  if (tid == 0) {
    if (xcd_units) {  // (see k_radix_onesweep)
      const uint32_t cls = blockIdx.x & 7u, tk = atomicAdd(ticket + cls, 1u);
      s_unit = ((tk >> 4) * 8 + cls) * 16 + (tk & 15u);
    } else {
      s_unit = atomicAdd(ticket, 1u);
    }
  }
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int i = 0; i < kSortWaves; ++i) cnt[t][i][tid] = 0;
  __syncthreads();
  const uint64_t unit = s_unit;
  const uint64_t unit_base = unit * (uint64_t)(kTile * UT);

  // 1. the unit's records -> registers (loaded: wave-blocked striped arrangement inside each tile; generated: the source's choice)
  Rec<S> rec[UT][NI];
  src.template get_unit<NI, UT>(unit_base, w, lane, n, rec);

  // 2. rank every record among the records of its (tile, wave) with the same digit.  Ranks, later positions, are kept two
  //    per register (16 bits each; 0xFFFF = no record): the unit's records already take 72 registers at 12 bytes
  uint32_t pk[UT][(NI + 1) / 2];
#pragma unroll
  for (int t = 0; t < UT; ++t) {
#pragma unroll
    for (int h = 0; h < (NI + 1) / 2; ++h) pk[t][h] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint64_t gi = src.template index<NI>(unit_base + (uint64_t)t * kTile, w, lane, j);  // (which item this is: the source's arrangement)
      bool valid = gi < n;
      if constexpr (Src::kMayDrop) valid = valid && src.is_record(rec[t][j]);
      const unsigned d = valid ? rec_digit_w<S, WI>(rec[t][j], ds) : 0u;
      uint32_t rk;
      bool by_atomic = RANK == 1;
      if constexpr (RANK == 2) {
        static_assert(WI >= 0, "RANK 2 needs the digit and the bits sorted before it in one word");
        const uint64_t vm = __ballot(valid);
        const uint32_t pv = valid ? (rec[t][j].w[WI >= 0 ? WI : 0] & ds.prev_mask) : 0u;
        const uint32_t p0 = __shfl(pv, vm ? __builtin_ctzll(vm) : 0, kWave);
        by_atomic = __ballot(valid && pv != p0) == 0;
      }
      if (by_atomic) {
        rk = valid ? atomicAdd(&cnt[t][w][d], 1u) : 0xFFFFu;
      } else {
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
          const bool bitset = (d >> b) & 1u;
          const uint64_t m = __ballot(bitset);
          peers &= bitset ? m : ~m;
        }
        const uint32_t before = cnt[t][w][d];
        rk = valid ? before + __builtin_popcountll(peers & lanemask_lt) : 0xFFFFu;
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lanemask_lt) == 0) cnt[t][w][d] = before + __builtin_popcountll(peers);
        __builtin_amdgcn_wave_barrier();
      }
      pk[t][j >> 1] = (j & 1) ? ((pk[t][j >> 1] & 0xFFFFu) | (rk << 16)) : ((pk[t][j >> 1] & 0xFFFF0000u) | rk);
      // finish this record's rank here: left alone the compiler keeps `before` and the peer mask of all 24 records alive
      // until the positions are formed (~70 registers = one workgroup per CU less)
      asm volatile("" : "+v"(pk[t][j >> 1]));
    }
  }
  __syncthreads();

  // 3. thread d: the unit's count of digit d -> status word; starts of the (tile, wave) cells inside the unit; look-back
  uint32_t unit_records = 0;  // records of the unit (all item slots below n, unless the source drops some)
  {
    uint32_t c[UT][kSortWaves], tot = 0;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int i = 0; i < kSortWaves; ++i) {
        c[t][i] = cnt[t][i][tid];
        tot += c[t][i];
      }
    unsigned long long *const st = status + unit * 256 + tid;
    const unsigned long long tagbits = tag << 58;
    __hip_atomic_store(st, tagbits | ((unit == 0 ? 2ull : 1ull) << 56) | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t start = block_exclusive_sum<uint32_t, kSortThreads>(tot, sm_scan, &unit_records);
    uint32_t run = start;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int i = 0; i < kSortWaves; ++i) {
        cnt[t][i][tid] = run;
        run += c[t][i];
      }
    unsigned long long excl = 0;
    if (unit > 0 && unit_base < n) {  // (a unit beyond the input — the grid is rounded up — publishes its zero counts and is done)
      for (uint64_t p = unit; p-- > 0;) {
        unsigned long long v;
        uint32_t polls = 0;  // per predecessor: a unit only ever waits for units that already run (ticket order)
        for (;;) {
          v = __hip_atomic_load(status + p * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((v >> 58) == tag && ((v >> 56) & 3ull)) break;
          if (++polls > (1u << 27)) {  // seconds of polling one predecessor (a wedged GPU): terminate, the host raises on *err
            atomicOr(err, 1u);
            v = 2ull << 56;
            break;
          }
          if (polls < 64) __builtin_amdgcn_s_sleep(1);
          else __builtin_amdgcn_s_sleep(8);
        }
        excl += v & kStValMask;
        if (((v >> 56) & 3ull) == 2ull) break;
      }
      __hip_atomic_store(st, tagbits | (2ull << 56) | (excl + (unsigned long long)tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    g_off[tid] = (long long)(bin_start[tid] + excl) - (long long)start;
  }
  __syncthreads();

  // 4. positions inside the unit's digit-sorted sequence, then the window slides over it
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint32_t rk = (pk[t][j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
      if (rk != 0xFFFFu) {
        const uint32_t p = cnt[t][w][rec_digit_w<S, WI>(rec[t][j], ds)] + rk;
        pk[t][j >> 1] = (j & 1) ? ((pk[t][j >> 1] & 0xFFFFu) | (p << 16)) : ((pk[t][j >> 1] & 0xFFFF0000u) | p);
      }
    }
  const uint64_t rem = n > unit_base ? n - unit_base : 0;
  const uint32_t unit_n = Src::kMayDrop ? unit_records : (rem < (uint64_t)(kTile * UT) ? (uint32_t)rem : (uint32_t)(kTile * UT));
#pragma unroll
  for (int r = 0; r < UT; ++r) {
    const uint32_t lo = (uint32_t)r * kTile;
    if (lo >= unit_n) break;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const uint32_t rel = ((pk[t][j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) - lo;  // (no record: 0xFFFF - lo, far outside)
        if (rel < (uint32_t)kTile) store_rec<S>(stage + (size_t)rel * S, rec[t][j]);
      }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint32_t li = (uint32_t)(j * kSortThreads + tid);
      if (lo + li < unit_n) {
        Rec<S> rr;
        load_rec<S>(stage + (size_t)li * S, rr);
        const unsigned d = rec_digit_w<S, WI>(rr, ds);
        store_rec<S>(out + (uint64_t)(g_off[d] + (long long)(lo + li)) * S, rr);
      }
    }
    __syncthreads();
  }
}

// what a generated first pass needs to be launched (sort.hip fills it, the generator's owner launches its instantiation)
struct OnesweepLaunch {
  unsigned grid;
  hipStream_t stream;
  uint32_t *out;
  uint64_t n;
  DigitSpec ds;
  int nbits;
  const unsigned long long *bin_start;
  unsigned long long *status;
  uint32_t *ticket, *err;
  unsigned long long tag;
  int xcd_units;
  int unit_runs;  // launch k_radix_onesweep_u (unit-wide runs) when the generator's owner has it for digit word `wi`
  int wi;         // digit_word_of(ds, nbits, ...)
};

}  // namespace mhx
