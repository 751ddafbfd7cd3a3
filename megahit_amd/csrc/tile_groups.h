// Tile-wise group processing of a sorted record array (the Lv2Postprocess step of every engine).
//
// A workgroup stages one tile of T consecutive records in LDS with coalesced 16-byte loads, finds
// the group heads inside the tile (record differs from its predecessor in the first cmp_bits key
// bits), and lets each thread walk whole groups out of LDS.  A group belongs to the tile that holds
// its head; its tail beyond the tile is read from global memory by the owning thread.
//
// Ordered output needs two launches of the same kernel:
//   EMIT=false  Op::count(group) -> three small counters, summed per tile  -> tile_tot[3][n_tiles]
//   (exclusive scan of the three rows on the host side)
//   EMIT=true   per-group counters again -> block scan in LDS -> Op::emit(group, offsets)
// so the only per-group state ever written to HBM is the output itself; no head array, no per-group
// arrays.  HBM traffic: each launch reads the records once.
#pragma once
#include "dev_prims.h"

namespace mhx {

constexpr int kTileThreads = 256;

template <int S>
struct TileCfg {
  // ~32 KiB of records per tile, a multiple of 256, at least 256
  static constexpr int kRaw = 32768 / (S * 4);
  static constexpr int kT = kRaw >= 4096 ? 4096 : (kRaw >= 256 ? (kRaw / 256) * 256 : 256);
};

// record accessor: LDS inside the tile, global memory beyond it
template <int S>
struct TileAcc {
  const uint32_t *lds;
  const uint32_t *glob;  // records of the whole array
  uint64_t base;         // index of the tile's first record
  uint64_t n;            // total records
  int t_n;               // records staged in LDS
  __device__ __forceinline__ uint32_t word(uint32_t rel, int w) const {
    return rel < (uint32_t)t_n ? lds[rel * S + w] : glob[(base + rel) * S + w];
  }
};

struct GroupCounts {
  uint32_t c0 = 0, c1 = 0, c2 = 0;
};

template <int S>
__device__ __forceinline__ bool key_differs(const uint32_t *a, const uint32_t *b, int full_words, uint32_t last_mask) {
  for (int w = 0; w < full_words; ++w)
    if (a[w] != b[w]) return true;
  return last_mask && ((a[full_words] ^ b[full_words]) & last_mask);
}

// Op interface:
//   __device__ void begin_block();                    (optional per-block LDS init, called by all threads)
//   __device__ GroupCounts count(const TileAcc<S>&, uint32_t b, uint32_t e);   b,e relative to the tile
//   __device__ void emit(const TileAcc<S>&, uint32_t b, uint32_t e, uint64_t o0, uint64_t o1, uint64_t o2);
//   __device__ void end_block();                      (optional flush, called by all threads after a barrier)
template <int S, class Op, bool EMIT>
__global__ __launch_bounds__(kTileThreads) void k_tile_groups(const uint32_t *__restrict__ items, uint64_t n, int full_words,
                                                             uint32_t last_mask, Op op, uint64_t *__restrict__ tile_tot,
                                                             const uint64_t *__restrict__ tile_base, uint64_t n_tiles) {
  constexpr int T = TileCfg<S>::kT;
  constexpr int PER = T / kTileThreads;
  __shared__ __attribute__((aligned(16))) uint32_t tile[T * S];
  __shared__ uint16_t hpos[T + 1];
  __shared__ uint64_t gcnt[EMIT ? T : 1];
  __shared__ uint32_t sm32[kTileThreads / kWave + 1];
  __shared__ uint64_t sm64[kTileThreads / kWave + 1];

  const int tid = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * T;
  const uint64_t rem = n - base;
  const int t_n = rem < (uint64_t)T ? (int)rem : T;
  op.begin_block();
  // 1. stage the tile (coalesced)
  for (int i = tid; i < t_n; i += kTileThreads) {
    const uint32_t *src = items + (base + i) * S;
    uint32_t *dst = tile + i * S;
    if constexpr (S % 4 == 0) {
#pragma unroll
      for (int q = 0; q < S / 4; ++q) reinterpret_cast<uint4 *>(dst)[q] = reinterpret_cast<const uint4 *>(src)[q];
    } else {
#pragma unroll
      for (int q = 0; q < S / 2; ++q) reinterpret_cast<uint2 *>(dst)[q] = reinterpret_cast<const uint2 *>(src)[q];
    }
  }
  __syncthreads();
  // 2. heads, in order: thread t owns records [t*PER, (t+1)*PER)
  uint32_t flags = 0, cnt = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid * PER + j;
    if (i < t_n) {
      bool head;
      if (i == 0) head = base == 0 || key_differs<S>(tile, items + (base - 1) * S, full_words, last_mask);
      else head = key_differs<S>(tile + i * S, tile + (i - 1) * S, full_words, last_mask);
      if (head) {
        flags |= 1u << j;
        ++cnt;
      }
    }
  }
  uint32_t n_heads;
  uint32_t hp = block_exclusive_sum<uint32_t, kTileThreads>(cnt, sm32, &n_heads);
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (flags & (1u << j)) hpos[hp++] = (uint16_t)(tid * PER + j);
  __syncthreads();

  TileAcc<S> acc{tile, items, base, n, t_n};
  // end of the last group: it may run past the tile
  auto group_end = [&](uint32_t g) -> uint32_t {
    if (g + 1 < n_heads) return hpos[g + 1];
    uint32_t e = (uint32_t)t_n;
    if (base + e < n) {  // walk the tail in global memory while the key stays the same
      const uint32_t *first = tile + (size_t)hpos[g] * S;
      while (base + e < n && !key_differs<S>(first, items + (base + e) * S, full_words, last_mask)) ++e;
    }
    return e;
  };

  if constexpr (!EMIT) {
    uint64_t t0 = 0, t1 = 0, t2 = 0;
    for (uint32_t g = tid; g < n_heads; g += kTileThreads) {
      GroupCounts c = op.count(acc, hpos[g], group_end(g));
      t0 += c.c0;
      t1 += c.c1;
      t2 += c.c2;
    }
    uint64_t s0, s1, s2;
    block_exclusive_sum<uint64_t, kTileThreads>(t0, sm64, &s0);
    block_exclusive_sum<uint64_t, kTileThreads>(t1, sm64, &s1);
    block_exclusive_sum<uint64_t, kTileThreads>(t2, sm64, &s2);
    if (tid == 0 && tile_tot) {
      tile_tot[blockIdx.x] = s0;
      tile_tot[n_tiles + blockIdx.x] = s1;
      tile_tot[2 * n_tiles + blockIdx.x] = s2;
    }
  } else {
    // per-group counters, packed 21 bits each (a tile never yields 2^21 outputs)
    for (uint32_t g = tid; g < n_heads; g += kTileThreads) {
      GroupCounts c = op.count(acc, hpos[g], group_end(g));
      gcnt[g] = (uint64_t)c.c0 | ((uint64_t)c.c1 << 21) | ((uint64_t)c.c2 << 42);
    }
    __syncthreads();
    // exclusive scan of gcnt[0..n_heads) in place: contiguous chunks per thread
    const uint32_t chunk = (n_heads + kTileThreads - 1) / kTileThreads;
    const uint32_t lo = min(n_heads, tid * chunk), hi = min(n_heads, lo + chunk);
    uint64_t s = 0;
    for (uint32_t g = lo; g < hi; ++g) s += gcnt[g];
    uint64_t run = block_exclusive_sum<uint64_t, kTileThreads>(s, sm64, nullptr);
    for (uint32_t g = lo; g < hi; ++g) {
      const uint64_t v = gcnt[g];
      gcnt[g] = run;
      run += v;
    }
    __syncthreads();
    const uint64_t b0 = tile_base[blockIdx.x], b1 = tile_base[n_tiles + blockIdx.x], b2 = tile_base[2 * n_tiles + blockIdx.x];
    for (uint32_t g = tid; g < n_heads; g += kTileThreads) {
      const uint64_t p = gcnt[g];
      op.emit(acc, hpos[g], group_end(g), b0 + (p & 0x1FFFFF), b1 + ((p >> 21) & 0x1FFFFF), b2 + (p >> 42));
    }
  }
  __syncthreads();
  op.end_block();
}

}  // namespace mhx
