// LSD radix sort of fixed-width multi-word records (replaces kmlib::kmsort behind
// SelectSortingFunc, reference src/sorting/kmsort_selector.cpp:39-63 / src/kmlib/kmsort.h:45-122).
//
// Device-wide, stable, 8-bit digits.  Per pass:
//   radix_hist     per-chunk digit histogram (LDS per-wavefront histograms)      reads  n*S*4 B
//   scan           exclusive scan of the digit-major chunk histograms            (small)
//   radix_scatter  per tile: wavefront match-any ranking (ballot) -> LDS-staged
//                  reorder by digit -> coalesced run writes                      reads+writes n*S*4 B
// Records are AoS with a stride of S uint32 words (S even): a record is moved with
// uint4/uint2 accesses, never word by word.  HBM-bound; no MFMA.
#include "dev_prims.h"
#include <cstdlib>
#include <cstring>
#include <string>

#include "mhx_internal.h"
#include "sort_digits.h"
#include "sort_kernels.h"

namespace mhx {

template <int S, int NI>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const uint32_t *__restrict__ items, uint64_t n, DigitSpec ds,
                                                             uint32_t *__restrict__ hist, uint64_t n_chunks,
                                                             const uint8_t *__restrict__ lut) {
  __shared__ uint32_t h[kSortWaves][256];
  for (int i = threadIdx.x; i < kSortWaves * 256; i += kSortThreads) (&h[0][0])[i] = 0;
  __syncthreads();
  const int w = threadIdx.x / kWave;
  const uint64_t base = (uint64_t)blockIdx.x * SortCfg<S, NI>::kChunk;
  for (int j = 0; j < SortCfg<S, NI>::kChunk / kSortThreads; ++j) {
    uint64_t idx = base + (uint64_t)j * kSortThreads + threadIdx.x;
    if (idx < n) {
      const uint32_t *p = items + idx * S;
      unsigned d;
      if (lut) d = lut[p[0] >> 16];  // digit = owner of the item's lv1 bucket
      else {
        d = mem_digit(p, ds.wi1, ds.bit1, ds.mask1);
        if (ds.mask2) d |= mem_digit(p, ds.wi2, ds.bit2, ds.mask2) << ds.sh2;
      }
      atomicAdd(&h[w][d], 1u);
    }
  }
  __syncthreads();
  {
    const int d = threadIdx.x;  // kSortThreads == 256 digits
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kSortWaves; ++i) s += h[i][d];
    hist[(uint64_t)d * n_chunks + blockIdx.x] = s;
  }
}

// Histogram of a pass from the 1-byte-per-record digit side array that the previous pass's scatter wrote
// (same chunking and output layout as k_radix_hist, 1/8 .. 1/16 of its HBM traffic).
template <int CHUNK>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist_bytes(const uint8_t *__restrict__ dig, uint64_t n, uint32_t *__restrict__ hist,
                                                                   uint64_t n_chunks) {
  __shared__ uint32_t h[kSortWaves][256];
  for (int i = threadIdx.x; i < kSortWaves * 256; i += kSortThreads) (&h[0][0])[i] = 0;
  __syncthreads();
  const int w = threadIdx.x / kWave;
  const uint64_t base = (uint64_t)blockIdx.x * CHUNK;
  static_assert(CHUNK % 16 == 0, "chunk shape");
  const uint64_t end = base + CHUNK < n ? base + CHUNK : n;
  for (uint64_t idx = base + (uint64_t)threadIdx.x * 16; idx < end; idx += (uint64_t)kSortThreads * 16) {
    if (idx + 16 <= end) {
      const uint4 v = *reinterpret_cast<const uint4 *>(dig + idx);
      const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        atomicAdd(&h[w][x[q] & 255u], 1u);
        atomicAdd(&h[w][(x[q] >> 8) & 255u], 1u);
        atomicAdd(&h[w][(x[q] >> 16) & 255u], 1u);
        atomicAdd(&h[w][x[q] >> 24], 1u);
      }
    } else {
      for (uint64_t i = idx; i < end; ++i) atomicAdd(&h[w][dig[i]], 1u);
    }
  }
  __syncthreads();
  {
    const int d = threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kSortWaves; ++i) s += h[i][d];
    hist[(uint64_t)d * n_chunks + blockIdx.x] = s;
  }
}

// Ranking inside a wavefront: match-any over the digit bits — stable by construction.
template <int S, int NI>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n,
                                                                DigitSpec ds, int nbits,
                                                                const uint64_t *__restrict__ offs, uint64_t n_chunks,
                                                                const uint8_t *__restrict__ lut, DigitSpec ds_next,
                                                                uint8_t *__restrict__ dnext) {
  using Cfg = SortCfg<S, NI>;
  constexpr int ITEMS = Cfg::kItems;
  __shared__ __attribute__((aligned(16))) uint32_t stage[Cfg::kTile * S];
  __shared__ uint32_t wave_cnt[kSortWaves][256];   // per-wave running digit counters, then wave bases
  __shared__ long long g_off[256];                 // global position minus position in tile, per digit
  __shared__ uint64_t g_base[256];                 // running global base per digit for this chunk
  __shared__ uint32_t sm_scan[kSortThreads / kWave + 1];

  const int tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
  g_base[tid] = offs[(uint64_t)tid * n_chunks + blockIdx.x];
  const uint64_t chunk_base = (uint64_t)blockIdx.x * Cfg::kChunk;
  const uint64_t lanemask_lt = (1ull << lane) - 1;

  // records of the next tile are fetched while the current one is ranked, staged and written
  // (wave-blocked striped arrangement: wave w owns [w*64*ITEMS, (w+1)*64*ITEMS) of a tile)
  Rec<S> nxt[ITEMS];
  auto fetch_tile = [&](int t) {
    const uint64_t tb = chunk_base + (uint64_t)t * Cfg::kTile;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint64_t gi = tb + (uint64_t)(w * (kWave * ITEMS) + j * kWave + lane);
      if (t < Cfg::kTilesPerChunk && gi < n) load_rec<S>(in + gi * S, nxt[j]);
    }
  };
  fetch_tile(0);

  for (int t = 0; t < Cfg::kTilesPerChunk; ++t) {
    const uint64_t tile_base = chunk_base + (uint64_t)t * Cfg::kTile;
    if (tile_base >= n) break;
    const uint64_t rem = n - tile_base;
    const int tile_n = rem < (uint64_t)Cfg::kTile ? (int)rem : Cfg::kTile;
#pragma unroll
    for (int i = 0; i < kSortWaves; ++i) wave_cnt[i][tid] = 0;
    __syncthreads();

    Rec<S> rec[ITEMS];
    uint32_t rank[ITEMS];
    unsigned dig[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) rec[j] = nxt[j];
    fetch_tile(t + 1);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int li = w * (kWave * ITEMS) + j * kWave + lane;
      const bool valid = li < tile_n;
      unsigned d = valid ? (lut ? (unsigned)lut[rec[j].w[0] >> 16] : rec_digit2<S>(rec[j], ds)) : 0u;
      dig[j] = d;
      {
        // match-any over the digit bits
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < (nbits < 0 ? -nbits : nbits); ++b) {
          const bool bitset = (d >> b) & 1u;
          const uint64_t m = __ballot(bitset);
          peers &= bitset ? m : ~m;
        }
        // every lane reads the running counter of its digit, then the lowest peer lane bumps it:
        // LDS operations of one wavefront execute in program order, so the read precedes the write.
        const uint32_t before = wave_cnt[w][d];
        rank[j] = before + __builtin_popcountll(peers & lanemask_lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lanemask_lt) == 0) wave_cnt[w][d] = before + __builtin_popcountll(peers);
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    // thread d: turn per-wave totals into bases inside the tile
    {
      uint32_t c[kSortWaves], tot = 0;
#pragma unroll
      for (int i = 0; i < kSortWaves; ++i) {
        c[i] = wave_cnt[i][tid];
        tot += c[i];
      }
      uint32_t start = block_exclusive_sum<uint32_t, kSortThreads>(tot, sm_scan, nullptr);
      uint32_t run = start;
#pragma unroll
      for (int i = 0; i < kSortWaves; ++i) {
        wave_cnt[i][tid] = run;
        run += c[i];
      }
      g_off[tid] = (long long)g_base[tid] - (long long)start;
      g_base[tid] += tot;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int li = w * (kWave * ITEMS) + j * kWave + lane;
      if (li < tile_n) store_rec<S>(stage + (size_t)(wave_cnt[w][dig[j]] + rank[j]) * S, rec[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int li = j * kSortThreads + tid;
      if (li < tile_n) {
        Rec<S> r;
        load_rec<S>(stage + (size_t)li * S, r);
        const unsigned d = lut ? (unsigned)lut[r.w[0] >> 16] : rec_digit2<S>(r, ds);
        // nbits < 0: timing experiment only (identity placement: same LDS work, perfectly coalesced stores)
        const uint64_t dst = nbits < 0 ? tile_base + li : (uint64_t)(g_off[d] + li);
        store_rec<S>(out + dst * S, r);
        if (dnext) dnext[dst] = (uint8_t)rec_digit2<S>(r, ds_next);  // next pass's digit, read by k_radix_hist_bytes
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Chained-scan ("onesweep") scatter: no per-pass histogram read.  The global digit histograms of ALL passes come
// from one read of the input (k_radix_hist_all; a digit histogram does not depend on the record order), and the
// scatter resolves "how many records with my digit precede my unit" with a decoupled look-back over per-unit
// status words instead of a scanned per-chunk table.  A unit = UT tiles held in registers, so a workgroup can
// publish its digit counts before it knows its own offsets and predecessors never wait for successors.
//   status[unit][digit] = tag<<58 | flag<<56 | value      flag 1: count of this unit, 2: inclusive prefix
// (tag = pass number, so one memset per sort).  Units are handed out by an atomic ticket, which guarantees that every
// predecessor of a running unit is running too.  Stable: units in input order, tiles of a unit in order.
// ---------------------------------------------------------------------------
constexpr int kMaxChainedPasses = 60;  // status tag = pass number + 1 in 6 bits

template <int S>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist_all(const uint32_t *__restrict__ items, uint64_t n, DigitSpecs specs,
                                                                 unsigned long long *__restrict__ ghist) {
  __shared__ uint32_t h[kMaxFusedPasses][256];
  for (int i = threadIdx.x; i < kMaxFusedPasses * 256; i += kSortThreads) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * kSortThreads;
  for (uint64_t idx = (uint64_t)blockIdx.x * kSortThreads + threadIdx.x; idx < n; idx += stride) {
    Rec<S> r;
    load_rec<S>(items + idx * S, r);
    for (int p = 0; p < specs.n; ++p) atomicAdd(&h[p][rec_digit2<S>(r, specs.d[p])], 1u);
  }
  __syncthreads();
  for (int p = 0; p < specs.n; ++p) {
    const uint32_t v = h[p][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)v);
  }
}
// per pass: exclusive scan over the 256 digits -> first output index of each digit
__global__ __launch_bounds__(256) void k_bin_starts(const unsigned long long *__restrict__ ghist, unsigned long long *__restrict__ starts) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const unsigned long long v = ghist[blockIdx.x * 256 + threadIdx.x];
  starts[blockIdx.x * 256 + threadIdx.x] = block_exclusive_sum<uint64_t, 256>((uint64_t)v, sm, nullptr);
}

std::vector<SortPass> make_passes(int key_words, int lo_bit, int hi_bit) {
  (void)key_words;
  std::vector<SortPass> p;
  for (int s = lo_bit; s < hi_bit; s += 8) p.push_back({s, std::min(8, hi_bit - s), 0, 0});
  return p;
}

uint64_t passes_signature(const std::vector<SortPass> &ps) {
  uint64_t h = 1469598103934665603ull;
  for (const SortPass &q : ps)
    for (int v : {q.shift, q.bits, q.shift2, q.bits2}) h = (h ^ (uint64_t)(v + 1)) * 1099511628211ull;
  return h;
}

DigitSpec spec_of_pass(const SortPass &ps, int key_words) {
  DigitSpec ds{key_words - 1 - ps.shift / 32, (unsigned)(ps.shift % 32), (1u << ps.bits) - 1, 0, 0u, 0u, 0u};
  if (ps.bits2) {
    ds.wi2 = key_words - 1 - ps.shift2 / 32;
    ds.bit2 = (unsigned)(ps.shift2 % 32);
    ds.mask2 = (1u << ps.bits2) - 1;
    ds.sh2 = (unsigned)ps.bits;
  }
  // bits [prev_lo, shift) sorted by the earlier passes (SortPass::prev_lo): usable when they lie in the digit's own word
  if (ps.prev_lo >= 0 && ps.prev_lo < ps.shift && !ps.bits2 && ps.prev_lo / 32 == (ps.shift + ps.bits - 1) / 32) {
    const unsigned hi = (unsigned)(ps.shift % 32), lo = (unsigned)(ps.prev_lo % 32);
    ds.prev_mask = ((1u << hi) - 1u) & ~((1u << lo) - 1u);  // bits [lo, hi) of the word
  }
  return ds;
}

// one decision per call about the shape of the sort (no function-local statics: a resident server answers requests with
// different environments, and the decision has to agree with sort_takes_generated_first_pass)
struct SortEnv {
  bool classic;
  std::string shape;  // "" = default
  int items;          // 0 = default
};
static SortEnv sort_env() {
  const char *e = getenv("MHX_SORT"), *sh = getenv("MHX_SORT_SHAPE"), *it = getenv("MHX_SORT_ITEMS");
  return SortEnv{e && !strcmp(e, "classic"), sh ? sh : "", it ? atoi(it) : 0};
}
// a generated first pass that this call does not consume would leave the buffer without records: never sort that
static void refuse_unconsumed_generator(mhx_ctx *c, const void *a, const char *who) {
  if (c->gen_first_pass && c->gen_buf == a) {
    c->gen_first_pass = nullptr;
    throw Error(std::string(who) + ": the records of this buffer are to be made by a generated first pass, which this sort path cannot run");
  }
}

// chained-scan sort (8/12/16-byte records, <= 8 passes); MHX_SORT=classic selects the histogram + scan + scatter passes
template <int S, int NI, int UT>
static uint32_t *radix_sort_onesweep(mhx_ctx *c, uint32_t *a, uint32_t *b, uint64_t n, int key_words, const std::vector<SortPass> &passes) {
  const int P = (int)passes.size();
  // a generated first pass (s1.hip): valid for exactly this buffer, item count and unit shape
  std::function<void(const OnesweepLaunch &)> gen;
  // (gen_slots: item slots the generator walks — more than n when it drops the items of filtered-out lv1 buckets)
  uint64_t gen_slots = n;
  if (c->gen_first_pass && c->gen_buf == (const void *)a && c->gen_n == n && S == 3 && NI == 8 && UT == 3) {
    gen = c->gen_first_pass;
    gen_slots = std::max<uint64_t>(c->gen_slots, n);
  }
  if (!gen) refuse_unconsumed_generator(c, a, "radix sort (chained scan)");
  c->gen_first_pass = nullptr;
  // Per-XCD tickets (see k_radix_onesweep): a unit may wait for a unit whose block id is up to 127 higher, so the scheme needs
  // the whole 8-XCD part with a couple of hundred workgroups resident at once (MI355X in SPX mode: 256 CUs x 3-4 workgroups).
  // On a partition (CPX: 32 CUs, one XCD) or an unknown device the single ticket counter is used: its look-back only ever
  // waits for lower tickets, which are running by construction.
  const int xcd_units = c->opt("sort_xcd_units", 1) != 0 && c->n_cus >= 192;
  const uint64_t unit = (uint64_t)kSortThreads * NI * UT;
  auto units_of = [&](uint64_t items) { return xcd_units ? (div_ceil(items, unit) + 127) / 128 * 128 : div_ceil(items, unit); };
  const uint64_t n_units = units_of(n), n_units_gen = units_of(gen_slots);
  hipStream_t st = c->stream;
  // sort_loaded_ut2 (12-byte records): the passes that LOAD their records take units of two tiles instead of three — 100 registers
  // instead of 140, four workgroups per CU instead of three (the SQ counters of round 6 show the waves of these passes parked at
  // waits and barriers for two thirds of their cycles: latency, not issue) — at the price of shorter unit-wide runs
  constexpr bool kCanUt2 = S == 3 && NI == 8 && UT == 3;
  const bool loaded_ut2 = kCanUt2 && c->opt("sort_loaded_ut2", 0) != 0;
  const uint64_t unit2 = (uint64_t)kSortThreads * NI * 2;
  const uint64_t n_units2 = xcd_units ? (div_ceil(n, unit2) + 127) / 128 * 128 : div_ceil(n, unit2);
  const uint64_t n_status = std::max(n_units_gen, loaded_ut2 ? n_units2 : 0);
  unsigned long long *status = c->ws("sort_status", n_status * 256 * 8).as<unsigned long long>();
  unsigned long long *gh = c->ws("sort_ghist", (size_t)2 * kMaxChainedPasses * 256 * 8).as<unsigned long long>();
  unsigned long long *starts = gh + kMaxChainedPasses * 256;
  constexpr int kErrSlot = kMaxChainedPasses * 8;
  uint32_t *tickets = c->ws("sort_tickets", (kErrSlot + 8) * 4).as<uint32_t>();  // 8 ticket counters per pass, then the error flag
  MHX_HIP(hipMemsetAsync(status, 0, n_status * 256 * 8, st));
  MHX_HIP(hipMemsetAsync(gh, 0, (size_t)kMaxChainedPasses * 256 * 8, st));
  MHX_HIP(hipMemsetAsync(tickets, 0, (kErrSlot + 8) * 4, st));
  const double bytes = (double)n * S * 4;
  static const std::string nm_hist = "radix_hist_all_" + std::to_string(S * 4) + "B", nm_scat = "radix_scatter_" + std::to_string(S * 4) + "B";
  // (>= 16 records per thread: every workgroup ends with 256 global atomics per pass, which is all a small input would do)
  const unsigned hgrid = (unsigned)std::min<uint64_t>(div_ceil(n, (uint64_t)kSortThreads * 16), 4096);
  std::vector<DigitSpec> all(P);
  for (int p = 0; p < P; ++p) all[p] = spec_of_pass(passes[p], key_words);
  // extraction may have taken the digit histograms while it produced the records (s1.hip): then no read at all
  const bool pre = c->pre_hist_buf == (const void *)a && c->pre_hist_n == n && c->pre_hist_passes == P && P <= kMaxFusedPasses &&
                   c->pre_hist_sig == passes_signature(passes);
  c->pre_hist_buf = nullptr;
  if (gen && !pre) throw Error("radix sort: a generated first pass needs the digit histograms of the plan (pre-hist)");
  if (pre) MHX_HIP(hipMemcpyAsync(gh, c->work["sort_pre_hist"].p, (size_t)P * 256 * 8, hipMemcpyDeviceToDevice, st));
  for (int p0 = 0; p0 < P && !pre; p0 += kMaxFusedPasses) {  // one read of the input per 16 passes
    DigitSpecs specs;
    specs.n = std::min(kMaxFusedPasses, P - p0);
    for (int p = 0; p < specs.n; ++p) specs.d[p] = all[p0 + p];
    MHX_LAUNCH(c, nm_hist.c_str(), bytes,
               hipLaunchKernelGGL((k_radix_hist_all<S>), dim3(hgrid), dim3(kSortThreads), 0, st, a, n, specs, gh + (size_t)p0 * 256));
  }
  hipLaunchKernelGGL(k_bin_starts, dim3(P), dim3(256), 0, st, gh, starts);
  // unit-wide runs (k_radix_onesweep_u) for the default unit shape of every width; MHX_SORT_UNIT_RUNS=0: the tile-by-tile kernel
  constexpr bool kHasUnitRuns = (S <= 3 && NI == 8 && (UT == 3 || UT == 2)) || (S == 4 && NI == 8 && UT == 2) || (S > 4 && NI == 4 && UT == 2);
  const bool unit_runs = kHasUnitRuns && c->opt("sort_unit_runs", 1) != 0;
  // sort_rank_uniform: passes whose plan declares the bits sorted before them (SortPass::prev_lo) rank with one LDS atomic per
  // record wherever all records of a wavefront instruction agree on those bits, with the ballots elsewhere (RANK 2 of
  // k_radix_onesweep_u: correct whatever order the LDS applies the lanes in).  Every other loading pass: the ballots.
  const bool rank_uniform = unit_runs && c->opt("sort_rank_uniform", 1) != 0;
  for (int p = 0; p < P; ++p) {
    const int nb = passes[p].bits + passes[p].bits2;
    const int wi = digit_word_of(all[p], nb, 1);
    if (p == 0 && gen) {  // the records of the first pass are made on the fly (no input array): the generator's owner launches
      static const std::string nm_gen = nm_scat + "_gen";
      MHX_LAUNCH(c, nm_gen.c_str(), bytes,
                 gen(OnesweepLaunch{(unsigned)n_units_gen, st, b, gen_slots, all[p], nb, starts + p * 256, status, tickets + p * 8,
                                    tickets + kErrSlot, (unsigned long long)(p + 1), xcd_units, unit_runs ? 1 : 0, wi}));
    } else if (unit_runs) {
      if constexpr (kHasUnitRuns) {
#define MHX_U(RANKV, WIV)                                                                                                                      \
  hipLaunchKernelGGL((k_radix_onesweep_u<S, NI, UT, SrcArray<S>, RANKV, WIV>), dim3((unsigned)n_units), dim3(kSortThreads), 0, st, SrcArray<S>{a}, b, \
                     n, all[p], nb, starts + p * 256, status, tickets + p * 8, tickets + kErrSlot, (unsigned long long)(p + 1), xcd_units)
#define MHX_U2(RANKV, WIV)                                                                                                                     \
  hipLaunchKernelGGL((k_radix_onesweep_u<S, NI, 2, SrcArray<S>, RANKV, WIV>), dim3((unsigned)n_units2), dim3(kSortThreads), 0, st, SrcArray<S>{a}, b, \
                     n, all[p], nb, starts + p * 256, status, tickets + p * 8, tickets + kErrSlot, (unsigned long long)(p + 1), xcd_units)
        if constexpr (kCanUt2) {
          if (loaded_ut2 && wi == 0) {
            MHX_LAUNCH(c, nm_scat.c_str(), 2 * bytes, {
              if (rank_uniform && all[p].prev_mask) MHX_U2(2, 0);
              else MHX_U2(0, 0);
            });
            std::swap(a, b);
            continue;
          }
        }
#undef MHX_U2
        MHX_LAUNCH(c, nm_scat.c_str(), 2 * bytes, {
          if (wi == 0 && rank_uniform && all[p].prev_mask) MHX_U(2, 0);
          else if (wi == 0) MHX_U(0, 0);
          else if (wi == 1) MHX_U(0, 1);
          else MHX_U(0, -1);
        });
#undef MHX_U
      }
    } else {
      MHX_LAUNCH(c, nm_scat.c_str(), 2 * bytes,
                 hipLaunchKernelGGL((k_radix_onesweep<S, NI, UT, SrcArray<S>>), dim3((unsigned)n_units), dim3(kSortThreads), 0, st, SrcArray<S>{a}, b, n,
                                    all[p], nb, starts + p * 256, status, tickets + p * 8, tickets + kErrSlot,
                                    (unsigned long long)(p + 1), xcd_units));
    }
    std::swap(a, b);
  }
  uint32_t e = 0;
  MHX_HIP(hipMemcpyAsync(&e, tickets + kErrSlot, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (e) throw Error("radix sort: chained scan timed out waiting for a predecessor unit (set MHX_SORT=classic)");
  return a;
}

template <int S, int NI>
static uint32_t *radix_sort_impl2(mhx_ctx *c, uint32_t *a, uint32_t *b, uint64_t n, int key_words,
                                  const std::vector<SortPass> &passes) {
  if (n == 0) {
    // nothing to sort: a generated first pass armed for this buffer (a bucket-range pass or a rank that keeps no record)
    // is spent with it — left armed it would meet the next sort of the same buffer
    if (c->gen_first_pass && c->gen_buf == (const void *)a) c->gen_first_pass = nullptr;
    return a;
  }
  const SortEnv env = sort_env();
  if constexpr (S <= 8 && NI == default_items<S>()) {
    const bool classic = env.classic;
    if (!classic && passes.size() <= (size_t)kMaxChainedPasses && div_ceil(n, (uint64_t)kSortThreads * 8) < (1ull << 31)) {
      if constexpr (S <= 4) {
        // unit shape: records per thread per tile x tiles per unit (MHX_SORT_SHAPE overrides).  Measured at 12 B, 1.33 G
        // records, ms per 6 passes on one box: 4x2 89, 4x4 66, 4x8 80, 8x1 88, 6x2 72, 12x1 75, 16x1 78, 8x2 59, 8x3 57, 8x4 70
        // (classic 3-kernel passes: 63 + 20 histogram): 2048-record tiles halve the number of scattered runs, units of
        // 4-6 K records amortise the look-back, more registers cost occupancy.
        const std::string shape = !env.shape.empty() ? env.shape : (S <= 3 ? "8x3" : "8x2");
        if (shape == "8x2") return radix_sort_onesweep<S, 8, 2>(c, a, b, n, key_words, passes);
        if (shape == "4x4") return radix_sort_onesweep<S, 4, 4>(c, a, b, n, key_words, passes);
        if (shape == "16x1") return radix_sort_onesweep<S, 16, 1>(c, a, b, n, key_words, passes);
        if (shape == "8x4" && S <= 3) return radix_sort_onesweep<S, 8, 4>(c, a, b, n, key_words, passes);
        if (shape == "8x3" || S <= 3) return radix_sort_onesweep<S, 8, 3>(c, a, b, n, key_words, passes);
        return radix_sort_onesweep<S, 8, 2>(c, a, b, n, key_words, passes);
      } else {
        return radix_sort_onesweep<S, 4, 2>(c, a, b, n, key_words, passes);  // 24/32-byte records: 8 per thread in registers
      }
    }
  }
  refuse_unconsumed_generator(c, a, "radix sort (classic passes)");
  const uint64_t n_chunks = div_ceil(n, SortCfg<S, NI>::kChunk);
  uint32_t *hist = c->ws("sort_hist", n_chunks * 256 * 4).as<uint32_t>();
  uint64_t *offs = c->ws("sort_offs", n_chunks * 256 * 8).as<uint64_t>();
  const double bytes = (double)n * S * 4;
  static const std::string nm_hist = "radix_hist_" + std::to_string(S * 4) + "B", nm_scat = "radix_scatter_" + std::to_string(S * 4) + "B";
  auto spec_of = [&](const SortPass &ps) {
    DigitSpec ds{key_words - 1 - ps.shift / 32, (unsigned)(ps.shift % 32), (1u << ps.bits) - 1, 0, 0u, 0u, 0u};
    if (ps.bits2) {
      ds.wi2 = key_words - 1 - ps.shift2 / 32;
      ds.bit2 = (unsigned)(ps.shift2 % 32);
      ds.mask2 = (1u << ps.bits2) - 1;
      ds.sh2 = (unsigned)ps.bits;
    }
    return ds;
  };
  // MHX_SORT_SIDE_DIGITS=1: scatter also writes the next pass's digit per record (1 B) so that the next histogram reads
  // bytes instead of records.  Measured on MI355X: histograms 52 -> 17 ms/step but scatter +33 ms/step: off by default.
  // (experiments of rounds 1-2, compiled only into the timing build — libmhx.so reads none of them: make timing)
#ifdef MHX_TILE_TIMING
  const bool use_side = getenv("MHX_SORT_SIDE_DIGITS") != nullptr;
#else
  constexpr bool use_side = false;
#endif
  uint8_t *dnext = use_side && passes.size() > 1 ? c->ws("sort_digits", n + 64).as<uint8_t>() : nullptr;
  static const std::string nm_histb = "radix_hist_bytes";
  for (size_t pi = 0; pi < passes.size(); ++pi) {
    const SortPass &ps = passes[pi];
    const DigitSpec ds = spec_of(ps);
    const bool last = pi + 1 == passes.size();
    const DigitSpec ds_next = last ? ds : spec_of(passes[pi + 1]);
#ifdef MHX_TILE_TIMING
    const bool dbg_identity = getenv("MHX_DEBUG_IDENTITY_SCATTER") != nullptr;  // WRONG RESULTS: timing experiment
#else
    constexpr bool dbg_identity = false;
#endif
    const int nbits = dbg_identity ? -(ps.bits + ps.bits2) : ps.bits + ps.bits2;
    if (pi == 0 || !dnext)
      MHX_LAUNCH(c, nm_hist.c_str(), bytes,
                 hipLaunchKernelGGL((k_radix_hist<S, NI>), dim3((unsigned)n_chunks), dim3(kSortThreads), 0, c->stream, a, n, ds, hist,
                                    n_chunks, (const uint8_t *)nullptr));
    else
      MHX_LAUNCH(c, nm_histb.c_str(), (double)n,
                 hipLaunchKernelGGL((k_radix_hist_bytes<SortCfg<S, NI>::kChunk>), dim3((unsigned)n_chunks), dim3(kSortThreads), 0,
                                    c->stream, dnext, n, hist, n_chunks));
    exclusive_scan_u32_u64(c, hist, offs, n_chunks * 256, nullptr);
    uint8_t *dn = last ? nullptr : dnext;
#ifdef MHX_TILE_TIMING
    const unsigned lds_pad = getenv("MHX_SORT_LDS_PAD") ? (unsigned)atoi(getenv("MHX_SORT_LDS_PAD")) : 0u;  // occupancy experiment
#else
    constexpr unsigned lds_pad = 0u;
#endif
    MHX_LAUNCH(c, nm_scat.c_str(), 2 * bytes + (dn ? (double)n : 0.0),
               hipLaunchKernelGGL((k_radix_scatter<S, NI>), dim3((unsigned)n_chunks), dim3(kSortThreads), lds_pad, c->stream, a, b, n, ds,
                                  nbits, offs, n_chunks, (const uint8_t *)nullptr, ds_next, dn));
    std::swap(a, b);
  }
  return a;
}

// tile-shape selection: MHX_SORT_ITEMS=8|16 overrides the default for 8- and 16-byte records (tuning knob)
template <int S>
static uint32_t *radix_sort_impl(mhx_ctx *c, uint32_t *a, uint32_t *b, uint64_t n, int key_words,
                                 const std::vector<SortPass> &passes) {
  if constexpr (S <= 4) {
    const SortEnv env = sort_env();
    const int items = env.items ? env.items : default_items<S>();
    if (items == 2) return radix_sort_impl2<S, 2>(c, a, b, n, key_words, passes);
    if (items == 3) return radix_sort_impl2<S, 3>(c, a, b, n, key_words, passes);
    if (items == 6) return radix_sort_impl2<S, 6>(c, a, b, n, key_words, passes);
    if (items == 16) return radix_sort_impl2<S, 16>(c, a, b, n, key_words, passes);
    if (items == 8) return radix_sort_impl2<S, 8>(c, a, b, n, key_words, passes);
    if (items == 12) return radix_sort_impl2<S, 12>(c, a, b, n, key_words, passes);
  }
  return radix_sort_impl2<S, default_items<S>()>(c, a, b, n, key_words, passes);
}

// ---------------------------------------------------------------------------
// Prefix sort + segment finish (kmlib::kmsort's job on wide keys without one pass per key byte; the reference's
// src/sorting/kmsort_selector.cpp:39-63 treats every width alike: n_bytes = 4*NWords - 2 radix levels).
// The LSD passes above cost 2 x record bytes of HBM traffic per 8 key bits: 33 passes over 40-byte records at k = 119.
// But after passes over only the top `pbits` bits of key word 0, every maximal stretch of equal prefix — a SEGMENT —
// already sits at its final place as a whole, and holds a handful of records when 2^pbits ~ n/10.  One more kernel
// finishes the job: one thread per record walks outwards from its own place until the prefix changes and counts the
// records of its segment that precede it in (whole key, input order) — its rank — then writes the record to
// segment start + rank.  Stable.  The neighbours' key words come out of the caches (the threads of a wavefront read
// adjacent records), no LDS tile, no barriers, ~24 registers: the latency of the walk hides behind dozens of resident
// wavefronts.  (A first version staged tiles in LDS and ranked there: 2 workgroups per CU, one dependent LDS chain per
// record — 9-15 ms for the 117 M 8-byte records that three LSD passes move in 2 ms.)
// A walk longer than kSegWalkLimit raises *err and the host sorts every bit with LSD passes instead (same result).
// ---------------------------------------------------------------------------
constexpr int kSegWalkLimit = 4096;

template <int S>
__global__ __launch_bounds__(256) void k_seg_finish(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n, int kw,
                                                    uint32_t pfx_mask, uint32_t *__restrict__ err) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t gi = (uint64_t)blockIdx.x * 256 + threadIdx.x; gi < n; gi += stride) {
    Rec<S> me;
    load_rec<S>(in + gi * S, me);
    const uint32_t r0 = me.w[0];
    // -1: other segment, 0: key_x < mine, 1: equal, 2: greater
    auto cmp = [&](uint64_t x) -> int {
      const uint32_t *p = in + x * S;
      uint32_t v = p[0];
      if ((v ^ r0) & pfx_mask) return -1;
      if (v != r0) return v < r0 ? 0 : 2;
#pragma unroll
      for (int w = 1; w < S; ++w) {
        if (w >= kw) break;
        v = p[w];
        if (v != me.w[w]) return v < me.w[w] ? 0 : 2;
      }
      return 1;
    };
    uint32_t before = 0;
    uint64_t x = gi;
    bool over = false;
    while (x > 0) {
      const int c = cmp(x - 1);
      if (c < 0) break;
      before += c <= 1;
      --x;
      if (gi - x > (uint64_t)kSegWalkLimit) {
        over = true;
        break;
      }
    }
    uint64_t y = gi + 1;
    while (!over && y < n) {
      const int c = cmp(y);
      if (c < 0) break;
      before += c == 0;
      ++y;
      if (y - gi > (uint64_t)kSegWalkLimit) over = true;
    }
    if (over) {
      atomicOr(err, 1u);
      continue;
    }
    store_rec<S>(out + (x + before) * S, me);
  }
}

template <int S>
static bool seg_finish_impl(mhx_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n, int key_words, int pbits) {
  uint32_t *err = c->ws("seg_finish_err", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(err, 0, 4, c->stream));
  const uint32_t pfx_mask = pbits >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> pbits);
  static const std::string nm = "seg_finish_" + std::to_string(S * 4) + "B";
  MHX_LAUNCH(c, nm.c_str(), 2.0 * (double)n * S * 4,
             hipLaunchKernelGGL((k_seg_finish<S>), dim3((unsigned)std::min<uint64_t>(div_ceil(n, 256), 256 * 64)), dim3(256), 0, c->stream, in, out, n,
                                key_words, pfx_mask, err));
  uint32_t e = 0;
  MHX_HIP(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  return e == 0;
}
static bool seg_finish(mhx_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n, int stride, int key_words, int pbits) {
  switch (stride) {
#define MHX_CASE(SV) \
  case SV: return seg_finish_impl<SV>(c, in, out, n, key_words, pbits);
    MHX_CASE(2) MHX_CASE(3) MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
#undef MHX_CASE
    default: throw Error("seg_finish: unsupported record stride");
  }
}

// Sort by the whole key (words [0, key_words) as one big-endian number), stably: LSD passes over the top bits only +
// k_seg_finish when that saves passes, else (or when a segment outgrows the look-ahead) the plain LSD plan `passes`.
// The caller vouches that `passes` orders the records exactly as the whole key does (bits it leaves out are constant or
// may order equal keys arbitrarily).
uint32_t *sort_whole_key(mhx_ctx *c, uint32_t *a, uint32_t *b, uint64_t n, int stride, int key_words, const std::vector<SortPass> &passes) {
  const int mode = (int)c->opt("sort_hybrid", 1);  // 0: never, 1: when the cost model says so, 2: always (tests)
  const uint64_t min_n = (uint64_t)c->opt("sort_hybrid_min", 1 << 16);
  if (!mode || n < min_n || n >= (1ull << 40)) return radix_sort(c, a, b, n, stride, key_words, passes);
  // Cost model, picoseconds per record on MI355X (measured at 0.8-1.2 x 10^8 records, round 3, profiles/r03_bench_klist.json):
  // an LSD pass moves the record twice, ~3 ps per 32-bit word (0.71 ms for 117 M 8-byte records, 1.9 ms for 82 M 40-byte
  // ones); the finish kernel moves it once and reads a couple of key words of every neighbour in its segment:
  // ~(14 + 4.5 S) ps at one record per segment plus ~(2 + 0.8 S) ps per further record of the segment (S = words per record:
  // 2.2 ms for 117 M 8-byte singletons, 3.8 ms for 117 M 16-byte ones, 7.9 ms for 111 M 24-byte records at ~7 per segment).
  // So: never for 8-byte records with 6 passes (stage 2 at k = 21), 32 prefix bits for the wide keys of seq2sdbg at k >= 29.
  const double pass_ps = 3.0 * stride;
  auto finish_ps = [&](int pbits) {
    const double avg = pbits >= 40 ? 0.0 : (double)n / (double)(1ull << pbits);  // records per prefix value if the keys were uniform
    return (14.0 + 4.5 * stride) + std::max(0.0, avg - 1.0) * (2.0 + 0.8 * stride);
  };
  int best_bits = 0;
  double best = (double)passes.size() * pass_ps;
  for (int pbits = 8; pbits <= 32; pbits += 8) {
    const double t = (pbits / 8) * pass_ps + finish_ps(pbits) + (double)c->opt("sort_hybrid_margin_ps", 6);
    if ((double)n / (double)(1ull << pbits) <= 64.0 && t < best) {  // (segments of hundreds of records: the walk is quadratic)
      best = t;
      best_bits = pbits;
    }
  }
  if (mode == 2 && !best_bits) {  // forced: the width the model would like best
    best_bits = 8;
    while (best_bits < 32 && (double)n / 4.0 > (double)(1ull << best_bits)) best_bits += 8;
  }
  if (const long long f = c->opt("sort_hybrid_bits", 0)) best_bits = (int)std::min<long long>(32, std::max<long long>(1, f));
  if (!best_bits) return radix_sort(c, a, b, n, stride, key_words, passes);
  uint32_t *sorted = radix_sort(c, a, b, n, stride, key_words, make_passes(key_words, key_words * 32 - best_bits, key_words * 32));
  uint32_t *other = sorted == a ? b : a;
  if (seg_finish(c, sorted, other, n, stride, key_words, best_bits)) return other;
  return radix_sort(c, sorted, other, n, stride, key_words, passes);  // a segment beyond the walk limit: every bit by LSD passes
}

// One stable multisplit pass: digit = owner_lut[item.w[0] >> 16].  Items of owner p end up contiguous
// in `b`, owners ascending; counts[p] receives their numbers.
template <int S>
static void partition_impl(mhx_ctx *c, const uint32_t *a, uint32_t *b, uint64_t n, const uint8_t *lut, int n_parts, uint64_t *counts) {
  for (int p = 0; p < n_parts; ++p) counts[p] = 0;
  if (n == 0) return;
  const uint64_t n_chunks = div_ceil(n, SortCfg<S, default_items<S>()>::kChunk);
  uint32_t *hist = c->ws("sort_hist", n_chunks * 256 * 4).as<uint32_t>();
  uint64_t *offs = c->ws("sort_offs", (n_chunks * 256 + 1) * 8).as<uint64_t>();
  const double bytes = (double)n * S * 4;
  int nbits = 1;
  while ((1 << nbits) < n_parts) ++nbits;
  MHX_LAUNCH(c, "owner_hist", bytes,
             hipLaunchKernelGGL((k_radix_hist<S, default_items<S>()>), dim3((unsigned)n_chunks), dim3(kSortThreads), 0, c->stream, a, n, DigitSpec{0, 0u, 0xFFu, 0, 0u, 0u, 0u}, hist,
                                n_chunks, lut));
  exclusive_scan_u32_u64(c, hist, offs, n_chunks * 256, offs + n_chunks * 256);
  MHX_LAUNCH(c, "owner_scatter", 2 * bytes,
             hipLaunchKernelGGL((k_radix_scatter<S, default_items<S>()>), dim3((unsigned)n_chunks), dim3(kSortThreads), 0, c->stream, a, b, n, DigitSpec{0, 0u, 0xFFu, 0, 0u, 0u, 0u},
                                nbits, offs, n_chunks, lut, DigitSpec{0, 0u, 0xFFu, 0, 0u, 0u, 0u}, (uint8_t *)nullptr));
  std::vector<uint64_t> starts(n_parts + 1);
  for (int p = 0; p <= n_parts; ++p)
    MHX_HIP(hipMemcpyAsync(&starts[p], offs + (uint64_t)p * n_chunks, 8, hipMemcpyDeviceToHost, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  for (int p = 0; p < n_parts; ++p) counts[p] = starts[p + 1] - starts[p];
}

void partition_by_owner(mhx_ctx *c, const uint32_t *a, uint32_t *b, uint64_t n, int stride, const uint8_t *lut, int n_parts,
                        uint64_t *counts) {
  if (n_parts > 256) throw Error("partition_by_owner: at most 256 owners");
  switch (stride) {
    case 2: return partition_impl<2>(c, a, b, n, lut, n_parts, counts);
    case 3: return partition_impl<3>(c, a, b, n, lut, n_parts, counts);
    case 4: return partition_impl<4>(c, a, b, n, lut, n_parts, counts);
    case 6: return partition_impl<6>(c, a, b, n, lut, n_parts, counts);
    case 8: return partition_impl<8>(c, a, b, n, lut, n_parts, counts);
    case 10: return partition_impl<10>(c, a, b, n, lut, n_parts, counts);
    case 12: return partition_impl<12>(c, a, b, n, lut, n_parts, counts);
    case 14: return partition_impl<14>(c, a, b, n, lut, n_parts, counts);
    case 16: return partition_impl<16>(c, a, b, n, lut, n_parts, counts);
    case 18: return partition_impl<18>(c, a, b, n, lut, n_parts, counts);
    case 20: return partition_impl<20>(c, a, b, n, lut, n_parts, counts);
    default: throw Error("partition_by_owner: unsupported record stride");
  }
}

// true when radix_sort(stride 3, these passes) runs the chained-scan passes with the default 8x3 unit shape, i.e. when it
// will take a generated first pass (mhx_ctx::gen_first_pass)
bool sort_takes_generated_first_pass(const mhx_ctx *c, uint64_t n, int stride, const std::vector<SortPass> &passes) {
  (void)c;
  const SortEnv env = sort_env();  // (the same per-call decision radix_sort_impl2 takes)
  if (env.classic) return false;
  if ((!env.shape.empty() && env.shape != "8x3") || (env.items && env.items != default_items<3>())) return false;
  return stride == 3 && n > 0 && passes.size() <= (size_t)kMaxChainedPasses && passes.size() <= (size_t)kMaxFusedPasses &&
         div_ceil(n, (uint64_t)kSortThreads * 8) < (1ull << 31);
}

uint32_t *radix_sort(mhx_ctx *c, uint32_t *a, uint32_t *b, uint64_t n, int stride, int key_words,
                     const std::vector<SortPass> &passes) {
  switch (stride) {
    case 2: return radix_sort_impl<2>(c, a, b, n, key_words, passes);
    case 3: return radix_sort_impl<3>(c, a, b, n, key_words, passes);
    case 4: return radix_sort_impl<4>(c, a, b, n, key_words, passes);
    case 6: return radix_sort_impl<6>(c, a, b, n, key_words, passes);
    case 8: return radix_sort_impl<8>(c, a, b, n, key_words, passes);
    case 10: return radix_sort_impl<10>(c, a, b, n, key_words, passes);
    case 12: return radix_sort_impl<12>(c, a, b, n, key_words, passes);
    case 14: return radix_sort_impl<14>(c, a, b, n, key_words, passes);
    case 16: return radix_sort_impl<16>(c, a, b, n, key_words, passes);
    case 18: return radix_sort_impl<18>(c, a, b, n, key_words, passes);
    case 20: return radix_sort_impl<20>(c, a, b, n, key_words, passes);
    default: throw Error("radix_sort: unsupported record stride");
  }
}

}  // namespace mhx
