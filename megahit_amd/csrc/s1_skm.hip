// read2sdbg stage 1 — and `count` — on super-k-mer records (round 6).  The no-mercy reduction of Read2SdbgS1 (reference
// src/sorting/read_to_sdbg_s1.cpp:208-366 makes one sort item per (k-1)-mer window of every read, :368-464 counts the groups of equal items)
// needs the (k+1)-mers GROUPED, not sorted: is_solid, the multiplicity histogram and the aggregated stage-2 items are functions of the
// multiset of canonical (k+1)-mers; the same holds for KmerCounter (src/sorting/kmer_counter.cpp:158-381), whose edges are ordered afterwards.
// So the grouping key need not be a prefix of the k-mer.  Here it is the MINIMIZER of the (k+1)-mer — the smallest hash of a canonical
// m-mer inside it — which consecutive windows of a read share: a run of windows with one minimizer (a "super-k-mer") leaves the read as
// ONE 16-byte record (its bases, where it starts, how many windows it holds) instead of one 12-byte record per window.
//   k_skm_make    one thread per aligned block of 8 windows: 17 canonical m-mer hashes from one 64-bit window of the store and its
//                 reverse complement, the 8 window minima by a suffix / prefix scan, a record per run of equal bins (3.5 windows per
//                 record at k = 21) -> the record array through one cursor (one atomic per workgroup and trip); the digit histograms of the
//                 sort passes on the way; windows of ONE base (poly-A, poly-G) counted beside the records; COUNT: a base more either side
//   radix_sort    two passes over the 16-bit bin (the chained-scan passes of sort.hip on 16-byte records; a third beyond 2^16 bins)
//   k_skm_bounds  where each bin starts
//   k_s1_skm      one workgroup per bin at a time: the windows of a wavefront's 64 records (~225) are dealt to the lanes 64 at a time (a bitmap
//                 of run heads, six shuffles per window), canonical key (read_to_sdbg_s1.cpp:228-292: the strand of the (k-1)-mer, then
//                 head / tail), LDS table with 64-bit keys, one walk over the table for the histogram (:430-436), the marks of the
//                 non-solid occurrences (:464, inverted: a key of count 1 < m has one record, whose position sits next to the key) and the
//                 aggregated stage-2 items; several sources per bin (several GPUs: one per sending rank) and the marks as a list there
//   k_count_skm   the same for `count`: in / out characters in the slot, packed edges out, the windows of flagged keys in a second expansion
// 6.0 GB of records at 10 M reads instead of 16: the two sort passes and the read of the group-by move 2.7 x fewer bytes.
// Shapes: no mercy, no lv1 bucket filter (the path cuts large jobs into passes over ranges of its OWN bins), reads of one length or of
// several, min count <= 2, 19 <= k <= 22 (count: 21), positions below 2^36; everything else — and any input with a bin that outgrows what
// one workgroup should stream (repeats other than homopolymers) — takes the prefix plan (s1_stream.hip).  DESIGN.md 4o.
#include <cstring>

#include "s1_shared.h"

namespace mhx {

constexpr int kSkmC = 8;                     // windows per aligned block = the longest run a record holds
constexpr int kSkmW = 10;                    // m-mers per window: m = k + 1 - 9
constexpr int kSkmNM = kSkmC + kSkmW - 1;    // m-mers a block looks at
constexpr int kSkmMinBinBits = 16, kSkmMaxBinBits = 20;  // two sort passes, or three

__device__ __forceinline__ uint32_t skm_mix(uint32_t c) {  // (a bijection of 32 bits: the order of the m-mers)
  uint32_t h = c * 0x9E3779B1u;
  return h ^ (h >> 15);
}
__device__ __forceinline__ uint32_t skm_mix2(uint32_t c) {  // (the second hash of a table key: which half of a bin that overflowed it belongs to)
  uint32_t h = c * 0x9E3779B1u;
  h ^= h >> 15;
  return h * 0x85EBCA6Bu;
}
__device__ __forceinline__ uint32_t skm_bin_of(uint32_t minh) {  // (the minimum of ten hashes is small: mixed once more; the bin = its top bits)
  return (minh ^ (minh >> 16)) * 0x7FEB352Du;
}

// record (16 bytes): w0 = position bits 32.. << 28 | bin << 8 (bits 0..7 zero: the sort passes may rank with LDS atomics, sort_kernels.h
// RANK 2), w1:w2 = the bases of the run, MSB first (k + windows of them: at most 30 at k <= 22) with windows - 1 in the three lowest bits,
// w3 = position of the run's first base in the store, low 32 bits.
// VAR: reads of several lengths — every read gets bpr = ceil((max_len - k) / 8) blocks, its place in the store comes from start[]
// (as the padded item slots of S1GenVarT); a block beyond the read's last window makes nothing.
// COUNT: the records of `count` (KmerCounter, kmer_counter.cpp:158-381) — a window's key is the canonical (k+1)-mer and the table wants the base
// in front of it and the base behind it: the record carries one more base either side of its run (k + windows + 2 <= 31 bases at k <= 21),
// two flags for a run that starts at the first / ends at the last window of its read ('$' there), and its length in the first word:
// w0 = position bits 32.. << 28 | bin << 8 | (windows - 1) << 5 | no_prev << 4 | no_next << 3.
template <int NT, int J, bool VAR, bool COUNT = false>
__global__ __launch_bounds__(NT) void k_skm_make(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint32_t L, uint32_t bpr, uint64_t n_blocks,
                                                 int k, int bin_bits, uint32_t bin_lo, uint32_t bin_hi, int count_items, uint64_t pos_base,
                                                 unsigned long long *__restrict__ hp, uint4 *__restrict__ out, unsigned long long cap,
                                                 unsigned long long *__restrict__ cursor, uint32_t *__restrict__ err, unsigned long long *__restrict__ digit_hist) {
  __shared__ uint32_t sm_scan[NT / kWave + 1];
  __shared__ unsigned long long s_base;
  __shared__ uint32_t dh[3][256];  // the digit histograms of the sort passes, taken while the records are made
  __shared__ uint32_t lbin[J * kSkmC][NT];  // a thread's bins, read back by window number when its records leave (no register array indexed by a variable)
  __shared__ uint32_t hpl[2][12];  // COUNT: the homopolymer windows of this workgroup per class: [0] windows, [1 + x] base x in front, [5 + x] base x behind
  const int tid = threadIdx.x;
  if (tid < 24) hpl[tid / 12][tid % 12] = 0;
  for (int i = tid; i < 768; i += NT) dh[i >> 8][i & 255] = 0;
  __syncthreads();
  const bool third_digit = bin_bits > 16;
  const int M = k + 1 - (kSkmW - 1);
  const uint32_t mmask = (1u << (2 * M)) - 1u;
  const int K1 = k + 1;
  const unsigned bin_sh = 32u - (unsigned)bin_bits;
  unsigned long long items = 0;  // VAR: what the reference sorts, L - k + 4 items per read that holds an edge (read_to_sdbg_s1.cpp:344-363)
  // HOMOPOLYMER windows — (k+1)-mers of one base: poly-A tails, the poly-G of a two-colour instrument's dark cycles — share ONE key by the
  // million and would all sit behind one minimizer in one bin.  They never enter a record: a window of one base is counted here (hp != null,
  // first pass only: hp[0] / hp[1] = windows of A or T / of C or G, hp[2] / hp[3] = the smallest position of one, should it be the only one)
  // and published as the one or two keys they are by k_skm_hp_publish.
  uint32_t hp_n[2] = {0, 0};
  unsigned long long hp_p[2] = {~0ull, ~0ull};
  for (uint64_t it = blockIdx.x; it * (uint64_t)(NT * J) < n_blocks; it += gridDim.x) {
    uint64_t Wv[J], pos0[J];
    uint32_t smask[J], emask[J];  // per block: the windows that start a run of this pass's bins, and those that end one
    uint32_t bflag[J];            // COUNT: bit 0 the block's first window is its read's first, bit 1 its last window is the read's last, bits 2.. its windows
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint64_t b = it * (uint64_t)(NT * J) + (uint64_t)j * NT + tid;
      smask[j] = 0;
      emask[j] = 0;
      bflag[j] = 0;
      Wv[j] = 0;
      pos0[j] = 0;
      if (b < n_blocks) {
        const uint64_t r = b / bpr;
        const uint32_t q0 = (uint32_t)(b - r * bpr) * kSkmC;
        uint64_t base;
        uint32_t nwin;
        if constexpr (VAR) {
          base = start[r];
          const uint64_t len_r = start[r + 1] - base;
          nwin = len_r >= (uint64_t)K1 ? (uint32_t)(len_r - k) : 0u;
          if (q0 == 0 && nwin && count_items) items += nwin + (COUNT ? 0 : 4);  // (count: one item per window, kmer_counter.cpp:158-206)
        } else {
          base = r * L;
          nwin = L - k;
        }
        if (q0 < nwin) {
          const uint64_t a = base + q0;
          const uint64_t wi = a >> 4;
          const unsigned sh = (unsigned)(a & 15) * 2;
          const uint32_t c0 = seq[wi], c1 = seq[wi + 1], c2 = seq[wi + 2];
          const uint64_t W = ((uint64_t)funnel_l(c0, c1, sh) << 32) | funnel_l(c1, c2, sh);
          const uint64_t R = rc64(W, 32);
          uint32_t h[kSkmNM];
#pragma unroll
          for (int i = 0; i < kSkmNM; ++i) {
            const uint32_t f = (uint32_t)(W >> (64 - 2 * (i + M))) & mmask;
            const uint32_t rv = (uint32_t)(R >> (2 * i)) & mmask;
            h[i] = skm_mix(min(f, rv));
          }
          // window j: the minimum of h[j .. j + 9] = min(suffix minimum inside h[0..9], prefix minimum inside h[10..16])
          uint32_t sfx[kSkmW];
          sfx[kSkmW - 1] = h[kSkmW - 1];
#pragma unroll
          for (int i = kSkmW - 2; i >= 0; --i) sfx[i] = min(h[i], sfx[i + 1]);
          uint32_t pfx = 0xFFFFFFFFu;
          uint32_t bins[kSkmC];
          bins[0] = skm_bin_of(sfx[0]) >> bin_sh;
#pragma unroll
          for (int w = 1; w < kSkmC; ++w) {
            pfx = min(pfx, h[kSkmW - 1 + w]);
            bins[w] = skm_bin_of(min(sfx[w], pfx)) >> bin_sh;
          }
#pragma unroll
          for (int w = 0; w < kSkmC; ++w) lbin[j * kSkmC + w][tid] = bins[w];
          // a pass of the memory plan keeps the bins [bin_lo, bin_hi): the windows of a run share their bin, so runs stay whole
          const uint32_t n_here = min((uint32_t)kSkmC, nwin - q0);
          uint32_t km = 0, sm = 0, hm = 0;
          if (hp) {  // (uniform) windows of one base: adjacent bases equal all along the window
            const uint64_t dx = W ^ (W << 2);  // base i differs from base i + 1: bits 63 - 2i, 62 - 2i
#pragma unroll
            for (int w = 0; w < kSkmC; ++w)
              if ((uint32_t)w < n_here && ((dx << (2 * w)) >> (64 - 2 * (K1 - 1))) == 0) hm |= 1u << w;
            if (hm && count_items) {
#pragma unroll
              for (int w = 0; w < kSkmC; ++w)
                if ((hm >> w) & 1u) {
                  const unsigned b = (unsigned)(W >> (62 - 2 * w)) & 3u;
                  const unsigned cls = b == 0 || b == 3 ? 0u : 1u;
                  if constexpr (COUNT) {
                    // count: the key is A...A (C...C); a window of T (G) is its reverse complement (strand 1: the bases either side swap and complement)
                    const bool strand = b >= 2;
                    const unsigned prev_raw = q0 + w == 0 ? kSentinel
                                              : (w == 0 ? (unsigned)((seq[(a - 1) >> 4] >> (30 - 2 * (unsigned)((a - 1) & 15))) & 3u) : (unsigned)(W >> (64 - 2 * w)) & 3u);
                    const unsigned next_raw = q0 + w + 1 == nwin ? kSentinel : (unsigned)(W >> (62 - 2 * (w + K1))) & 3u;
                    const unsigned pv = strand ? comp_or_sentinel(next_raw) : prev_raw, nx = strand ? comp_or_sentinel(prev_raw) : next_raw;
                    atomicAdd(&hpl[cls][0], 1u);
                    if (pv < 4) atomicAdd(&hpl[cls][1 + pv], 1u);
                    if (nx < 4) atomicAdd(&hpl[cls][5 + nx], 1u);
                  } else {
                    ++hp_n[cls];
                    hp_p[cls] = min(hp_p[cls], pos_base + a + (uint64_t)w);
                  }
                }
            }
          }
#pragma unroll
          for (int w = 0; w < kSkmC; ++w)
            if ((uint32_t)w < n_here && !((hm >> w) & 1u) && bins[w] >= bin_lo && bins[w] < bin_hi) km |= 1u << w;
#pragma unroll
          for (int w = 0; w < kSkmC; ++w)
            if (((km >> w) & 1u) && (w == 0 || !((km >> (w - 1)) & 1u) || bins[w] != bins[w > 0 ? w - 1 : 0])) sm |= 1u << w;
          smask[j] = sm;
          emask[j] = km & ((sm >> 1) | ~(km >> 1));  // window w ends a run: the next one starts one, or is not this pass's (bit 8 of km is clear)
          if constexpr (COUNT) {  // the window that starts one base earlier: the base in front of the block's first window, then 31 bases
            const uint64_t pb = a > 0 ? (uint64_t)((seq[(a - 1) >> 4] >> (30 - 2 * (unsigned)((a - 1) & 15))) & 3u) : 0ull;
            Wv[j] = (W >> 2) | (pb << 62);
            bflag[j] = (q0 == 0 ? 1u : 0u) | (q0 + n_here == nwin ? 2u : 0u) | (n_here << 2);
          } else {
            Wv[j] = W;
          }
          pos0[j] = pos_base + a;  // (several GPUs: this rank's reads start at pos_base of the global read set)
        }
      }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) cnt += (uint32_t)__builtin_popcount(smask[j]);
    uint32_t total = 0;
    const uint32_t excl = block_exclusive_sum<uint32_t, NT>(cnt, sm_scan, &total);
    if (tid == 0) s_base = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
    __syncthreads();
    const unsigned long long base = s_base;
    if (base + total > cap) {  // (uniform) more records than the array was sized for: the host takes the prefix plan
      if (tid == 0) atomicOr(err, 1u);
      return;
    }
    unsigned long long at = base + excl;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      uint32_t sm = smask[j], em = emask[j];
      while (sm) {  // (runs are disjoint and in order: the i-th start goes with the i-th end)
        const int rs = __builtin_ctz(sm), re = __builtin_ctz(em);
        sm &= sm - 1;
        em &= em - 1;
        const int len = re - rs + 1;
        const uint32_t run_bin = lbin[j * kSkmC + rs][tid];
        const uint64_t p = pos0[j] + (uint64_t)rs;
        if constexpr (COUNT) {
          const uint64_t bases = (Wv[j] << (2 * rs)) & (~0ull << (64 - 2 * (K1 + len + 1)));
          const uint32_t n_blk = bflag[j] >> 2;
          const uint32_t fl = ((bflag[j] & 1u) && rs == 0 ? 16u : 0u) | ((bflag[j] & 2u) && (uint32_t)(re + 1) == n_blk ? 8u : 0u);
          out[at++] = make_uint4((run_bin << 8) | ((uint32_t)(p >> 32) << 28) | ((uint32_t)(len - 1) << 5) | fl, (uint32_t)(bases >> 32), (uint32_t)bases, (uint32_t)p);
        } else {
        const uint64_t bases = ((Wv[j] << (2 * rs)) & (~0ull << (64 - 2 * (K1 + len - 1)))) | (uint64_t)(len - 1);
        out[at++] = make_uint4((run_bin << 8) | ((uint32_t)(p >> 32) << 28), (uint32_t)(bases >> 32), (uint32_t)bases, (uint32_t)p);
        }
        atomicAdd(&dh[0][run_bin & 255u], 1u);
        atomicAdd(&dh[1][(run_bin >> 8) & 255u], 1u);
        if (third_digit) atomicAdd(&dh[2][run_bin >> 16], 1u);
      }
    }
  }
  if constexpr (VAR) {
    items = wave_sum(items);
    if ((tid & (kWave - 1)) == 0 && items) atomicAdd(cursor + 3, items);
  }
  if constexpr (COUNT) {
    __syncthreads();
    if (hp && count_items && tid < 24 && hpl[tid / 12][tid % 12]) atomicAdd(hp + 8 + (tid / 12) * 16 + (tid % 12), (unsigned long long)hpl[tid / 12][tid % 12]);
  }
  if (!COUNT && hp && count_items) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const uint32_t n_w = wave_sum(hp_n[x]);
      if (n_w) {  // (uniform per wavefront)
        unsigned long long pmin = hp_p[x];
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) pmin = min(pmin, (unsigned long long)__shfl_xor((long long)pmin, d, kWave));
        if ((tid & (kWave - 1)) == 0) {
          atomicAdd(hp + x, (unsigned long long)n_w);
          atomicMin(hp + 2 + x, pmin);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 768; i += NT)
    if (dh[i >> 8][i & 255]) atomicAdd(&digit_hist[i], (unsigned long long)dh[i >> 8][i & 255]);
}

// the one or two keys the homopolymer windows are (A...A with T...T, C...C with G...G): histogram, the mark of a key seen once, the
// aggregated stage-2 items of a solid one, behind the items of the group-by — as the walk over the table does for every other key
__global__ void k_skm_hp_publish(const unsigned long long *__restrict__ hp, int k, uint32_t m, uint8_t *__restrict__ solid_bytes,
                                 unsigned long long *__restrict__ hist, uint2 *__restrict__ agg_items, uint64_t *__restrict__ agg_cursor, int agg) {
  if (threadIdx.x || blockIdx.x) return;
  for (int x = 0; x < 2; ++x) {
    const unsigned long long cnt = hp[x];
    if (!cnt) continue;
    const unsigned long long hb = cnt > MHX_MAX_MUL ? (unsigned long long)MHX_MAX_MUL : cnt;
    hist[hb] += 1;  // :430-436
    if (cnt < m) {
      solid_bytes[hp[2 + x]] = 1;  // its only window is a non-solid occurrence
    } else if (agg) {
      const uint64_t e0 = x == 0 ? 0ull : 0x5555555555555555ull;  // A...A / C...C, MSB first; the other strand: T...T / G...G
      const uint64_t keep = ~0ull << (64 - 2 * (k + 1));
      const uint64_t xs[2] = {e0 & keep, ~e0 & keep};
      const uint64_t mask_k = ~0ull << (64 - 2 * k);
      uint64_t at = *agg_cursor;
      for (int y = 0; y < 2; ++y) {
        const uint64_t v = xs[y];
        const uint64_t f = ((v << 2) & mask_k) | (1ull << 19) | ((v >> 62) << 16) | hb;
        agg_items[at++] = make_uint2((uint32_t)(f >> 32), (uint32_t)f);
      }
      *agg_cursor = at;
    }
  }
}

// (Natural super-k-mers — runs cut at eight windows from their OWN start, the windows' minimizer hashes exchanged between the threads of a
//  workgroup through LDS — were built and measured in round 6: 0.23 instead of 0.28 records per window, the two sort passes 5.7 -> 4.8 ms,
//  but the make kernel 3.1 -> 7.9 ms (three barriers, a walk back and a walk forward through LDS per block, 1024-thread workgroups) and the
//  group-by 7.7 -> 8.4 (282 windows per wavefront trip instead of 225: a second, nearly empty round of four chunks).  Not kept:
//  profiles/r06-interim_ab_skm_natural.jsonl.)
// bounds[b] = the first record whose bin is >= b (b = 0 .. n_bins); a thread per bin, a binary search each
__global__ __launch_bounds__(256) void k_skm_bounds(const uint4 *__restrict__ recs, uint64_t n, uint32_t n_bins, uint64_t *__restrict__ bounds,
                                                    uint32_t *__restrict__ max_bin) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b > n_bins) return;
  auto lower = [&](uint32_t v) -> uint64_t {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const uint32_t bin = (reinterpret_cast<const uint32_t *>(recs + mid)[0] >> 8) & 0xFFFFFu;
      if (bin < v) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  const uint64_t at = b == n_bins ? n : lower(b);
  bounds[b] = at;
  if (b < n_bins) {
    const uint64_t nx = b + 1 == n_bins ? n : lower(b + 1);
    const uint64_t d = nx - at;
    atomicMax(max_bin, d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d);
  }
}

struct SkmArgs {
  int k;
  uint32_t m;
  uint8_t *solid_bytes;
  unsigned long long *hist;
  uint2 *agg_raw;
  uint32_t agg_cap;
  uint32_t *agg_counts;
  uint32_t *err;
  uint32_t max_fill;
  int probe_limit;
  uint32_t bin_lo, bin_hi;  // the bins of this pass
  // several GPUs: the reads live on other ranks — a mark is the global position itself, appended to the workgroup's region
  // marks_raw[blockIdx.x * marks_cap ...] (count in marks_counts[blockIdx.x]); the host packs the regions and routes them (comm.hip)
  unsigned long long *marks_raw;
  uint32_t marks_cap;
  uint32_t *marks_counts;
};
// the records of a bin arrive as one sub-range per source: one array on a single GPU; on several, one per sending rank (each ordered by bin)
struct SkmSrcs {
  const uint4 *ptr[kSkmSrcMax];
  const uint64_t *bounds[kSkmSrcMax];  // [n_bins + 1] each
  int n;
};

constexpr int kSkmThreads = 1024, kSkmLogSlots = 13;

// TAGS: read sets of 2^32 bases and more — the position bits above 32 ride in the record's first word and next to the key in the table
template <bool AGG, bool TAGS, bool DEAL>
__global__ __launch_bounds__(kSkmThreads) void k_s1_skm(SkmSrcs srcs, SkmArgs a, uint32_t *__restrict__ ticket) {
  constexpr int NT = kSkmThreads, NSLOT = 1 << kSkmLogSlots, W = NSLOT / NT;
  constexpr unsigned long long kEmpty = ~0ull;  // never a key: head / tail bits 63 do not occur
  constexpr int NLIST = 1024;
  __shared__ unsigned long long keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  __shared__ uint32_t fpos[NSLOT];
  __shared__ uint8_t ftag[TAGS ? NSLOT : 4];
  __shared__ uint32_t lhist[kSegHist];
  __shared__ unsigned long long slist_k[AGG ? NLIST : 1];
  __shared__ uint32_t slist_c[AGG ? NLIST : 1];
  __shared__ unsigned long long heads[DEAL ? kSkmThreads / kWave : 1][8];  // DEAL, per wavefront: bit g set = window g of the trip is the first of its record
  __shared__ uint32_t s_bad2[2], s_nclaimed2[2], s_list_n2[2];  // per round, double-buffered: the next round's are cleared while this round's are read
  __shared__ uint32_t s_agg_cur, s_mark_cur, s_tk;
  __shared__ uint64_t s_lo[kSkmSrcMax][kSkmBatch + 1];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = DEAL ? tid / kWave : 0;
  const uint32_t le_lo = lane >= 31 ? 0xFFFFFFFFu : (2u << lane) - 1u, le_hi = lane < 32 ? 0u : (lane == 63 ? 0xFFFFFFFFu : (2u << (lane - 32)) - 1u);  // lanes <= this one
  const int k = a.k, K1 = k + 1;
  const uint32_t m = a.m;
  const uint64_t kmask = ~0ull << (64 - 2 * (k - 1));
  uint2 *const agg_end = AGG ? a.agg_raw + (size_t)(blockIdx.x + 1) * a.agg_cap : nullptr;
  for (int i = tid; i < NSLOT; i += NT) {
    keys[i] = kEmpty;
    cnts[i] = 0;
  }
  for (int i = tid; i < kSegHist; i += NT) lhist[i] = 0;
  if (tid == 0) {
    s_bad2[0] = s_bad2[1] = 0;
    s_nclaimed2[0] = s_nclaimed2[1] = 0;
    s_list_n2[0] = s_list_n2[1] = 0;
    s_agg_cur = 0;
    s_mark_cur = 0;
  }
  const int n_src = srcs.n;
  unsigned long long *const marks_out = a.marks_raw ? a.marks_raw + (size_t)blockIdx.x * a.marks_cap : nullptr;
  int rp = 0;  // the round's parity
  __syncthreads();

  // the aggregated stage-2 items of a solid key (one per strand; one for a palindrome) -> this workgroup's region, from its end
  auto emit_items = [&](unsigned long long key, uint32_t cnt, bool dense, bool valid) {
    uint64_t x = 0, xr = 0;
    uint32_t n_out = 0;
    if (valid) {
      const uint64_t smer = key & kmask;
      x = ((uint64_t)((key >> 3) & 7u) << 62) | (smer >> 2) | ((uint64_t)(key & 7u) << (62 - 2 * k));
      xr = rc64(x, K1);
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
      ok = wbase + tot <= a.agg_cap;
      at = wbase + incl - n_out;
    } else {
      at = atomicAdd(&s_agg_cur, n_out);
      ok = at + n_out <= a.agg_cap;
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

  uint4 pre = make_uint4(0u, 0u, 0u, 0u);  // a first trip requested ahead (of the bin that starts at pre_at)
  uint64_t pre_at = 0;
  bool pre_ok = false;
  for (;;) {
    if (tid == 0) s_tk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint64_t bin0 = (uint64_t)a.bin_lo + (uint64_t)s_tk * kSkmBatch;
    if (bin0 >= a.bin_hi) break;
    if (tid < n_src * (kSkmBatch + 1)) {
      const int q = tid / (kSkmBatch + 1), x = tid - q * (kSkmBatch + 1);
      s_lo[q][x] = srcs.bounds[q][min(bin0 + x, (uint64_t)a.bin_hi)];
    }
    __syncthreads();
    for (int bb = 0; bb < kSkmBatch; ++bb) {
      bool any = false;
      for (int q = 0; q < n_src; ++q) any = any || s_lo[q][bb] != s_lo[q][bb + 1];
      if (!any) continue;
      const uint64_t lo0 = s_lo[0][bb], hi0 = s_lo[0][bb + 1];
      // the bin in rounds: round (sub, rj) takes the keys whose top `sub` bits of a second hash are rj (a round that overflows the
      // table is redone in two halves)
      uint32_t sub = 0, rj = 0;
      for (;;) {
        uint32_t seen = 0;
        // A: insert, source by source (the round's first trip was requested before the walk of the round before it)
        for (int q = 0; q < n_src; ++q) {
        const uint64_t lo = s_lo[q][bb], hi = s_lo[q][bb + 1];
        if (lo == hi) continue;
        const uint4 *__restrict__ recs = srcs.ptr[q];
        uint4 nxt = q == 0 && pre_ok && pre_at == lo ? pre : (lo + tid < hi ? recs[lo + tid] : make_uint4(0u, 0u, 0u, 0u));
        if (q == 0) pre_ok = false;
        for (uint64_t base = lo; base < hi; base += NT) {
          const uint4 r = nxt;
          const bool in = base + tid < hi;
          if (base + NT + tid < hi) nxt = recs[base + NT + tid];
          if (seen > a.max_fill) continue;  // (uniform per wavefront; the round is redone in halves anyway)
          const uint32_t len = in ? (r.z & 7u) + 1u : 0u;
          const uint8_t tag = TAGS ? (uint8_t)(r.x >> 28) : (uint8_t)0;
          uint32_t claims = 0;
          // (the slot found — the key's own, or a free one claimed: count it, and remember the window that claimed it)
          auto settle = [&](unsigned long long old, unsigned long long key, uint32_t hh, uint32_t pos, uint8_t tg) -> bool {
            if (old != kEmpty && old != key) return false;
            atomicAdd(&cnts[hh], 1u);
            if (old == kEmpty) {  // only read back when the count stays 1: then this window is the key's only one
              fpos[hh] = pos;
              if constexpr (TAGS) ftag[hh] = tg;
              ++claims;
            }
            return true;
          };
          const uint64_t rec_b = ((uint64_t)r.y << 32) | r.z;
          const uint64_t rec_r = rc64(rec_b, 32);  // (of the record's 32 base slots, once: a window's reverse complement is a sub-window of it)
          if constexpr (DEAL) {
            // The windows of the wavefront's 64 records (~225) are DEALT to the lanes, 64 at a time: window g belongs to the record whose
            // run of windows starts at or before g — a bitmap of run heads per wavefront, one population count per window — and comes
            // over with six shuffles.  All lanes work on every step (own-record expansion: 44 % of the lane steps, below); four steps
            // share one round of compare-and-swaps and one retry loop.
            const uint32_t incl = wave_inclusive_sum(len);
            const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
            const uint32_t start = incl - len;
            // (the LDS executes a wavefront's operations in order: clear, set, read — no barrier between them)
            if (lane < 8) heads[wv][lane] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (len) atomicOr(&heads[wv][start >> 6], 1ull << (start & 63u));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long hv = __hip_atomic_load(&heads[wv][lane & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t hv_lo = (uint32_t)hv, hv_hi = (uint32_t)(hv >> 32);
            const uint32_t bh = r.y, bl = r.z, qh = (uint32_t)(rec_r >> 32), ql = (uint32_t)rec_r;
            const uint32_t pd = r.w - start;  // position of window g of the trip = pd + g
            uint32_t cbase_m1 = 0xFFFFFFFFu;  // heads in front of this step's windows, minus one
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
              if ((uint32_t)(grp * 4 * kWave) >= T) break;  // (uniform)
              unsigned long long key[4], old1[4];
              uint32_t h1[4], pos[4], mine = 0;
              uint8_t tg[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int c = grp * 4 + u;
                const uint32_t g = (uint32_t)(c * kWave) + (uint32_t)lane;
                const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readlane((int)hv_lo, c), w_hi = (uint32_t)__builtin_amdgcn_readlane((int)hv_hi, c);
                const uint32_t o = cbase_m1 + (uint32_t)__builtin_popcount(w_lo & le_lo) + (uint32_t)__builtin_popcount(w_hi & le_hi);
                cbase_m1 += (uint32_t)__builtin_popcount(w_lo) + (uint32_t)__builtin_popcount(w_hi);
                const uint32_t xbh = __shfl(bh, (int)o, kWave), xbl = __shfl(bl, (int)o, kWave);
                const uint32_t xqh = __shfl(qh, (int)o, kWave), xql = __shfl(ql, (int)o, kWave);
                const uint32_t so = __shfl(start, (int)o, kWave), pdo = __shfl(pd, (int)o, kWave);
                const uint32_t j = g - so;
                const uint64_t win = (((uint64_t)xbh << 32) | xbl) << (2 * j);  // the (k+1)-mer head.S.tail, MSB first
                const uint64_t f = (win << 2) & kmask;
                const uint64_t rc = ((((uint64_t)xqh << 32) | xql) << (2 * (32 - k - j))) & kmask;
                const unsigned head = (unsigned)(win >> 62), tail = ((uint32_t)win >> (62 - 2 * k)) & 3u;  // (k <= 22: the tail sits in the low word)
                const bool rev = f > rc || (f == rc && head > 3u - tail);  // read_to_sdbg_s1.cpp:228-292
                const uint64_t kf = f | ((uint64_t)head << 3) | tail, kr = rc | ((uint64_t)(3u - tail) << 3) | (3u - head);
                key[u] = rev ? kr : kf;
                const uint32_t klo = (uint32_t)key[u], khi = (uint32_t)(key[u] >> 32);
                bool mn = g < T;
                if (sub) mn = mn && (skm_mix2((klo * 0xC2B2AE35u) ^ (khi * 0x27D4EB2Fu)) >> (32 - sub)) == rj;  // (uniform branch)
                mine |= mn ? 1u << u : 0u;
                h1[u] = ((klo * 0x9E3779B1u) ^ (khi * 0x85EBCA6Bu)) >> (32 - kSkmLogSlots);
                pos[u] = pdo + g;
                if constexpr (TAGS) {
                  const uint32_t p0 = pdo + so;  // the record's own position word: a run across a multiple of 2^32 carries into the tag
                  tg[u] = (uint8_t)(__shfl((uint32_t)tag, (int)o, kWave) + (pos[u] < p0 ? 1u : 0u));
                } else {
                  tg[u] = 0;
                }
              }
              if (a.probe_limit <= 0) {
                if (mine) s_bad2[rp] = 1;
                mine = 0;
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                old1[u] = kEmpty;
                if ((mine >> u) & 1u) old1[u] = atomicCAS(&keys[h1[u]], kEmpty, key[u]);
              }
              bool has = false;
              unsigned long long pk = 0;
              uint32_t ph = 0, pp = 0;
              uint8_t pt = 0;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if ((mine >> u) & 1u) {
                  if (!settle(old1[u], key[u], h1[u], pos[u], tg[u])) {
                    uint32_t hh = (h1[u] + 1) & (NSLOT - 1);
                    if (!has) {
                      has = true;
                      pk = key[u], ph = hh, pp = pos[u], pt = tg[u];
                    } else {
                      int n = 0;
                      while (!settle(atomicCAS(&keys[hh], kEmpty, key[u]), key[u], hh, pos[u], tg[u])) {
                        hh = (hh + 1) & (NSLOT - 1);
                        if (++n >= a.probe_limit) {
                          s_bad2[rp] = 1;
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
                  if (settle(atomicCAS(&keys[ph], kEmpty, pk), pk, ph, pp, pt)) has = false;
                  else ph = (ph + 1) & (NSLOT - 1);
                }
                if (++turns > a.probe_limit) {  // (uniform: every lane counts the same turns)
                  if (has) s_bad2[rp] = 1;
                  break;
                }
              }
            }
          } else {
            // Every lane expands its own record, four windows at a time: the keys are independent of each other, so their compare-and-swaps
            // go out back to back and one LDS round trip serves four windows; lanes whose record is shorter idle.
  #pragma unroll
            for (int half = 0; half < 2; ++half) {
              if (half == 1 && __ballot(len > 4u) == 0) break;  // (uniform)
              unsigned long long key[4], old1[4];
              uint32_t h1[4], mine = 0;
  #pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int j = half * 4 + u;
                const uint64_t win = rec_b << (2 * j);  // the (k+1)-mer head.S.tail, MSB first
                const uint64_t f = (win << 2) & kmask;
                const uint64_t rc = (rec_r << (2 * (32 - k - j))) & kmask;
                const unsigned head = (unsigned)(win >> 62), tail = (unsigned)(win >> (62 - 2 * k)) & 3u;
                const bool rev = f > rc || (f == rc && head > 3u - tail);  // read_to_sdbg_s1.cpp:228-292
                const uint64_t kf = f | ((uint64_t)head << 3) | tail, kr = rc | ((uint64_t)(3u - tail) << 3) | (3u - head);
                key[u] = rev ? kr : kf;
                const uint32_t klo = (uint32_t)key[u], khi = (uint32_t)(key[u] >> 32);
                bool mn = (uint32_t)j < len;
                if (sub) mn = mn && (skm_mix2((klo * 0xC2B2AE35u) ^ (khi * 0x27D4EB2Fu)) >> (32 - sub)) == rj;  // (uniform branch)
                mine |= mn ? 1u << u : 0u;
                h1[u] = ((klo * 0x9E3779B1u) ^ (khi * 0x85EBCA6Bu)) >> (32 - kSkmLogSlots);
              }
              if (a.probe_limit <= 0) {
                if (mine) s_bad2[rp] = 1;
                mine = 0;
              }
  #pragma unroll
              for (int u = 0; u < 4; ++u) {
                old1[u] = kEmpty;
                if ((mine >> u) & 1u) old1[u] = atomicCAS(&keys[h1[u]], kEmpty, key[u]);
              }
              // a lane that met another key keeps the window pending — one per lane; a second one of the same four is seen to on the spot —
              // and the pending windows of all lanes are retried together
              bool has = false;
              unsigned long long pk = 0;
              uint32_t ph = 0, pp = 0;
              uint8_t pt = 0;
  #pragma unroll
              for (int u = 0; u < 4; ++u) {
                if ((mine >> u) & 1u) {
                  const uint32_t pos = r.w + (uint32_t)(half * 4 + u);
                  const uint8_t ptag = TAGS ? (uint8_t)(tag + (pos < r.w ? 1 : 0)) : (uint8_t)0;  // (a run across a multiple of 2^32)
                  if (!settle(old1[u], key[u], h1[u], pos, ptag)) {
                    uint32_t hh = (h1[u] + 1) & (NSLOT - 1);
                    if (!has) {
                      has = true;
                      pk = key[u], ph = hh, pp = pos, pt = ptag;
                    } else {
                      int n = 0;
                      while (!settle(atomicCAS(&keys[hh], kEmpty, key[u]), key[u], hh, pos, ptag)) {
                        hh = (hh + 1) & (NSLOT - 1);
                        if (++n >= a.probe_limit) {
                          s_bad2[rp] = 1;
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
                  if (settle(atomicCAS(&keys[ph], kEmpty, pk), pk, ph, pp, pt)) has = false;
                  else ph = (ph + 1) & (NSLOT - 1);
                }
                if (++turns > a.probe_limit) {  // (uniform: every lane counts the same turns)
                  if (has) s_bad2[rp] = 1;
                  break;
                }
              }
            }
          }
          {
            const uint32_t c = wave_sum(claims);
            if (lane == 0 && c) atomicAdd(&s_nclaimed2[rp], c);
            seen = __hip_atomic_load(&s_nclaimed2[rp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen);
          }
        }
        }  // (sources)
        __syncthreads();  // the table is complete
        const bool bad = s_bad2[rp] != 0 || s_nclaimed2[rp] > a.max_fill;
        if (tid == 0) {  // (the other parity: last read behind the first barrier of the round before this one)
          s_bad2[rp ^ 1] = 0;
          s_nclaimed2[rp ^ 1] = 0;
          s_list_n2[rp ^ 1] = 0;
        }
        uint32_t nsub = sub, nrj = rj;
        bool done = false, give_up = false;
        if (bad) {
          if (sub >= 31) {
            give_up = true;
            done = true;
          } else {
            nsub = sub + 1;
            nrj = rj << 1;
          }
        } else {
          nrj = rj + 1;
          while (nsub > 0 && (nrj & 1u) == 0) {
            --nsub;
            nrj >>= 1;
          }
          done = nsub == 0 && nrj == 1u;
        }
        if (give_up && tid == 0) atomicOr(a.err, 1u);
        {  // the first trip of what comes next — this bin again, or the next bin of the batch that holds records — is requested here (first source)
          uint64_t nlo = lo0, nhi = hi0;
          if (done) {
            nlo = nhi = 0;
            for (int nb = bb + 1; nb < kSkmBatch; ++nb)
              if (s_lo[0][nb] != s_lo[0][nb + 1]) {
                nlo = s_lo[0][nb];
                nhi = s_lo[0][nb + 1];
                break;
              }
          }
          if (nlo != nhi) {
            pre = nlo + tid < nhi ? srcs.ptr[0][nlo + tid] : make_uint4(0u, 0u, 0u, 0u);
            pre_at = nlo;
            pre_ok = true;
          }
        }
        // C: one walk over the table — per distinct key: histogram, the mark of a non-solid key's only record, the solid keys listed
        unsigned long long wk[W];
        uint32_t wc[W], wp[W];
        uint8_t wt[W];
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          wk[it] = keys[sl];
          wc[it] = cnts[sl];
          wp[it] = fpos[sl];
          wt[it] = TAGS ? ftag[sl] : (uint8_t)0;
        }
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          keys[sl] = kEmpty;
          cnts[sl] = 0;
        }
        uint32_t want_bits = 0, mark_bits = 0;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          if (wk[it] != kEmpty && !bad) {
            const uint32_t cnt = wc[it];
            const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;  // :430-436
            if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
            else atomicAdd(&a.hist[hb], 1ull);
            if (cnt < m) mark_bits |= 1u << it;  // count 1 < m <= 2: the key's only window is a non-solid occurrence
            else if (AGG) want_bits |= 1u << it;
          }
        }
        if (!marks_out) {  // (uniform)
#pragma unroll
          for (int it = 0; it < W; ++it)
            if ((mark_bits >> it) & 1u) a.solid_bytes[((uint64_t)wt[it] << 32) | wp[it]] = 1;
        } else {  // several GPUs: the mark is the global position itself, appended to this workgroup's region
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
                if (at < a.marks_cap) marks_out[at] = ((unsigned long long)wt[it] << 32) | wp[it];
                else atomicOr(a.err, 2u);
                ++at;
              }
          }
        }
        if constexpr (AGG) {
          const uint32_t n_w = (uint32_t)__builtin_popcount(want_bits);
          const uint32_t incl = wave_inclusive_sum(n_w);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t lbase = 0;
            if (lane == 0) lbase = atomicAdd(&s_list_n2[rp], tot);
            lbase = __shfl(lbase, 0, kWave);
            uint32_t at = lbase + incl - n_w;
#pragma unroll
            for (int it = 0; it < W; ++it)
              if ((want_bits >> it) & 1u) {
                if (at < (uint32_t)NLIST) {
                  slist_k[at] = wk[it];
                  slist_c[at] = wc[it];
                } else {
                  emit_items(wk[it], wc[it], false, true);  // (more solid keys in one round than the list holds: in place)
                }
                ++at;
              }
          }
        }
        __syncthreads();  // the table is empty, the list complete
        if constexpr (AGG) {
          const uint32_t n_list = min(s_list_n2[rp], (uint32_t)NLIST);
          for (uint32_t base = 0; base < n_list; base += NT) {
            const uint32_t i = base + tid;
            const bool v = i < n_list;
            emit_items(v ? slist_k[i] : 0ull, v ? slist_c[i] : 0u, true, v);
          }
        }
        // (no barrier here: the next round's inserts touch the table and the other parity's counters only; its walk, which writes the
        //  list again, comes behind that round's first barrier — by then every thread has left this loop)
        rp ^= 1;
        sub = nsub;
        rj = nrj;
        if (done) break;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < kSegHist; i += NT)
    if (lhist[i]) atomicAdd(&a.hist[i], (unsigned long long)lhist[i]);
  if (AGG && tid == 0) a.agg_counts[blockIdx.x] = s_agg_cur < a.agg_cap ? s_agg_cur : a.agg_cap;
  if (marks_out && tid == 0) a.marks_counts[blockIdx.x] = s_mark_cur < a.marks_cap ? s_mark_cur : a.marks_cap;
}

// count: the one or two keys the homopolymer windows are -> histogram, the packed edge of a solid one behind the passes' edges.  A solid one
// WITHOUT an in- or out-edge would have to move first_0_out / last_0_in of every read that holds such a window: res[2] says so and the
// caller takes the prefix plan (poly-A / poly-G stretches are preceded and followed by their own base: not seen in practice)
__global__ void k_count_hp_publish(const unsigned long long *__restrict__ hp, int k, uint32_t m, unsigned long long *__restrict__ hist,
                                   unsigned long long *__restrict__ edges_at, unsigned long long *__restrict__ res) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned long long n_edges = 0, n_keys = 0, flagged = 0;
  for (int x = 0; x < 2; ++x) {
    const unsigned long long *t = hp + 8 + x * 16;
    const unsigned long long cnt = t[0];
    if (!cnt) continue;
    ++n_keys;
    const unsigned long long hb = cnt > MHX_MAX_MUL ? (unsigned long long)MHX_MAX_MUL : cnt;
    hist[hb] += 1;
    if (cnt >= m) {
      bool has_in = false, has_out = false;
      for (int c = 0; c < 4; ++c) {
        has_in = has_in || t[1 + c] >= m;
        has_out = has_out || t[5 + c] >= m;
      }
      if (!has_in || !has_out) flagged = 1;
      const uint64_t key = (x == 0 ? 0ull : 0x5555555555555555ull) & (~0ull << (64 - 2 * (k + 1)));
      edges_at[n_edges++] = key | hb;  // PackEdge, kmer_counter.cpp:32-52
    }
  }
  res[0] = n_edges;
  res[1] = n_keys;
  res[2] = flagged;
}

// ---------------------------------------------------------------------------------------------------------------
// `count` on super-k-mer records (KmerCounter::Lv2Postprocess, kmer_counter.cpp:254-381, on the records k_skm_make<.., COUNT> makes): the
// table key is the canonical (k+1)-mer (the smaller of the window and its reverse complement, :179), the slot's third word holds, per base
// in front and base behind — in the key's orientation — "seen once" and "seen twice" bits (min count <= 2: has_in / has_out need no more);
// the walk over the table takes the histogram, sends a solid key's packed edge (PackEdge, :32-52) to the workgroup's region and leaves two
// flag bits in the slot of a solid key without an in- or out-edge; a bin that has such a key is expanded once more, and the windows of the
// flagged keys move first_0_out / last_0_in of their reads (:307-368).  One GPU; the windows are dealt to the lanes as in k_s1_skm.
struct CountSkmArgs {
  int k;
  uint32_t m;
  unsigned long long *hist, *ctr;  // ctr[4]: distinct keys
  unsigned long long *edges_raw;   // per-workgroup regions of packed edges, filled from the front
  uint32_t edges_cap;
  uint32_t *edges_counts;
  uint32_t *err;
  uint32_t max_fill;
  int probe_limit;
  uint32_t bin_lo, bin_hi;
  const uint64_t *c_start;
  uint64_t c_n_seqs;
  uint32_t c_fixed_len;
  uint32_t *first_0_out, *last_0_in_p1;
};

template <bool TAGS, int G>
__global__ __launch_bounds__(kSkmThreads) void k_count_skm(SkmSrcs srcs, CountSkmArgs a, uint32_t *__restrict__ ticket) {
  constexpr int NT = kSkmThreads, NSLOT = 1 << kSkmLogSlots, W = NSLOT / NT;
  constexpr unsigned long long kEmpty = ~0ull;  // never a key: the bits below the (k+1)-mer are zero
  __shared__ unsigned long long keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  __shared__ uint32_t chw[NSLOT];  // base in front x: bit 2x seen once, 2x + 1 seen twice; base behind x: bits 8 + 2x, 9 + 2x; after the walk: flags << 30
  __shared__ uint32_t lhist[kSegHist];
  __shared__ unsigned long long heads[kSkmThreads / kWave][8];
  __shared__ uint32_t s_bad, s_nclaimed, s_edge_cur, s_flagged, s_tk;
  __shared__ uint64_t s_lo[kSkmSrcMax][kSkmBatch + 1];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const uint32_t le_lo = lane >= 31 ? 0xFFFFFFFFu : (2u << lane) - 1u, le_hi = lane < 32 ? 0u : (lane == 63 ? 0xFFFFFFFFu : (2u << (lane - 32)) - 1u);
  const int k = a.k, K1 = k + 1;
  const uint32_t m = a.m;
  const uint64_t kmask = ~0ull << (64 - 2 * K1);
  const int n_src = srcs.n;
  unsigned long long *const eout = a.edges_raw + (size_t)blockIdx.x * a.edges_cap;
  for (int i = tid; i < NSLOT; i += NT) {
    keys[i] = kEmpty;
    cnts[i] = 0;
    chw[i] = 0;
  }
  for (int i = tid; i < kSegHist; i += NT) lhist[i] = 0;
  if (tid == 0) {
    s_bad = 0;
    s_nclaimed = 0;
    s_edge_cur = 0;
    s_flagged = 0;
  }
  __syncthreads();
  unsigned long long n_dist_all = 0;

  // One window of the trip, dealt to this lane: window g of the wavefront's records belongs to the record whose run starts at or before g
  // (bitmap of run heads, hv) and comes over with seven shuffles.
  struct Win {
    bool valid, fwd;
    unsigned long long key;
    unsigned pv, nx;   // base in front / behind in the key's orientation (4: none)
    uint32_t pos;      // low word of the window's position
    uint8_t tag;
  };
  struct Trip {
    uint32_t T, start, pd, hv_lo, hv_hi, bh, bl, qh, ql, meta;
    uint8_t tag;
  };
  auto open_trip = [&](const uint4 &r, bool in) -> Trip {
    Trip t;
    const uint32_t len = in ? ((r.x >> 5) & 7u) + 1u : 0u;
    const uint32_t incl = wave_inclusive_sum(len);
    t.T = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
    t.start = incl - len;
    if (lane < 8) heads[wv][lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (len) atomicOr(&heads[wv][t.start >> 6], 1ull << (t.start & 63u));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long hv = __hip_atomic_load(&heads[wv][lane & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t.hv_lo = (uint32_t)hv;
    t.hv_hi = (uint32_t)(hv >> 32);
    const uint64_t rec_b = ((uint64_t)r.y << 32) | r.z;
    const uint64_t rec_r = rc64(rec_b, 32);
    t.bh = r.y, t.bl = r.z, t.qh = (uint32_t)(rec_r >> 32), t.ql = (uint32_t)rec_r;
    t.pd = r.w - t.start;
    t.meta = (r.x & 0xFFu) | (len << 8);  // flags, and the run's length for its last window
    t.tag = TAGS ? (uint8_t)(r.x >> 28) : (uint8_t)0;
    return t;
  };
  auto deal = [&](const Trip &t, int c, uint32_t &cbase_m1) -> Win {
    Win w;
    const uint32_t g = (uint32_t)(c * kWave) + (uint32_t)lane;
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readlane((int)t.hv_lo, c), w_hi = (uint32_t)__builtin_amdgcn_readlane((int)t.hv_hi, c);
    const uint32_t o = cbase_m1 + (uint32_t)__builtin_popcount(w_lo & le_lo) + (uint32_t)__builtin_popcount(w_hi & le_hi);
    cbase_m1 += (uint32_t)__builtin_popcount(w_lo) + (uint32_t)__builtin_popcount(w_hi);
    const uint32_t xbh = __shfl(t.bh, (int)o, kWave), xbl = __shfl(t.bl, (int)o, kWave);
    const uint32_t xqh = __shfl(t.qh, (int)o, kWave), xql = __shfl(t.ql, (int)o, kWave);
    const uint32_t so = __shfl(t.start, (int)o, kWave), pdo = __shfl(t.pd, (int)o, kWave), mo = __shfl(t.meta, (int)o, kWave);
    const uint32_t j = g - so;
    const uint64_t B = ((uint64_t)xbh << 32) | xbl, R = ((uint64_t)xqh << 32) | xql;
    const uint64_t x = (B << (2 * (1 + j))) & kmask;                 // the window: slots 1 + j .. j + k + 1 of the record
    const uint64_t rcx = (R << (2 * (32 - 1 - K1 - j))) & kmask;     // its reverse complement: a sub-window of the record's
    const unsigned pb = (unsigned)(B >> (62 - 2 * j)) & 3u, nb = (unsigned)(B >> (60 - 2 * K1 - 2 * j)) & 3u;
    const unsigned prev = j == 0 && (mo & 16u) ? kSentinel : pb, next = j + 1 == (mo >> 8) && (mo & 8u) ? kSentinel : nb;
    const bool strand = rcx < x;  // rev_edge.cmp(edge) < 0, kmer_counter.cpp:179
    w.fwd = !strand;
    w.key = strand ? rcx : x;
    w.pv = strand ? comp_or_sentinel(next) : prev;
    w.nx = strand ? comp_or_sentinel(prev) : next;
    w.pos = pdo + g;
    if constexpr (TAGS) w.tag = (uint8_t)(__shfl((uint32_t)t.tag, (int)o, kWave) + (w.pos < pdo + so ? 1u : 0u));
    else w.tag = 0;
    w.valid = g < t.T;
    return w;
  };
  auto slot_of = [&](unsigned long long key) -> uint32_t {
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    return ((klo * 0x9E3779B1u) ^ (khi * 0x85EBCA6Bu)) >> (32 - kSkmLogSlots);
  };
  auto in_round = [&](unsigned long long key, uint32_t sub, uint32_t rj) -> bool {
    if (!sub) return true;
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    return (skm_mix2((klo * 0xC2B2AE35u) ^ (khi * 0x27D4EB2Fu)) >> (32 - sub)) == rj;
  };

  for (;;) {
    if (tid == 0) s_tk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint64_t bin0 = (uint64_t)a.bin_lo + (uint64_t)s_tk * kSkmBatch;
    if (bin0 >= a.bin_hi) break;
    if (tid < n_src * (kSkmBatch + 1)) {
      const int q = tid / (kSkmBatch + 1), x = tid - q * (kSkmBatch + 1);
      s_lo[q][x] = srcs.bounds[q][min(bin0 + x, (uint64_t)a.bin_hi)];
    }
    __syncthreads();
    for (int bb = 0; bb < kSkmBatch; ++bb) {
      bool any = false;
      for (int q = 0; q < n_src; ++q) any = any || s_lo[q][bb] != s_lo[q][bb + 1];
      if (!any) continue;
      uint32_t sub = 0, rj = 0;
      for (;;) {
        uint32_t seen = 0;
        // A: insert
        for (int q = 0; q < n_src; ++q) {
          const uint64_t lo = s_lo[q][bb], hi = s_lo[q][bb + 1];
          const uint4 *__restrict__ recs = srcs.ptr[q];
          for (uint64_t base = lo; base < hi; base += NT) {
            const bool in = base + tid < hi;
            const uint4 r = in ? recs[base + tid] : make_uint4(0u, 0u, 0u, 0u);
            if (seen > a.max_fill) continue;
            const Trip t = open_trip(r, in);
            uint32_t claims = 0;
            // (the slot found — the key's own, or a free one claimed: count the window and the bases either side of it)
            auto settle = [&](unsigned long long old, unsigned long long key, uint32_t hh, unsigned pv, unsigned nx) -> bool {
              if (old != kEmpty && old != key) return false;
              atomicAdd(&cnts[hh], 1u);
              if (old == kEmpty) ++claims;
              const uint32_t add1 = (pv < 4 ? 1u << (2 * pv) : 0u) | (nx < 4 ? 1u << (8 + 2 * nx) : 0u);  // ('$' counts for nothing)
              if (add1) {
                const uint32_t o = atomicOr(&chw[hh], add1);
                const uint32_t again = ((o & add1) << 1) & ~o;  // a base seen before and now again: seen twice
                if (again) atomicOr(&chw[hh], again);
              }
              return true;
            };
            uint32_t cbase_m1 = 0xFFFFFFFFu;
#pragma unroll
            for (int grp = 0; grp < 8 / G; ++grp) {  // G chunks of 64 windows share one round of compare-and-swaps and one retry loop
              if ((uint32_t)(grp * G * kWave) >= t.T) break;  // (uniform)
              Win w[G];
              unsigned long long old1[G];
              uint32_t h1[G], mine = 0;
#pragma unroll
              for (int u = 0; u < G; ++u) {
                w[u] = deal(t, grp * G + u, cbase_m1);
                const bool mn = w[u].valid && in_round(w[u].key, sub, rj);
                mine |= mn ? 1u << u : 0u;
                h1[u] = slot_of(w[u].key);
              }
              if (a.probe_limit <= 0) {
                if (mine) s_bad = 1;
                mine = 0;
              }
#pragma unroll
              for (int u = 0; u < G; ++u) {
                old1[u] = kEmpty;
                if ((mine >> u) & 1u) old1[u] = atomicCAS(&keys[h1[u]], kEmpty, w[u].key);
              }
              bool has = false;
              unsigned long long pk = 0;
              uint32_t ph = 0;
              unsigned ppv = 0, pnx = 0;
#pragma unroll
              for (int u = 0; u < G; ++u) {
                if ((mine >> u) & 1u) {
                  if (!settle(old1[u], w[u].key, h1[u], w[u].pv, w[u].nx)) {
                    uint32_t hh = (h1[u] + 1) & (NSLOT - 1);
                    if (!has) {
                      has = true;
                      pk = w[u].key, ph = hh, ppv = w[u].pv, pnx = w[u].nx;
                    } else {
                      int n = 0;
                      while (!settle(atomicCAS(&keys[hh], kEmpty, w[u].key), w[u].key, hh, w[u].pv, w[u].nx)) {
                        hh = (hh + 1) & (NSLOT - 1);
                        if (++n >= a.probe_limit) {
                          s_bad = 1;
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
                  if (settle(atomicCAS(&keys[ph], kEmpty, pk), pk, ph, ppv, pnx)) has = false;
                  else ph = (ph + 1) & (NSLOT - 1);
                }
                if (++turns > a.probe_limit) {  // (uniform: every lane counts the same turns)
                  if (has) s_bad = 1;
                  break;
                }
              }
            }
            {
              const uint32_t c = wave_sum(claims);
              if (lane == 0 && c) atomicAdd(&s_nclaimed, c);
              seen = __hip_atomic_load(&s_nclaimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen);
            }
          }
        }
        __syncthreads();  // the table is complete
        const bool bad = s_bad != 0 || s_nclaimed > a.max_fill;
        uint32_t nsub = sub, nrj = rj;
        bool done = false, give_up = false;
        if (bad) {
          if (sub >= 31) {
            give_up = true;
            done = true;
          } else {
            nsub = sub + 1;
            nrj = rj << 1;
          }
        } else {
          nrj = rj + 1;
          while (nsub > 0 && (nrj & 1u) == 0) {
            --nsub;
            nrj >>= 1;
          }
          done = nsub == 0 && nrj == 1u;
        }
        if (give_up && tid == 0) atomicOr(a.err, 1u);
        // C: one walk over the table — per distinct (k+1)-mer: histogram, has_in / has_out from the seen-twice (m = 2) or seen-once (m = 1)
        // bits, the packed edge of a solid key -> this workgroup's region; a solid key without an in- or out-edge keeps two flag bits
        unsigned long long wk[W];
        uint32_t wc[W], wf[W];
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          wk[it] = keys[sl];
          wc[it] = cnts[sl];
          wf[it] = chw[sl];
        }
        const uint32_t lvl = m >= 2 ? 0xAAu : 0x55u;
        uint32_t solid_bits = 0, n_dist = 0;
        bool any_flag = false;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          uint32_t fb = 0;
          if (wk[it] != kEmpty && !bad) {
            ++n_dist;
            const uint32_t cnt = wc[it];
            const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;
            if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
            else atomicAdd(&a.hist[hb], 1ull);
            if (cnt >= m) {
              solid_bits |= 1u << it;
              const bool has_in = (wf[it] & lvl) != 0, has_out = ((wf[it] >> 8) & lvl) != 0;
              fb = (has_in ? 0u : 1u) | (has_out ? 0u : 2u);
              any_flag = any_flag || fb != 0;
            }
          }
          chw[it * NT + tid] = fb << 30;
        }
        n_dist_all += n_dist;
        if (__ballot(any_flag) && lane == 0) s_flagged = 1;
        {  // the solid keys' packed edges (PackEdge, kmer_counter.cpp:32-52: multiplicity in the low 16 bits) -> the region, from its front
          const uint32_t n_e = (uint32_t)__builtin_popcount(solid_bits);
          const uint32_t incl = wave_inclusive_sum(n_e);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t ebase = 0;
            if (lane == 0) ebase = atomicAdd(&s_edge_cur, tot);
            ebase = __shfl(ebase, 0, kWave);
            if (ebase + tot > a.edges_cap) {
              if (lane == 0) atomicOr(a.err, 1u);
            } else {
              uint32_t at = ebase + incl - n_e;
#pragma unroll
              for (int it = 0; it < W; ++it)
                if ((solid_bits >> it) & 1u) {
                  const uint32_t mul = wc[it] > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : wc[it];
                  eout[at++] = wk[it] | mul;
                }
            }
          }
        }
        __syncthreads();  // the flags are in the slots
        if (s_flagged && !bad) {  // the windows of the flagged keys: first_0_out / last_0_in of their reads (kmer_counter.cpp:307-368)
          for (int q = 0; q < n_src; ++q) {
            const uint64_t lo = s_lo[q][bb], hi = s_lo[q][bb + 1];
            const uint4 *__restrict__ recs = srcs.ptr[q];
            for (uint64_t base = lo; base < hi; base += NT) {
              const bool in = base + tid < hi;
              const uint4 r = in ? recs[base + tid] : make_uint4(0u, 0u, 0u, 0u);
              const Trip t = open_trip(r, in);
              uint32_t cbase_m1 = 0xFFFFFFFFu;
              for (uint32_t c = 0; c * kWave < t.T; ++c) {
                const Win w = deal(t, (int)c, cbase_m1);
                uint32_t f = 0;
                if (w.valid && in_round(w.key, sub, rj)) {  // (a key of another round is not in the table)
                  uint32_t h = slot_of(w.key);
                  while (keys[h] != w.key) h = (h + 1) & (NSLOT - 1);
                  f = chw[h] >> 30;
                }
                if (f) {
                  // no in-edge (f & 1): strand 0 -> last_0_in = max(off), strand 1 -> first_0_out = min(off + 1); no out-edge (f & 2): the roles swap
                  const uint64_t abs = ((uint64_t)w.tag << 32) | w.pos;
                  const uint64_t rid = seq_of_offset(a.c_start, a.c_n_seqs, a.c_fixed_len, abs);
                  const uint32_t off = (uint32_t)(abs - (a.c_fixed_len ? rid * a.c_fixed_len : a.c_start[rid]));
                  if (f & 1u) {
                    if (w.fwd) atomicMax(&a.last_0_in_p1[rid], off + 1);
                    else atomicMin(&a.first_0_out[rid], off + 1);
                  }
                  if (f & 2u) {
                    if (w.fwd) atomicMin(&a.first_0_out[rid], off + 1);
                    else atomicMax(&a.last_0_in_p1[rid], off + 1);
                  }
                }
              }
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          keys[sl] = kEmpty;
          cnts[sl] = 0;
          chw[sl] = 0;
        }
        if (tid == 0) {
          s_bad = 0;
          s_nclaimed = 0;
          s_flagged = 0;
        }
        __syncthreads();
        sub = nsub;
        rj = nrj;
        if (done) break;
      }
    }
  }
  n_dist_all = wave_sum(n_dist_all);
  if (lane == 0 && n_dist_all) atomicAdd(a.ctr + 4, n_dist_all);
  __syncthreads();
  for (int i = tid; i < kSegHist; i += NT)
    if (lhist[i]) atomicAdd(&a.hist[i], (unsigned long long)lhist[i]);
  if (tid == 0) a.edges_counts[blockIdx.x] = s_edge_cur < a.edges_cap ? s_edge_cur : a.edges_cap;
}

// ---- host ----
bool s1_skm_applies(const mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy) {
  const SeqSet &s = c->seqs;
  const long long knob = c->opt("s1_skm", 1);
  if (!knob || want_mercy || c->global_bases || c->n_parts > 1 || c->filter_on || c->accumulate || c->pos_base) return false;
  if (k < 19 || k > 22 || m < 1 || m > 2) return false;
  if (!s.n_seqs || s.max_len < k + 1 || (s.n_bases >> 36)) return false;
  // (reads of several lengths: every read takes the blocks of the longest — while at least s1_var_min_fill per cent of them hold windows)
  if (!s.fixed_len && (double)s.n_bases * 100.0 < (double)c->opt("s1_var_min_fill", 50) * (double)s.n_seqs * s.max_len) return false;
  const char *e = getenv("MHX_S1_MARK");
  if (e && strcmp(e, "nonsolid")) return false;  // (a caller that asks for another polarity of the marks, or atomics into the bitmap)
  const uint64_t n_win = s.n_bases > s.n_seqs * (uint64_t)k ? s.n_bases - s.n_seqs * (uint64_t)k : 0;
  return knob >= 2 || n_win >= (uint64_t)c->opt("s1_skm_min_windows", 1 << 22);
}

// several GPUs (comm.hip dist_s1_skm): this rank's say — the same shape with a global layout; at most kSkmSrcMax ranks (one source per sender)
bool s1_skm_dist_applies(const mhx_ctx *c, uint32_t k, uint32_t m) {
  const SeqSet &s = c->seqs;
  if (!c->opt("s1_skm", 1) || !c->opt("dist_skm", 1) || !c->global_bases || c->n_parts > kSkmSrcMax || c->filter_on || c->accumulate) return false;
  if (k < 19 || k > 22 || m < 1 || m > 2 || (c->global_bases >> 36)) return false;
  if (!s.n_seqs || s.max_len < k + 1) return false;
  if (!s.fixed_len && (double)s.n_bases * 100.0 < (double)c->opt("s1_var_min_fill", 50) * (double)s.n_seqs * s.max_len) return false;
  if (getenv("MHX_S1_MARK") || !c->opt("dist_sparse_marks", 1)) return false;
  return s1_skm_passes(c, k) == 1;
}

// make the records, order them by bin, find the bins.  -> false: gave up (more records than the array was sized for, or a bin that one
// workgroup should not stream alone: low-complexity reads) — nothing published
int s1_skm_passes(const mhx_ctx *c, uint32_t k) {
  // the two record arrays of a pass together take s1_skm_pass_gb (hipMalloc costs per byte on this driver — seconds per 100 GB — while one
  // more pass costs one more scan of the reads by k_skm_make: 28 ms at 100 M reads); s1_skm_passes forces a count (tests)
  const SeqSet &s = c->seqs;
  if (const long long f = c->opt("s1_skm_passes", 0)) return (int)std::min<long long>(64, std::max<long long>(1, f));
  const uint64_t n_win = s.n_bases > s.n_seqs * (uint64_t)k ? s.n_bases - s.n_seqs * (uint64_t)k : 0;
  const double need = (double)n_win * (double)std::max<long long>(c->opt("s1_skm_cap_pct", 36), 1) / 100.0 * 32.0;
  const double budget = (double)std::max<long long>(c->opt("s1_skm_pass_gb", 48), 1) * 1e9;
  return (int)std::min(64.0, std::max(1.0, std::ceil(need / budget)));
}

bool s1_skm_front(mhx_ctx *c, uint32_t k, SkmFront *f, int pass, int n_passes, int bin_bits_agreed, bool for_count) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const bool var = s.fixed_len == 0;
  const uint32_t L = var ? s.max_len : s.fixed_len, bpr = (L - k + kSkmC - 1) / kSkmC;
  const uint64_t n_blocks = s.n_seqs * (uint64_t)bpr;
  const uint64_t n_win = var ? (s.n_bases > s.n_seqs * (uint64_t)k ? s.n_bases - s.n_seqs * (uint64_t)k : 0) : s.n_seqs * (uint64_t)(L - k);  // (var: an estimate)
  // the array: 0.284 records per window of random sequence at m = k - 8 (fewer in repeats); s1_skm_cap_pct per cent of the windows
  // (a pass of several: its share of the bins, which the hash fills evenly, and 15 % on top)
  const uint64_t cap_all = n_win * (uint64_t)std::max<long long>(c->opt("s1_skm_cap_pct", 36), 1) / 100;
  const uint64_t cap = std::max<uint64_t>(n_passes > 1 ? cap_all / n_passes + cap_all / n_passes * 15 / 100 : cap_all, 1u << 16);
  // bins: ~5000 records each (20 000 windows: what the table of one workgroup takes in one round) — 2^16 up to 14 M reads of 150 bases,
  // up to 2^20 and a third sort pass beyond
  int bin_bits = kSkmMinBinBits;
  while (bin_bits < kSkmMaxBinBits && (double)n_win * 0.29 / (double)(1ull << bin_bits) > 8192.0) ++bin_bits;
  if (const long long fb = c->opt("s1_skm_bin_bits", 0)) bin_bits = (int)std::min<long long>(kSkmMaxBinBits, std::max<long long>(8, fb));
  if (bin_bits_agreed > 0) bin_bits = bin_bits_agreed;  // (several GPUs: the ranks made it from the size of the whole job)
  uint4 *buf_a = c->ws("items_a", cap * 16 + 64).as<uint4>();
  uint4 *buf_b = c->ws("items_b", cap * 16 + 64).as<uint4>();
  unsigned long long *cursor = c->ws("skm_cursor", 64).as<unsigned long long>();
  uint32_t *err = reinterpret_cast<uint32_t *>(cursor + 1);
  uint32_t *max_bin = err + 1;
  MHX_HIP(hipMemsetAsync(cursor, 0, 64, st));
  // (bits 0..7 of the first word are zero in every record: all passes may rank with LDS atomics)
  std::vector<SortPass> passes;
  for (int lo = 0; lo < bin_bits; lo += 8) passes.push_back(SortPass{8 + lo, std::min(8, bin_bits - lo), 0, 0, 0});
  unsigned long long *pre_hist = c->ws("sort_pre_hist", (size_t)kMaxFusedPasses * 256 * 8).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(pre_hist, 0, (size_t)3 * 256 * 8, st));
  constexpr int NT = 512, J = 2;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  const unsigned grid = (unsigned)std::min<uint64_t>(div_ceil(n_blocks, (uint64_t)NT * J), cus * 8);
  const uint32_t n_bins = 1u << bin_bits;
  const uint32_t bin_lo = (uint32_t)((uint64_t)n_bins * pass / n_passes), bin_hi = (uint32_t)((uint64_t)n_bins * (pass + 1) / n_passes);
  // homopolymer windows are counted beside the records on one GPU (several GPUs: they stay in the records; a job with many gives the path up)
  unsigned long long *hp = nullptr;
  if (!c->global_bases && c->opt("s1_skm_hp", 1)) {
    hp = c->ws("skm_hp", 512).as<unsigned long long>();  // [0..3] stage 1: windows per class, a position each; [8 + 16 class + ..] count: windows, bases in front, bases behind
    if (pass == 0) {
      unsigned long long init[48] = {0};
      init[2] = init[3] = ~0ull;
      MHX_HIP(hipMemcpyAsync(hp, init, sizeof init, hipMemcpyHostToDevice, st));
      MHX_HIP(hipStreamSynchronize(st));  // (init is a stack variable)
    }
  }
#define MHX_MAKE(VARV, COUNTV)                                                                                                                                   \
  hipLaunchKernelGGL((k_skm_make<NT, J, VARV, COUNTV>), dim3(grid), dim3(NT), 0, st, s.words.as<uint32_t>(), s.start.as<uint64_t>(), L, bpr, n_blocks, (int)k, \
                     bin_bits, bin_lo, bin_hi, pass == 0 ? 1 : 0, c->pos_base, hp, buf_a, (unsigned long long)cap, cursor, err, pre_hist)
  MHX_LAUNCH(c, for_count ? "count_skm_make" : "s1_skm_make", (double)s.n_bases / 4 + (double)n_win * 16 / 3.5, {
    if (var && for_count) MHX_MAKE(true, true);
    else if (var) MHX_MAKE(true, false);
    else if (for_count) MHX_MAKE(false, true);
    else MHX_MAKE(false, false);
  });
#undef MHX_MAKE
  unsigned long long h[4] = {0, 0, 0, 0};
  std::vector<unsigned long long> h_dh(passes.size() * 256);
  MHX_HIP(hipMemcpyAsync(h, cursor, 32, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_dh.data(), pre_hist, h_dh.size() * 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  f->n_records = 0;
  f->hp = hp;
  if ((uint32_t)h[1] != 0 || h[0] > cap) return false;
  const uint64_t n = h[0];
  {  // a bin far over the limit shows in the digit histograms already — one value of EVERY digit stands out by its records: no need to
     // order the records to find it (low-complexity reads: 1 % of (AC)n reads put 1.7 M records behind one minimizer)
    const uint64_t n_bins_pass = std::max<uint64_t>(1, ((1ull << bin_bits) + n_passes - 1) / n_passes);
    const uint64_t limit = std::max<uint64_t>((uint64_t)c->opt("s1_skm_max_bin", 1 << 16), 16 * (n / n_bins_pass + 1));
    bool all = n > 0;
    uint64_t smallest_excess = ~0ull;
    for (size_t p = 0; p < passes.size() && all; ++p) {
      const uint64_t vals = 1ull << passes[p].bits;
      uint64_t mx = 0;
      for (uint64_t d = 0; d < vals; ++d) mx = std::max<uint64_t>(mx, h_dh[p * 256 + d]);
      const uint64_t mean = n / std::max<uint64_t>(1, (p + 1 == passes.size() && n_passes > 1) ? std::max<uint64_t>(1, vals / n_passes) : vals);
      const uint64_t excess = mx > mean ? mx - mean : 0;
      smallest_excess = std::min(smallest_excess, excess);
      all = excess >= limit + limit / 2;  // (1.5 x: the values of a digit differ by a few per cent of their mean — 256 bins each — far below the limit)
    }
    if (all) {
      f->n_records = n;
      f->max_bin = (uint32_t)std::min<uint64_t>(smallest_excess, 0xFFFFFFFFu);
      return false;
    }
  }
  c->pre_hist_buf = buf_a;  // (radix_sort: no histogram read of its own)
  c->pre_hist_n = n;
  c->pre_hist_passes = (int)passes.size();
  c->pre_hist_sig = passes_signature(passes);
  uint32_t *sorted = radix_sort(c, reinterpret_cast<uint32_t *>(buf_a), reinterpret_cast<uint32_t *>(buf_b), n, 4, 1, passes);
  c->pre_hist_buf = nullptr;
  uint64_t *bounds = c->ws("s1_bucket_bounds", ((size_t)n_bins + 1) * 8 + 64).as<uint64_t>();
  MHX_LAUNCH(c, "s1_skm_bounds", (double)n_bins * 8 * 30,
             hipLaunchKernelGGL(k_skm_bounds, dim3((n_bins + 1 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4 *>(sorted), n, n_bins, bounds, max_bin));
  uint32_t h_max = 0;
  MHX_HIP(hipMemcpyAsync(&h_max, max_bin, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  f->n_src = 1;
  f->src[0] = reinterpret_cast<const uint4 *>(sorted);
  f->src_bounds[0] = bounds;
  f->spare = sorted == reinterpret_cast<uint32_t *>(buf_a) ? reinterpret_cast<uint32_t *>(buf_b) : reinterpret_cast<uint32_t *>(buf_a);
  f->spare_bytes = cap * 16;
  f->n_records = n;
  f->n_windows = n_win;
  if (pass == 0) f->n_items = var ? h[3] : s.n_seqs * (uint64_t)(L - k + (for_count ? 0 : 4));
  f->n_bins = n_bins;
  f->bin_lo = bin_lo;
  f->bin_hi = bin_hi;
  f->bin_bits = bin_bits;
  f->max_bin = h_max;
  // a bin of many times the mean is low-complexity sequence (one minimizer for millions of windows): the prefix plan has the giant path
  const uint64_t limit = std::max<uint64_t>((uint64_t)c->opt("s1_skm_max_bin", 1 << 16), 16 * (n / (bin_hi - bin_lo) + 1));
  return h_max <= limit;
}

void s1_skm_groups_launch(mhx_ctx *c, bool agg, unsigned grid, const SkmFront &f, uint32_t k, uint32_t m, uint8_t *solid_bytes, unsigned long long *hist,
                          uint2 *agg_raw, uint32_t agg_cap, uint32_t *agg_counts, uint32_t *err, unsigned long long *marks_raw, uint32_t marks_cap,
                          uint32_t *marks_counts) {
  hipStream_t st = c->stream;
  uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
  const uint32_t nslot = 1u << kSkmLogSlots;
  SkmArgs a{(int)k, m, solid_bytes, hist, agg_raw, agg_cap, agg_counts, err,
            (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", nslot * 7 / 8), 1), nslot), (int)std::min<long long>(c->opt("s1_stream_probes", 1024), 1024),
            f.bin_lo, f.bin_hi, marks_raw, marks_cap, marks_counts};
  SkmSrcs srcs{};
  srcs.n = f.n_src;
  for (int q = 0; q < f.n_src; ++q) {
    srcs.ptr[q] = f.src[q];
    srcs.bounds[q] = f.src_bounds[q];
  }
  const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
  const bool tags = (n_bits >> 32) != 0 || c->opt("s1_skm_tags", 0) != 0;
  const bool deal = c->opt("s1_skm_deal", 1) != 0;
#define MHX_SKM(AGGV, TAGV)                                                                                                  \
  do {                                                                                                                       \
    if (deal) hipLaunchKernelGGL((k_s1_skm<AGGV, TAGV, true>), dim3(grid), dim3(kSkmThreads), 0, st, srcs, a, ticket);        \
    else hipLaunchKernelGGL((k_s1_skm<AGGV, TAGV, false>), dim3(grid), dim3(kSkmThreads), 0, st, srcs, a, ticket);            \
  } while (0)
  MHX_LAUNCH(c, "s1_skm_groups", (double)f.n_records * 16, {
    if (agg && tags) MHX_SKM(true, true);
    else if (agg) MHX_SKM(true, false);
    else if (tags) MHX_SKM(false, true);
    else MHX_SKM(false, false);
  });
#undef MHX_SKM
}

// ---- `count` on super-k-mer records: one GPU, 19 <= k <= 21 (k + 10 bases and two flags in a record), min count <= 2, one pass ----
bool count_skm_applies(const mhx_ctx *c, uint32_t k, uint32_t m) {
  const SeqSet &s = c->seqs;
  const long long knob = c->opt("count_skm", 1);
  if (!knob || !c->opt("s1_skm", 1) || c->global_bases || c->n_parts > 1 || c->filter_on || c->accumulate || c->pos_base || c->count_edges_only) return false;
  if (k < 19 || k > 21 || m < 1 || m > 2) return false;
  if (!s.n_seqs || s.max_len < k + 1 || (s.n_bases >> 36)) return false;
  if (!s.fixed_len && (double)s.n_bases * 100.0 < (double)c->opt("s1_var_min_fill", 50) * (double)s.n_seqs * s.max_len) return false;
  const uint64_t n_win = s.n_bases > s.n_seqs * (uint64_t)k ? s.n_bases - s.n_seqs * (uint64_t)k : 0;
  return knob >= 2 || c->opt("s1_skm", 1) >= 2 || n_win >= (uint64_t)c->opt("s1_skm_min_windows", 1 << 22);
}
// -> false: gave up (record array, a bin of low-complexity reads, an edge region): the caller's arrays may hold partial results of this attempt
// (pass of n_passes: a job whose record arrays would take more than s1_skm_pass_gb runs in passes over ranges of bins, as stage 1 does; the
//  caller packs every pass's edge regions behind the earlier passes' edges and orders them once at the end)
bool count_skm_groups(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *first_0_out, uint32_t *last_0_in_p1, unsigned long long *hist, CountStreamOut *o, bool *touched,
                      int pass, int n_passes) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  SkmFront f{};
  c->gen_first_pass = nullptr;
  c->pre_hist_buf = nullptr;
  if (!s1_skm_front(c, k, &f, pass, n_passes, 0, true)) return false;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  const unsigned grid = (unsigned)std::min<uint64_t>(cus, std::max<uint64_t>(1, div_ceil((uint64_t)(f.bin_hi - f.bin_lo), (uint64_t)kSkmBatch)));
  const uint32_t region = (uint32_t)std::min<uint64_t>(f.spare_bytes / 8 / grid, 0xFFFFFFF0u);
  uint32_t *counts = c->ws("cs_edge_counts", (size_t)grid * 4).as<uint32_t>();
  unsigned long long *ctr = c->ws("s1_counters", 64).as<unsigned long long>();
  uint32_t *seg_err = c->ws("s1_seg_err", 64).as<uint32_t>();
  uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  MHX_HIP(hipMemsetAsync(seg_err, 0, 4, st));
  MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
  const uint32_t nslot = 1u << kSkmLogSlots;
  CountSkmArgs a{(int)k, m, hist, ctr, reinterpret_cast<unsigned long long *>(f.spare), region, counts, seg_err,
                 (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", nslot * 7 / 8), 1), nslot), (int)std::min<long long>(c->opt("s1_stream_probes", 1024), 1024),
                 f.bin_lo, f.bin_hi, s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, first_0_out, last_0_in_p1};
  SkmSrcs srcs{};
  srcs.n = 1;
  srcs.ptr[0] = f.src[0];
  srcs.bounds[0] = f.src_bounds[0];
  const bool tags = (s.n_bases >> 32) != 0 || c->opt("s1_skm_tags", 0) != 0;
  *touched = true;
  const bool g2 = c->opt("count_skm_group", 2) == 2;  // (four chunks per round of compare-and-swaps spill registers here: measured 13.6 against 13.1 ms)
  MHX_LAUNCH(c, "count_skm_groups", (double)f.n_records * 16, {
    if (tags && g2) hipLaunchKernelGGL((k_count_skm<true, 2>), dim3(grid), dim3(kSkmThreads), 0, st, srcs, a, ticket);
    else if (tags) hipLaunchKernelGGL((k_count_skm<true, 4>), dim3(grid), dim3(kSkmThreads), 0, st, srcs, a, ticket);
    else if (g2) hipLaunchKernelGGL((k_count_skm<false, 2>), dim3(grid), dim3(kSkmThreads), 0, st, srcs, a, ticket);
    else hipLaunchKernelGGL((k_count_skm<false, 4>), dim3(grid), dim3(kSkmThreads), 0, st, srcs, a, ticket);
  });
  uint32_t e = 0;
  unsigned long long h_ctr[8] = {0};
  MHX_HIP(hipMemcpyAsync(&e, seg_err, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_ctr, ctr, 64, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (e) return false;
  o->grid = grid;
  o->cap = region;
  o->counts = counts;
  o->spare = f.spare;
  o->sorted = nullptr;
  if (pass == 0) o->n_items = f.n_items;
  o->n_distinct = h_ctr[4];
  o->events = nullptr;
  o->n_events = 0;
  o->skm_hp = f.hp;
  o->skm_records = f.n_records;
  o->skm_windows = f.n_windows;
  o->skm_max_bin = f.max_bin;
  o->skm_bin_bits = f.bin_bits;
  return true;
}

// count: the homopolymer keys -> histogram, edges behind `edges_at`.  -> edges appended, keys, and whether a solid one lacks an in- or out-edge
void count_skm_hp_publish(mhx_ctx *c, const unsigned long long *hp, uint32_t k, uint32_t m, unsigned long long *hist, unsigned long long *edges_at,
                          uint64_t *n_edges, uint64_t *n_keys, bool *flagged) {
  unsigned long long *res = c->ws("skm_hp_res", 64).as<unsigned long long>();
  hipLaunchKernelGGL(k_count_hp_publish, dim3(1), dim3(64), 0, c->stream, hp, (int)k, m, hist, edges_at, res);
  unsigned long long h[3] = {0, 0, 0};
  MHX_HIP(hipMemcpyAsync(h, res, 24, hipMemcpyDeviceToHost, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  *n_edges = h[0];
  *n_keys = h[1];
  *flagged = h[2] != 0;
}

// the homopolymer keys counted by k_skm_make -> histogram, mark or aggregated items (after the last pass of the group-by)
void s1_skm_hp_publish(mhx_ctx *c, const SkmFront &f, uint32_t k, uint32_t m, uint8_t *solid_bytes, unsigned long long *hist, uint2 *agg_items,
                       uint64_t *agg_cursor, bool agg) {
  if (!f.hp) return;
  MHX_LAUNCH(c, "s1_skm_hp", 64.0, hipLaunchKernelGGL(k_skm_hp_publish, dim3(1), dim3(64), 0, c->stream, f.hp, (int)k, m, solid_bytes, hist, agg_items, agg_cursor,
                                                       agg ? 1 : 0));
}

// several GPUs: where the bins start in an array of records ordered by bin (a source of the owner's group-by)
void s1_skm_bounds_of(mhx_ctx *c, const uint4 *recs, uint64_t n, uint32_t n_bins, uint64_t *bounds) {
  uint32_t *scratch = c->ws("skm_cursor", 64).as<uint32_t>() + 8;
  hipLaunchKernelGGL(k_skm_bounds, dim3((n_bins + 1 + 255) / 256), dim3(256), 0, c->stream, recs, n, n_bins, bounds, scratch);
}

}  // namespace mhx
