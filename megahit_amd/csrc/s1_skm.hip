// read2sdbg stage 1 on super-k-mer records (round 6): the no-mercy reduction of Read2SdbgS1 (reference src/sorting/read_to_sdbg_s1.cpp:208-366
// makes one sort item per (k-1)-mer window of every read, :368-464 counts the groups of equal items) needs the (k+1)-mers GROUPED, not
// sorted: is_solid, the multiplicity histogram and the aggregated stage-2 items are functions of the multiset of canonical (k+1)-mers.
// So the grouping key need not be a prefix of the k-mer.  Here it is the MINIMIZER of the (k+1)-mer — the smallest hash of a canonical
// m-mer inside it — which consecutive windows of a read share: a run of windows with one minimizer (a "super-k-mer") leaves the read as
// ONE 16-byte record (its bases, where it starts, how many windows it holds) instead of one 12-byte record per window.
//   k_skm_make    one thread per aligned block of 8 windows: 17 canonical m-mer hashes from one 64-bit window of the store and its
//                 reverse complement, the 8 window minima by a suffix / prefix scan, a record per run of equal bins (3.5 windows per
//                 record at k = 21) -> the record array through one cursor (one atomic per workgroup and trip)
//   radix_sort    two passes over the 16-bit bin (the chained-scan passes of sort.hip on 16-byte records)
//   k_skm_bounds  where each bin starts
//   k_s1_skm      one workgroup per bin at a time: the windows of its records are dealt to the lanes one by one (a wavefront's 64
//                 records hold ~225 windows: every lane takes the g-th, finds its record through a bitmap of run heads and two shuffles),
//                 canonical key (read_to_sdbg_s1.cpp:228-292: the strand of the (k-1)-mer, then head / tail), LDS table with 64-bit
//                 keys, one walk over the table for the histogram (:430-436), the marks of the non-solid occurrences (:464, inverted:
//                 a key of count 1 < m has one record, whose position sits next to the key) and the aggregated stage-2 items.
// 5.9 GB of records at 10 M reads instead of 16: the two sort passes and the read of the group-by move 2.7 x fewer bytes.
// Shapes: one GPU, no bucket filter, reads of one length, min count <= 2, 19 <= k <= 22, positions below 2^32; everything else — and
// any bin that outgrows what one workgroup should stream (low-complexity reads) — takes the prefix plan (s1_stream.hip).
#include <cstring>

#include "s1_shared.h"

namespace mhx {

constexpr int kSkmC = 8;                     // windows per aligned block = the longest run a record holds
constexpr int kSkmW = 10;                    // m-mers per window: m = k + 1 - 9
constexpr int kSkmNM = kSkmC + kSkmW - 1;    // m-mers a block looks at
constexpr int kSkmBinBits = 16;

__device__ __forceinline__ uint32_t skm_mix(uint32_t c) {  // (a bijection of 32 bits: the order of the m-mers)
  uint32_t h = c * 0x9E3779B1u;
  h ^= h >> 15;
  return h * 0x85EBCA6Bu;
}
__device__ __forceinline__ uint32_t skm_bin_of(uint32_t minh) {  // (the minimum of ten hashes is small: mixed once more before its top bits are taken)
  uint32_t h = minh ^ (minh >> 16);
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  return (h * 0x846CA68Bu) >> (32 - kSkmBinBits);
}

// record: w0 = bin << 8 | (windows - 1) << 24 (bits 0..7 zero: the first sort pass may rank with LDS atomics, sort_kernels.h RANK 2),
// w1:w2 = the bases of the run, MSB first (k + windows of them), w3 = position of the run's first base in the store
template <int NT, int J>
__global__ __launch_bounds__(NT) void k_skm_make(const uint32_t *__restrict__ seq, uint32_t L, uint32_t nwin, uint32_t bpr, uint64_t n_blocks, int k,
                                                 uint4 *__restrict__ out, unsigned long long cap, unsigned long long *__restrict__ cursor,
                                                 uint32_t *__restrict__ err) {
  __shared__ uint32_t sm_scan[NT / kWave + 1];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x;
  const int M = k + 1 - (kSkmW - 1);
  const uint32_t mmask = (1u << (2 * M)) - 1u;
  const int K1 = k + 1;
  for (uint64_t it = blockIdx.x; it * (uint64_t)(NT * J) < n_blocks; it += gridDim.x) {
    uint64_t Wv[J];
    uint32_t binp[J][4], smask[J], pos0[J], nv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint64_t b = it * (uint64_t)(NT * J) + (uint64_t)j * NT + tid;
      smask[j] = 0;
      nv[j] = 0;
      Wv[j] = 0;
      pos0[j] = 0;
#pragma unroll
      for (int x = 0; x < 4; ++x) binp[j][x] = 0;
      if (b < n_blocks) {
        const uint64_t r = b / bpr;
        const uint32_t q0 = (uint32_t)(b - r * bpr) * kSkmC;
        const uint64_t a = r * L + q0;
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
        bins[0] = skm_bin_of(sfx[0]);
#pragma unroll
        for (int w = 1; w < kSkmC; ++w) {
          pfx = min(pfx, h[kSkmW - 1 + w]);
          bins[w] = skm_bin_of(min(sfx[w], pfx));
        }
        const uint32_t n_here = min((uint32_t)kSkmC, nwin - q0);
        uint32_t sm = 1u;
#pragma unroll
        for (int w = 1; w < kSkmC; ++w)
          if ((uint32_t)w < n_here && bins[w] != bins[w - 1]) sm |= 1u << w;
        smask[j] = sm;
        nv[j] = n_here;
        Wv[j] = W;
        pos0[j] = (uint32_t)a;
#pragma unroll
        for (int x = 0; x < 4; ++x) binp[j][x] = bins[2 * x] | (bins[2 * x + 1] << 16);
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
    if (base + total > cap) {  // (uniform) the array is sized for one record per two windows: the host takes the prefix plan
      if (tid == 0) atomicOr(err, 1u);
      return;
    }
    unsigned long long at = base + excl;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      uint32_t sm = smask[j];
      while (sm) {
        const int s = __builtin_ctz(sm);
        sm &= sm - 1;
        const int e = sm ? __builtin_ctz(sm) : (int)nv[j];
        const int len = e - s;
        const int nb = K1 + len - 1;
        const uint64_t bases = (Wv[j] << (2 * s)) & (~0ull << (64 - 2 * nb));
        const uint32_t pw = s < 2 ? binp[j][0] : (s < 4 ? binp[j][1] : (s < 6 ? binp[j][2] : binp[j][3]));
        const uint32_t bin = (pw >> ((s & 1) * 16)) & 0xFFFFu;
        out[at++] = make_uint4((bin << 8) | ((uint32_t)(len - 1) << 24), (uint32_t)(bases >> 32), (uint32_t)bases, pos0[j] + (uint32_t)s);
      }
    }
  }
}

// bounds[b] = the first record whose bin is >= b (b = 0 .. n_bins); a thread per bin, a binary search each
__global__ __launch_bounds__(256) void k_skm_bounds(const uint4 *__restrict__ recs, uint64_t n, uint32_t n_bins, uint64_t *__restrict__ bounds,
                                                    uint32_t *__restrict__ max_bin) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b > n_bins) return;
  auto lower = [&](uint32_t v) -> uint64_t {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const uint32_t bin = (reinterpret_cast<const uint32_t *>(recs + mid)[0] >> 8) & 0xFFFFu;
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
  uint32_t n_bins;
};

constexpr int kSkmThreads = 1024, kSkmLogSlots = 13, kSkmBatch = 4;

template <bool AGG>
__global__ __launch_bounds__(kSkmThreads) void k_s1_skm(const uint4 *__restrict__ recs, const uint64_t *__restrict__ bounds, SkmArgs a,
                                                        uint32_t *__restrict__ ticket) {
  constexpr int NT = kSkmThreads, NSLOT = 1 << kSkmLogSlots, NW = NT / kWave, W = NSLOT / NT;
  constexpr unsigned long long kEmpty = ~0ull;  // never a key: head / tail bits 63 do not occur
  constexpr int NLIST = 1024;
  __shared__ unsigned long long keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  __shared__ uint32_t fpos[NSLOT];
  __shared__ uint32_t lhist[kSegHist];
  __shared__ unsigned long long slist_k[AGG ? NLIST : 1];
  __shared__ uint32_t slist_c[AGG ? NLIST : 1];
  __shared__ uint32_t heads[NW][16];  // per wavefront: bit g set = window g of the trip is the first of its record
  __shared__ uint32_t s_bad, s_nclaimed, s_list_n, s_agg_cur, s_tk;
  __shared__ uint64_t s_lo[kSkmBatch + 1];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
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
    s_bad = 0;
    s_nclaimed = 0;
    s_list_n = 0;
    s_agg_cur = 0;
  }
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

  for (;;) {
    if (tid == 0) s_tk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint64_t bin0 = (uint64_t)s_tk * kSkmBatch;
    if (bin0 >= a.n_bins) break;
    if (tid <= kSkmBatch) s_lo[tid] = bounds[min(bin0 + tid, (uint64_t)a.n_bins)];
    __syncthreads();
    for (int bb = 0; bb < kSkmBatch; ++bb) {
      const uint64_t lo = s_lo[bb], hi = s_lo[bb + 1];
      if (lo == hi) continue;
      // the bin in rounds: round (sub, rj) takes the keys whose top `sub` bits of a second hash are rj (a round that overflows the
      // table is redone in two halves)
      uint32_t sub = 0, rj = 0;
      for (;;) {
        uint32_t seen = 0;
        // A: insert
        uint4 nxt = lo + tid < hi ? recs[lo + tid] : make_uint4(0u, 0u, 0u, 0u);
        for (uint64_t base = lo; base < hi; base += NT) {
          const uint4 r = nxt;
          const bool in = base + tid < hi;
          if (base + NT + tid < hi) nxt = recs[base + NT + tid];
          if (seen > a.max_fill) continue;  // (uniform per wavefront; the round is redone in halves anyway)
          const uint32_t len = in ? ((r.x >> 24) & 7u) + 1u : 0u;
          const uint32_t incl = wave_inclusive_sum(len);
          const uint32_t T = __shfl(incl, kWave - 1, kWave);
          const uint32_t start = incl - len;
          if (lane < 16) heads[wv][lane] = 0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (len) atomicOr(&heads[wv][start >> 5], 1u << (start & 31u));
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          uint32_t cbase = 0, claims = 0;
          for (uint32_t g0 = 0; g0 < T; g0 += kWave) {
            const uint32_t g = g0 + lane;
            const volatile uint32_t *hw = &heads[wv][(g0 >> 5)];
            const uint64_t word = (uint64_t)hw[0] | ((uint64_t)hw[1] << 32);
            const uint64_t le = lane == kWave - 1 ? ~0ull : ((2ull << lane) - 1ull);
            const bool valid = g < T;
            const uint32_t o = valid ? cbase + (uint32_t)__builtin_popcountll(word & le) - 1u : 0u;
            cbase += (uint32_t)__builtin_popcountll(word);
            const uint32_t bh = __shfl(r.y, (int)o, kWave), bl = __shfl(r.z, (int)o, kWave), ps = __shfl(r.w, (int)o, kWave);
            const uint32_t so = __shfl(start, (int)o, kWave);
            const uint32_t j = g - so;
            const uint64_t win = (((uint64_t)bh << 32) | bl) << (2 * j);  // the (k+1)-mer head.S.tail, MSB first
            const uint64_t f = (win << 2) & kmask;
            const uint64_t rc = rc64(f, k - 1);
            const unsigned head = (unsigned)(win >> 62), tail = (unsigned)(win >> (62 - 2 * k)) & 3u;
            const int strand = f > rc ? 1 : (f < rc ? 0 : (head <= 3 - tail ? 0 : 1));
            const unsigned long long key = strand ? (rc | ((uint64_t)(3u - tail) << 3) | (3u - head)) : (f | ((uint64_t)head << 3) | tail);
            const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
            const uint32_t h2 = (klo * 0xC2B2AE35u) ^ (khi * 0x27D4EB2Fu);
            bool pend = valid && (sub == 0 || (skm_mix(h2) >> (32 - sub)) == rj);
            uint32_t hh = ((klo * 0x9E3779B1u) ^ (khi * 0x85EBCA6Bu)) >> (32 - kSkmLogSlots);
            if (a.probe_limit <= 0 && pend) {
              s_bad = 1;
              pend = false;
            }
            int turns = 0;
            while (__ballot(pend)) {
              if (pend) {
                const unsigned long long old = atomicCAS(&keys[hh], kEmpty, key);
                if (old == kEmpty) {
                  fpos[hh] = ps + j;
                  atomicAdd(&cnts[hh], 1u);
                  ++claims;
                  pend = false;
                } else if (old == key) {
                  atomicAdd(&cnts[hh], 1u);
                  pend = false;
                } else {
                  hh = (hh + 1) & (NSLOT - 1);
                }
              }
              if (++turns > a.probe_limit) {
                if (pend) s_bad = 1;
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
        // C: one walk over the table — per distinct key: histogram, the mark of a non-solid key's only record, the solid keys listed
        unsigned long long wk[W];
        uint32_t wc[W], wp[W];
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
          keys[sl] = kEmpty;
          cnts[sl] = 0;
        }
        uint32_t want_bits = 0;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          if (wk[it] != kEmpty && !bad) {
            const uint32_t cnt = wc[it];
            const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;  // :430-436
            if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
            else atomicAdd(&a.hist[hb], 1ull);
            if (cnt < m) a.solid_bytes[wp[it]] = 1;  // count 1 < m <= 2: the key's only record is a non-solid occurrence
            else if (AGG) want_bits |= 1u << it;
          }
        }
        if constexpr (AGG) {
          const uint32_t n_w = (uint32_t)__builtin_popcount(want_bits);
          const uint32_t incl = wave_inclusive_sum(n_w);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t lbase = 0;
            if (lane == 0) lbase = atomicAdd(&s_list_n, tot);
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
          const uint32_t n_list = min(s_list_n, (uint32_t)NLIST);
          for (uint32_t base = 0; base < n_list; base += NT) {
            const uint32_t i = base + tid;
            const bool v = i < n_list;
            emit_items(v ? slist_k[i] : 0ull, v ? slist_c[i] : 0u, true, v);
          }
        }
        __syncthreads();
        if (tid == 0) {
          s_bad = 0;
          s_nclaimed = 0;
          s_list_n = 0;
        }
        __syncthreads();
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
}

// ---- host ----
bool s1_skm_applies(const mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy) {
  const SeqSet &s = c->seqs;
  const long long knob = c->opt("s1_skm", 1);
  if (!knob || want_mercy || c->global_bases || c->n_parts > 1 || c->filter_on || c->accumulate || c->pos_base) return false;
  if (k < 19 || k > 22 || m < 1 || m > 2) return false;
  if (!s.n_seqs || s.fixed_len < k + 1 || s.n_bases >= (1ull << 32)) return false;
  const char *e = getenv("MHX_S1_MARK");
  if (e && strcmp(e, "nonsolid")) return false;  // (a caller that asks for another polarity of the marks, or atomics into the bitmap)
  const uint64_t n_win = s.n_seqs * (uint64_t)(s.fixed_len - k);
  return knob >= 2 || n_win >= (uint64_t)c->opt("s1_skm_min_windows", 1 << 22);
}

// make the records, order them by bin, find the bins.  -> false: gave up (more records than the array was sized for, or a bin that one
// workgroup should not stream alone: low-complexity reads) — nothing published
bool s1_skm_front(mhx_ctx *c, uint32_t k, SkmFront *f) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint32_t L = s.fixed_len, nwin = L - k, bpr = (nwin + kSkmC - 1) / kSkmC;
  const uint64_t n_blocks = s.n_seqs * (uint64_t)bpr, n_win = s.n_seqs * (uint64_t)nwin;
  const uint64_t cap = std::max<uint64_t>(n_win / 2, 1u << 16);
  uint4 *buf_a = c->ws("items_a", cap * 16 + 64).as<uint4>();
  uint4 *buf_b = c->ws("items_b", cap * 16 + 64).as<uint4>();
  unsigned long long *cursor = c->ws("skm_cursor", 64).as<unsigned long long>();
  uint32_t *err = reinterpret_cast<uint32_t *>(cursor + 1);
  uint32_t *max_bin = err + 1;
  MHX_HIP(hipMemsetAsync(cursor, 0, 64, st));
  constexpr int NT = 512, J = 2;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  const unsigned grid = (unsigned)std::min<uint64_t>(div_ceil(n_blocks, (uint64_t)NT * J), cus * 8);
  MHX_LAUNCH(c, "s1_skm_make", (double)s.n_bases / 4 + (double)n_win * 16 / 3.5,
             hipLaunchKernelGGL((k_skm_make<NT, J>), dim3(grid), dim3(NT), 0, st, s.words.as<uint32_t>(), L, nwin, bpr, n_blocks, (int)k, buf_a,
                                (unsigned long long)cap, cursor, err));
  unsigned long long h[2] = {0, 0};
  MHX_HIP(hipMemcpyAsync(h, cursor, 16, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if ((uint32_t)h[1] != 0 || h[0] > cap) return false;
  const uint64_t n = h[0];
  std::vector<SortPass> passes(2);
  passes[0] = SortPass{8, 8, 0, 0, 0};
  passes[1] = SortPass{16, 8, 0, 0, 0};
  uint32_t *sorted = radix_sort(c, reinterpret_cast<uint32_t *>(buf_a), reinterpret_cast<uint32_t *>(buf_b), n, 4, 1, passes);
  const uint32_t n_bins = 1u << kSkmBinBits;
  uint64_t *bounds = c->ws("s1_bucket_bounds", ((size_t)n_bins + 1) * 8 + 64).as<uint64_t>();
  MHX_LAUNCH(c, "s1_skm_bounds", (double)n_bins * 8 * 30,
             hipLaunchKernelGGL(k_skm_bounds, dim3((n_bins + 1 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4 *>(sorted), n, n_bins, bounds, max_bin));
  uint32_t h_max = 0;
  MHX_HIP(hipMemcpyAsync(&h_max, max_bin, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  f->sorted = reinterpret_cast<const uint4 *>(sorted);
  f->spare = sorted == reinterpret_cast<uint32_t *>(buf_a) ? reinterpret_cast<uint32_t *>(buf_b) : reinterpret_cast<uint32_t *>(buf_a);
  f->spare_bytes = cap * 16;
  f->n_records = n;
  f->n_windows = n_win;
  f->bounds = bounds;
  f->n_bins = n_bins;
  f->max_bin = h_max;
  // a bin of many times the mean is low-complexity sequence (one minimizer for millions of windows): the prefix plan has the giant path
  const uint64_t limit = std::max<uint64_t>((uint64_t)c->opt("s1_skm_max_bin", 1 << 16), 16 * (n / n_bins + 1));
  return h_max <= limit;
}

void s1_skm_groups_launch(mhx_ctx *c, bool agg, unsigned grid, const SkmFront &f, uint32_t k, uint32_t m, uint8_t *solid_bytes, unsigned long long *hist,
                          uint2 *agg_raw, uint32_t agg_cap, uint32_t *agg_counts, uint32_t *err) {
  hipStream_t st = c->stream;
  uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
  const uint32_t nslot = 1u << kSkmLogSlots;
  SkmArgs a{(int)k, m, solid_bytes, hist, agg_raw, agg_cap, agg_counts, err,
            (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", nslot * 7 / 8), 1), nslot), (int)std::min<long long>(c->opt("s1_stream_probes", 1024), 1024),
            f.n_bins};
  MHX_LAUNCH(c, "s1_skm_groups", (double)f.n_records * 16, {
    if (agg) hipLaunchKernelGGL((k_s1_skm<true>), dim3(grid), dim3(kSkmThreads), 0, st, f.sorted, f.bounds, a, ticket);
    else hipLaunchKernelGGL((k_s1_skm<false>), dim3(grid), dim3(kSkmThreads), 0, st, f.sorted, f.bounds, a, ticket);
  });
}

}  // namespace mhx
