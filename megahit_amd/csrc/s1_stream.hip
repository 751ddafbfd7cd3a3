// read2sdbg stage 1 and `count`, bucket streaming: the no-mercy reduction of Read2SdbgS1::Lv2Postprocess (reference
// src/sorting/read_to_sdbg_s1.cpp:368-464) and KmerCounter::Lv2Postprocess (src/sorting/kmer_counter.cpp:254-381) as an LDS
// group-by per bucket of the sort plan's prefix, with giant buckets cut into slices.
#include "s1_shared.h"
#include "tile_groups.h"  // (MHX_TT: the phase clocks of the timing build)

namespace mhx {

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

// local key of a record inside a bucket of the pbits-bit prefix: the (k-1)-mer bits below the prefix, then head/tail.  32 bits hold it
// up to k = 22 at a 16-bit prefix; wider (k-1)-mers (k <= 29: the 12-byte record's two key words) take the 64-bit form (K64, round 6)
template <bool K64>
struct StreamKey {
  typedef uint32_t T;
  static constexpr T kEmpty = kStreamEmpty;  // never a key: head/tail bits 63 do not occur
};
template <>
struct StreamKey<true> {
  typedef unsigned long long T;
  static constexpr T kEmpty = ~0ull;
};
template <bool K64>
__device__ __forceinline__ typename StreamKey<K64>::T s1_stream_local_key(uint32_t w0, uint32_t w1, int k, int pbits) {
  const int rem = 2 * (k - 1) - pbits, mer_sh = 64 - 2 * (k - 1);
  const uint64_t key = ((uint64_t)w0 << 32) | w1;
  typedef typename StreamKey<K64>::T T;
  const T lo = (T)(key >> mer_sh);
  return (rem ? (lo & (((T)1 << rem) - 1)) << 6 : (T)0) | (T)(w1 & 63u);
}
template <bool K64>
__device__ __forceinline__ uint32_t stream_hash(typename StreamKey<K64>::T lk, int logs) {
  if constexpr (K64) return (((uint32_t)lk * 0x9E3779B1u) ^ ((uint32_t)(lk >> 32) * 0x85EBCA6Bu)) >> (32 - logs);
  else return (lk * 0x9E3779B1u) >> (32 - logs);
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
// COUNT (round 6): the slices of a giant bucket of `count`'s records.  A key's partial entry then carries, instead of its first record's
// position, what KmerCounter::Lv2Postprocess needs of the key's prev / next chars (kmer_counter.cpp:283-305): per char a 4-bit counter
// that stops at 15 (prev char x: bits [4x, 4x + 4), next char x: [16 + 4x, 20 + 4x)) — enough for any min count the streaming form takes.
template <bool K64, bool COUNT>
__global__ __launch_bounds__(256) void k_s1_giant_reduce(const uint32_t *__restrict__ items0, const uint32_t *const *__restrict__ srcs,
                                                         const uint64_t *__restrict__ bounds, int n_src, uint32_t n_buckets, int pbits, int k, S1Giant g) {
  constexpr int NS = 4096, NT = 256, kFlushAt = NS / 2;
  typedef typename StreamKey<K64>::T KeyT;
  constexpr KeyT kEmpty = StreamKey<K64>::kEmpty;
  __shared__ KeyT keys[NS];
  __shared__ uint32_t cnts[NS], fidx[NS];
  __shared__ uint32_t cinfo[COUNT ? NS : 1];
  __shared__ uint32_t s_claims, s_out, s_start, s_stop;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const size_t bstride = (size_t)n_buckets + 1;
  const uint32_t n_g = min(g.ctr[0], g.gcap);
  for (int i = tid; i < NS; i += NT) {
    keys[i] = kEmpty;
    cnts[i] = 0;
    if (COUNT) cinfo[i] = 0;
  }
  if (tid == 0) s_claims = 0;
  __syncthreads();
  // count: the (k+1)-mer below the prefix is the key (no head / tail field)
  const int c_rem = 2 * (k + 1) - pbits, c_sh = 64 - 2 * (k + 1);
  const unsigned long long c_mask = c_rem >= 64 ? ~0ull : (c_rem > 0 ? (1ull << c_rem) - 1ull : 0ull);
  // saturating add of per-char counts (one byte per char in `pv` / `nx`) to a slot's eight 4-bit counters
  auto add_chars = [&](uint32_t h, uint32_t pv, uint32_t nx) {
    uint32_t cur = __hip_atomic_load(&cinfo[COUNT ? h : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      uint32_t nv = cur;
#pragma unroll
      for (unsigned x = 0; x < 4; ++x) {
        const uint32_t fp = (cur >> (4 * x)) & 15u, fn = (cur >> (16 + 4 * x)) & 15u;
        nv += (min(fp + ((pv >> (8 * x)) & 0xFFu), 15u) - fp) << (4 * x);
        nv += (min(fn + ((nx >> (8 * x)) & 0xFFu), 15u) - fn) << (16 + 4 * x);
      }
      if (nv == cur) break;
      const uint32_t got = atomicCAS(&cinfo[COUNT ? h : 0], cur, nv);
      if (got == cur) break;
      cur = got;
    }
  };
  for (uint32_t gi = 0; gi < n_g; ++gi) {
    const uint32_t ns = g.ns[gi];
    if (!ns) continue;
    const uint32_t b = g.bucket[gi], sl_len = g.sl[gi], cap = g.cap[gi];
    uint4 *const region = g.partial + g.off[gi];
    for (uint32_t sl = blockIdx.x; sl < ns; sl += gridDim.x) {
      // (a giant whose region overflowed gave its bucket back to the streaming launch: its other slices are not worth reducing.  The
      //  flag is read by one thread and handed out through LDS: every wave of the workgroup takes the same way.)
      if (tid == 0) s_stop = ((const volatile uint8_t *)g.flag)[b] == 0;
      __syncthreads();
      const bool given_back = s_stop != 0;
      __syncthreads();
      if (given_back) break;
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
        for (int i = tid; i < NS; i += NT) mine += keys[i] != kEmpty;
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
          s_stop = (uint64_t)start + tot > (uint64_t)cap;
          if (s_stop) g.flag[b] = 0;
          s_claims = 0;
        }
        __syncthreads();
        uint32_t at = s_start + wbase + incl - mine;
        const bool write = !s_stop;
        for (int i = tid; i < NS; i += NT) {
          const KeyT key = keys[i];
          if (key != kEmpty) {
            if (write) {
              const uint32_t *r = src + (lo + fidx[i]) * 3;
              region[at++] = make_uint4(r[0], r[1], COUNT ? cinfo[i] : r[2], cnts[i]);
            }
            keys[i] = kEmpty;
            cnts[i] = 0;
            if (COUNT) cinfo[i] = 0;
          }
        }
        __syncthreads();
      };
      bool stop = false;
      for (uint64_t base = lo; base < hi && !stop; base += NT) {
        const uint64_t idx = base + tid;
        const bool in = idx < hi;
        KeyT lk = 0;
        uint32_t w1v = 0;
        if (in) {
          const uint32_t *r = src + idx * 3;
          w1v = r[1];
          if constexpr (COUNT) lk = (KeyT)(((((unsigned long long)r[0] << 32) | w1v) >> c_sh) & c_mask);
          else lk = s1_stream_local_key<K64>(r[0], w1v, k, pbits);
        }
        // a wavefront whose records all carry one key (poly-A): one lane inserts for all
        KeyT lk0 = (KeyT)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lk);
        if constexpr (K64) lk0 |= (KeyT)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)lk >> 32)) << 32;
        const bool uniform = __ballot(in && lk == lk0) == ~0ull;
        const uint32_t mult = uniform ? (uint32_t)kWave : 1u;
        uint32_t c_pv = 0, c_nx = 0;  // count: what this insert adds per prev / next char (one byte each)
        if constexpr (COUNT) {
          const unsigned pv = (w1v >> 3) & 7u, nx = w1v & 7u;
          if (uniform) {
#pragma unroll
            for (unsigned x = 0; x < 4; ++x) {
              c_pv |= min((uint32_t)__builtin_popcountll(__ballot(pv == x)), 255u) << (8 * x);
              c_nx |= min((uint32_t)__builtin_popcountll(__ballot(nx == x)), 255u) << (8 * x);
            }
          } else {
            c_pv = pv < 4 ? 1u << (8 * pv) : 0u;
            c_nx = nx < 4 ? 1u << (8 * nx) : 0u;
          }
        }
        if (in && (!uniform || lane == 0)) {
          uint32_t h = stream_hash<K64>(lk, 12);
          for (;;) {
            const KeyT old = atomicCAS(&keys[h], kEmpty, lk);
            if (old == kEmpty) {
              fidx[h] = (uint32_t)(idx - lo);
              atomicAdd(&s_claims, 1u);
            }
            if (old == kEmpty || old == lk) {
              atomicAdd(&cnts[h], mult);
              if constexpr (COUNT) {
                if (c_pv | c_nx) add_chars(h, c_pv, c_nx);
              }
              break;
            }
            h = (h + 1) & (NS - 1);  // (the table is flushed at half full: a free slot exists)
          }
        }
        __syncthreads();
        // (latched between two barriers: a wavefront that runs ahead into the next trip's inserts raises s_claims while a slower one
        //  is still reading it here — the waves would disagree about entering flush(), whose barriers then pair up wrongly)
        const uint32_t claims = s_claims;
        __syncthreads();
        if (claims >= (uint32_t)kFlushAt - NT) {  // (uniform; at most NT more keys before the next look)
          flush();
          stop = s_stop != 0;
        }
      }
      if (!stop) flush();
    }
  }
}


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
template <bool AGG, int UNR, int NT, int LOGS, bool TAGS, bool GIANT = false, bool COUNT = false, bool K64 = false>
__global__ __launch_bounds__(NT) void k_s1_stream(const uint32_t *__restrict__ items0, const uint64_t *__restrict__ bounds, S1SegArgs a,
                                                  S1StreamGeom geo, uint32_t bucket_stride, uint32_t *__restrict__ ticket,
                                                  const uint32_t *const *__restrict__ srcs, int n_src) {
  // Multi-GPU: the records of a bucket arrive as n_src sub-ranges, one per sending rank, each rank's records sorted by
  // bucket in an array of its own (srcs[q], bounds[q * (n_buckets + 1) + bucket]); single GPU: one source, items0.
  constexpr int NSLOT = 1 << LOGS;
  constexpr int TRIP = NT * UNR;
  static_assert(NSLOT % NT == 0 && UNR <= 8, "table walk / pending mask");
  typedef typename StreamKey<K64>::T KeyT;       // K64: local keys wider than 32 bits (k > 22 at a 16-bit prefix): 64-bit compare-and-swap
  constexpr KeyT kEmpty = StreamKey<K64>::kEmpty;
  static_assert(!K64 || !AGG || COUNT, "aggregated stage-2 items exist up to k = 22: their keys fit 32 bits");
  __shared__ KeyT keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  __shared__ uint32_t fpos[NSLOT];             // position word of the record that claimed the slot (direct_marks)
  __shared__ uint8_t ftag[TAGS ? NSLOT : 4];   // ... and the position bits above it (s1_pos_tag), when the read set has any
  __shared__ uint32_t lhist[kSegHist];
  __shared__ uint32_t s_bad[2], s_nclaimed[2];  // per round, double-buffered: the next round's are cleared while this round's are read
  constexpr int NLIST = NSLOT / 4;              // solid keys of a round waiting for their aggregated items (more: worked off in place)
  __shared__ uint2 slist[AGG && !COUNT ? NLIST : 1];
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
    keys[i] = kEmpty;
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
  const bool count_wide = COUNT && m > 2;  // count with min count 3..15: per-char counters instead of the seen-once / seen-twice bits
  // count: per-char occurrences (one byte per prev char in pv8, per next char in nx8) -> the slot's char word, in either form
  auto count_add_chars = [&](uint32_t *slot, uint32_t pv8, uint32_t nx8) {
    if (count_wide) {
      uint32_t cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (;;) {
        uint32_t nv = cur;
#pragma unroll
        for (unsigned x = 0; x < 4; ++x) {
          const uint32_t fp = (cur >> (4 * x)) & 15u, fn = (cur >> (16 + 4 * x)) & 15u;
          nv += (min(fp + ((pv8 >> (8 * x)) & 0xFFu), m) - fp) << (4 * x);
          nv += (min(fn + ((nx8 >> (8 * x)) & 0xFFu), m) - fn) << (16 + 4 * x);
        }
        if (nv == cur) break;
        const uint32_t got = atomicCAS(slot, cur, nv);
        if (got == cur) break;
        cur = got;
      }
    } else {  // prev char x: bit 2x = seen once, 2x + 1 = seen twice; next char x: bits 8 + 2x, 9 + 2x
      uint32_t add1 = 0, add2 = 0;
#pragma unroll
      for (unsigned x = 0; x < 4; ++x) {
        const uint32_t cp = (pv8 >> (8 * x)) & 0xFFu, cn = (nx8 >> (8 * x)) & 0xFFu;
        add1 |= (cp ? 1u << (2 * x) : 0u) | (cn ? 1u << (8 + 2 * x) : 0u);
        add2 |= (cp >= 2 ? 2u << (2 * x) : 0u) | (cn >= 2 ? 2u << (8 + 2 * x) : 0u);
      }
      if (add1) {
        const uint32_t o = atomicOr(slot, add1 | add2);
        const uint32_t again = ((o & add1) << 1) & ~(o | add2);  // a char seen before and now again: seen twice
        if (again) atomicOr(slot, again);
      }
    }
  };
  const int k = a.k;
  const int pbits = geo.pbits;
  const size_t bstride = (size_t)geo.n_buckets + 1;
  // local key: the (k-1)-mer bits below the prefix, then head/tail (the position tag bits in between dropped)
  static_assert(!COUNT || AGG, "count: edges leave through the regions of the aggregated items");
  const int key_chars = COUNT ? k + 1 : k - 1;
  const int rem = 2 * key_chars - pbits;         // 0..26 bits (count: up to 32)
  const int lk_bits = COUNT ? rem : rem + 6;     // <= 32
  const int mer_sh = 64 - 2 * key_chars;
  const uint32_t mer_mask = rem >= 32 ? 0xFFFFFFFFu : (rem ? (1u << rem) - 1u : 0u);
  const unsigned long long mer_mask64 = rem >= 64 ? ~0ull : (rem ? (1ull << rem) - 1ull : 0ull);
  // (the low 32 bits of (w0:w1) >> mer_sh: one funnel shift while the (k-1)-mer reaches into the second word, k >= 18)
  const bool mer_two_words = mer_sh < 32;
  const uint32_t mer_sh1 = (uint32_t)(mer_two_words ? mer_sh : mer_sh - 32);
  auto local_key = [&](uint32_t w0, uint32_t w1) -> KeyT {
    if constexpr (K64) {
      const unsigned long long lo = ((((unsigned long long)w0 << 32) | w1) >> mer_sh) & mer_mask64;
      if constexpr (COUNT) return lo;
      else return lo << 6 | (w1 & 63u);
    } else {
      const uint32_t lo = mer_two_words ? __builtin_amdgcn_alignbit(w0, w1, mer_sh1) : w0 >> mer_sh1;
      if constexpr (COUNT) return lo & mer_mask;
      else return (lo & mer_mask) << 6 | (w1 & 63u);
    }
  };
  // (a value the lanes of a wavefront agree on: the first lane's)
  auto first_lane_key = [](KeyT v) -> KeyT {
    KeyT r = (KeyT)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    if constexpr (K64) r |= (KeyT)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)v >> 32)) << 32;
    return r;
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
  auto hash_of = [&](KeyT lk) -> uint32_t { return stream_hash<K64>(lk, LOGS); };
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
    uint32_t sub = (uint32_t)min(geo.sub0, lk_bits);
    KeyT rj = 0;  // (as wide as the local key: a round that keeps overflowing splits until every key bit is fixed)
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
        KeyT lk[UNR];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          lk[u] = local_key(rw0[u], rw1[u]);
          const bool mn = ((inm >> u) & 1u) && (sub == 0 || (KeyT)(lk[u] >> sub_sh) == (KeyT)rj);
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
        one_key = __ballot(one_key && lk[0] == first_lane_key(lk[0])) == ~0ull;
        uint32_t mult = 1;
        uint32_t wave_add1 = 0, wave_add2 = 0;  // COUNT: the seen-once / seen-twice bits of all records of a one-key trip
        uint32_t wave_cp = 0, wave_cn = 0;      // COUNT, min count > 2: per char the trip's occurrences, capped at 255, one byte each
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
              wave_cp |= min(cp, 255u) << (8 * x);
              wave_cn |= min(cn, 255u) << (8 * x);
            }
          }
          mine = lane == 0 ? 1u : 0u;
          mult = (uint32_t)(kWave * UNR);
        }
        // First probe of every record, straight-line.  A lane that met another key there keeps the record pending — one per lane;
        // a second one of the same trip (one lane in twenty) is seen to on the spot — and the pending records of all lanes are
        // retried together afterwards: the retries cost their instructions per turn, however few lanes take part.
        // (the slot found — the key's own, or a free one claimed: count it, and remember the record that claimed it)
        auto settle = [&](KeyT old, KeyT key, uint32_t hh, uint32_t pos, uint32_t w1v) -> bool {
          if (old != kEmpty && old != key) return false;
          atomicAdd(&cnts[hh], mult);
          if constexpr (COUNT) {
            if (old == kEmpty) ++claims;
            const unsigned pv = (w1v >> 3) & 7u, nx = w1v & 7u;
            if (count_wide) {
              // min count 3..15: per char a 4-bit counter that stops at m (prev char x: bits [4x, 4x + 4), next char x: [16 + 4x, 20 + 4x)),
              // moved by compare-and-swap so that a field never runs over into its neighbour
              uint32_t inc[2] = {0, 0};  // what this record (or, one key per wavefront: the whole trip) adds to its prev / next char
              if (one_key) {
                inc[0] = wave_cp;
                inc[1] = wave_cn;
              } else {
                inc[0] = pv < 4 ? 1u << (8 * pv) : 0u;
                inc[1] = nx < 4 ? 1u << (8 * nx) : 0u;
              }
              if (inc[0] | inc[1]) {
                uint32_t cur = __hip_atomic_load(&fpos[hh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (;;) {
                  uint32_t nv = cur;
#pragma unroll
                  for (unsigned x = 0; x < 4; ++x) {
                    const uint32_t ap = (inc[0] >> (8 * x)) & 0xFFu, an = (inc[1] >> (8 * x)) & 0xFFu;
                    const uint32_t fp = (cur >> (4 * x)) & 15u, fn = (cur >> (16 + 4 * x)) & 15u;
                    nv += (min(fp + ap, m) - fp) << (4 * x);
                    nv += (min(fn + an, m) - fn) << (16 + 4 * x);
                  }
                  if (nv == cur) break;
                  const uint32_t got = atomicCAS(&fpos[hh], cur, nv);
                  if (got == cur) break;
                  cur = got;
                }
              }
            } else {
            // prev char x: bit 2x = seen once, 2x + 1 = seen twice; next char x: bits 8 + 2x, 9 + 2x ('$' counts for nothing)
            const uint32_t add1 = one_key ? wave_add1 : ((pv < 4 ? 1u << (2 * pv) : 0u) | (nx < 4 ? 1u << (8 + 2 * nx) : 0u));
            const uint32_t add2 = one_key ? wave_add2 : 0u;
            if (add1) {
              const uint32_t o = atomicOr(&fpos[hh], add1 | add2);
              const uint32_t again = ((o & add1) << 1) & ~(o | add2);  // a char seen before and now again: seen twice
              if (again) atomicOr(&fpos[hh], again);
            }
            }
          } else if (old == kEmpty) {  // only read back when the count stays 1: then this record is the key's only one
            fpos[hh] = pos;
            if (TAGS) ftag[hh] = (uint8_t)(w1v >> 6);
            ++claims;
          }
          return true;
        };
        auto probe = [&](KeyT key, uint32_t hh, uint32_t pos, uint32_t w1v) -> bool {
          return settle(atomicCAS(&keys[hh], kEmpty, key), key, hh, pos, w1v);
        };
        // the UNR compare-and-swaps go out back to back: one LDS round trip per trip instead of UNR (with four wavefronts per SIMD
        // the round trips, ~250 cycles each under load, are not hidden)
        uint32_t h1[UNR];
        KeyT old1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          h1[u] = hash_of(lk[u]);
          old1[u] = kEmpty;
          if ((mine >> u) & 1u) old1[u] = atomicCAS(&keys[h1[u]], kEmpty, lk[u]);
        }
        bool has = false;
        KeyT pk = 0;
        uint32_t ph = 0, pw = 0, pt = 0;
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
          const KeyT lk = local_key(en.x, en.y);
          if (sub != 0 && (KeyT)(lk >> sub_sh) != (KeyT)rj) continue;
          if (probe_limit <= 0) {
            s_bad[rp] = 1;
            continue;
          }
          uint32_t hh = hash_of(lk);
          for (int n = 0;; ++n) {
            const KeyT old = atomicCAS(&keys[hh], kEmpty, lk);
            if (old == kEmpty || old == lk) {
              atomicAdd(&cnts[hh], en.w);
              if constexpr (COUNT) {  // the slice's per-char counters (4 bits each, k_s1_giant_reduce) -> the slot's char word
                if (old == kEmpty) ++my_claims;
                uint32_t pv8 = 0, nx8 = 0;
#pragma unroll
                for (unsigned x = 0; x < 4; ++x) {
                  pv8 |= ((en.z >> (4 * x)) & 15u) << (8 * x);
                  nx8 |= ((en.z >> (16 + 4 * x)) & 15u) << (8 * x);
                }
                if (pv8 | nx8) count_add_chars(&fpos[hh], pv8, nx8);
              } else if (old == kEmpty) {
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
      uint32_t nsub = sub;
      KeyT nrj = rj;
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
        bucket_done = nsub == sub_first && nrj == ((KeyT)1 << sub_first);
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
                const KeyT lk = local_key(w0, w1);
                in = sub == 0 || (KeyT)(lk >> sub_sh) == (KeyT)rj;  // (a key of another round is not in the table)
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
        KeyT wk[W];
        uint32_t wc[W], wf[W];
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          wk[it] = keys[sl];
          wc[it] = cnts[sl];
          wf[it] = fpos[sl];
        }
        __syncthreads();  // (s_flagged cleared before anybody sets it)
        const uint32_t lvl = m >= 2 ? 0xAAu : 0x55u;  // which bit of a char's pair says "at least m" (min count <= 2)
        uint32_t solid_bits = 0, n_dist = 0;
        bool any_flag = false;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const KeyT lk = wk[it];
          const uint32_t cnt = wc[it];
          uint32_t fb = 0;
          if (lk != kEmpty && !bad) {
            ++n_dist;
            const uint32_t hb = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;
            if (hb < kSegHist) atomicAdd(&lhist[hb], 1u);
            else atomicAdd(&a.hist[hb], 1ull);
            if (cnt >= m) {
              solid_bits |= 1u << it;
              bool has_in, has_out;
              if (count_wide) {  // some char's counter reached m
                has_in = has_out = false;
#pragma unroll
                for (unsigned x = 0; x < 4; ++x) {
                  has_in = has_in || ((wf[it] >> (4 * x)) & 15u) >= m;
                  has_out = has_out || ((wf[it] >> (16 + 4 * x)) & 15u) >= m;
                }
              } else {
                has_in = (wf[it] & lvl) != 0;
                has_out = ((wf[it] >> 8) & lvl) != 0;
              }
              fb = (has_in ? 0u : 1u) | (has_out ? 0u : 2u);
              any_flag = any_flag || fb != 0;
              if constexpr (GIANT) {  // listed for k_count_giant_look (a list that runs over: this workgroup looks itself, below)
                if (fb && a.giant.fl_cnt && !a.c_edges_only) {
                  const uint32_t at = atomicAdd(&a.giant.fl_cnt[(uint32_t)bi64], 1u);
                  if (at < a.giant.fl_cap) a.giant.fl_key[(size_t)bi64 * a.giant.fl_cap + at] = ((unsigned long long)lk << 2) | fb;
                }
              }
            }
            fpos[it * NT + tid] = fb << 30;
          }
        }
        st_solid += n_dist;  // (count: distinct keys)
        if (__ballot(any_flag) && lane == 0 && !a.c_edges_only) s_flagged = 1;
        {  // the solid keys' packed edges (PackEdge, kmer_counter.cpp:32-52: multiplicity in the low 16 bits) -> the region, from its front
          const uint32_t n_e = (uint32_t)__builtin_popcount(solid_bits);
          const uint32_t incl = wave_inclusive_sum(n_e);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t ebase = 0;
            if (lane == 0) ebase = atomicAdd(&s_agg_cur, tot);
            ebase = __shfl(ebase, 0, kWave);
            const bool wide_edges = a.c_wpe == 3;  // k >= 24: the (k+1)-mer and the 16-bit multiplicity no longer share 64 bits
            if (ebase + tot > (wide_edges ? a.agg_cap / 2 : a.agg_cap)) {
              if (lane == 0) atomicOr(a.err, 1u);
            } else {
              unsigned long long *const eout = reinterpret_cast<unsigned long long *>(a.agg_raw + (size_t)blockIdx.x * a.agg_cap);
              uint32_t at = ebase + incl - n_e;
#pragma unroll
              for (int it = 0; it < W; ++it)
                if ((solid_bits >> it) & 1u) {
                  const uint32_t cnt = wc[it];
                  const unsigned long long edge = ((unsigned long long)bi << (64 - pbits)) | (rem ? (unsigned long long)wk[it] << mer_sh : 0ull);
                  const uint32_t mul = cnt > MHX_MAX_MUL ? (uint32_t)MHX_MAX_MUL : cnt;
                  if (wide_edges) reinterpret_cast<uint4 *>(eout)[at++] = make_uint4((uint32_t)(edge >> 32), (uint32_t)edge, mul, 0u);  // (words in edge order)
                  else eout[at++] = edge | mul;
                }
            }
          }
        }
        __syncthreads();
        if constexpr (GIANT) {  // the listed keys are k_count_giant_look's; only a list that ran over leaves the look to this workgroup
          if (a.giant.fl_cnt) {
            if (tid == 0 && s_flagged)
              s_flagged = __hip_atomic_load(&a.giant.fl_cnt[(uint32_t)bi64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > a.giant.fl_cap ? 1u : 0u;
            __syncthreads();
          }
        }
        if (s_flagged && !bad) {  // the records of the flagged keys: first_0_out / last_0_in of their reads (kmer_counter.cpp:307-368)
          for (int q = 0; q < n_src; ++q) {
            // (GIANT: the table was filled from the slices' partial entries; the records themselves lie where the bucket's bounds say)
            const uint64_t lo = GIANT ? gbounds[(size_t)q * bstride + bi] : lo_of(par, q), hi = GIANT ? gbounds[(size_t)q * bstride + bi + 1] : hi_of(par, q);
            const gptr items = (gptr)src_of(q);
            for (uint64_t base = lo; base < hi; base += NT) {
              const uint64_t gi = base + tid;
              uint32_t f = 0, w1 = 0, w2 = 0;
              if (gi < hi) {
                const gptr p = items + gi * 3;
                const uint32_t w0 = p[0];
                w1 = p[1];
                w2 = p[2];
                const KeyT lk = local_key(w0, w1);
                if (sub == 0 || (KeyT)(lk >> sub_sh) == (KeyT)rj) {  // (a key of another round is not in the table)
                  uint32_t h = hash_of(lk);
                  while (keys[h] != lk) h = (h + 1) & (NSLOT - 1);
                  f = fpos[h] >> 30;
                }
              }
              const uint64_t abs = w2 + (TAGS ? (uint64_t)((w1 >> 7) & 0xFFu) * a.pos_stride : 0ull);
              const bool fwd = (w1 & kCountStrandBit) == 0;
              // no in-edge (f & 1): strand 0 -> last_0_in = max(off), strand 1 -> first_0_out = min(off + 1); no out-edge (f & 2): the
              // roles swap.  As an event: position << 1 | (1: first_0_out, 0: last_0_in) — count.hip k_apply_count_events
              if (!marks_out) {
                if (f) {
                  const uint64_t rid = seq_of_offset(a.c_start, a.c_n_seqs, a.c_fixed_len, abs);
                  const uint32_t off = (uint32_t)(abs - a.c_start[rid]);
                  if (f & 1u) {
                    if (fwd) atomicMax(&a.last_0_in_p1[rid], off + 1);
                    else atomicMin(&a.first_0_out[rid], off + 1);
                  }
                  if (f & 2u) {
                    if (fwd) atomicMin(&a.first_0_out[rid], off + 1);
                    else atomicMax(&a.last_0_in_p1[rid], off + 1);
                  }
                }
              } else {  // several GPUs: the reads live on other ranks — the events go to this workgroup's region and are routed to the read owners
                const uint32_t ne = (f & 1u) + (f >> 1);
                const uint32_t incl = wave_inclusive_sum(ne);
                const uint32_t tot = __shfl(incl, kWave - 1, kWave);
                if (tot) {
                  uint32_t ebase = 0;
                  if (lane == 0) ebase = atomicAdd(&s_mark_cur, tot);
                  ebase = __shfl(ebase, 0, kWave);
                  uint32_t at = ebase + incl - ne;
                  if (ebase + tot > a.marks_cap) {
                    if (lane == 0) atomicOr(a.err, 2u);
                  } else {
                    if (f & 1u) marks_out[at++] = (abs << 1) | (fwd ? 0ull : 1ull);
                    if (f & 2u) marks_out[at] = (abs << 1) | (fwd ? 1ull : 0ull);
                  }
                }
              }
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const int sl = it * NT + tid;
          keys[sl] = kEmpty;
          cnts[sl] = 0;
          fpos[sl] = 0;
        }
      } else {
        constexpr int W = NSLOT / NT;
        KeyT wk[W];
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
        uint32_t want_bits = 0, mark_bits = 0;
#pragma unroll
        for (int it = 0; it < W; ++it) {
          const KeyT lk = wk[it];
          const uint32_t cnt = wc[it];
          if (lk != kEmpty && !bad && (lk & 0x24u) == 0) {
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
                if (at < (uint32_t)NLIST) slist[at] = make_uint2((uint32_t)wk[it], wc[it]);
                else emit_items(bi, (uint32_t)wk[it], wc[it], false);  // (more solid keys in one round than the list holds: in place)
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


// count, giant buckets: the records of the listed keys (solid, without an in- or out-edge) move first_0_out / last_0_in of their reads
// (kmer_counter.cpp:307-368) — the slices of the giant again, many workgroups, the giant's list as an LDS table
__global__ __launch_bounds__(256) void k_count_giant_look(const uint32_t *__restrict__ items0, const uint32_t *const *__restrict__ srcs,
                                                          const uint64_t *__restrict__ bounds, int n_src, uint32_t n_buckets, int pbits, S1SegArgs a) {
  constexpr int NS = 4096, NT = 256;
  static_assert(NS >= 2 * (int)kGiantFlagged, "the list at half load");
  __shared__ unsigned long long tk[NS];
  const S1Giant &g = a.giant;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const size_t bstride = (size_t)n_buckets + 1;
  const uint32_t n_g = min(g.ctr[0], g.gcap);
  const int k = a.k;
  const int c_rem = 2 * (k + 1) - pbits, c_sh = 64 - 2 * (k + 1);
  const unsigned long long c_mask = c_rem >= 64 ? ~0ull : (c_rem > 0 ? (1ull << c_rem) - 1ull : 0ull);
  const bool tags = a.pos_stride != 0;
  for (uint32_t gi = 0; gi < n_g; ++gi) {
    const uint32_t ns = g.ns[gi], nf = g.fl_cnt[gi];
    const uint32_t b = g.bucket[gi];
    if (!ns || !nf || nf > g.fl_cap || !g.flag[b]) continue;  // (nothing listed; a list that ran over or a bucket given back: looked at by the streaming kernel)
    if (blockIdx.x >= ns) continue;
    __syncthreads();
    for (int i = tid; i < NS; i += NT) tk[i] = ~0ull;
    __syncthreads();
    for (uint32_t i = tid; i < nf; i += NT) {
      const unsigned long long e = g.fl_key[(size_t)gi * g.fl_cap + i];
      uint32_t h = stream_hash<true>(e >> 2, 12);
      while (atomicCAS(&tk[h], ~0ull, e) != ~0ull) h = (h + 1) & (NS - 1);
    }
    __syncthreads();
    const uint32_t sl_len = g.sl[gi];
    for (uint32_t sl = blockIdx.x; sl < ns; sl += gridDim.x) {
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
      for (uint64_t base = lo; base < hi; base += NT) {
        const uint64_t idx = base + tid;
        uint32_t f = 0, w1 = 0, w2 = 0;
        if (idx < hi) {
          const uint32_t *r = src + idx * 3;
          w1 = r[1];
          w2 = r[2];
          const unsigned long long lk = ((((unsigned long long)r[0] << 32) | w1) >> c_sh) & c_mask;
          uint32_t h = stream_hash<true>(lk, 12);
          for (;;) {
            const unsigned long long e = tk[h];
            if (e == ~0ull) break;
            if ((e >> 2) == lk) {
              f = (uint32_t)e & 3u;
              break;
            }
            h = (h + 1) & (NS - 1);
          }
        }
        const uint64_t abs = w2 + (tags ? (uint64_t)((w1 >> 7) & 0xFFu) * a.pos_stride : 0ull);
        const bool fwd = (w1 & kCountStrandBit) == 0;
        if (!g.ev) {
          if (f) {
            const uint64_t rid = seq_of_offset(a.c_start, a.c_n_seqs, a.c_fixed_len, abs);
            const uint32_t off = (uint32_t)(abs - a.c_start[rid]);
            if (f & 1u) {
              if (fwd) atomicMax(&a.last_0_in_p1[rid], off + 1);
              else atomicMin(&a.first_0_out[rid], off + 1);
            }
            if (f & 2u) {
              if (fwd) atomicMin(&a.first_0_out[rid], off + 1);
              else atomicMax(&a.last_0_in_p1[rid], off + 1);
            }
          }
        } else {  // several GPUs: events for the ranks that hold the reads (k_s1_stream<COUNT> writes the same)
          const uint32_t ne = (f & 1u) + (f >> 1);
          const uint32_t incl = wave_inclusive_sum(ne);
          const uint32_t tot = __shfl(incl, kWave - 1, kWave);
          if (tot) {
            uint32_t ebase = 0;
            if (lane == 0) ebase = atomicAdd(g.ev_cur, tot);
            ebase = __shfl(ebase, 0, kWave);
            uint32_t at = ebase + incl - ne;
            if ((uint64_t)ebase + tot > g.ev_cap) {
              if (lane == 0) atomicOr(a.err, 2u);
            } else {
              if (f & 1u) g.ev[at++] = (abs << 1) | (fwd ? 0ull : 1ull);
              if (f & 2u) g.ev[at] = (abs << 1) | (fwd ? 1ull : 0ull);
            }
          }
        }
      }
    }
  }
}

// ---- launchers (the only way into this unit's kernels) ----
void count_giant_look_launch(mhx_ctx *c, const uint32_t *items0, const uint32_t *const *srcs, const uint64_t *bounds, int n_src, uint64_t n_buckets, int pbits,
                             const S1SegArgs &a) {
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  MHX_LAUNCH(c, "count_giant_look", 0.0,
             hipLaunchKernelGGL(k_count_giant_look, dim3((unsigned)(3 * cus)), dim3(256), 0, c->stream, items0, srcs, bounds, n_src, (uint32_t)n_buckets, pbits, a));
}
void s1_giant_launch(mhx_ctx *c, const uint32_t *items0, const uint32_t *const *srcs, const uint64_t *bounds, int n_src, uint64_t n_buckets, int pbits, int k,
                     const S1Giant &g, bool key64, bool count) {
  hipStream_t st = c->stream;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  MHX_LAUNCH(c, "s1_giant_find", (double)n_src * n_buckets * 8,
             hipLaunchKernelGGL(k_s1_giant_find, dim3((unsigned)div_ceil(n_buckets, 256)), dim3(256), 0, st, bounds, n_src, (uint32_t)n_buckets, g));
#define MHX_GR(K64V, COUNTV)                                                                                                                          \
  MHX_LAUNCH(c, "s1_giant_reduce", 0.0,                                                                                                                 \
             hipLaunchKernelGGL((k_s1_giant_reduce<K64V, COUNTV>), dim3((unsigned)(3 * cus)), dim3(256), 0, st, items0, srcs, bounds, n_src, (uint32_t)n_buckets, \
                                pbits, k, g))
  if (key64 && count) MHX_GR(true, true);
  else if (key64) MHX_GR(true, false);
  else if (count) MHX_GR(false, true);
  else MHX_GR(false, false);
#undef MHX_GR
}

void s1_stream_launch(mhx_ctx *c, const char *name, double bytes, const S1StreamLaunch &l) {
  hipStream_t st = c->stream;
#define MHX_STREAM_K(AGGV, NTV, LOGV, TAGV, GIANTV, COUNTV, K64V)                                                                                     \
  MHX_LAUNCH(c, name, bytes, hipLaunchKernelGGL((k_s1_stream<AGGV, 4, NTV, LOGV, TAGV, GIANTV, COUNTV, K64V>), dim3(l.grid), dim3(NTV), 0, st, l.items0, \
                                                l.bounds, l.a, l.geo, l.stride, l.ticket, l.srcs, l.n_src))
#define MHX_STREAM(AGGV, NTV, LOGV, TAGV, GIANTV, COUNTV) MHX_STREAM_K(AGGV, NTV, LOGV, TAGV, GIANTV, COUNTV, false)
#define MHX_STREAM_T(AGGV, NTV, LOGV, GIANTV, COUNTV)             \
  do {                                                            \
    if (l.tags) MHX_STREAM(AGGV, NTV, LOGV, true, GIANTV, COUNTV); \
    else MHX_STREAM(AGGV, NTV, LOGV, false, GIANTV, COUNTV);       \
  } while (0)
  if (l.key64 && l.count) {  // count at k = 23..27 (or under a forced narrow prefix): no position tags in those records
    if (!l.agg || l.half || l.tags) throw Error("s1_stream_launch: count with 64-bit local keys runs on full tables, edge regions, no position tags");
    if (l.giant) MHX_STREAM_K(true, kStreamThreads, 13, false, true, true, true);
    else MHX_STREAM_K(true, kStreamThreads, 13, false, false, true, true);
  } else if (l.key64) {  // local keys of more than 32 bits (stage 1 at k = 23..29: no aggregated items there)
    if (l.agg || l.half) throw Error("s1_stream_launch: 64-bit local keys serve stage 1 without aggregated items on full tables");
    if (l.giant && l.tags) MHX_STREAM_K(false, kStreamThreads, 13, true, true, false, true);
    else if (l.giant) MHX_STREAM_K(false, kStreamThreads, 13, false, true, false, true);
    else if (l.tags) MHX_STREAM_K(false, kStreamThreads, 13, true, false, false, true);
    else MHX_STREAM_K(false, kStreamThreads, 13, false, false, false, true);
  } else if (l.count) {
    if (!l.agg || l.half) throw Error("s1_stream_launch: count runs on full tables with edge regions");
    if (l.giant) MHX_STREAM_T(true, kStreamThreads, 13, true, true);
    else MHX_STREAM_T(true, kStreamThreads, 13, false, true);
  } else if (l.giant) {
    if (l.half) throw Error("s1_stream_launch: the giant path runs on full tables");
    if (l.agg) MHX_STREAM_T(true, kStreamThreads, 13, true, false);
    else MHX_STREAM_T(false, kStreamThreads, 13, true, false);
  } else if (l.half) {
    throw Error("s1_stream_launch: the half-size tables were retired in round 6");
  } else {
    if (l.agg) MHX_STREAM_T(true, kStreamThreads, 13, false, false);
    else MHX_STREAM_T(false, kStreamThreads, 13, false, false);
  }
#undef MHX_STREAM_T
#undef MHX_STREAM
#undef MHX_STREAM_K
}

#ifdef MHX_TILE_TIMING
// debug build only: this unit's copy of the phase clocks (phases 10..14), added to out16
int s1_stream_phases(unsigned long long *out16, int reset) {
  unsigned long long v[16];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_tile_phase), 16 * 8) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out16[i] += v[i];
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_phase), z, 16 * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif

}  // namespace mhx
