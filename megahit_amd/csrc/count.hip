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

// items of the local reads -> c->ws("items_a"); returns their number
uint64_t count_extract(mhx_ctx *c, uint32_t k) {
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
  if (n_items) {
    const unsigned grid = 256 * 8;
    MHX_DISPATCH_KW(KWv, {
      if (S == KW + 2)
        MHX_LAUNCH(c, "count_extract", (double)n_items * item_bytes + (double)s.n_bases / 4,
                   hipLaunchKernelGGL((k_count_extract<KW, KW + 2>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(),
                                      s.start.as<uint64_t>(), item_start, ns, (int)k, c->pos_base, buf_a));
      else
        MHX_LAUNCH(c, "count_extract", (double)n_items * item_bytes + (double)s.n_bases / 4,
                   hipLaunchKernelGGL((k_count_extract<KW, KW + 3>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(),
                                      s.start.as<uint64_t>(), item_start, ns, (int)k, c->pos_base, buf_a));
    });
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
  uint32_t *sorted = radix_sort(c, buf_a, buf_b, n_items, S, KWv, make_passes(KWv, KWv * 32 - key_bits, KWv * 32));
  uint32_t *spare = sorted == buf_a ? buf_b : buf_a;

  // results
  // accumulate (bucket-range passes after the first): first_0_out, the raw last_0_in (+1) values and the histogram of
  // the earlier passes are kept; the published last_0_in is re-derived from the raw values after every pass
  const bool acc = c->accumulate && c->results.count(MHX_BUF_FIRST_0_OUT) && c->results[MHX_BUF_FIRST_0_OUT].used == ns * 4 &&
                   c->work.count("last_p1") && c->results.count(MHX_BUF_MUL_HIST);
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
  switch (S) {
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

int run_count(mhx_ctx *c, uint32_t k, uint32_t m, mhx_count_result *out) {
  if (c->global_bases) throw Error("count: the global layout is set; use the mhx_dist_* entry points (or mhx_set_global_layout(0, 0))");
  const StageItems it = extract_stage(c, MHX_STAGE_COUNT, k, m);
  uint32_t *buf_a = c->work["items_a"].as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
  return count_process(c, k, m, buf_a, buf_b, it.n, out);
}

}  // namespace mhx
