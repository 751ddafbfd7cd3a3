// seq2sdbg: replaces SeqToSdbg (reference src/sorting/seq_to_sdbg.cpp:530-807) on the GPU.
//
//   extract  2(n-k+2) items per sequence of n >= k+1 bases: every k-mer (k-1 chars at the two ends)
//            on both strands, with the W char and 65535-multiplicity in the low 20 key bits
//            (Lv1FillOffsets :579-628 + Lv2ExtractSubString :630-700 fused)
//   sort / groups / emit: shared with read2sdbg S2 (s2.hip)
#include <algorithm>

#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

__global__ void k_seq_item_counts2(const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t k, uint32_t *__restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seqs) {
    uint64_t n = start[i + 1] - start[i];
    cnt[i] = n >= k + 1 ? (uint32_t)(2 * (n - k + 2)) : 0u;
  }
}

template <int KW, int S>
__global__ __launch_bounds__(256) void k_seq_extract(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start,
                                                     const uint64_t *__restrict__ item_start, uint64_t n_seqs, uint32_t fixed_items,
                                                     const uint16_t *__restrict__ mult, int k, uint64_t n_items,
                                                     uint32_t *__restrict__ items, uint64_t first_item) {
  const uint64_t idx = first_item + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // a launch covers < 2^31 items
  if (idx >= n_items) return;
  uint64_t sid;
  if (fixed_items) sid = idx / fixed_items;
  else {
    uint64_t lo = 0, hi = n_seqs;  // item_start[lo] <= idx < item_start[hi]
    while (hi - lo > 1) {
      uint64_t mid = (lo + hi) >> 1;
      if (item_start[mid] <= idx) lo = mid;
      else hi = mid;
    }
    sid = lo;
  }
  const uint64_t st = start[sid];
  const int64_t n = (int64_t)(start[sid + 1] - st);
  const uint64_t t = idx - (fixed_items ? sid * fixed_items : item_start[sid]);
  const int64_t o = (int64_t)(t >> 1);
  const int strand = (int)(t & 1);
  const int nc = k - (o + k > n ? 1 : 0);
  const uint32_t counting = (o > 0 && o + k <= n) ? mult[sid] : 0u;  // :638-643
  unsigned prev;
  uint32_t f[KW], out[S];
  if (!strand) {
    prev = o == 0 ? kSentinel : base_at(seq, st + o - 1);
    load_chars<KW>(seq, st + o, nc, f);
#pragma unroll
    for (int i = 0; i < KW; ++i) out[i] = f[i];
  } else {
    prev = o == 0 ? kSentinel : 3 - base_at(seq, st + n - o);
    int64_t off = n - 1 - o - (k - 1);  // switch to the forward strand, :676-681
    if (off < 0) off = 0;
    load_chars<KW>(seq, st + off, nc, f);
    uint32_t rc[KW];
    rc_chars<KW>(f, nc, rc);
#pragma unroll
    for (int i = 0; i < KW; ++i) out[i] = rc[i];
  }
  out[KW - 1] |= (nc == k ? 1u << 19 : 0u) | (prev << 16) | (MHX_MAX_MUL - counting);
  if constexpr (S > KW) out[KW] = 0;
  uint32_t *dst = items + idx * S;
  if constexpr (S % 4 == 0) {
#pragma unroll
    for (int i = 0; i < S / 4; ++i)
      reinterpret_cast<uint4 *>(dst)[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < S / 2; ++i) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(out[2 * i], out[2 * i + 1]);
  }
}

static int seq2sdbg_kw(uint32_t k) { return (int)div_ceil(k * 2 + 3 + 1 + 16, 32); }  // seq_to_sdbg.cpp:511-513
int seq2sdbg_stride(uint32_t k) { return round_up2(seq2sdbg_kw(k)); }

// items of the local sequences -> c->ws("items_a"); returns their number
uint64_t seq2sdbg_extract(mhx_ctx *c, uint32_t k) {
  SeqSet &s = c->seqs;
  if (k < 9 || k > MHX_MAX_K) throw Error("seq2sdbg: kmer size must be >= 9 and <= 255");  // main_sdbg_build.cpp:205-207
  if (s.mult.used < s.n_seqs * 2) throw Error("seq2sdbg: multiplicities not loaded (mhx_load_multiplicity)");
  const int KWv = seq2sdbg_kw(k), S = seq2sdbg_stride(k);
  const uint64_t ns = s.n_seqs;
  hipStream_t st = c->stream;

  // sequences of one length >= k + 1 (an edge file, reads): item i of sequence s is number s * fixed_items + i — no counts, no scan
  const uint32_t fixed_items = (s.fixed_len >= k + 1) ? 2 * (s.fixed_len - k + 2) : 0;
  uint64_t *item_start = nullptr;
  uint64_t n_items = (uint64_t)fixed_items * ns;
  if (ns && !fixed_items) {
    uint32_t *cnt = c->ws("seq_item_cnt", (ns + 1) * 4).as<uint32_t>();
    item_start = c->ws("seq_item_start", (ns + 2) * 8).as<uint64_t>();
    MHX_LAUNCH(c, "item_counts", (double)ns * 12,
               hipLaunchKernelGGL(k_seq_item_counts2, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, s.start.as<uint64_t>(), ns, k, cnt));
    exclusive_scan_u32_u64(c, cnt, item_start, ns, item_start + ns + 1);
    MHX_HIP(hipMemcpyAsync(&n_items, item_start + ns + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  const size_t item_bytes = (size_t)S * 4;
  uint32_t *buf_a = c->ws("items_a", n_items * item_bytes + 64).as<uint32_t>();
  // one thread per item; a grid holds fewer than 2^32 threads, so large inputs take several launches
  const uint64_t per_launch = 1ull << 31;
  for (uint64_t first = 0; first < n_items; first += per_launch) {
    const uint64_t n_now = std::min(per_launch, n_items - first);
    const unsigned grid = (unsigned)div_ceil(n_now, 256);
    const double bytes_now = ((double)n_items * item_bytes + (double)s.n_bases / 4) * (double)n_now / (double)n_items;
    MHX_DISPATCH_KW(KWv, {
      if (S == KW)
        MHX_LAUNCH(c, "seq_extract", bytes_now,
                   hipLaunchKernelGGL((k_seq_extract<KW, KW>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(), s.start.as<uint64_t>(),
                                      item_start, ns, fixed_items, s.mult.as<uint16_t>(), (int)k, n_items, buf_a, first));
      else
        MHX_LAUNCH(c, "seq_extract", bytes_now,
                   hipLaunchKernelGGL((k_seq_extract<KW, KW + 1>), dim3(grid), dim3(256), 0, st, s.words.as<uint32_t>(),
                                      s.start.as<uint64_t>(), item_start, ns, fixed_items, s.mult.as<uint16_t>(), (int)k, n_items, buf_a, first));
    });
  }
  return n_items;
}

// sort + SdBG emission of n_items items held in buf_a (items carry no positions: any rank can process any bucket)
int seq2sdbg_process(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_sdbg_result *out) {
  const int KWv = seq2sdbg_kw(k), S = seq2sdbg_stride(k);
  const int char_bits = (int)k * 2;
  // whole-key order = chars, then flag/W/multiplicity (the bits between them are zero padding)
  uint32_t *sorted = sort_whole_key(c, buf_a, buf_b, n_items, S, KWv, make_passes_ranges(KWv, {{0, 20}, {KWv * 32 - char_bits, KWv * 32}}));
  emit_sdbg(c, sorted, n_items, S, KWv, k, 1, out);
  return 0;
}

int run_seq2sdbg(mhx_ctx *c, uint32_t k, mhx_sdbg_result *out) {
  const StageItems it = extract_stage(c, MHX_STAGE_SEQ2SDBG, k, 0);
  uint32_t *buf_a = c->work["items_a"].as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
  return seq2sdbg_process(c, k, buf_a, buf_b, it.n, out);
}

}  // namespace mhx
