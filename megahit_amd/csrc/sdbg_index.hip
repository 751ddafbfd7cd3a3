// SURVEY.md §8f N1 — device-resident SdBG hand-over.
//
// Replaces, on the GPU, what the consumer of the path does first and serially on one CPU thread: LoadSdbgRawContent
// (reference src/sdbg/sdbg_raw_content.cpp:18-96: parse the variable-length record stream bucket by bucket into the
// W / last / tip / multiplicity / tip-label arrays) and the rank/select construction of SDBG::LoadFromFile
// (src/sdbg/sdbg.h:26-61 over kmlib::RankAndSelect::from_packed_array, src/kmlib/kmrns.h:118-175).  The input is the
// byte stream + per-bucket tables that stage 2 / seq2sdbg left in HBM (or any stream installed with
// mhx_sdbg_load_bytes), the outputs are result buffers in exactly the reference's in-memory layouts, so that a
// downstream stage can adopt them instead of reading .sdbg files back:
//   W            4 bits per item, item i at bits 4*(i%16) of uint64 word i/16           (CompactVector<4, uint64>)
//   last, tip    1 bit per item, LSB-first uint64 words; invalid = tip | (W == 0)        (sdbg.h:56-60)
//   mul          uint16 per item (EdgeMultiplicity), small_mul uint8 with the 255 sentinel (sdbg_raw_content.cpp:72-83)
//   tip labels   words_per_tip_label uint32 per tip, chars reversed inside each word      (:85-91)
//   rank/select  per character c: l2 (int64 every 16384 items / 65536 bits), l1 (uint16 every 256 items / 1024 bits,
//                relative to l2), select samples (interval of every 4096th occurrence)   (kmrns.h:118-175)
//   prefix table, f, rank_f                                                             (sdbg.h:38-55)
// Bit-exact against the reference's own loader (oracle/ref_sdbg_dump.cpp, tests/test_gpu_sdbg_index.py).
#include <algorithm>

#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

// One wavefront per bucket walks the bucket's records 64 two-byte slots at a time: every lane decodes "the record
// that would start at my slot", lane 0's slot is a true start, and the chain of true starts inside the window is
// followed with scalar reads of the per-lane jumps.
__global__ __launch_bounds__(256) void k_sdbg_parse(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ b_off,
                                                    const uint64_t *__restrict__ b_items, const uint64_t *__restrict__ b_tips,
                                                    const uint64_t *__restrict__ b_large, const uint64_t *__restrict__ acc_items,
                                                    const uint64_t *__restrict__ acc_tips, uint32_t wpt, uint8_t *__restrict__ o_b0,
                                                    uint16_t *__restrict__ o_mul, uint32_t *__restrict__ o_labels, uint32_t *__restrict__ bad) {
  const int lane = lane_id();
  const uint64_t lanemask_lt = (1ull << lane) - 1;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave, n_waves = gridDim.x * blockDim.x / kWave;
  for (uint32_t b = wave; b < MHX_NUM_BUCKETS; b += n_waves) {
    const uint64_t n_it = b_items[b];
    if (!n_it) continue;
    uint64_t pos = b_off[b];
    const uint64_t end = pos + 2 * (n_it + b_large[b]) + 4ull * wpt * b_tips[b];
    uint64_t item = acc_items[b], tip = acc_tips[b];
    const uint64_t item_end = item + n_it;
    while (pos < end) {
      const uint64_t p = pos + 2 * (uint64_t)lane;
      const bool in = p < end;
      const unsigned b0 = in ? bytes[p] : 0u, b1 = in ? bytes[p + 1] : 0u;
      const bool is_tip = (b0 >> 5) & 1u, is_large = b1 == 255u;  // kSmallMulSentinel, sdbg_item.h:14-24
      const int jump = 1 + (is_large ? 1 : 0) + (is_tip ? 2 * (int)wpt : 0);  // slots this record occupies
      uint64_t starts = 0;
      int cur = 0, last_jump = 1;
      while (cur < kWave) {
        const int c = __builtin_amdgcn_readfirstlane(cur);
        if (pos + 2 * (uint64_t)c >= end) break;
        starts |= 1ull << c;
        last_jump = __builtin_amdgcn_readlane(jump, c);
        cur = c + last_jump;
      }
      const bool mine = (starts >> lane) & 1ull;
      const uint64_t tips_here = __ballot(mine && is_tip);
      if (mine) {
        const uint64_t i = item + (uint64_t)__builtin_popcountll(starts & lanemask_lt);
        if (i < item_end) {
          o_b0[i] = (uint8_t)b0;
          uint64_t q = p + 2;
          unsigned mul = b1;
          if (is_large) {
            mul = (unsigned)bytes[q] | ((unsigned)bytes[q + 1] << 8);
            q += 2;
          }
          o_mul[i] = (uint16_t)mul;
          if (is_tip) {
            const uint64_t t = tip + (uint64_t)__builtin_popcountll(tips_here & lanemask_lt);
            for (uint32_t j = 0; j < wpt; ++j) {
              const uint8_t *s = bytes + q + 4 * j;
              const uint32_t wv = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
              o_labels[t * wpt + j] = rev_word(wv);  // kmlib::bit::Reverse<2>, sdbg_raw_content.cpp:87-90
            }
          }
        } else {
          atomicOr(bad, 1u);  // more records in the byte range than the bucket table announces
        }
      }
      item += (uint64_t)__builtin_popcountll(starts);
      tip += (uint64_t)__builtin_popcountll(tips_here);
      pos += 2 * (uint64_t)cur;  // cur = first slot behind the last record that started in this window
    }
    if (lane == 0 && (item != item_end || pos != end)) atomicOr(bad, 2u);
  }
}

// 64 items per thread: byte0 of every item -> one word of last / tip / invalid, four words of W, small_mul
__global__ __launch_bounds__(256) void k_sdbg_pack(const uint8_t *__restrict__ b0, const uint16_t *__restrict__ mul, uint64_t n,
                                                   unsigned long long *__restrict__ w, unsigned long long *__restrict__ last,
                                                   unsigned long long *__restrict__ tip, unsigned long long *__restrict__ invalid,
                                                   uint8_t *__restrict__ small_mul) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i0 = t * 64;
  if (i0 >= n) return;
  unsigned long long wl = 0, wt = 0, wi = 0, ww[4] = {0, 0, 0, 0};
  for (int j = 0; j < 64; ++j) {
    const uint64_t i = i0 + j;
    if (i >= n) break;
    const unsigned x = b0[i], wc = x & 15u;
    ww[j >> 4] |= (unsigned long long)wc << (4 * (j & 15));
    wl |= (unsigned long long)((x >> 4) & 1u) << j;
    wt |= (unsigned long long)((x >> 5) & 1u) << j;
    wi |= (unsigned long long)(((x >> 5) & 1u) | (wc == 0 ? 1u : 0u)) << j;  // tips and W == 0 edges start invalid (sdbg.h:33-60)
    const unsigned m = mul[i];
    small_mul[i] = m < 254u ? (uint8_t)m : (uint8_t)255;  // kMaxSmallMul / kSmallMulSentinel, sdbg_raw_content.cpp:76-81
  }
  last[t] = wl;
  tip[t] = wt;
  invalid[t] = wi;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (i0 + 16 * q < n) w[t * 4 + q] = ww[q];
}

// occurrences of every character in every level-1 interval (whole words, as from_packed_array counts them: the zero
// padding of the last word counts as character 0).  NIB: 4-bit characters 0..8 (16 words per interval); else bits.
template <bool NIB>
__global__ __launch_bounds__(256) void k_rs_counts(const unsigned long long *__restrict__ words, uint64_t n_words, uint64_t n_itv,
                                                   uint32_t *__restrict__ counts /* [n_chars][n_itv] */) {
  const uint64_t itv = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (itv >= n_itv) return;
  const uint64_t w0 = itv * 16, w1 = std::min<uint64_t>(n_words, w0 + 16);
  if constexpr (NIB) {
    uint32_t c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t wi = w0; wi < w1; ++wi) {
      const unsigned long long x = words[wi];
#pragma unroll
      for (int ch = 0; ch < 9; ++ch) {
        unsigned long long y = x ^ ~(0x1111111111111111ull * (unsigned long long)ch);  // nibble == ch -> 0xF
        y &= y >> 2;
        y &= y >> 1;
        c[ch] += (uint32_t)__builtin_popcountll(y & 0x1111111111111111ull);
      }
    }
    for (int ch = 0; ch < 9; ++ch) counts[(uint64_t)ch * n_itv + itv] = c[ch];
  } else {
    uint32_t c = 0;
    for (uint64_t wi = w0; wi < w1; ++wi) c += (uint32_t)__builtin_popcountll(words[wi]);
    counts[itv] = c;
  }
}
// occ[i] (exclusive prefix over intervals, occ[n_itv] = total) -> l2 / l1 exactly as from_packed_array leaves them
__global__ void k_rs_tables(const uint64_t *__restrict__ occ, uint64_t n_itv, uint64_t num_l1, uint64_t num_l2, long long *__restrict__ l2,
                            uint16_t *__restrict__ l1) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = occ[n_itv];
  if (i < num_l2) l2[i] = i + 1 == num_l2 ? (long long)total : (long long)occ[i * 64];
  if (i < num_l1) {
    const uint64_t j = i / 64;
    const uint64_t base = j + 1 == num_l2 ? total : occ[j * 64];
    const uint64_t v = i + 1 == num_l1 ? total : occ[i];
    l1[i] = (uint16_t)(v - base);
  }
}
// select samples: sel[s] = (first interval index i with Occ(i) > s * 4096) - 1 for s < n_samples - 1; last = num_l1 - 1,
// where Occ(i) = occ[i] for i < num_l1 - 1 and the total for the last entry (kmrns.h:160-168)
__global__ void k_rs_select(const uint64_t *__restrict__ occ, uint64_t n_itv, uint64_t num_l1, uint64_t n_samples, uint32_t *__restrict__ sel) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_samples) return;
  if (s + 1 == n_samples) {
    sel[s] = (uint32_t)(num_l1 - 1);
    return;
  }
  const uint64_t key = s * 4096;
  uint64_t lo = 0, hi = num_l1 - 1;  // Occ(hi) = total > key by construction
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const uint64_t v = mid == num_l1 - 1 ? occ[n_itv] : occ[mid];
    if (v > key) hi = mid;
    else lo = mid + 1;
  }
  sel[s] = (uint32_t)(lo - 1);
}
// ones in bits [0, pos) of a bit vector, for a handful of positions (rank_f, sdbg.h:52-54)
__global__ __launch_bounds__(256) void k_rank_at(const unsigned long long *__restrict__ words, const uint64_t *__restrict__ pos, uint64_t *__restrict__ out) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint64_t p = pos[blockIdx.x];
  uint64_t acc = 0;
  for (uint64_t w = threadIdx.x; w * 64 < p; w += 256) {
    unsigned long long x = words[w];
    if ((w + 1) * 64 > p) x &= (1ull << (p & 63)) - 1;
    acc += (uint64_t)__builtin_popcountll(x);
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>(acc, sm, &tot);
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

struct RsResult {
  uint64_t num_l1, num_l2;
  std::vector<uint64_t> char_count, sel_off;  // per character
};
// rank/select tables of one packed array -> result buffers buf_l2 / buf_l1 / buf_sel (0 = rank only)
static RsResult build_rs(mhx_ctx *c, const unsigned long long *words, uint64_t size, bool nib, int buf_l2, int buf_l1, int buf_sel) {
  hipStream_t st = c->stream;
  const uint64_t per_word = nib ? 16 : 64, l1_bases = nib ? 256 : 1024, l2_bases = nib ? 16384 : 65536;
  const int n_chars = nib ? 9 : 1;
  const uint64_t n_words = div_ceil(size, per_word), n_itv = div_ceil(size, l1_bases);
  RsResult r;
  r.num_l1 = n_itv + 1;
  r.num_l2 = div_ceil(size, l2_bases) + 1;
  uint32_t *counts = c->ws("rs_counts", (size_t)n_chars * (n_itv + 1) * 4 + 64).as<uint32_t>();
  uint64_t *occ = c->ws("rs_occ", (size_t)n_chars * (n_itv + 2) * 8 + 64).as<uint64_t>();
  if (n_itv) {
    if (nib)
      MHX_LAUNCH(c, "rs_counts", (double)n_words * 8,
                 hipLaunchKernelGGL(k_rs_counts<true>, dim3((unsigned)div_ceil(n_itv, 256)), dim3(256), 0, st, words, n_words, n_itv, counts));
    else
      MHX_LAUNCH(c, "rs_counts", (double)n_words * 8,
                 hipLaunchKernelGGL(k_rs_counts<false>, dim3((unsigned)div_ceil(n_itv, 256)), dim3(256), 0, st, words, n_words, n_itv, counts));
  }
  long long *l2 = c->result(buf_l2, (size_t)n_chars * r.num_l2 * 8).as<long long>();
  uint16_t *l1 = c->result(buf_l1, (size_t)n_chars * r.num_l1 * 2).as<uint16_t>();
  r.char_count.resize(n_chars);
  for (int ch = 0; ch < n_chars; ++ch) {
    uint64_t *o = occ + (size_t)ch * (n_itv + 2);
    if (n_itv) exclusive_scan_u32_u64(c, counts + (size_t)ch * n_itv, o, n_itv, o + n_itv);
    else MHX_HIP(hipMemsetAsync(o, 0, 8, st));
    hipLaunchKernelGGL(k_rs_tables, dim3((unsigned)div_ceil(std::max(r.num_l1, r.num_l2), 256)), dim3(256), 0, st, o, n_itv, r.num_l1, r.num_l2,
                       l2 + (size_t)ch * r.num_l2, l1 + (size_t)ch * r.num_l1);
    MHX_HIP(hipMemcpyAsync(&r.char_count[ch], o + n_itv, 8, hipMemcpyDeviceToHost, st));
  }
  MHX_HIP(hipStreamSynchronize(st));
  if (buf_sel) {
    r.sel_off.assign(n_chars + 1, 0);
    for (int ch = 0; ch < n_chars; ++ch) r.sel_off[ch + 1] = r.sel_off[ch] + div_ceil(r.char_count[ch], 4096) + 1;
    uint32_t *sel = c->result(buf_sel, r.sel_off[n_chars] * 4).as<uint32_t>();
    for (int ch = 0; ch < n_chars; ++ch) {
      const uint64_t ns = r.sel_off[ch + 1] - r.sel_off[ch];
      hipLaunchKernelGGL(k_rs_select, dim3((unsigned)div_ceil(ns, 256)), dim3(256), 0, st, occ + (size_t)ch * (n_itv + 2), n_itv, r.num_l1, ns,
                         sel + r.sel_off[ch]);
    }
    MHX_HIP(hipGetLastError());
  }
  return r;
}

int sdbg_build_index(mhx_ctx *c, uint32_t k, mhx_sdbg_index_info *out) {
  hipStream_t st = c->stream;
  auto need = [&](int which) -> DevBuf & {
    auto it = c->results.find(which);
    if (it == c->results.end() || !it->second.p) throw Error("sdbg_build_index: no SdBG in the handle (run stage 2 / seq2sdbg or mhx_sdbg_load_bytes first)");
    return it->second;
  };
  DevBuf &bytes = need(MHX_BUF_SDBG_BYTES);
  const uint64_t *b_off = need(MHX_BUF_BUCKET_OFFSET).as<uint64_t>(), *b_items = need(MHX_BUF_BUCKET_COUNT).as<uint64_t>();
  const uint64_t *b_tips = need(MHX_BUF_BUCKET_TIPS).as<uint64_t>(), *b_large = need(MHX_BUF_BUCKET_LARGE).as<uint64_t>();
  const uint32_t wpt = (k + 15) / 16;
  // item / tip index of every bucket's first record (accumulate_item_count / accumulate_tip_count, sdbg_meta.cpp:28-41)
  uint64_t *acc_items = c->ws("sx_acc_items", (MHX_NUM_BUCKETS + 2) * 8).as<uint64_t>();
  uint64_t *acc_tips = c->ws("sx_acc_tips", (MHX_NUM_BUCKETS + 2) * 8).as<uint64_t>();
  exclusive_scan_u64(c, b_items, acc_items, MHX_NUM_BUCKETS, acc_items + MHX_NUM_BUCKETS);
  exclusive_scan_u64(c, b_tips, acc_tips, MHX_NUM_BUCKETS, acc_tips + MHX_NUM_BUCKETS);
  std::vector<uint64_t> h_items(MHX_NUM_BUCKETS), h_acc(MHX_NUM_BUCKETS + 1), h_tip_tot(1), h_large(MHX_NUM_BUCKETS);
  MHX_HIP(hipMemcpyAsync(h_items.data(), b_items, MHX_NUM_BUCKETS * 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_large.data(), b_large, MHX_NUM_BUCKETS * 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_acc.data(), acc_items, (MHX_NUM_BUCKETS + 1) * 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_tip_tot.data(), acc_tips + MHX_NUM_BUCKETS, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  const uint64_t n = h_acc[MHX_NUM_BUCKETS], n_tips = h_tip_tot[0];
  uint64_t n_large = 0;
  for (uint64_t v : h_large) n_large += v;

  uint8_t *b0 = c->ws("sx_b0", n + 64).as<uint8_t>();
  uint16_t *mul = c->result(MHX_BUF_SDBG_MUL, n * 2 + 2).as<uint16_t>();
  c->results[MHX_BUF_SDBG_MUL].used = n * 2;
  uint32_t *labels = c->result(MHX_BUF_SDBG_TIP_LABELS, n_tips * wpt * 4 + 4).as<uint32_t>();
  c->results[MHX_BUF_SDBG_TIP_LABELS].used = n_tips * wpt * 4;
  uint32_t *bad = c->ws("sx_bad", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(bad, 0, 4, st));
  if (n)
    MHX_LAUNCH(c, "sdbg_parse", (double)bytes.used + (double)n * 3,
               hipLaunchKernelGGL(k_sdbg_parse, dim3(2048), dim3(256), 0, st, bytes.as<uint8_t>(), b_off, b_items, b_tips, b_large, acc_items, acc_tips,
                                  wpt, b0, mul, labels, bad));
  const uint64_t nw64 = div_ceil(n, 64);
  unsigned long long *w = c->result(MHX_BUF_SDBG_W, div_ceil(n, 16) * 8 + 8).as<unsigned long long>();
  c->results[MHX_BUF_SDBG_W].used = div_ceil(n, 16) * 8;
  unsigned long long *last = c->result(MHX_BUF_SDBG_LAST, nw64 * 8 + 8).as<unsigned long long>();
  unsigned long long *tip = c->result(MHX_BUF_SDBG_TIP, nw64 * 8 + 8).as<unsigned long long>();
  unsigned long long *inv = c->result(MHX_BUF_SDBG_INVALID, nw64 * 8 + 8).as<unsigned long long>();
  for (int which : {MHX_BUF_SDBG_LAST, MHX_BUF_SDBG_TIP, MHX_BUF_SDBG_INVALID}) c->results[which].used = nw64 * 8;
  uint8_t *small = c->result(MHX_BUF_SDBG_SMALL_MUL, n + 1).as<uint8_t>();
  c->results[MHX_BUF_SDBG_SMALL_MUL].used = n;
  if (n)
    MHX_LAUNCH(c, "sdbg_pack", (double)n * 5,
               hipLaunchKernelGGL(k_sdbg_pack, dim3((unsigned)div_ceil(nw64, 256)), dim3(256), 0, st, b0, mul, n, w, last, tip, inv, small));
  uint32_t h_bad = 0;
  MHX_HIP(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h_bad) throw Error("sdbg_build_index: the byte stream does not match its bucket table");

  const RsResult rw = build_rs(c, w, n, true, MHX_BUF_SDBG_RS_W_L2, MHX_BUF_SDBG_RS_W_L1, MHX_BUF_SDBG_RS_W_SEL);
  const RsResult rl = build_rs(c, last, n, false, MHX_BUF_SDBG_RS_LAST_L2, MHX_BUF_SDBG_RS_LAST_L1, MHX_BUF_SDBG_RS_LAST_SEL);
  const RsResult rt = build_rs(c, tip, n, false, MHX_BUF_SDBG_RS_TIP_L2, MHX_BUF_SDBG_RS_TIP_L1, 0);

  // prefix table (first, last item of every bucket; (0, 0) for empty ones as std::vector value-initialises), f, rank_f
  std::vector<long long> lkt(2 * MHX_NUM_BUCKETS, 0);
  long long f[6] = {-1, 0, 0, 0, 0, 0};
  for (int b = 0; b < MHX_NUM_BUCKETS; ++b) {
    if (!h_items[b]) continue;
    f[b / (MHX_NUM_BUCKETS / 4) + 2] += (long long)h_items[b];
    lkt[2 * b] = (long long)h_acc[b];
    lkt[2 * b + 1] = (long long)(h_acc[b] + h_items[b] - 1);
  }
  for (int i = 2; i < 6; ++i) f[i] += f[i - 1];
  DevBuf &dl = c->result(MHX_BUF_SDBG_PREFIX_LKT, lkt.size() * 8);
  MHX_HIP(hipMemcpyAsync(dl.p, lkt.data(), lkt.size() * 8, hipMemcpyHostToDevice, st));
  uint64_t h_pos[5], h_rank[5] = {0, 0, 0, 0, 0};
  for (int i = 1; i < 6; ++i) h_pos[i - 1] = (uint64_t)f[i];  // rank(f[i] - 1) = ones in [0, f[i])
  uint64_t *d_pos = c->ws("sx_rank_pos", 128).as<uint64_t>();
  MHX_HIP(hipMemcpyAsync(d_pos, h_pos, 40, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_rank_at, dim3(5), dim3(256), 0, st, last, d_pos, d_pos + 8);
  MHX_HIP(hipMemcpyAsync(h_rank, d_pos + 8, 40, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (out) {
    memset(out, 0, sizeof *out);
    out->n_items = n;
    out->n_tips = n_tips;
    out->n_large = n_large;
    out->k = k;
    out->words_per_tip_label = wpt;
    out->use_full_mul = (double)n_large >= (double)n * 0.08 ? 1 : 0;  // sdbg_raw_content.cpp:29-30
    out->num_l1_w = rw.num_l1;
    out->num_l2_w = rw.num_l2;
    out->num_l1_bits = rl.num_l1;
    out->num_l2_bits = rl.num_l2;
    for (int ch = 0; ch < 9; ++ch) out->w_char_count[ch] = rw.char_count[ch];
    for (int ch = 0; ch <= 9; ++ch) out->w_sel_offset[ch] = rw.sel_off[ch];
    out->ones_in_last = rl.char_count[0];
    out->ones_in_tip = rt.char_count[0];
    out->last_sel_count = rl.sel_off[1];
    for (int i = 0; i < 6; ++i) out->f[i] = f[i];
    out->rank_f[0] = 0;
    for (int i = 1; i < 6; ++i) out->rank_f[i] = (long long)h_rank[i - 1];
  }
  return 0;
}

// install an SdBG byte stream that was produced elsewhere (e.g. read back from .sdbg files) as the handle's current SdBG
int sdbg_load_bytes(mhx_ctx *c, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *off, const uint64_t *items, const uint64_t *tips,
                    const uint64_t *large) {
  hipStream_t st = c->stream;
  DevBuf &d = c->result(MHX_BUF_SDBG_BYTES, n_bytes + 64);
  d.used = n_bytes;
  if (n_bytes) upload_pinned(c, d.p, bytes, n_bytes);
  const struct {
    int which;
    const uint64_t *src;
  } tabs[] = {{MHX_BUF_BUCKET_OFFSET, off}, {MHX_BUF_BUCKET_COUNT, items}, {MHX_BUF_BUCKET_TIPS, tips}, {MHX_BUF_BUCKET_LARGE, large}};
  for (const auto &t : tabs) {
    DevBuf &b = c->result(t.which, MHX_NUM_BUCKETS * 8);
    MHX_HIP(hipMemcpyAsync(b.p, t.src, MHX_NUM_BUCKETS * 8, hipMemcpyHostToDevice, st));
  }
  MHX_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // namespace mhx
