// SURVEY.md §8f N4 — SdBG-level tip trimming on the device-resident graph of N1.
//
// sdbg_pruning::RemoveTips (reference src/assembly/sdbg_pruning.cpp:61-179) over the succinct de Bruijn graph's
// navigation (src/sdbg/sdbg.h:106-121 Forward/Backward on rank/select, :240-330 ComputeIncomings/ComputeOutgoings):
// one thread per edge, the graph = the MHX_BUF_SDBG_* buffers that mhx_sdbg_build_index left in HBM, the result = the
// updated MHX_BUF_SDBG_INVALID bit vector (+ the number of tips removed) — the first graph-cleaning pass of `assemble`
// without the graph ever leaving the GPU.  rank / select are answered from the reference-layout tables (l2 + l1 +
// in-interval popcounts; select = binary search over the intervals between two select samples, then a word scan).
#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

struct DevSdbg {
  const unsigned long long *w, *last, *tip;
  unsigned long long *invalid;
  uint64_t n;
  const long long *w_l2;      // [9][num_l2_w]
  const uint16_t *w_l1;       // [9][num_l1_w]
  const uint32_t *w_sel;      // concatenated, offsets w_sel_off[c]
  const long long *last_l2;
  const uint16_t *last_l1;
  const uint32_t *last_sel;
  uint64_t num_l1_w, num_l2_w, num_l1_b, num_l2_b;
  uint64_t w_sel_off[10];
  uint64_t w_count[9], last_count;
  long long f[6], rank_f[6];
};
constexpr uint64_t kNull = ~0ull;

__device__ __forceinline__ unsigned sd_w(const DevSdbg &g, uint64_t x) { return (unsigned)(g.w[x >> 4] >> (4 * (x & 15))) & 15u; }
__device__ __forceinline__ bool sd_bit(const unsigned long long *v, uint64_t x) { return (v[x >> 6] >> (x & 63)) & 1ull; }
__device__ __forceinline__ bool sd_last_or_tip(const DevSdbg &g, uint64_t x) { return ((g.last[x >> 6] | g.tip[x >> 6]) >> (x & 63)) & 1ull; }
__device__ __forceinline__ bool sd_valid(const DevSdbg &g, uint64_t x) { return !sd_bit(g.invalid, x); }

__device__ __forceinline__ unsigned nib_count(unsigned long long x, unsigned c) {  // nibbles of x equal to c
  unsigned long long y = x ^ ~(0x1111111111111111ull * (unsigned long long)c);
  y &= y >> 2;
  y &= y >> 1;
  return (unsigned)__builtin_popcountll(y & 0x1111111111111111ull);
}
// occurrences of character c in W[0 .. pos]  (RankAndSelect::rank(c, pos), kmrns.h:177-183)
__device__ uint64_t sd_rank_w(const DevSdbg &g, unsigned c, uint64_t pos) {
  const uint64_t itv = (pos + 1) >> 8;  // 256 items per level-1 interval
  uint64_t r = (uint64_t)g.w_l2[c * g.num_l2_w + (itv >> 6)] + g.w_l1[c * g.num_l1_w + itv];
  const uint64_t first = itv << 8, cnt = pos + 1 - first;  // items first .. pos
  const uint64_t w0 = first >> 4;
  uint64_t full = cnt >> 4;
  for (uint64_t i = 0; i < full; ++i) r += nib_count(g.w[w0 + i], c);
  const unsigned rem = (unsigned)(cnt & 15);
  if (rem) {
    // count only the low `rem` nibbles: make the others differ from every c by a per-nibble mask
    unsigned long long x = g.w[w0 + full], y = x ^ ~(0x1111111111111111ull * (unsigned long long)c);
    y &= y >> 2;
    y &= y >> 1;
    r += (unsigned)__builtin_popcountll(y & 0x1111111111111111ull & ((1ull << (4 * rem)) - 1));
  }
  return r;
}
// ones in last[0 .. pos]
__device__ uint64_t sd_rank_last(const DevSdbg &g, uint64_t pos) {
  const uint64_t itv = (pos + 1) >> 10;  // 1024 bits per level-1 interval
  uint64_t r = (uint64_t)g.last_l2[itv >> 6] + g.last_l1[itv];
  const uint64_t first = itv << 10, cnt = pos + 1 - first;
  const uint64_t w0 = first >> 6;
  const uint64_t full = cnt >> 6;
  for (uint64_t i = 0; i < full; ++i) r += (uint64_t)__builtin_popcountll(g.last[w0 + i]);
  const unsigned rem = (unsigned)(cnt & 63);
  if (rem) r += (uint64_t)__builtin_popcountll(g.last[w0 + full] & ((1ull << rem) - 1));
  return r;
}
// position of the (k+1)-th one of last (k 0-based); n if k == #ones  (RankAndSelect::select, kmrns.h:185-191,282-320)
__device__ uint64_t sd_select_last(const DevSdbg &g, uint64_t k) {
  if (k > g.last_count) return kNull;
  if (k == g.last_count) return g.n;
  uint64_t lo = g.last_sel[k >> 12], hi = g.last_sel[(k + 4095) >> 12];
  auto occ = [&](uint64_t i) -> uint64_t { return (uint64_t)g.last_l2[i >> 6] + g.last_l1[i]; };
  while (hi > lo) {  // largest interval whose start count is <= k
    const uint64_t mid = (lo + hi + 1) >> 1;
    if (occ(mid) > k) hi = mid - 1;
    else lo = mid;
  }
  uint64_t remain = k + 1 - occ(lo);
  uint64_t wi = (lo << 10) >> 6;
  for (;; ++wi) {
    const unsigned pc = (unsigned)__builtin_popcountll(g.last[wi]);
    if (pc >= remain) break;
    remain -= pc;
  }
  unsigned long long x = g.last[wi];
  for (uint64_t t = 1; t < remain; ++t) x &= x - 1;  // drop the lowest remain-1 ones
  return (wi << 6) + (uint64_t)__builtin_ctzll(x);
}
// position of the (k+1)-th occurrence of character c in W
__device__ uint64_t sd_select_w(const DevSdbg &g, unsigned c, uint64_t k) {
  if (k > g.w_count[c]) return kNull;
  if (k == g.w_count[c]) return g.n;
  const uint32_t *sel = g.w_sel + g.w_sel_off[c];
  uint64_t lo = sel[k >> 12], hi = sel[(k + 4095) >> 12];
  auto occ = [&](uint64_t i) -> uint64_t { return (uint64_t)g.w_l2[c * g.num_l2_w + (i >> 6)] + g.w_l1[c * g.num_l1_w + i]; };
  while (hi > lo) {
    const uint64_t mid = (lo + hi + 1) >> 1;
    if (occ(mid) > k) hi = mid - 1;
    else lo = mid;
  }
  uint64_t remain = k + 1 - occ(lo);
  uint64_t wi = (lo << 8) >> 4;
  unsigned long long y;
  for (;; ++wi) {
    y = g.w[wi] ^ ~(0x1111111111111111ull * (unsigned long long)c);
    y &= y >> 2;
    y &= y >> 1;
    y &= 0x1111111111111111ull;
    const unsigned pc = (unsigned)__builtin_popcountll(y);
    if (pc >= remain) break;
    remain -= pc;
  }
  for (uint64_t t = 1; t < remain; ++t) y &= y - 1;
  return (wi << 4) + (uint64_t)(__builtin_ctzll(y) >> 2);
}
__device__ __forceinline__ unsigned sd_last_char_of(const DevSdbg &g, uint64_t x) {  // sdbg.h:83-90
  for (unsigned i = 1; i < 6; ++i)
    if (g.f[i] > (long long)x) return i - 1;
  return 6;
}
__device__ uint64_t sd_forward(const DevSdbg &g, uint64_t e) {  // sdbg.h:106-113
  unsigned a = sd_w(g, e);
  if (a > 4) a -= 4;
  const uint64_t count_a = sd_rank_w(g, a, e);
  return sd_select_last(g, (uint64_t)g.rank_f[a] + count_a - 1);
}
__device__ uint64_t sd_backward(const DevSdbg &g, uint64_t e) {  // sdbg.h:115-121
  const unsigned a = sd_last_char_of(g, e);
  const uint64_t count_a = (e == 0 ? 0 : sd_rank_last(g, e - 1)) - (uint64_t)g.rank_f[a];
  return sd_select_w(g, a, count_a);
}
// ComputeIncomings (sdbg.h:240-283).  mode 0: the in-degree; kMustEq0: -1 as soon as one exists; kUnique: the in-degree,
// -1 as soon as a second exists, *one = the incoming edge when there is exactly one
enum { kAny = 0, kMustEq0 = 1, kUnique = 2 };
__device__ int sd_incomings(const DevSdbg &g, uint64_t e, int mode, uint64_t *one) {
  if (!sd_valid(g, e)) return -1;
  const uint64_t first = sd_backward(g, e);
  const unsigned c = sd_w(g, first);
  unsigned count_ones = sd_last_or_tip(g, first);
  int indeg = sd_valid(g, first) ? 1 : 0;
  if (mode == kMustEq0 && indeg) return -1;
  if (indeg && one) *one = first;
  for (uint64_t y = first + 1; count_ones < 5 && y < g.n; ++y) {
    count_ones += sd_last_or_tip(g, y);
    const unsigned cur = sd_w(g, y);
    if (cur == c) break;
    if (cur == c + 4 && sd_valid(g, y)) {
      if (mode == kMustEq0) return -1;
      if (mode == kUnique && indeg == 1) return -1;
      if (one) *one = y;  // (only meaningful when it stays the single one)
      ++indeg;
    }
  }
  return indeg;
}
// ComputeOutgoings (sdbg.h:294-323)
__device__ int sd_outgoings(const DevSdbg &g, uint64_t e, int mode, uint64_t *one) {
  if (!sd_valid(g, e)) return -1;
  int outdeg = 0;
  uint64_t next = sd_forward(g, e);
  do {
    if (sd_valid(g, next)) {
      if (mode == kMustEq0) return -1;
      if (mode == kUnique && outdeg == 1) return -1;
      if (one) *one = next;
      ++outdeg;
    }
    --next;
  } while (next != kNull && !sd_last_or_tip(g, next));
  return outdeg;
}
__device__ __forceinline__ bool sd_indeg_zero(const DevSdbg &g, uint64_t e) { return sd_incomings(g, e, kMustEq0, nullptr) == 0; }
__device__ __forceinline__ bool sd_outdeg_zero(const DevSdbg &g, uint64_t e) { return sd_outgoings(g, e, kMustEq0, nullptr) == 0; }
__device__ __forceinline__ uint64_t sd_unique_prev(const DevSdbg &g, uint64_t e) {
  uint64_t r = 0;
  return sd_incomings(g, e, kUnique, &r) == 1 ? r : kNull;
}
__device__ __forceinline__ uint64_t sd_unique_next(const DevSdbg &g, uint64_t e) {
  uint64_t r = 0;
  return sd_outgoings(g, e, kUnique, &r) == 1 ? r : kNull;
}
__device__ __forceinline__ void bit_set(unsigned long long *v, uint64_t x) { atomicOr(&v[x >> 6], 1ull << (x & 63)); }
__device__ __forceinline__ void bit_unset(unsigned long long *v, uint64_t x) { atomicAnd(&v[x >> 6], ~(1ull << (x & 63))); }

// RemoveTips, first loop (sdbg_pruning.cpp:150-157): everything that is neither a source nor a sink is ignored
__global__ void k_tips_init(DevSdbg g, unsigned long long *__restrict__ ignored) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= g.n) return;
  if (!sd_indeg_zero(g, id) && !sd_outdeg_zero(g, id)) bit_set(ignored, id);
}
// Trim (sdbg_pruning.cpp:61-145), the two walking loops: backward from the sinks (dir 0), forward from the sources (dir 1).
// A path is at most `len` edges, so it is walked twice instead of stored: once to decide, once to mark.
// The nodes a walk starts from are the few that `ignored` does not cover (sources and sinks: ~1 % of the edges): they are
// listed first (one popcount pass over the bitmap + a scan), and the walk kernel runs one thread per LISTED node instead of
// one per node of the graph — 12 launches over 6 x 10^7 threads each were 48 ms, nearly all of it threads that returned at once.
__global__ void k_tips_cand_count(const unsigned long long *__restrict__ ignored, uint64_t n, uint64_t n_words, uint32_t *__restrict__ cnt) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  unsigned long long free_bits = ~ignored[w];
  if ((w + 1) * 64 > n) free_bits &= (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  cnt[w] = (uint32_t)__builtin_popcountll(free_bits);
}
__global__ void k_tips_cand_write(const unsigned long long *__restrict__ ignored, uint64_t n, uint64_t n_words, const uint64_t *__restrict__ off,
                                  uint64_t *__restrict__ cand) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  unsigned long long free_bits = ~ignored[w];
  if ((w + 1) * 64 > n) free_bits &= (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  uint64_t o = off[w];
  for (; free_bits; free_bits &= free_bits - 1) cand[o++] = w * 64 + (uint64_t)__builtin_ctzll(free_bits);
}

__global__ void k_tips_walk(DevSdbg g, int len, int dir, unsigned long long *__restrict__ ignored, unsigned long long *__restrict__ to_remove,
                            unsigned long long *__restrict__ n_tips, const uint64_t *__restrict__ cand, uint64_t n_cand) {
  const uint64_t ci = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= (cand ? n_cand : g.n)) return;
  const uint64_t id = cand ? cand[ci] : ci;
  if (sd_bit(ignored, id)) return;
  if (dir == 0 ? !sd_outdeg_zero(g, id) : !sd_indeg_zero(g, id)) return;
  uint64_t other = kNull, cur = id;
  bool is_tip = false;
  int steps = 0;  // edges appended to the path behind `id`
  for (int i = 1; i < len; ++i) {
    other = dir == 0 ? sd_unique_prev(g, cur) : sd_unique_next(g, cur);
    if (other == kNull) {
      is_tip = dir == 0 ? sd_indeg_zero(g, cur) : sd_outdeg_zero(g, cur);
      break;
    } else if ((dir == 0 ? sd_unique_next(g, other) : sd_unique_prev(g, other)) == kNull) {
      is_tip = true;
      break;
    } else {
      ++steps;
      cur = other;
    }
  }
  if (!is_tip) return;
  // the path: id, then `steps` unique predecessors / successors (the graph does not change inside a Trim call)
  uint64_t p = id;
  bit_set(to_remove, p);
  for (int s = 0; s < steps; ++s) {
    p = dir == 0 ? sd_unique_prev(g, p) : sd_unique_next(g, p);
    bit_set(to_remove, p);
  }
  atomicAdd(n_tips, 1ull);
  bit_set(ignored, id);
  bit_set(ignored, p);  // path.back()
  if (other != kNull) bit_unset(ignored, other);
}
__global__ void k_tips_apply(unsigned long long *__restrict__ invalid, unsigned long long *__restrict__ to_remove, uint64_t n_words) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) {
    invalid[i] |= to_remove[i];
    to_remove[i] = 0;
  }
}

int sdbg_remove_tips(mhx_ctx *c, const mhx_sdbg_index_info *info, int max_tip_len, uint64_t *n_removed) {
  hipStream_t st = c->stream;
  auto buf = [&](int which) -> DevBuf & {
    auto it = c->results.find(which);
    if (it == c->results.end() || !it->second.p) throw Error("sdbg_remove_tips: run mhx_sdbg_build_index first");
    return it->second;
  };
  DevSdbg g{};
  g.w = buf(MHX_BUF_SDBG_W).as<unsigned long long>();
  g.last = buf(MHX_BUF_SDBG_LAST).as<unsigned long long>();
  g.tip = buf(MHX_BUF_SDBG_TIP).as<unsigned long long>();
  g.invalid = buf(MHX_BUF_SDBG_INVALID).as<unsigned long long>();
  g.n = info->n_items;
  g.w_l2 = buf(MHX_BUF_SDBG_RS_W_L2).as<long long>();
  g.w_l1 = buf(MHX_BUF_SDBG_RS_W_L1).as<uint16_t>();
  g.w_sel = buf(MHX_BUF_SDBG_RS_W_SEL).as<uint32_t>();
  g.last_l2 = buf(MHX_BUF_SDBG_RS_LAST_L2).as<long long>();
  g.last_l1 = buf(MHX_BUF_SDBG_RS_LAST_L1).as<uint16_t>();
  g.last_sel = buf(MHX_BUF_SDBG_RS_LAST_SEL).as<uint32_t>();
  g.num_l1_w = info->num_l1_w;
  g.num_l2_w = info->num_l2_w;
  g.num_l1_b = info->num_l1_bits;
  g.num_l2_b = info->num_l2_bits;
  for (int i = 0; i < 10; ++i) g.w_sel_off[i] = info->w_sel_offset[i];
  for (int i = 0; i < 9; ++i) g.w_count[i] = info->w_char_count[i];
  g.last_count = info->ones_in_last;
  for (int i = 0; i < 6; ++i) {
    g.f[i] = info->f[i];
    g.rank_f[i] = info->rank_f[i];
  }
  if (n_removed) *n_removed = 0;
  if (!g.n || max_tip_len <= 0) return 0;
  const uint64_t nw = div_ceil(g.n, 64);
  unsigned long long *ignored = c->ws("tips_ignored", nw * 8 + 8).as<unsigned long long>();
  unsigned long long *to_remove = c->ws("tips_remove", nw * 8 + 8).as<unsigned long long>();
  unsigned long long *cnt = c->ws("tips_count", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(ignored, 0, nw * 8, st));
  MHX_HIP(hipMemsetAsync(to_remove, 0, nw * 8, st));
  MHX_HIP(hipMemsetAsync(cnt, 0, 8, st));
  const unsigned grid = (unsigned)div_ceil(g.n, 256);
  MHX_LAUNCH(c, "tips_init", (double)g.n * 2, hipLaunchKernelGGL(k_tips_init, dim3(grid), dim3(256), 0, st, g, ignored));
  const bool listed = c->opt("tips_candidate_list", 1) != 0;
  uint32_t *wcnt = listed ? c->ws("tips_word_cnt", (nw + 1) * 4).as<uint32_t>() : nullptr;
  uint64_t *woff = listed ? c->ws("tips_word_off", (nw + 2) * 8).as<uint64_t>() : nullptr;
  auto walk = [&](int len, int dir) {
    if (!listed) {
      MHX_LAUNCH(c, "tips_walk", (double)g.n, hipLaunchKernelGGL(k_tips_walk, dim3(grid), dim3(256), 0, st, g, len, dir, ignored, to_remove, cnt,
                                                                   (const uint64_t *)nullptr, (uint64_t)0));
      return;
    }
    const unsigned gw = (unsigned)div_ceil(nw, 256);
    uint64_t n_cand = 0;
    MHX_LAUNCH(c, "tips_candidates", (double)nw * 16,
               hipLaunchKernelGGL(k_tips_cand_count, dim3(gw), dim3(256), 0, st, ignored, g.n, nw, wcnt));
    exclusive_scan_u32_u64(c, wcnt, woff, nw, woff + nw + 1);
    MHX_HIP(hipMemcpyAsync(&n_cand, woff + nw + 1, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    if (!n_cand) return;
    uint64_t *cand = c->ws("tips_cand", n_cand * 8 + 64).as<uint64_t>();
    hipLaunchKernelGGL(k_tips_cand_write, dim3(gw), dim3(256), 0, st, ignored, g.n, nw, woff, cand);
    MHX_LAUNCH(c, "tips_walk", (double)n_cand * 64,
               hipLaunchKernelGGL(k_tips_walk, dim3((unsigned)div_ceil(n_cand, 256)), dim3(256), 0, st, g, len, dir, ignored, to_remove, cnt, cand, n_cand));
  };
  auto trim = [&](int len) {
    walk(len, 0);
    walk(len, 1);
    hipLaunchKernelGGL(k_tips_apply, dim3((unsigned)div_ceil(nw, 256)), dim3(256), 0, st, g.invalid, to_remove, nw);
  };
  for (int len = 2; len < max_tip_len; len *= 2) trim(len);  // sdbg_pruning.cpp:159-166
  trim(max_tip_len);
  MHX_HIP(hipGetLastError());
  unsigned long long h = 0;
  MHX_HIP(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (n_removed) *n_removed = h;
  return 0;
}

}  // namespace mhx
