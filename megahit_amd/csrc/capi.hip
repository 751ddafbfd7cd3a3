// C-ABI entry points of libmhx.so (declared in include/mhx.h) and context plumbing.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdarg>
#include <cstdlib>
#include <dlfcn.h>
#include <fstream>
#include <sstream>

#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// host time spent inside hipMalloc / hipFree by this process (mhx_alloc_stats): a process that starts while the driver
// still reclaims the device memory of its predecessor waits HERE, not in its kernels
static std::atomic<uint64_t> g_alloc_ns{0}, g_alloc_bytes{0}, g_alloc_calls{0}, g_free_ns{0};
static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void DevBuf::reserve(size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (cap >= bytes && p) return;
  if (p && cap) {
    const uint64_t t0 = now_ns();
    (void)hipFree(p);
    g_free_ns += now_ns() - t0;
  }
  p = nullptr;
  cap = 0;
  // grow with a little headroom so that repeated calls with slightly different sizes do not re-allocate
  size_t want = bytes + bytes / 16 + 256;
  const uint64_t t0 = now_ns();
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    want = bytes;
    e = hipMalloc(&p, want);
  }
  g_alloc_ns += now_ns() - t0;
  g_alloc_calls += 1;
  if (e == hipSuccess) g_alloc_bytes += want;
  if (e != hipSuccess) {
    p = nullptr;
    char b[256];
    snprintf(b, sizeof b, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    throw Error(b);
  }
  cap = want;
}
void DevBuf::release() {
  if (p && cap) {
    const uint64_t t0 = now_ns();
    (void)hipFree(p);
    g_free_ns += now_ns() - t0;
  }
  p = nullptr;
  cap = used = 0;
}

}  // namespace mhx

void mhx_ctx::prof_begin(const char *name, double bytes) {
  if (!profiling) return;
  hipEvent_t a, b;
  if (event_pool.size() >= 2) {
    a = event_pool.back(); event_pool.pop_back();
    b = event_pool.back(); event_pool.pop_back();
  } else {
    MHX_HIP(hipEventCreate(&a));
    MHX_HIP(hipEventCreate(&b));
  }
  MHX_HIP(hipEventRecord(a, stream));
  pending.push_back({name, a, b, bytes});
}
void mhx_ctx::prof_end() {
  if (!profiling) return;
  MHX_HIP(hipEventRecord(pending.back().b, stream));
  if (pending.size() > 4096) prof_collect();
}
void mhx_ctx::prof_collect() {
  if (pending.empty()) return;
  MHX_HIP(hipStreamSynchronize(stream));
  for (auto &pe : pending) {
    float ms = 0;
    MHX_HIP(hipEventElapsedTime(&ms, pe.a, pe.b));
    mhx::KernelStat &ks = stats[pe.name];
    ks.launches++;
    ks.ms += ms;
    ks.bytes += pe.bytes;
    event_pool.push_back(pe.a);
    event_pool.push_back(pe.b);
  }
  pending.clear();
}

namespace mhx {

// ---- sequence upload -------------------------------------------------------
__global__ void k_fixed_starts(uint64_t *start, uint64_t n_seqs, uint32_t fixed_len) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_seqs) start[i] = i * fixed_len;
}

constexpr size_t kSeqPadWords = 64;

void upload_fixed_starts(mhx_ctx *c) {
  SeqSet &s = c->seqs;
  hipLaunchKernelGGL(k_fixed_starts, dim3((unsigned)div_ceil(s.n_seqs + 1, 256)), dim3(256), 0, c->stream, s.start.as<uint64_t>(), s.n_seqs,
                     s.fixed_len);
  MHX_HIP(hipGetLastError());
  MHX_HIP(hipStreamSynchronize(c->stream));
}

void upload_sequences(mhx_ctx *c, const uint32_t *packed, uint64_t n_words, uint64_t n_seqs, uint32_t fixed_len,
                      const uint64_t *start_pos) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  c->agg_valid = false; c->solid_plain_k = 0;
  s.n_seqs = n_seqs;
  s.fixed_len = start_pos ? 0 : fixed_len;
  s.n_words = n_words;
  s.words.reserve((n_words + kSeqPadWords) * 4);
  s.start.reserve((n_seqs + 2) * 8);
  if (n_words) upload_pinned(c, s.words.p, packed, n_words * 4);
  MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + n_words, 0, kSeqPadWords * 4, st));
  if (start_pos) {
    MHX_HIP(hipMemcpyAsync(s.start.p, start_pos, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    s.n_bases = start_pos[n_seqs];
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n_seqs; ++i) {
      uint64_t L = start_pos[i + 1] - start_pos[i];
      if (L > mx) mx = (uint32_t)L;
    }
    s.max_len = mx;
    // (a start array that describes reads of ONE length is a fixed-length library: the paths that need no per-read table)
    if (n_seqs && start_pos[0] == 0 && s.n_bases == (uint64_t)mx * n_seqs) s.fixed_len = mx;
  } else {
    hipLaunchKernelGGL(k_fixed_starts, dim3((unsigned)div_ceil(n_seqs + 1, 256)), dim3(256), 0, st, s.start.as<uint64_t>(), n_seqs,
                       fixed_len);
    MHX_HIP(hipGetLastError());
    s.n_bases = n_seqs * (uint64_t)fixed_len;
    s.max_len = fixed_len;
  }
  if (s.n_bases > n_words * 16) throw Error("load_sequences: start_pos/n_seqs exceed the packed buffer");
  s.mult.used = 0;
  MHX_HIP(hipStreamSynchronize(st));
}

// append host sequences behind the device-resident set (bit-unaligned: one thread per output word)
__global__ void k_append_words(uint32_t *__restrict__ dst, uint64_t old_bases, uint64_t new_bases, const uint32_t *__restrict__ src) {
  const uint64_t w = old_bases / 16 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 16 >= new_bases) return;
  uint32_t word = 0;
  for (int j = 0; j < 16; ++j) {
    const uint64_t b = w * 16 + j;
    unsigned ch = 0;
    if (b < old_bases) ch = (dst[w] >> (30 - 2 * j)) & 3u;
    else if (b < new_bases) ch = base_at(src, b - old_bases);
    word |= ch << (30 - 2 * j);
  }
  dst[w] = word;
}
__global__ void k_offset_starts(uint64_t *__restrict__ start, uint64_t from, uint64_t n_new, const uint64_t *__restrict__ rel, uint32_t fixed_len,
                                uint64_t base) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_new) start[from + i] = base + (rel ? rel[i] : i * fixed_len);
}

static void grow_keep(mhx_ctx *c, DevBuf &b, size_t new_bytes, size_t keep_bytes) {
  if (b.cap >= new_bytes) return;
  DevBuf nb;
  nb.reserve(new_bytes);
  if (keep_bytes && b.p) MHX_HIP(hipMemcpyAsync(nb.p, b.p, keep_bytes, hipMemcpyDeviceToDevice, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  nb.used = b.used;
  b.release();
  b = nb;
}

void append_sequences(mhx_ctx *c, const uint32_t *packed, uint64_t n_words, uint64_t n_new, uint32_t fixed_len, const uint64_t *start_pos,
                      const uint16_t *mult) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  c->agg_valid = false; c->solid_plain_k = 0;
  if (n_new == 0) return;
  const uint64_t add_bases = start_pos ? start_pos[n_new] : n_new * (uint64_t)fixed_len;
  if (add_bases > n_words * 16) throw Error("append_sequences: start_pos/n_seqs exceed the packed buffer");
  const uint64_t old_bases = s.n_bases, new_bases = old_bases + add_bases, new_words = div_ceil(new_bases, 16);
  const uint64_t old_seqs = s.n_seqs, new_seqs = old_seqs + n_new;
  const bool had_mult = s.mult.used >= old_seqs * 2 && (old_seqs > 0 || mult);
  grow_keep(c, s.words, (new_words + kSeqPadWords) * 4, (s.n_words + 1) * 4);
  grow_keep(c, s.start, (new_seqs + 2) * 8, (old_seqs + 1) * 8);
  uint32_t *src = c->ws("append_words", (n_words + kSeqPadWords) * 4).as<uint32_t>();
  MHX_HIP(hipMemcpyAsync(src, packed, n_words * 4, hipMemcpyHostToDevice, st));
  const uint64_t first_w = old_bases / 16;
  const uint64_t zero_from = (old_bases % 16 == 0) ? first_w : first_w + 1;
  MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + zero_from, 0, (new_words + kSeqPadWords - zero_from) * 4, st));
  const uint64_t n_w = new_words - first_w;
  MHX_LAUNCH(c, "append_words", (double)n_w * 8,
             hipLaunchKernelGGL(k_append_words, dim3((unsigned)div_ceil(n_w, 256)), dim3(256), 0, st, s.words.as<uint32_t>(), old_bases,
                                new_bases, src));
  uint64_t *rel = nullptr;
  if (start_pos) {
    rel = c->ws("append_start", (n_new + 1) * 8).as<uint64_t>();
    MHX_HIP(hipMemcpyAsync(rel, start_pos, (n_new + 1) * 8, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(k_offset_starts, dim3((unsigned)div_ceil(n_new + 1, 256)), dim3(256), 0, st, s.start.as<uint64_t>(), old_seqs, n_new, rel,
                     fixed_len, old_bases);
  MHX_HIP(hipGetLastError());
  if (had_mult || mult) {
    grow_keep(c, s.mult, (new_seqs + 1) * 2, old_seqs * 2);
    if (mult) MHX_HIP(hipMemcpyAsync(s.mult.as<uint16_t>() + old_seqs, mult, n_new * 2, hipMemcpyHostToDevice, st));
    else MHX_HIP(hipMemsetAsync(s.mult.as<uint16_t>() + old_seqs, 0, n_new * 2, st));
    s.mult.used = new_seqs * 2;
  }
  uint32_t mx = s.max_len;
  if (start_pos) {
    for (uint64_t i = 0; i < n_new; ++i) mx = std::max<uint32_t>(mx, (uint32_t)(start_pos[i + 1] - start_pos[i]));
  } else mx = std::max(mx, fixed_len);
  const bool still_fixed = !start_pos && (old_seqs == 0 || s.fixed_len == fixed_len);
  s.fixed_len = still_fixed ? fixed_len : 0;
  s.max_len = mx;
  s.n_seqs = new_seqs;
  s.n_bases = new_bases;
  s.n_words = new_words;
  MHX_HIP(hipStreamSynchronize(st));
}

// .bin record stream -> (reversed) concatenated store, on the GPU.
// Host walks the record headers (one uint32 per read) to get lengths; bases are moved by a kernel.
__global__ void k_unpack_records(const uint32_t *__restrict__ rec, const uint64_t *__restrict__ rec_off, const uint64_t *__restrict__ start,
                                 uint64_t n_seqs, int reverse, uint32_t *__restrict__ out_words, uint64_t n_out_words) {
  // one thread per OUTPUT word: gathers its 16 bases (no atomics, coalesced stores)
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_out_words) return;
  const uint64_t b0 = w * 16;
  // sequence containing base b0
  uint64_t lo = 0, hi = n_seqs;
  while (hi - lo > 1) {
    uint64_t mid = (lo + hi) >> 1;
    if (start[mid] <= b0) lo = mid;
    else hi = mid;
  }
  uint64_t sid = lo;
  uint32_t word = 0;
  const uint64_t total = start[n_seqs];
  for (int j = 0; j < 16; ++j) {
    const uint64_t b = b0 + j;
    if (b >= total) break;
    while (b >= start[sid + 1]) ++sid;
    const uint64_t L = start[sid + 1] - start[sid];
    uint64_t off = b - start[sid];
    if (reverse) off = L - 1 - off;
    const uint32_t *r = rec + rec_off[sid] + 1;  // skip the length word
    const unsigned ch = (r[off >> 4] >> (30 - 2 * (unsigned)(off & 15))) & 3u;
    word |= ch << (30 - 2 * j);
  }
  out_words[w] = word;
}

// packed (k+1)-mer edges as EdgeWriter lays them out (words_per_edge words per edge: chars MSB-first, the multiplicity in
// the low 16 bits of the last word; kmer_counter.cpp:32-52, edge_reader.h:24-52) -> gap-free sequence store + multiplicities
__global__ void k_unpack_edges(const uint32_t *__restrict__ raw, uint64_t n_edges, uint32_t len, uint32_t wpe, uint32_t *__restrict__ out_words,
                               uint64_t n_out_words, uint16_t *__restrict__ mult) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_edges) mult[w] = (uint16_t)(raw[w * wpe + wpe - 1] & 0xFFFFu);
  if (w >= n_out_words) return;
  const uint64_t b0 = w * 16, total = n_edges * (uint64_t)len;
  uint64_t e = b0 / len;
  uint32_t off = (uint32_t)(b0 - e * len);
  const uint32_t *r = raw + e * wpe;
  uint32_t word = 0;
  for (int j = 0; j < 16; ++j) {
    if (b0 + j >= total) break;
    word |= ((r[off >> 4] >> (30 - 2 * (off & 15))) & 3u) << (30 - 2 * j);
    if (++off == len) {
      off = 0;
      r += wpe;
    }
  }
  out_words[w] = word;
}
void upload_edges(mhx_ctx *c, const uint32_t *raw, uint64_t n_edges, uint32_t k, uint32_t wpe) {
  hipStream_t st = c->stream;
  SeqSet &s = c->seqs;
  c->agg_valid = false; c->solid_plain_k = 0;
  const uint32_t len = k + 1;
  if (wpe != (len * 2 + 16 + 31) / 32) throw Error("load_edges: words_per_edge does not match k");
  uint32_t *d_raw = c->ws("edges_raw", (n_edges * wpe + 4) * 4).as<uint32_t>();
  if (n_edges) upload_pinned(c, d_raw, raw, n_edges * wpe * 4);
  const uint64_t bases = n_edges * (uint64_t)len, n_out = div_ceil(bases, 16);
  // room for the mercy edges that may be appended behind them (SeqToSdbg::Initialize, seq_to_sdbg.cpp:370-378: 25 % by default,
  // MEGAHIT_NUM_MERCY_FACTOR otherwise — the CLI passes it on as edges_reserve_permille): the store does not have to grow then
  const uint64_t permille = (uint64_t)std::min<long long>(std::max<long long>(c->opt("edges_reserve_permille", 1000), 1000), 100000);
  const uint64_t n_res = n_edges * permille / 1000;
  s.words.reserve((div_ceil(n_res * (uint64_t)len, 16) + kSeqPadWords) * 4);
  s.start.reserve((n_res + 2) * 8);
  s.mult.reserve((n_res + 1) * 2);
  s.mult.used = n_edges * 2;
  MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + n_out, 0, kSeqPadWords * 4, st));
  const uint64_t n_thr = std::max(n_out, n_edges);
  if (n_thr)
    MHX_LAUNCH(c, "unpack_edges", (double)n_edges * wpe * 4 + (double)n_out * 4,
               hipLaunchKernelGGL(k_unpack_edges, dim3((unsigned)div_ceil(n_thr, 256)), dim3(256), 0, st, d_raw, n_edges, len, wpe, s.words.as<uint32_t>(),
                                  n_out, s.mult.as<uint16_t>()));
  s.n_seqs = n_edges;
  s.n_bases = bases;
  s.n_words = n_out;
  s.max_len = n_edges ? len : 0;
  s.fixed_len = len;
  s.h_start.clear();
  upload_fixed_starts(c);
}

// Host -> device copy of a large pageable (e.g. mmap'ed) buffer through two pinned staging buffers: a few host threads
// fill one buffer from the page cache while the DMA engine drains the other.  A plain hipMemcpy of pageable memory
// staged 400 MB of reads in ~0.3 s; this takes what the slower of (page-cache memcpy, PCIe) takes.
void upload_pinned(mhx_ctx *c, void *d_dst, const void *h_src, size_t bytes) {
  hipStream_t st = c->stream;
  if (bytes < (8u << 20)) {
    MHX_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
    MHX_HIP(hipStreamSynchronize(st));
    return;
  }
  constexpr size_t kChunk = 32u << 20;
  constexpr int kThreads = 4;
  if (!c->pinned[0]) {
    for (int i = 0; i < 2; ++i) {
      MHX_HIP(hipHostMalloc(&c->pinned[i], kChunk, hipHostMallocDefault));
      MHX_HIP(hipEventCreateWithFlags(&c->pinned_free[i], hipEventDisableTiming));
    }
  }
  const char *src = static_cast<const char *>(h_src);
  char *dst = static_cast<char *>(d_dst);
  int b = 0;
  bool used[2] = {false, false};
  for (size_t off = 0; off < bytes; off += kChunk, b ^= 1) {
    const size_t len = std::min(kChunk, bytes - off);
    if (used[b]) MHX_HIP(hipEventSynchronize(c->pinned_free[b]));
    char *stage = static_cast<char *>(c->pinned[b]);
    std::thread th[kThreads - 1];
    const size_t part = (len / kThreads + 63) & ~(size_t)63;
    for (int t = 1; t < kThreads; ++t) {
      const size_t lo = std::min(len, part * t), hi = std::min(len, part * (t + 1));
      th[t - 1] = std::thread([=] { if (hi > lo) memcpy(stage + lo, src + off + lo, hi - lo); });
    }
    memcpy(stage, src + off, std::min(len, part));
    for (auto &t : th) t.join();
    MHX_HIP(hipMemcpyAsync(dst + off, stage, len, hipMemcpyHostToDevice, st));
    MHX_HIP(hipEventRecord(c->pinned_free[b], st));
    used[b] = true;
  }
  MHX_HIP(hipStreamSynchronize(st));
}

// every record holds exactly L bases (rw = 1 + ceil(L/16) words per record): one thread per OUTPUT word
__global__ void k_unpack_fixed(const uint32_t *__restrict__ rec, uint64_t n_seqs, uint32_t L, uint32_t rw, int reverse,
                               uint32_t *__restrict__ out_words, uint64_t n_out_words) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_out_words) return;
  const uint64_t b0 = w * 16, total = n_seqs * (uint64_t)L;
  uint64_t sid = b0 / L;
  uint32_t off = (uint32_t)(b0 - sid * L);
  const uint32_t *r = rec + sid * rw + 1;
  uint32_t word = 0;
  for (int j = 0; j < 16; ++j) {
    if (b0 + j >= total) break;
    const uint32_t o = reverse ? L - 1 - off : off;
    word |= ((r[o >> 4] >> (30 - 2 * (o & 15))) & 3u) << (30 - 2 * j);
    if (++off == L) {
      off = 0;
      r += rw;
    }
  }
  out_words[w] = word;
}
__global__ void k_check_fixed(const uint32_t *__restrict__ rec, uint64_t n_seqs, uint32_t L, uint32_t rw, uint32_t *__restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seqs && rec[i * rw] != L) atomicOr(bad, 1u);
}

// the common library: every read has the same length (checked on the device) -> no per-record tables at all
static bool upload_bin_records_fixed(mhx_ctx *c, const uint32_t *records, uint64_t n_words, uint64_t n_seqs, int reverse) {
  if (!n_seqs || !n_words) return false;
  const uint32_t L = records[0], rw = 1 + (L + 15) / 16;
  if (L == 0 || n_words != n_seqs * (uint64_t)rw) return false;
  hipStream_t st = c->stream;
  SeqSet &s = c->seqs;
  DevBuf &d_rec = c->ws("bin_records", (n_words + 4) * 4);
  upload_pinned(c, d_rec.p, records, n_words * 4);
  uint32_t *bad = c->ws("bin_bad", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(bad, 0, 4, st));
  hipLaunchKernelGGL(k_check_fixed, dim3((unsigned)div_ceil(n_seqs, 256)), dim3(256), 0, st, d_rec.as<uint32_t>(), n_seqs, L, rw, bad);
  const uint64_t bases = n_seqs * (uint64_t)L, n_out_words = div_ceil(bases, 16);
  s.words.reserve((n_out_words + kSeqPadWords) * 4);
  MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + n_out_words, 0, kSeqPadWords * 4, st));
  MHX_LAUNCH(c, "unpack_records", (double)n_words * 4 + (double)n_out_words * 4,
             hipLaunchKernelGGL(k_unpack_fixed, dim3((unsigned)div_ceil(n_out_words, 256)), dim3(256), 0, st, d_rec.as<uint32_t>(), n_seqs, L, rw,
                                reverse, s.words.as<uint32_t>(), n_out_words));
  uint32_t h_bad = 1;
  MHX_HIP(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h_bad) return false;  // some record has another length: the general path
  s.n_seqs = n_seqs;
  s.n_bases = bases;
  s.n_words = n_out_words;
  s.max_len = L;
  s.fixed_len = L;
  s.mult.used = 0;
  s.h_start.clear();
  s.start.reserve((n_seqs + 2) * 8);
  upload_fixed_starts(c);
  return true;
}

void upload_bin_records(mhx_ctx *c, const uint32_t *records, uint64_t n_words, uint64_t n_seqs, int reverse) {
  c->agg_valid = false; c->solid_plain_k = 0;
  if (upload_bin_records_fixed(c, records, n_words, n_seqs, reverse)) return;
  // lengths (host): an empty read becomes a 1-base 'A' (sequence_package.h:275-281)
  std::vector<uint64_t> rec_off(n_seqs + 1), start(n_seqs + 1);
  uint64_t pos = 0, bases = 0;
  uint32_t mx = 0;
  bool has_empty = false;
  for (uint64_t i = 0; i < n_seqs; ++i) {
    if (pos >= n_words) throw Error("load_bin_records: truncated record stream");
    uint32_t L = records[pos];
    rec_off[i] = pos;
    start[i] = bases;
    pos += 1 + (L + 15) / 16;
    if (L == 0) { has_empty = true; L = 1; }
    bases += L;
    if (L > mx) mx = L;
  }
  if (pos > n_words) throw Error("load_bin_records: truncated record stream");
  start[n_seqs] = bases;
  hipStream_t st = c->stream;
  SeqSet &s = c->seqs;
  // an empty read's record has no payload word: give it one zero word by copying records with a pad
  std::vector<uint32_t> patched;
  const uint32_t *src = records;
  uint64_t src_words = n_words;
  if (has_empty) {
    patched.reserve(n_words + 1024);
    for (uint64_t i = 0; i < n_seqs; ++i) {
      uint32_t L = records[rec_off[i]];
      uint64_t nw = (L + 15) / 16;
      uint64_t new_off = patched.size();
      patched.push_back(L ? L : 1u);
      if (L == 0) patched.push_back(0u);
      else patched.insert(patched.end(), records + rec_off[i] + 1, records + rec_off[i] + 1 + nw);
      rec_off[i] = new_off;
    }
    src = patched.data();
    src_words = patched.size();
  }
  DevBuf &d_rec = c->ws("bin_records", (src_words + 4) * 4);
  DevBuf &d_off = c->ws("bin_rec_off", (n_seqs + 1) * 8);
  const uint64_t n_out_words = div_ceil(bases, 16);
  s.words.reserve((n_out_words + kSeqPadWords) * 4);
  s.start.reserve((n_seqs + 2) * 8);
  upload_pinned(c, d_rec.p, src, src_words * 4);
  MHX_HIP(hipMemcpyAsync(d_off.p, rec_off.data(), (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
  MHX_HIP(hipMemcpyAsync(s.start.p, start.data(), (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
  MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + n_out_words, 0, kSeqPadWords * 4, st));
  if (n_out_words) {
    MHX_LAUNCH(c, "unpack_records", (double)src_words * 4 + (double)n_out_words * 4,
               hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)div_ceil(n_out_words, 256)), dim3(256), 0, st, d_rec.as<uint32_t>(),
                                  d_off.as<uint64_t>(), s.start.as<uint64_t>(), n_seqs, reverse, s.words.as<uint32_t>(), n_out_words));
  }
  MHX_HIP(hipStreamSynchronize(st));
  s.n_seqs = n_seqs;
  s.n_bases = bases;
  s.n_words = n_out_words;
  s.max_len = mx;
  s.fixed_len = 0;
  // fixed-length fast path when every read has the same length
  if (n_seqs && bases == (uint64_t)mx * n_seqs) s.fixed_len = mx;
  s.mult.used = 0;
}

}  // namespace mhx

// ---------------------------------------------------------------------------
#define MHX_TRY(...)                          \
  try {                                       \
    __VA_ARGS__;                              \
    return 0;                                 \
  } catch (const std::exception &e) {         \
    mhx::set_error("%s", e.what());           \
    return -1;                                \
  }

extern "C" {

const char *mhx_last_error(void) { return mhx::g_err; }
const char *mhx_version(void) { return "mhx 0.1 (gfx950)"; }

int mhx_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    mhx::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}

static void load_tuning(mhx_ctx *c);

mhx_ctx *mhx_create(int device) {
  try {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw mhx::Error("no HIP device available (libmhx has no CPU fallback)");
    if (device < 0 || device >= n) throw mhx::Error("invalid device ordinal");
    MHX_HIP(hipSetDevice(device));
    mhx_ctx *c = new mhx_ctx();
    c->device = device;
    MHX_HIP(hipStreamCreate(&c->stream));
    {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    }
    load_tuning(c);
    return c;
  } catch (const std::exception &e) {
    mhx::set_error("%s", e.what());
    return nullptr;
  }
}

int mhx_trim(mhx_ctx *c) {
  MHX_TRY({
    MHX_HIP(hipStreamSynchronize(c->stream));
    for (auto &kv : c->work) kv.second.release();
    c->work.clear();
    // the sorted-items view aliases a workspace
    auto it = c->results.find(MHX_BUF_SORTED_ITEMS);
    if (it != c->results.end()) c->results.erase(it);
  })
}

// Forget every input, result and mode of the handle but keep its device buffers (grow-only allocations, the pinned
// staging buffers, the streams): the next job on this handle starts like on a new one without paying for allocations.
int mhx_reset(mhx_ctx *c) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    MHX_HIP(hipStreamSynchronize(c->stream));
    c->seqs.n_seqs = c->seqs.n_bases = c->seqs.n_words = 0;
    c->seqs.fixed_len = c->seqs.max_len = 0;
    c->seqs.mult.used = 0;
    c->seqs.h_start.clear();
    for (auto &kv : c->results) kv.second.used = 0;
    auto it = c->results.find(MHX_BUF_SORTED_ITEMS);  // a view into a workspace, not an allocation
    if (it != c->results.end()) c->results.erase(it);
    c->sorted_item_words = 0;
    c->my_part = 0;
    c->n_parts = 1;
    c->part_begin.clear();
    c->pos_base = c->global_bases = 0;
    c->agg_valid = false; c->solid_plain_k = 0;
    c->agg_n = 0;
    c->n_route = 0;
    c->pre_hist_buf = nullptr;
    // (whatever a failed request left half-way: a deferred first pass, a filter handed to an extraction)
    c->gen_first_pass = nullptr;
    c->gen_buf = nullptr;
    c->gen_n = c->gen_slots = 0;
    c->s1_defer_items = c->s2_filter_in_extract = c->s1_filter_in_gen = false;
    c->s1_density = 0;
    c->filter_kept = 0;
    c->filter_on = c->accumulate = false;
    c->filter_expected = c->filter_batch_bytes = 0;
    c->global_marks_inverted = false;
    c->s1_acc_bits = c->mercy_acc_n = 0;
    c->s1_acc_k = c->s1_acc_m = c->count_acc_k = c->count_acc_m = 0;
    c->n_marks = c->dist_local_solid = 0;
    c->dist_s2_agg = false;
    c->options.clear();
    c->prof_collect();
    c->stats.clear();
  })
}

void mhx_destroy(mhx_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto &kv : c->work) kv.second.release();
  for (auto &kv : c->results) kv.second.release();
  c->seqs.words.release();
  c->seqs.start.release();
  c->seqs.mult.release();
  for (auto &pe : c->pending) {
    (void)hipEventDestroy(pe.a);
    (void)hipEventDestroy(pe.b);
  }
  for (auto ev : c->event_pool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 2; ++i) {
    if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
    if (c->pinned_free[i]) (void)hipEventDestroy(c->pinned_free[i]);
  }
  for (hipStream_t ss : c->side_streams) (void)hipStreamDestroy(ss);
  for (hipEvent_t ev : c->side_events) (void)hipEventDestroy(ev);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

int mhx_synchronize(mhx_ctx *c) { MHX_TRY(MHX_HIP(hipStreamSynchronize(c->stream))) }

extern "C" const char *mhx_last_s1_plan(const mhx_ctx *c) { return c ? c->last_s1_plan.c_str() : ""; }

long long mhx_ctx::opt(const char *name, long long dflt) const {
  auto it = options.find(name);
  if (it != options.end()) return it->second;
  std::string env = "MHX_";
  for (const char *p = name; *p; ++p) env += (char)toupper((unsigned char)*p);
  const char *e = getenv(env.c_str());
  if (e && *e) return atoll(e);
  auto tu = tuned.find(name);
  return tu != tuned.end() ? tu->second : dflt;
}

// mhx_tuning.conf beside libmhx.so: the knob settings measured best on this installation (tools/ab_options.py writes it from
// an A/B on the box); `name = value` or `name value` per line, '#' starts a comment.  Every knob only chooses between
// code paths that produce identical results.
static void load_tuning(mhx_ctx *c) {
  if (getenv("MHX_NO_TUNING")) return;
  std::string path;
  if (const char *f = getenv("MHX_TUNING_FILE")) path = f;
  else {
    Dl_info info;
    if (!dladdr((const void *)&mhx_synchronize, &info) || !info.dli_fname) return;
    path = info.dli_fname;
    const size_t slash = path.rfind('/');
    path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/mhx_tuning.conf";
  }
  std::ifstream in(path);
  if (!in) return;
  std::string line;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.resize(hash);
    for (char &ch : line)
      if (ch == '=') ch = ' ';
    std::istringstream ls(line);
    std::string name;
    long long v;
    if (ls >> name >> v) c->tuned[name] = v;
  }
  if (getenv("MHX_VERBOSE")) fprintf(stderr, "[mhx] %zu tuned defaults from %s\n", c->tuned.size(), path.c_str());
}
long long mhx_get_option(mhx_ctx *c, const char *name, long long dflt) { return c && name ? c->opt(name, dflt) : dflt; }
int mhx_set_option(mhx_ctx *c, const char *name, long long value) {
  if (!c || !name) return -1;
  c->options[name] = value;
  return 0;
}

int mhx_load_sequences(mhx_ctx *c, const uint32_t *packed, uint64_t n_words, uint64_t n_seqs, uint32_t fixed_len,
                       const uint64_t *start_pos) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::upload_sequences(c, packed, n_words, n_seqs, fixed_len, start_pos);
  })
}
int mhx_append_sequences(mhx_ctx *c, const uint32_t *packed, uint64_t n_words, uint64_t n_seqs, uint32_t fixed_len,
                         const uint64_t *start_pos, const uint16_t *mult) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::append_sequences(c, packed, n_words, n_seqs, fixed_len, start_pos, mult);
  })
}
int mhx_load_bin_records(mhx_ctx *c, const uint32_t *records, uint64_t n_words, uint64_t n_seqs, int reverse) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::upload_bin_records(c, records, n_words, n_seqs, reverse);
  })
}
int mhx_load_edges(mhx_ctx *c, const uint32_t *edges, uint64_t n_edges, uint32_t k, uint32_t words_per_edge) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::upload_edges(c, edges, n_edges, k, words_per_edge);
  })
}
int mhx_load_multiplicity(mhx_ctx *c, const uint16_t *mult, uint64_t n_seqs) {
  MHX_TRY({
    if (n_seqs != c->seqs.n_seqs) throw mhx::Error("load_multiplicity: n_seqs differs from the loaded sequence set");
    c->seqs.mult.reserve((n_seqs + 1) * 2);
    c->seqs.mult.used = n_seqs * 2;
    if (n_seqs) MHX_HIP(hipMemcpyAsync(c->seqs.mult.p, mult, n_seqs * 2, hipMemcpyHostToDevice, c->stream));
    MHX_HIP(hipStreamSynchronize(c->stream));
  })
}
uint32_t mhx_fixed_length(const mhx_ctx *c) { return c ? c->seqs.fixed_len : 0; }
uint64_t mhx_num_sequences(const mhx_ctx *c) { return c->seqs.n_seqs; }
uint64_t mhx_num_bases(const mhx_ctx *c) { return c->seqs.n_bases; }

uint64_t mhx_buffer_bytes(const mhx_ctx *c, int which) {
  auto it = c->results.find(which);
  return it == c->results.end() ? 0 : it->second.used;
}
// device -> pageable host memory through the two pinned staging buffers of upload_pinned: the copy of chunk i + 1 off the device runs while
// four threads move chunk i from its staging buffer to the destination (whose pages they touch for the first time on the way).  A plain
// hipMemcpy into a fresh std::vector ran at ~1 GB/s: 2.4 GB of solid edges at 100 M reads were 3 s of `count`'s 4.7 s (round 6).
static void download_pinned(mhx_ctx *c, void *h_dst, const void *d_src, size_t bytes) {
  hipStream_t st = c->stream;
  constexpr size_t kChunk = 32u << 20;
  constexpr int kThreads = 4;
  if (!c->pinned[0]) {
    for (int i = 0; i < 2; ++i) {
      MHX_HIP(hipHostMalloc(&c->pinned[i], kChunk, hipHostMallocDefault));
      MHX_HIP(hipEventCreateWithFlags(&c->pinned_free[i], hipEventDisableTiming));
    }
  }
  const char *src = static_cast<const char *>(d_src);
  char *dst = static_cast<char *>(h_dst);
  const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
  auto issue = [&](size_t i) {
    const size_t off = i * kChunk, len = std::min(kChunk, bytes - off);
    MHX_HIP(hipMemcpyAsync(c->pinned[i & 1], src + off, len, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipEventRecord(c->pinned_free[i & 1], st));
  };
  issue(0);
  for (size_t i = 0; i < n_chunks; ++i) {
    MHX_HIP(hipEventSynchronize(c->pinned_free[i & 1]));
    if (i + 1 < n_chunks) issue(i + 1);
    const size_t off = i * kChunk, len = std::min(kChunk, bytes - off);
    const char *stage = static_cast<const char *>(c->pinned[i & 1]);
    std::thread th[kThreads - 1];
    const size_t part = (len / kThreads + 63) & ~(size_t)63;
    for (int t = 1; t < kThreads; ++t) {
      const size_t lo = std::min(len, part * t), hi = std::min(len, part * (t + 1));
      th[t - 1] = std::thread([=] { if (hi > lo) memcpy(dst + off + lo, stage + lo, hi - lo); });
    }
    memcpy(dst + off, stage, std::min(len, part));
    for (auto &t : th) t.join();
  }
  MHX_HIP(hipStreamSynchronize(st));
}
int mhx_fetch(mhx_ctx *c, int which, void *dst, uint64_t offset, uint64_t bytes) {
  MHX_TRY({
    auto it = c->results.find(which);
    if (it == c->results.end() || !it->second.p) throw mhx::Error("fetch: buffer not present");
    if (offset + bytes > it->second.used) throw mhx::Error("fetch: range exceeds buffer");
    if (bytes >= (16u << 20) && c->opt("fetch_pinned", 1)) {
      MHX_HIP(hipStreamSynchronize(c->stream));
      download_pinned(c, dst, (const char *)it->second.p + offset, bytes);
    } else if (bytes) {
      MHX_HIP(hipMemcpyAsync(dst, (const char *)it->second.p + offset, bytes, hipMemcpyDeviceToHost, c->stream));
      MHX_HIP(hipStreamSynchronize(c->stream));
    }
  })
}

int mhx_count(mhx_ctx *c, uint32_t k, uint32_t min_count, mhx_count_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::run_count(c, k, min_count, out);
  })
}
int mhx_read2sdbg_s1(mhx_ctx *c, uint32_t k, uint32_t min_count, int want_mercy, mhx_s1_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::run_s1(c, k, min_count, want_mercy, out);
  })
}
int mhx_s1_self_planned(mhx_ctx *c, uint32_t k, uint32_t min_count, int want_mercy) {
  if (!c) return 0;
  try {
    return mhx::s1_skm_applies(c, k, min_count, want_mercy) ? 1 : 0;
  } catch (...) {
    return 0;
  }
}
int mhx_count_self_planned(mhx_ctx *c, uint32_t k, uint32_t min_count) {
  if (!c) return 0;
  try {
    return mhx::count_skm_applies(c, k, min_count) ? 1 : 0;
  } catch (...) {
    return 0;
  }
}
int mhx_read2sdbg_add_mercy(mhx_ctx *c, uint32_t k, uint64_t *num_mercy) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::run_s1_mercy(c, k, num_mercy);
  })
}
int mhx_set_is_solid(mhx_ctx *c, const uint64_t *bits, uint64_t n_words) {
  MHX_TRY({
    uint64_t need = mhx::div_ceil(c->seqs.n_bases, 64);
    if (n_words < need) throw mhx::Error("set_is_solid: bitmap too short");
    c->agg_valid = false; c->solid_plain_k = 0;
    mhx::DevBuf &b = c->result(MHX_BUF_IS_SOLID, (need + 1) * 8);
    b.used = need * 8;
    if (need) MHX_HIP(hipMemcpyAsync(b.p, bits, need * 8, hipMemcpyHostToDevice, c->stream));
    MHX_HIP(hipStreamSynchronize(c->stream));
  })
}
int mhx_read2sdbg_s2(mhx_ctx *c, uint32_t k, uint32_t min_count, mhx_sdbg_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::run_s2(c, k, min_count, out);
  })
}
int mhx_seq2sdbg(mhx_ctx *c, uint32_t k, mhx_sdbg_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::run_seq2sdbg(c, k, out);
  })
}
int mhx_gen_mercy_edges(mhx_ctx *c, uint32_t k, const uint32_t *cand_packed, uint64_t cand_words, uint64_t n_cand,
                        const uint64_t *cand_start, uint64_t *n_mercy) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::run_gen_mercy(c, k, cand_packed, cand_words, n_cand, cand_start, n_mercy);
  })
}

int mhx_iterate(mhx_ctx *c, uint32_t k, uint32_t step, const uint32_t *contig_packed, uint64_t contig_words, uint64_t n_contigs,
                const uint64_t *contig_start, mhx_iterate_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    if (!out || (!contig_start && n_contigs)) throw mhx::Error("iterate: bad arguments");
    static const uint64_t zero = 0;
    mhx::iterate_edges(c, k, step, contig_packed, contig_words, n_contigs, n_contigs ? contig_start : &zero, out);
  })
}
int mhx_fastx_to_records(mhx_ctx *c, const char *text1, uint64_t n1, const char *text2, uint64_t n2, mhx_fastx_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    if (!out || !text1) throw mhx::Error("fastx_to_records: bad arguments");
    mhx::fastx_to_records(c, text1, n1, text2, n2, out);
  })
}
int mhx_sdbg_build_index(mhx_ctx *c, uint32_t k, mhx_sdbg_index_info *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::sdbg_build_index(c, k, out);
  })
}
int mhx_sdbg_remove_tips(mhx_ctx *c, const mhx_sdbg_index_info *info, int max_tip_len, uint64_t *n_removed) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    if (!info) throw mhx::Error("sdbg_remove_tips: no index info");
    mhx::sdbg_remove_tips(c, info, max_tip_len, n_removed);
  })
}
int mhx_sdbg_load_bytes(mhx_ctx *c, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *bucket_offset, const uint64_t *bucket_items,
                        const uint64_t *bucket_tips, const uint64_t *bucket_large) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    mhx::sdbg_load_bytes(c, bytes, n_bytes, bucket_offset, bucket_items, bucket_tips, bucket_large);
  })
}
int mhx_sort_records(mhx_ctx *c, uint32_t *host_items, uint64_t n, uint32_t key_words, uint32_t aux_words) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    const int w = (int)(key_words + aux_words), S = mhx::round_up2(w);
    if (key_words == 0 || S > 20) throw mhx::Error("sort_records: unsupported record width");
    if (n == 0) return 0;
    uint32_t *a = c->ws("items_a", n * S * 4 + 64).as<uint32_t>();
    uint32_t *b = c->ws("items_b", n * S * 4 + 64).as<uint32_t>();
    if (S == w) {
      MHX_HIP(hipMemcpyAsync(a, host_items, n * w * 4, hipMemcpyHostToDevice, c->stream));
    } else {
      MHX_HIP(hipMemsetAsync(a, 0, n * S * 4, c->stream));
      MHX_HIP(hipMemcpy2DAsync(a, S * 4, host_items, w * 4, w * 4, n, hipMemcpyHostToDevice, c->stream));
    }
    uint32_t *r = mhx::sort_whole_key(c, a, b, n, S, (int)key_words, mhx::make_passes((int)key_words, 0, (int)key_words * 32));
    if (S == w) MHX_HIP(hipMemcpyAsync(host_items, r, n * w * 4, hipMemcpyDeviceToHost, c->stream));
    else MHX_HIP(hipMemcpy2DAsync(host_items, w * 4, r, S * 4, w * 4, n, hipMemcpyDeviceToHost, c->stream));
    MHX_HIP(hipStreamSynchronize(c->stream));
  })
}

int mhx_set_partition(mhx_ctx *c, int my_part, int n_parts, const uint32_t *bucket_begin) {
  MHX_TRY({
    if (n_parts < 1 || n_parts > 256 || my_part < 0 || my_part >= n_parts) throw mhx::Error("set_partition: bad arguments");
    std::vector<uint32_t> pb(bucket_begin, bucket_begin + n_parts + 1);
    if (pb.front() != 0 || pb.back() != MHX_NUM_BUCKETS) throw mhx::Error("set_partition: bucket_begin must start at 0 and end at 65536");
    std::vector<uint8_t> lut(MHX_NUM_BUCKETS);
    for (int p = 0; p < n_parts; ++p) {
      if (pb[p] > pb[p + 1]) throw mhx::Error("set_partition: bucket_begin must be non-decreasing");
      for (uint32_t b = pb[p]; b < pb[p + 1]; ++b) lut[b] = (uint8_t)p;
    }
    c->my_part = my_part;
    c->n_parts = n_parts;
    c->part_begin = pb;
    mhx::DevBuf &d = c->ws("owner_lut", MHX_NUM_BUCKETS);
    MHX_HIP(hipMemcpyAsync(d.p, lut.data(), MHX_NUM_BUCKETS, hipMemcpyHostToDevice, c->stream));
    MHX_HIP(hipStreamSynchronize(c->stream));
  })
}
int mhx_set_global_layout(mhx_ctx *c, uint64_t pos_base, uint64_t global_bases) {
  MHX_TRY({
    if (global_bases && pos_base + c->seqs.n_bases > global_bases) throw mhx::Error("set_global_layout: local reads exceed the global set");
    c->pos_base = pos_base;
    c->global_bases = global_bases;
  })
}
int mhx_dist_extract(mhx_ctx *c, int stage, uint32_t k, uint32_t min_count, mhx_dist_items *out, uint64_t *counts) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    if (c->work.find("owner_lut") == c->work.end()) throw mhx::Error("dist_extract: call mhx_set_partition first");
    const mhx::StageItems it = mhx::extract_stage(c, stage, k, min_count);
    if (stage == MHX_STAGE_S2) c->dist_s2_agg = it.agg;
    c->pre_hist_buf = nullptr;  // the items are about to be partitioned and exchanged
    const uint64_t n = it.n;
    const int S = it.S;
    uint32_t *a = c->work["items_a"].as<uint32_t>();
    uint32_t *send = c->ws("items_send", n * (size_t)S * 4 + 64).as<uint32_t>();
    mhx::partition_by_owner(c, a, send, n, S, c->work["owner_lut"].as<uint8_t>(), c->n_parts, counts);
    out->d_items = send;
    out->n_items = n;
    out->item_bytes = (uint32_t)S * 4;
  })
}
void *mhx_dist_recv_buffer(mhx_ctx *c, uint64_t n_items, uint32_t item_bytes) {
  try {
    MHX_HIP(hipSetDevice(c->device));
    return c->ws("items_recv", n_items * (size_t)item_bytes + 64).p;
  } catch (const std::exception &e) {
    mhx::set_error("%s", e.what());
    return nullptr;
  }
}
int mhx_dist_process_s1(mhx_ctx *c, uint32_t k, uint32_t min_count, int want_mercy, uint64_t n_items, mhx_s1_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    const int S = mhx::s1_stride(k, mhx::s1_compact(c, k, want_mercy));
    uint32_t *a = c->ws("items_recv", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    uint32_t *b = c->ws("items_b", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    mhx::s1_process(c, k, min_count, want_mercy, a, b, n_items, out);
  })
}
int mhx_dist_process_count(mhx_ctx *c, uint32_t k, uint32_t min_count, uint64_t n_items, mhx_count_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    if (!c->global_bases) throw mhx::Error("dist_process_count: call mhx_set_global_layout first");
    const int S = mhx::count_stride(k);
    uint32_t *a = c->ws("items_recv", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    uint32_t *b = c->ws("items_b", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    mhx::count_process(c, k, min_count, a, b, n_items, out);
  })
}
int mhx_dist_process_seq2sdbg(mhx_ctx *c, uint32_t k, uint64_t n_items, mhx_sdbg_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    const int S = mhx::seq2sdbg_stride(k);
    uint32_t *a = c->ws("items_recv", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    uint32_t *b = c->ws("items_b", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    mhx::seq2sdbg_process(c, k, a, b, n_items, out);
  })
}
// first record whose position (value >> shift) is >= r * stride, r = 0..n_parts
__global__ void k_route_bounds(const unsigned long long *__restrict__ v, uint64_t n, int shift, uint64_t stride, int n_parts,
                               uint64_t *__restrict__ bounds) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_parts) return;
  const uint64_t key = (uint64_t)r * stride;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((v[mid] >> shift) < key) lo = mid + 1;
    else hi = mid;
  }
  bounds[r] = r == n_parts ? n : lo;
}
int mhx_dist_route_records(mhx_ctx *c, int which, uint64_t stride_bases, mhx_dist_items *out, uint64_t *counts) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    if (!stride_bases || c->n_parts < 1) throw mhx::Error("dist_route_records: bad stride or no partition");
    void *p = nullptr;
    uint64_t n = 0;
    int shift = 0;
    if (which == MHX_ROUTE_COUNT_EVENTS) {
      p = c->ws("route_records", 64).p;
      n = c->n_route;
      shift = 1;
    } else if (which == MHX_ROUTE_MERCY_CAND) {
      auto it = c->results.find(MHX_BUF_MERCY_CAND);
      if (it == c->results.end()) throw mhx::Error("dist_route_records: no mercy candidates");
      p = it->second.p;
      n = it->second.used / 8;
      shift = 2;
    } else if (which == MHX_ROUTE_S1_MARKS) {
      n = c->n_marks;
      int hi_bit = 1;
      while (hi_bit < 64 && (c->global_bases >> hi_bit)) ++hi_bit;
      // The marks only have to be grouped by the rank that holds the read, rank = position / stride.  mhx_dist_setup makes
      // the stride a multiple of 2^j with global_bases <= 2^(j+8): the 8 position bits from j upwards then order the ranks,
      // and ONE pass over them (a two-field digit when it straddles the 32-bit words of the little-endian uint64; no copy,
      // no word swaps) replaces the full sort of the positions (5 passes + 3 copies of ~10^8 marks per rank and step).
      const int j = hi_bit > 8 ? hi_bit - 8 : 0;
      if (c->opt("dist_marks_one_pass", 1) && n && j > 0 && j + 8 <= 64 && stride_bases % (1ull << j) == 0) {
        mhx::SortPass ps{0, 0, 0, 0};
        // key words = the two memory words of the record: word 0 (low half) is the "high" key word of the record sort
        if (j >= 32) ps = {j - 32, 8, 0, 0};
        else if (j + 8 <= 32) ps = {32 + j, 8, 0, 0};
        else ps = {32 + j, 32 - j, 0, j + 8 - 32};
        uint32_t *src = c->ws("s1_marks", n * 8 + 64).as<uint32_t>();
        uint32_t *tmp = c->ws("s1_marks_sorted", n * 8 + 64).as<uint32_t>();
        p = mhx::radix_sort(c, src, tmp, n, 2, 2, std::vector<mhx::SortPass>{ps});
      } else {
        p = const_cast<uint64_t *>(mhx::sort_u64(c, c->ws("s1_marks", 64).p, n, hi_bit));
      }
      shift = 0;
    } else throw mhx::Error("dist_route_records: unknown record kind");
    uint64_t *bounds = c->ws("route_bounds", (c->n_parts + 2) * 8).as<uint64_t>();
    hipLaunchKernelGGL(k_route_bounds, dim3((c->n_parts + 1 + 63) / 64), dim3(64), 0, c->stream, (const unsigned long long *)p, n, shift,
                       stride_bases, c->n_parts, bounds);
    std::vector<uint64_t> h(c->n_parts + 1);
    MHX_HIP(hipMemcpyAsync(h.data(), bounds, (c->n_parts + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    MHX_HIP(hipStreamSynchronize(c->stream));
    for (int r = 0; r < c->n_parts; ++r) counts[r] = h[r + 1] - h[r];
    out->d_items = p;
    out->n_items = n;
    out->item_bytes = 8;
  })
}
int mhx_dist_apply_routed(mhx_ctx *c, int which, uint64_t n_records) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    void *recv = c->ws("items_recv", n_records * 8 + 64).p;
    if (which == MHX_ROUTE_COUNT_EVENTS) mhx::count_apply_events(c, (const unsigned long long *)recv, n_records);
    else if (which == MHX_ROUTE_MERCY_CAND) mhx::mercy_adopt_routed(c, (const long long *)recv, n_records);
    else if (which == MHX_ROUTE_S1_MARKS) mhx::s1_apply_marks(c, (const unsigned long long *)recv, n_records);
    else throw mhx::Error("dist_apply_routed: unknown record kind");
  })
}
int mhx_dist_process_s2(mhx_ctx *c, uint32_t k, uint64_t n_items, mhx_sdbg_result *out) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    const int S = c->dist_s2_agg ? 2 : mhx::round_up2((int)mhx::div_ceil(k * 2 + 4, 32));
    uint32_t *a = c->ws("items_recv", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    uint32_t *b = c->ws("items_b", n_items * (size_t)S * 4 + 64).as<uint32_t>();
    if (c->dist_s2_agg) mhx::s2_agg_process(c, k, a, b, n_items, out);
    else mhx::s2_process(c, k, a, b, n_items, out);
  })
}
void *mhx_device_pointer(mhx_ctx *c, int which) {
  auto it = c->results.find(which);
  return it == c->results.end() ? nullptr : it->second.p;
}
int mhx_adopt_is_solid_slice(mhx_ctx *c, const void *d_words, uint64_t n_words) {
  MHX_TRY({
    const uint64_t need = mhx::div_ceil(c->seqs.n_bases, 64);
    if (n_words < need) throw mhx::Error("adopt_is_solid_slice: slice shorter than the local reads");
    mhx::DevBuf &b = c->result(mhx::MHX_BUF_IS_SOLID_LOCAL, (need + 1) * 8);
    b.used = need * 8;
    if (need) MHX_HIP(hipMemcpyAsync(b.p, d_words, need * 8, hipMemcpyDeviceToDevice, c->stream));
    if (c->global_marks_inverted) mhx::invert_local_marks(c, b.as<unsigned long long>(), need);
    MHX_HIP(hipStreamSynchronize(c->stream));
  })
}

void mhx_alloc_stats(double *malloc_s, double *free_s, uint64_t *bytes, uint64_t *calls) {
  if (malloc_s) *malloc_s = (double)mhx::g_alloc_ns.load() * 1e-9;
  if (free_s) *free_s = (double)mhx::g_free_ns.load() * 1e-9;
  if (bytes) *bytes = mhx::g_alloc_bytes.load();
  if (calls) *calls = mhx::g_alloc_calls.load();
}

uint64_t mhx_device_free_bytes(mhx_ctx *c) {
  size_t free_b = 0, total_b = 0;
  if (hipSetDevice(c->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  return (uint64_t)free_b;
}
uint64_t mhx_stage_pass_bytes(mhx_ctx *c, int stage, uint32_t k, uint32_t min_count, uint64_t n_items) {
  // what a pass over n_items (kept) items of `stage` holds on the device besides the stage's fixed state
  try {
    (void)min_count;
    if (stage == MHX_STAGE_S1 && c->seqs.n_seqs) {
      // (the question is about a filtered pass: asked with the filter fields set, restored whatever happens)
      struct Restore {
        mhx_ctx *c;
        bool on;
        uint64_t exp;
        ~Restore() {
          c->filter_on = on;
          c->filter_expected = exp;
        }
      } restore{c, c->filter_on, c->filter_expected};
      c->filter_on = true;
      c->filter_expected = n_items;
      const bool gen = mhx::s1_filter_in_gen_applies(c, k);
      // the generating first pass: two 12-byte record buffers, nothing staged, nothing split; its status words walk ALL item slots
      if (gen) return n_items * 24 + n_items / 2 + (c->seqs.n_bases + 4 * c->seqs.n_seqs) / 3;
      return n_items * (3 * (uint64_t)mhx::s1_stride(k, mhx::s1_compact(c, k, 0)) * 4 + 1);
    }
    if (stage == MHX_STAGE_COUNT && c->seqs.n_seqs) {
      struct Restore {
        mhx_ctx *c;
        bool on;
        uint64_t exp;
        ~Restore() {
          c->filter_on = on;
          c->filter_expected = exp;
        }
      } restore{c, c->filter_on, c->filter_expected};
      c->filter_on = true;
      c->filter_expected = n_items;
      // count on the stage-1 design: two 12-byte record buffers (the solid edges and the events live in the spare one)
      if (mhx::count_stream_applies(c, k, min_count) && (2 * (k + 1) + 16 + 31) / 32 <= 3)
        return n_items * 24 + n_items / 2 + c->seqs.n_bases / 3;
    }
    // stage 2 from a count of the (k+1)-mers (s2.hip s2_agg_from_count): two 12-byte record buffers per edge occurrence — about one per
    // base, where the caller's item estimate (per-occurrence items) is ~2.2 per base — and the few aggregated items behind them
    if (stage == MHX_STAGE_S2 && c->seqs.n_seqs && mhx::s2_agg_from_count_applies(c, k, min_count)) return n_items * 13;
    uint64_t ib = 16;
    if (stage == MHX_STAGE_S1_MERCY) ib = (uint64_t)mhx::s1_stride(k, false) * 4;
    else if (stage == MHX_STAGE_COUNT) ib = (uint64_t)mhx::count_stride(k) * 4;
    else if (stage == MHX_STAGE_SEQ2SDBG) ib = (uint64_t)mhx::seq2sdbg_stride(k) * 4;
    else if (stage == MHX_STAGE_S2) ib = (uint64_t)mhx::s2_stride(k) * 4;
    return n_items * (3 * ib + 1);  // 2 sort buffers + the filtered copy (+ status words)
  } catch (...) {
    return 0;
  }
}
int mhx_bucket_histogram(mhx_ctx *c, int stage, uint32_t k, uint32_t min_count, uint64_t *hist) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    const bool was = c->filter_on;
    c->filter_on = false;
    try {
      mhx::bucket_histogram(c, stage, k, min_count, hist);
    } catch (...) {
      c->filter_on = was;
      throw;
    }
    c->filter_on = was;
  })
}
int mhx_set_bucket_filter(mhx_ctx *c, const uint8_t *keep, uint64_t expected_items, uint64_t batch_bytes, int accumulate) {
  MHX_TRY({
    MHX_HIP(hipSetDevice(c->device));
    c->accumulate = accumulate != 0;
    c->filter_batch_bytes = batch_bytes;
    if (!keep) {
      c->filter_on = false;
      return 0;
    }
    std::vector<uint8_t> lut(MHX_NUM_BUCKETS);
    std::vector<uint32_t> bits(MHX_NUM_BUCKETS / 32, 0);  // the same as a bitmap (bit b of word b / 32): the generating pass of stage 1 reads it
    uint32_t kept = 0;
    for (int b = 0; b < MHX_NUM_BUCKETS; ++b) {
      lut[b] = keep[b] ? 0 : 1;  // owner 0 = keep, 1 = drop (partition_by_owner)
      if (keep[b]) {
        bits[b >> 5] |= 1u << (b & 31);
        ++kept;
      }
    }
    mhx::DevBuf &d = c->ws("filter_lut", MHX_NUM_BUCKETS);
    mhx::DevBuf &db = c->ws("filter_bits", MHX_NUM_BUCKETS / 8);
    MHX_HIP(hipMemcpyAsync(d.p, lut.data(), MHX_NUM_BUCKETS, hipMemcpyHostToDevice, c->stream));
    MHX_HIP(hipMemcpyAsync(db.p, bits.data(), MHX_NUM_BUCKETS / 8, hipMemcpyHostToDevice, c->stream));
    MHX_HIP(hipStreamSynchronize(c->stream));
    c->filter_on = true;
    c->filter_expected = expected_items;
    c->filter_kept = kept;
  })
}

int mhx_profile_enable(mhx_ctx *c, int on) {
  MHX_TRY({
    if (!on) c->prof_collect();
    c->profiling = on != 0;
  })
}
int mhx_profile_reset(mhx_ctx *c) {
  MHX_TRY({
    c->prof_collect();
    c->stats.clear();
  })
}
int mhx_profile_get(mhx_ctx *c, mhx_kernel_stat *out, int cap) {
  try {
    c->prof_collect();
    int i = 0;
    for (auto &kv : c->stats) {
      if (i < cap) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = kv.second.launches;
        out[i].total_ms = kv.second.ms;
        out[i].algo_bytes = kv.second.bytes;
      }
      ++i;
    }
    return i;
  } catch (const std::exception &e) {
    mhx::set_error("%s", e.what());
    return -1;
  }
}

}  // extern "C"
