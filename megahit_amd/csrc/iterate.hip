// SURVEY.md §8f N2 — `iterate`: the (k+step+1)-mer edges that reads support between the contigs of round k.
//
// Replaces the flank hash index + per-read scan + hash-set collection of the reference
// (src/iterate/contig_flank_index.h:16-219, src/iterate/kmer_collector.h:49-69, src/main_iterate.cpp:117-145) with the
// machinery of the SdBG-construction path:
//   flanks   per contig and strand the first (k+1)-mer + up to step-1 following bases (FeedBatchContigs :34-88) as
//            records (key = (k+1)-mer, aux = ~(ext_len, ext_seq)), radix-sorted: the best flank of a k-mer (longest,
//            then largest extension, :73-80) is the first of its run, a lookup is a lower_bound
//   reads    one thread per read walks the read exactly as FindNextKmersFromReads does (:90-188): flank hits in either
//            orientation and their matching extensions set `exist` bits; every position that ends a run of step+1 set
//            bits yields the canonical (k+step+1)-mer ending there
//   collect  the hash set of the reference = sort + unique of those k-mers (the tile-free way: head flags + scan)
// Multiplicities: the reference's flank records carry mul = 0 (FlankInfo{ext_seq, ext_len} leaves it value-initialised,
// :70), so every iterative edge is written with multiplicity 0; we write the same.
// Output order of the reference is its hash set's iteration order (unspecified); ours is sorted.
#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

constexpr uint64_t kIterPadWords = 64;  // window loads read up to KW + 1 words past a sequence's first word

// flank record: KW key words | aux hi | aux lo  (aux = ~((ext_len << 58) | ext_seq): ascending sort = best first)
template <int KW>
struct FlankRec {
  static constexpr int S = (KW + 2 + 1) & ~1;  // key words + 2 aux words, padded to an even stride for the record sort
};
template <int KW>
__global__ __launch_bounds__(256) void k_flanks(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint64_t n_seqs, int k, int step,
                                                uint32_t *__restrict__ out, unsigned long long *__restrict__ n_out) {
  constexpr int S = FlankRec<KW>::S;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i = t >> 1;
  const int strand = (int)(t & 1);
  if (i >= n_seqs) return;
  const uint64_t st = start[i];
  const uint64_t L = start[i + 1] - st;
  if (L < (uint64_t)k + 1) return;
  if (strand == 1 && L == (uint64_t)k + 1) return;  // `if (seq_len == k_ + 1) break;` after strand 0
  uint32_t f[KW], rc[KW];
  unsigned long long ext = 0;
  const unsigned ext_len = (unsigned)(L - (k + 1) < (uint64_t)(step - 1) ? L - (k + 1) : (uint64_t)(step - 1));
  if (strand == 0) {
    load_chars<KW>(seq, st, k + 1, f);
    for (unsigned j = 0; j < ext_len; ++j) ext |= (unsigned long long)base_at(seq, st + k + 1 + j) << (2 * j);
  } else {  // reverse complement of the contig's tail
    uint32_t tail[KW];
    load_chars<KW>(seq, st + L - (k + 1), k + 1, tail);
    rc_chars<KW>(tail, k + 1, f);
    for (unsigned j = 0; j < ext_len; ++j) ext |= (unsigned long long)(3u ^ base_at(seq, st + L - 1 - (k + 1 + j))) << (2 * j);
  }
  rc_chars<KW>(f, k + 1, rc);
  if (((k + 1) & 1) == 0 && cmp_words<KW>(f, rc) == 0) return;  // palindrome (Kmer::IsPalindrome: even length only)
  const unsigned long long aux = ~(((unsigned long long)ext_len << 58) | ext);
  uint32_t *o = out + atomicAdd(n_out, 1ull) * S;
#pragma unroll
  for (int w = 0; w < KW; ++w) o[w] = f[w];
  o[KW] = (uint32_t)(aux >> 32);
  o[KW + 1] = (uint32_t)aux;
  if constexpr (S > KW + 2) o[KW + 2] = 0;
}

// first flank record whose key is >= q; *found = its key equals q
template <int KW>
__device__ __forceinline__ uint64_t flank_lower_bound(const uint32_t *__restrict__ fl, uint64_t n, const uint32_t (&q)[KW], bool *found) {
  constexpr int S = FlankRec<KW>::S;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const uint32_t *r = fl + mid * S;
    int c = 0;
#pragma unroll
    for (int w = 0; w < KW; ++w) {
      if (c == 0 && r[w] != q[w]) c = r[w] < q[w] ? -1 : 1;
    }
    if (c < 0) lo = mid + 1;
    else hi = mid;
  }
  bool eq = lo < n;
  if (eq) {
    const uint32_t *r = fl + lo * S;
#pragma unroll
    for (int w = 0; w < KW; ++w) eq = eq && r[w] == q[w];
  }
  *found = eq;
  return lo;
}

// Prefix filter of the flank table: bit p is set iff some flank key starts with the 12 bases p (24 bits; the flank keys are
// (k+1)-mers with k + 1 >= 12).  A few 10^5 flanks set ~1 % of the 2^24 bits (2 MB: resident in the L2s), so 99 % of the
// 2 x (read positions) look-ups of k_iter_scan end after ONE cached load instead of a 17-step binary search through HBM
// (measured before: 130 ms for 10 M reads, 0.0014 of the HBM roofline).
constexpr int kIterPfxBits = 24;
template <int S>
__global__ void k_iter_prefix_bits(const uint32_t *__restrict__ fl, uint64_t n_fl, uint32_t *__restrict__ bits) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_fl) return;
  const uint32_t p = fl[i * S] >> (32 - kIterPfxBits);
  atomicOr(&bits[p >> 5], 1u << (p & 31));
}

// FindNextKmersFromReads, first half (:98-160): the exist bits of one read (bit j of exist[read * words_per_read ...])
template <int KW>
__global__ __launch_bounds__(256) void k_iter_scan(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint64_t n_reads, int k,
                                                   int step, const uint32_t *__restrict__ fl, uint64_t n_fl, unsigned long long *__restrict__ exist,
                                                   uint32_t words_per_read, uint32_t *__restrict__ n_emit, const uint32_t *__restrict__ pfx_bits) {
  auto maybe = [&](const uint32_t (&q)[KW]) -> bool {
    if (!pfx_bits) return true;
    const uint32_t p = q[0] >> (32 - kIterPfxBits);
    return (pfx_bits[p >> 5] >> (p & 31)) & 1u;
  };
  constexpr int S = FlankRec<KW>::S;
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t st = start[r];
  const uint32_t L = (uint32_t)(start[r + 1] - st);
  n_emit[r] = 0;
  if (L < (uint32_t)(k + step + 1)) return;
  unsigned long long *ex = exist + r * words_per_read;
  for (uint32_t w = 0; w < words_per_read; ++w) ex[w] = 0;
  auto get = [&](uint32_t j) -> bool { return (ex[j >> 6] >> (j & 63)) & 1ull; };
  auto set = [&](uint32_t j) { ex[j >> 6] |= 1ull << (j & 63); };
  uint32_t cur = 0;
  while (cur + k + 1 <= L) {
    uint32_t next = cur + 1;
    if (!get(cur)) {
      uint32_t f[KW], rc[KW];
      load_chars<KW>(seq, st + cur, k + 1, f);
      rc_chars<KW>(f, k + 1, rc);
      bool found = false;
      uint64_t at = 0;
      if (maybe(f)) at = flank_lower_bound<KW>(fl, n_fl, f, &found);
      if (found) {
        set(cur);
        const uint32_t *rec = fl + at * S;
        const unsigned long long aux = ~(((unsigned long long)rec[KW] << 32) | rec[KW + 1]);
        const unsigned ext_len = (unsigned)(aux >> 58);
        for (unsigned j = 0; j < ext_len && cur + k + 1 + j < L; ++j, ++next) {
          if (base_at(seq, st + cur + k + 1 + j) == (unsigned)((aux >> (2 * j)) & 3ull)) set(cur + j + 1);
          else break;
        }
      }
      found = false;
      if (maybe(rc)) at = flank_lower_bound<KW>(fl, n_fl, rc, &found);
      if (found) {
        set(cur);
        const uint32_t *rec = fl + at * S;
        const unsigned long long aux = ~(((unsigned long long)rec[KW] << 32) | rec[KW + 1]);
        const unsigned ext_len = (unsigned)(aux >> 58);
        for (unsigned j = 0; j < ext_len && cur >= j + 1; ++j) {
          if ((3u ^ base_at(seq, st + cur - 1 - j)) == (unsigned)((aux >> (2 * j)) & 3ull)) set(cur - 1 - j);
          else break;
        }
      }
    }
    if (next + k + 1 <= L) cur = next;
    else break;
  }
  // second half (:166-186): positions that end a run of step+1 exist bits
  uint32_t acc = 0, cnt = 0;
  for (uint32_t j = 0; j + k < L; ++j) {
    acc = get(j) ? acc + 1 : 0;
    if (acc >= (uint32_t)step + 1) ++cnt;
  }
  n_emit[r] = cnt;
}
// the canonical (k+step+1)-mer ending at every such position, reversed as the reference writes it -> records of NW words
template <int NW>
__global__ __launch_bounds__(256) void k_iter_emit(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, uint64_t n_reads, int k,
                                                   int step, const unsigned long long *__restrict__ exist, uint32_t words_per_read,
                                                   const uint64_t *__restrict__ out_off, uint32_t *__restrict__ out, int S) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t st = start[r];
  const uint32_t L = (uint32_t)(start[r + 1] - st);
  if (L < (uint32_t)(k + step + 1)) return;
  const unsigned long long *ex = exist + r * words_per_read;
  uint64_t o = out_off[r];
  if (o == out_off[r + 1]) return;
  uint32_t acc = 0;
  const int nk = k + step + 1;
  for (uint32_t j = 0; j + k < L; ++j) {
    acc = ((ex[j >> 6] >> (j & 63)) & 1ull) ? acc + 1 : 0;
    if (acc >= (uint32_t)step + 1) {
      uint32_t f[NW], rc[NW];
      load_chars<NW>(seq, st + j - step, nk, f);
      rc_chars<NW>(f, nk, rc);
      const bool fwd = cmp_words<NW>(f, rc) < 0;  // new_kmer < new_rkmer ? new_kmer : new_rkmer
      // KmerCollector::WriteToFile (kmer_collector.h:56-62) writes base k-1 first: the record holds the chosen k-mer
      // reversed = the complement of the other strand
      const int last_chars = nk - 16 * (NW - 1);
      const uint32_t last_mask = last_chars == 16 ? 0xffffffffu : ~(0xffffffffu >> (2 * last_chars));
      uint32_t *dst = out + o * S;
#pragma unroll
      for (int w = 0; w < NW; ++w) dst[w] = ~(fwd ? rc[w] : f[w]) & (w == NW - 1 ? last_mask : 0xffffffffu);
      for (int w = NW; w < S; ++w) dst[w] = 0;
      ++o;
    }
  }
}
// sorted records -> head flags (first of every run of equal keys)
__global__ void k_iter_heads(const uint32_t *__restrict__ rec, uint64_t n, int S, int kw, uint32_t *__restrict__ head) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool h = i == 0;
  if (!h) {
    for (int w = 0; w < kw; ++w) h = h || rec[i * S + w] != rec[(i - 1) * S + w];
  }
  head[i] = h ? 1u : 0u;
}
// unique k-mers -> edge records (words_per_edge words: chars MSB-first, multiplicity 0 in the low 16 bits of the last word)
__global__ void k_iter_edges(const uint32_t *__restrict__ rec, uint64_t n, int S, int kw, const uint32_t *__restrict__ head,
                             const uint64_t *__restrict__ pos, int wpe, uint32_t *__restrict__ edges) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  uint32_t *e = edges + pos[i] * wpe;
  for (int w = 0; w < wpe; ++w) e[w] = w < kw ? rec[i * S + w] : 0u;
}

int iterate_edges(mhx_ctx *c, uint32_t k, uint32_t step, const uint32_t *ctg_words, uint64_t ctg_n_words, uint64_t n_ctg, const uint64_t *ctg_start,
                  mhx_iterate_result *out) {
  hipStream_t st = c->stream;
  SeqSet &s = c->seqs;  // the reads (forward orientation)
  memset(out, 0, sizeof *out);
  if (step == 0 || step > 28 || (step & 1)) throw Error("iterate: invalid step");
  const int KWv = (int)div_ceil(k + 1, 16), NWv = (int)div_ceil(k + step + 1, 16);
  if (NWv > 17) throw Error("iterate: k + step too large");
  const int wpe = (int)div_ceil((k + step + 1) * 2 + 16, 32);
  // contigs -> device
  uint32_t *cw = c->ws("it_ctg_words", (ctg_n_words + kIterPadWords) * 4).as<uint32_t>();
  uint64_t *cs = c->ws("it_ctg_start", (n_ctg + 2) * 8).as<uint64_t>();
  MHX_HIP(hipMemsetAsync(cw, 0, (ctg_n_words + kIterPadWords) * 4, st));
  if (ctg_n_words) MHX_HIP(hipMemcpyAsync(cw, ctg_words, ctg_n_words * 4, hipMemcpyHostToDevice, st));
  MHX_HIP(hipMemcpyAsync(cs, ctg_start, (n_ctg + 1) * 8, hipMemcpyHostToDevice, st));
  // 1. flank records
  const int FS = round_up2(KWv + 2);
  uint32_t *fa = c->ws("it_flanks_a", (2 * n_ctg + 2) * (size_t)FS * 4 + 64).as<uint32_t>();
  uint32_t *fb = c->ws("it_flanks_b", (2 * n_ctg + 2) * (size_t)FS * 4 + 64).as<uint32_t>();
  unsigned long long *ctr = c->ws("it_counters", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  uint64_t n_fl = 0;
  if (n_ctg) {
    MHX_DISPATCH_KW(KWv, {
      MHX_LAUNCH(c, "iter_flanks", (double)n_ctg * 64,
                 hipLaunchKernelGGL((k_flanks<KW>), dim3((unsigned)div_ceil(2 * n_ctg, 256)), dim3(256), 0, st, cw, cs, n_ctg, (int)k, (int)step, fa, ctr));
    });
    MHX_HIP(hipMemcpyAsync(&n_fl, ctr, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  out->n_flanks = n_fl;
  uint32_t *fl = fa;
  if (n_fl > 1) fl = sort_whole_key(c, fa, fb, n_fl, FS, KWv + 2, make_passes(KWv + 2, 0, (KWv + 2) * 32));
  // 2. reads
  const uint64_t n_reads = s.n_seqs;
  const uint32_t wpr = (uint32_t)div_ceil(s.max_len ? s.max_len : 1, 64);
  unsigned long long *exist = c->ws("it_exist", (n_reads + 1) * (size_t)wpr * 8).as<unsigned long long>();
  uint32_t *n_emit = c->ws("it_n_emit", (n_reads + 1) * 4).as<uint32_t>();
  uint64_t *off = c->ws("it_off", (n_reads + 2) * 8).as<uint64_t>();
  uint64_t n_new = 0;
  if (n_reads) {
    uint32_t *pfx = nullptr;
    if (k + 1 >= 12 && c->opt("iterate_prefix_filter", 1)) {
      pfx = c->ws("it_prefix_bits", (size_t)(1u << kIterPfxBits) / 8).as<uint32_t>();
      MHX_HIP(hipMemsetAsync(pfx, 0, (size_t)(1u << kIterPfxBits) / 8, st));
      if (n_fl) {
        MHX_DISPATCH_KW(KWv, {
          hipLaunchKernelGGL((k_iter_prefix_bits<FlankRec<KW>::S>), dim3((unsigned)div_ceil(n_fl, 256)), dim3(256), 0, st, fl, n_fl, pfx);
        });
      }
    }
    MHX_DISPATCH_KW(KWv, {
      MHX_LAUNCH(c, "iter_scan", (double)s.n_bases,
                 hipLaunchKernelGGL((k_iter_scan<KW>), dim3((unsigned)div_ceil(n_reads, 256)), dim3(256), 0, st, s.words.as<uint32_t>(),
                                    s.start.as<uint64_t>(), n_reads, (int)k, (int)step, fl, n_fl, exist, wpr, n_emit, pfx));
    });
    exclusive_scan_u32_u64(c, n_emit, off, n_reads, off + n_reads);
    MHX_HIP(hipMemcpyAsync(&n_new, off + n_reads, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  out->n_kmers = n_new;
  // 3. emit, sort, unique
  const int NS = round_up2(NWv);
  uint32_t *ka = c->ws("it_kmers_a", (n_new + 1) * (size_t)NS * 4 + 64).as<uint32_t>();
  uint32_t *kb = c->ws("it_kmers_b", (n_new + 1) * (size_t)NS * 4 + 64).as<uint32_t>();
  uint64_t n_edges = 0;
  DevBuf &res = c->result(MHX_BUF_EDGES, 8);
  res.used = 0;
  if (n_new) {
    MHX_DISPATCH_KW(NWv, {
      MHX_LAUNCH(c, "iter_emit", (double)n_new * NS * 4,
                 hipLaunchKernelGGL((k_iter_emit<KW>), dim3((unsigned)div_ceil(n_reads, 256)), dim3(256), 0, st, s.words.as<uint32_t>(),
                                    s.start.as<uint64_t>(), n_reads, (int)k, (int)step, exist, wpr, off, ka, NS));
    });
    uint32_t *ks = sort_whole_key(c, ka, kb, n_new, NS, NWv, make_passes(NWv, NWv * 32 - 2 * (int)(k + step + 1), NWv * 32));
    uint32_t *head = c->ws("it_head", (n_new + 1) * 4).as<uint32_t>();
    uint64_t *pos = c->ws("it_pos", (n_new + 2) * 8).as<uint64_t>();
    hipLaunchKernelGGL(k_iter_heads, dim3((unsigned)div_ceil(n_new, 256)), dim3(256), 0, st, ks, n_new, NS, NWv, head);
    exclusive_scan_u32_u64(c, head, pos, n_new, pos + n_new);
    MHX_HIP(hipMemcpyAsync(&n_edges, pos + n_new, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    DevBuf &e = c->result(MHX_BUF_EDGES, (n_edges ? n_edges : 1) * (size_t)wpe * 4);
    e.used = n_edges * (size_t)wpe * 4;
    hipLaunchKernelGGL(k_iter_edges, dim3((unsigned)div_ceil(n_new, 256)), dim3(256), 0, st, ks, n_new, NS, NWv, head, pos, wpe, e.as<uint32_t>());
    MHX_HIP(hipGetLastError());
    MHX_HIP(hipStreamSynchronize(st));
  }
  out->n_edges = n_edges;
  out->words_per_edge = (uint32_t)wpe;
  // reads that yielded at least one k-mer ("aligned", main_iterate.cpp:131)
  return 0;
}

}  // namespace mhx
