// Mercy-edge paths (tip-rescue heuristics), both on the GPU:
//   run_s1_mercy   mercy block of Read2SdbgS2::Initialize (reference src/sorting/read_to_sdbg_s2.cpp:122-266)
//   run_gen_mercy  SeqToSdbg::GenMercyEdges (reference src/sorting/seq_to_sdbg.cpp:100-357)
#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

// ---------------------------------------------------------------------------
// read2sdbg: candidates -> extra is_solid bits.  One thread per candidate; the thread holding the
// first candidate of a read walks that read (candidates are sorted, so a read's candidates are
// contiguous and ascending by offset, which lets the position loop merge-walk them without the
// reference's per-read no_in/no_out/has_solid_kmer vectors).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_s1_mercy(const long long *__restrict__ cands, uint64_t n, const uint64_t *__restrict__ start,
                                                  uint64_t n_seqs, uint32_t fixed_len, int k, int max_len,
                                                  unsigned long long *__restrict__ is_solid, unsigned long long *__restrict__ num_mercy) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t abs0 = (uint64_t)cands[i] >> 2;
  const uint64_t rid = seq_of_offset(start, n_seqs, fixed_len, abs0);
  const uint64_t base = start[rid], end = start[rid + 1];
  if (i > 0 && ((uint64_t)cands[i - 1] >> 2) >= base) return;  // not the first candidate of its read
  const uint32_t L = (uint32_t)(end - base);
  int first_0_out = max_len + 1, last_0_in = -1;  // :196-197
  for (uint64_t j = i; j < n && ((uint64_t)cands[j] >> 2) < end; ++j) {
    const int off = (int)(((uint64_t)cands[j] >> 2) - base);
    const int flag = (int)(cands[j] & 3);
    if (flag == 2) first_0_out = min(first_0_out, off);
    else if (flag == 1) last_0_in = max(last_0_in, off);
  }
  if (last_0_in < first_0_out) return;  // :222-224
  int last_no_out = -1;
  uint64_t cj = i;
  unsigned long long added = 0;
  for (uint32_t p = 0; p + k <= L; ++p) {
    bool no_in = false, no_out = false, has_solid = false;
    while (cj < n && ((uint64_t)cands[cj] >> 2) < end && ((uint64_t)cands[cj] >> 2) - base == p) {
      const int flag = (int)(cands[cj] & 3);
      no_out |= flag == 2;
      no_in |= flag == 1;
      has_solid = true;
      ++cj;
    }
    // has_solid_kmer[p] also holds when the (k+1)-mer at p or p-1 is solid (:229-233); those two
    // bits cannot have been touched by this read's own mercy fill yet (it only writes below p-1)
    if (p + k < L && ((is_solid[(base + p) >> 6] >> ((base + p) & 63)) & 1ull)) has_solid = true;
    if (p >= 1 && ((is_solid[(base + p - 1) >> 6] >> ((base + p - 1) & 63)) & 1ull)) has_solid = true;
    if (no_in && last_no_out != -1) {
      for (uint32_t x = (uint32_t)last_no_out; x < p; ++x) atomicOr(&is_solid[(base + x) >> 6], 1ull << ((base + x) & 63));
      added += p - last_no_out;
    }
    if (has_solid) last_no_out = -1;
    if (no_out) last_no_out = (int)p;
  }
  if (added) atomicAdd(num_mercy, added);
}

__global__ void k_rebase_cands(long long *__restrict__ v, uint64_t n, long long delta) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] -= delta;
}

// multi-GPU: candidates routed to this rank (global positions, one sorted run per source rank) -> sorted, local positions
void mercy_adopt_routed(mhx_ctx *c, const long long *recv, uint64_t n) {
  hipStream_t st = c->stream;
  DevBuf &res = c->result(MHX_BUF_MERCY_CAND_LOCAL, n * 8 + 8);
  res.used = n * 8;
  if (n) {
    int hi_bit = 3;
    while (hi_bit < 64 && ((c->global_bases << 2) >> hi_bit)) ++hi_bit;
    const uint64_t *sorted = sort_u64(c, recv, n, hi_bit);
    MHX_HIP(hipMemcpyAsync(res.p, sorted, n * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_rebase_cands, dim3((unsigned)div_ceil(n, 256)), dim3(256), 0, st, res.as<long long>(), n, (long long)(c->pos_base << 2));
  }
  MHX_HIP(hipStreamSynchronize(st));
}

int run_s1_mercy(mhx_ctx *c, uint32_t k, uint64_t *num_mercy) {
  SeqSet &s = c->seqs;
  c->agg_valid = false; c->solid_plain_k = 0;  // mercy turns non-solid occurrences solid: stage 2 must look at every occurrence again
  hipStream_t st = c->stream;
  // multi-GPU: the local slice of the bitmap and the candidates routed to this rank
  auto itc = c->results.find(c->global_bases ? MHX_BUF_MERCY_CAND_LOCAL : MHX_BUF_MERCY_CAND);
  auto its = c->results.find(c->global_bases ? MHX_BUF_IS_SOLID_LOCAL : MHX_BUF_IS_SOLID);
  if (its == c->results.end() || its->second.used < div_ceil(s.n_bases, 64) * 8) throw Error("add_mercy: no is_solid bitmap");
  if (itc == c->results.end()) throw Error("add_mercy: no mercy candidates (run mhx_read2sdbg_s1 with want_mercy)");
  const uint64_t n = itc->second.used / 8;
  unsigned long long *ctr = c->ws("s1_counters", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  if (n) {
    MHX_LAUNCH(c, "s1_mercy", (double)n * 16,
               hipLaunchKernelGGL(k_s1_mercy, dim3((unsigned)div_ceil(n, 256)), dim3(256), 0, st, itc->second.as<long long>(), n,
                                  s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, (int)k, (int)s.max_len,
                                  its->second.as<unsigned long long>(), ctr));
  }
  unsigned long long h = 0;
  MHX_HIP(hipMemcpyAsync(&h, ctr, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (num_mercy) *num_mercy = h;
  return 0;
}

// ---------------------------------------------------------------------------
// seq2sdbg mercy: for every k-mer of every candidate read decide has_in / has_out by binary search
// in the sorted (k+1)-mer edge list, then add the (k+1)-mers between a "no out" and the next "no in".
// The reference prunes some probes by comparing against the reverse complement (:233-247,:270-297);
// on a sorted list of canonical edges the pruned probes cannot hit, so the un-pruned statement
//   has_in[i]  <=> an edge X.kmer_i exists (as itself or as its reverse complement)
//   has_out[i] <=> an edge kmer_i.Y exists (idem)
// gives identical results (proved against the reference in tests/test_oracle_vs_ref.py).
// ---------------------------------------------------------------------------
template <int KW>
__device__ __forceinline__ bool find_edge(const uint32_t *__restrict__ eseq, uint64_t n_edges, int edge_len, const uint32_t (&q)[KW],
                                          int n) {
  int64_t l = 0, r = (int64_t)n_edges - 1;
  while (l <= r) {
    const int64_t mid = (l + r) >> 1;
    uint32_t e[KW];
    load_chars<KW>(eseq, (uint64_t)mid * edge_len, n, e);
    const int cmp = cmp_words<KW>(q, e);
    if (cmp > 0) l = mid + 1;
    else if (cmp < 0) r = mid - 1;
    else return true;
  }
  return false;
}

// q (k chars) -> c.q (k+1 chars)
template <int KW>
__device__ __forceinline__ void prepend_char(const uint32_t (&in)[KW], unsigned ch, uint32_t (&out)[KW]) {
#pragma unroll
  for (int i = KW - 1; i > 0; --i) out[i] = (in[i] >> 2) | (in[i - 1] << 30);
  out[0] = (in[0] >> 2) | (ch << 30);
}

template <int KW>
__global__ __launch_bounds__(256) void k_mercy_flags(const uint32_t *__restrict__ eseq, uint64_t n_edges, const uint32_t *__restrict__ cseq,
                                                     const uint64_t *__restrict__ cstart, uint64_t n_cand, int k,
                                                     uint8_t *__restrict__ flags) {
  // one thread per base position of the candidate reads
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = cstart[n_cand];
  if (b >= total) return;
  const uint64_t rid = seq_of_offset(cstart, n_cand, 0, b);
  const uint32_t L = (uint32_t)(cstart[rid + 1] - cstart[rid]);
  const uint32_t i = (uint32_t)(b - cstart[rid]);
  uint8_t f = 0;
  if (L >= (uint32_t)k + 2 && i + k <= L) {
    uint32_t q[KW], rq[KW], t[KW];
    load_chars<KW>(cseq, b, k, q);
    rc_chars<KW>(q, k, rq);
    bool has_in = find_edge<KW>(eseq, n_edges, k + 1, rq, k);
    for (unsigned ch = 0; ch < 4 && !has_in; ++ch) {
      prepend_char<KW>(q, ch, t);
      has_in = find_edge<KW>(eseq, n_edges, k + 1, t, k + 1);
    }
    bool has_out = find_edge<KW>(eseq, n_edges, k + 1, q, k);
    for (unsigned ch = 0; ch < 4 && !has_out; ++ch) {
      prepend_char<KW>(rq, ch, t);
      has_out = find_edge<KW>(eseq, n_edges, k + 1, t, k + 1);
    }
    f = (has_in ? 1 : 0) | (has_out ? 2 : 0);
  }
  flags[b] = f;
}

// per candidate read: the state machine of seq_to_sdbg.cpp:310-345.  WRITE=false counts mercy edges,
// WRITE=true stores their absolute start offsets (in the candidate store).
template <bool WRITE>
__global__ void k_mercy_scan(const uint8_t *__restrict__ flags, const uint64_t *__restrict__ cstart, uint64_t n_cand, int k,
                             uint32_t *__restrict__ cnt, const uint64_t *__restrict__ pos, uint64_t *__restrict__ mercy_abs) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_cand) return;
  const uint64_t st = cstart[r];
  const uint32_t L = (uint32_t)(cstart[r + 1] - st);
  uint32_t n = 0;
  uint64_t o = WRITE ? pos[r] : 0;
  if (L >= (uint32_t)k + 2) {
    int last_no_out = -1;
    for (uint32_t i = 0; i + k <= L; ++i) {
      const int state = flags[st + i];
      if (state == 1) last_no_out = (int)i;
      else if (state == 2) {
        if (last_no_out >= 0) {
          if (WRITE)
            for (uint32_t j = (uint32_t)last_no_out; j < i; ++j) mercy_abs[o++] = st + j;
          n += i - last_no_out;
        }
        last_no_out = -1;
      } else if (state == 3) last_no_out = -1;
    }
  }
  if (!WRITE) cnt[r] = n;
}

// append n_mercy (k+1)-mers (taken from the candidate store at mercy_abs[]) after the n_edges edges
// of the fixed-length edge store; one thread per output word
__global__ void k_mercy_append(uint32_t *__restrict__ eseq, uint64_t old_bases, uint64_t new_bases, int edge_len,
                               const uint32_t *__restrict__ cseq, const uint64_t *__restrict__ mercy_abs) {
  const uint64_t w = old_bases / 16 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 16 >= new_bases) return;
  uint32_t word = 0;
  for (int j = 0; j < 16; ++j) {
    const uint64_t b = w * 16 + j;
    unsigned ch = 0;
    if (b < old_bases) ch = (eseq[w] >> (30 - 2 * j)) & 3u;
    else if (b < new_bases) {
      const uint64_t mi = (b - old_bases) / edge_len, off = (b - old_bases) % edge_len;
      ch = base_at(cseq, mercy_abs[mi] + off);
    }
    word |= ch << (30 - 2 * j);
  }
  eseq[w] = word;
}

__global__ void k_fill_u16(uint16_t *p, uint64_t from, uint64_t to, uint16_t v) {
  uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < to) p[i] = v;
}

static void grow_preserve(mhx_ctx *c, DevBuf &b, size_t new_bytes, size_t keep_bytes) {
  if (b.cap >= new_bytes) return;
  DevBuf nb;
  nb.reserve(new_bytes);
  if (keep_bytes && b.p) MHX_HIP(hipMemcpyAsync(nb.p, b.p, keep_bytes, hipMemcpyDeviceToDevice, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  nb.used = b.used;
  b.release();
  b = nb;
}

// Multi-GPU (share != nullptr): the sorted edges are sharded over the ranks, every rank holds ALL candidate reads.  A rank's
// binary searches only see its shard, so has_in / has_out are OR-ed over the ranks (share->reduce_flags) before the per-read
// state machine runs — identically on every rank — and the mercy edges it yields are dealt out round robin: rank r appends
// edge i iff i % n_parts == my_part.  *n_mercy = the number of ALL mercy edges (the reference's "Number of mercy edges").
int run_gen_mercy(mhx_ctx *c, uint32_t k, const uint32_t *cand_packed, uint64_t cand_words, uint64_t n_cand, const uint64_t *cand_start,
                  uint64_t *n_mercy, const MercyShare *share) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  if (n_mercy) *n_mercy = 0;
  if (s.n_seqs && s.fixed_len != k + 1) throw Error("gen_mercy_edges: the loaded sequences must be (k+1)-mer edges");
  if (s.n_seqs && s.mult.used < s.n_seqs * 2) throw Error("gen_mercy_edges: multiplicities not loaded");
  if (n_cand == 0) return 0;
  const uint64_t cand_bases = cand_start[n_cand];
  uint32_t *cseq = c->ws("cand_words", (cand_words + 64) * 4).as<uint32_t>();
  uint64_t *cstart = c->ws("cand_start", (n_cand + 2) * 8).as<uint64_t>();
  MHX_HIP(hipMemcpyAsync(cseq, cand_packed, cand_words * 4, hipMemcpyHostToDevice, st));
  MHX_HIP(hipMemsetAsync(cseq + cand_words, 0, 64 * 4, st));
  MHX_HIP(hipMemcpyAsync(cstart, cand_start, (n_cand + 1) * 8, hipMemcpyHostToDevice, st));
  uint8_t *flags = c->ws("cand_flags", cand_bases + 16).as<uint8_t>();
  const int KWv = (int)div_ceil(k + 1, 16);
  if (cand_bases) {
    MHX_DISPATCH_KW(KWv, {
      MHX_LAUNCH(c, "mercy_flags", (double)cand_bases * 8,
                 hipLaunchKernelGGL((k_mercy_flags<KW>), dim3((unsigned)div_ceil(cand_bases, 256)), dim3(256), 0, st,
                                    s.words.as<uint32_t>(), s.n_seqs, cseq, cstart, n_cand, (int)k, flags));
    });
  }
  if (share && share->reduce_flags && cand_bases) share->reduce_flags(flags, cand_bases);
  uint32_t *cnt = c->ws("cand_cnt", (n_cand + 1) * 4).as<uint32_t>();
  uint64_t *pos = c->ws("cand_pos", (n_cand + 2) * 8).as<uint64_t>();
  const unsigned g = (unsigned)div_ceil(n_cand, 256);
  MHX_LAUNCH(c, "mercy_scan", (double)cand_bases,
             hipLaunchKernelGGL(k_mercy_scan<false>, dim3(g), dim3(256), 0, st, flags, cstart, n_cand, (int)k, cnt, nullptr, nullptr));
  exclusive_scan_u32_u64(c, cnt, pos, n_cand, pos + n_cand + 1);
  uint64_t nm = 0;
  MHX_HIP(hipMemcpyAsync(&nm, pos + n_cand + 1, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  const uint64_t nm_all = nm;
  if (nm) {
    uint64_t *mercy_abs = c->ws("mercy_abs", nm * 8).as<uint64_t>();
    MHX_LAUNCH(c, "mercy_scan", (double)cand_bases + (double)nm * 8,
               hipLaunchKernelGGL(k_mercy_scan<true>, dim3(g), dim3(256), 0, st, flags, cstart, n_cand, (int)k, nullptr, pos, mercy_abs));
    if (share && share->n_parts > 1) {  // this rank's share of the list (a few 10^5 entries: through the host)
      std::vector<uint64_t> all(nm), mine;
      MHX_HIP(hipMemcpyAsync(all.data(), mercy_abs, nm * 8, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      for (uint64_t i = (uint64_t)share->my_part; i < nm; i += (uint64_t)share->n_parts) mine.push_back(all[i]);
      nm = mine.size();
      if (nm) MHX_HIP(hipMemcpyAsync(mercy_abs, mine.data(), nm * 8, hipMemcpyHostToDevice, st));
      MHX_HIP(hipStreamSynchronize(st));
    }
  }
  if (nm) {
    uint64_t *mercy_abs = c->ws("mercy_abs", nm * 8).as<uint64_t>();
    if (!s.n_seqs) {  // a rank without edges of its own: an empty fixed-length (k+1)-mer set to append to
      s.fixed_len = k + 1;
      s.n_bases = s.n_words = 0;
      s.words.reserve(64 * 4);
      MHX_HIP(hipMemsetAsync(s.words.p, 0, 64 * 4, st));
    }
    const uint64_t old_bases = s.n_bases, new_seqs = s.n_seqs + nm, new_bases = new_seqs * (k + 1);
    const uint64_t new_words = div_ceil(new_bases, 16);
    grow_preserve(c, s.words, (new_words + 64) * 4, (s.n_words + 1) * 4);
    grow_preserve(c, s.mult, (new_seqs + 1) * 2, s.n_seqs * 2);
    // zero the words after the old data, then gather
    const uint64_t first_w = old_bases / 16;
    if (old_bases % 16 == 0) MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + first_w, 0, (new_words + 64 - first_w) * 4, st));
    else MHX_HIP(hipMemsetAsync(s.words.as<uint32_t>() + first_w + 1, 0, (new_words + 64 - first_w - 1) * 4, st));
    const uint64_t n_w = new_words - first_w;
    MHX_LAUNCH(c, "mercy_append", (double)n_w * 4 + (double)nm * (k + 1),
               hipLaunchKernelGGL(k_mercy_append, dim3((unsigned)div_ceil(n_w, 256)), dim3(256), 0, st, s.words.as<uint32_t>(), old_bases,
                                  new_bases, (int)(k + 1), cseq, mercy_abs));
    hipLaunchKernelGGL(k_fill_u16, dim3((unsigned)div_ceil(nm, 256)), dim3(256), 0, st, s.mult.as<uint16_t>(), s.n_seqs, new_seqs,
                       (uint16_t)1);  // multiplicity 1, seq_to_sdbg.cpp:353
    // rebuild the start array for the longer fixed-length set
    grow_preserve(c, s.start, (new_seqs + 2) * 8, 0);
    s.n_seqs = new_seqs;
    s.n_bases = new_bases;
    s.n_words = new_words;
    s.mult.used = new_seqs * 2;
    MHX_HIP(hipStreamSynchronize(st));
    upload_fixed_starts(c);
  }
  if (n_mercy) *n_mercy = nm_all;
  return 0;
}

}  // namespace mhx
