// SURVEY.md §8f N3 — buildlib on the GPU: FASTA / FASTQ text -> the read library's record stream (.bin).
//
// Replaces the single-threaded parse + pack of SequenceLibCollection::Build (reference
// src/sequence/io/sequence_lib.cpp:8-91 over FastxReader / kseq: src/sequence/io/fastx_reader.cpp:28-71,
// src/sequence/io/kseq.h:193-247) for the formats sequencers actually write:
//   FASTA   header lines start with '>', every other non-empty line is sequence (any number of lines per record)
//   FASTQ   four lines per record: '@' header, sequence, '+' line, quality of the same length
// The whole file text sits in HBM; line starts come from a scan over the newline flags, records from a scan over the
// header flags, then per record: TrimN (the first N-free stretch, fastx_reader.cpp:56-71), the all-trimmed read faked as
// one 'A' (sequence_package.h:262-267), chars mapped ACGT/acgt -> 0..3, N/n -> 2, anything else -> 0 (:78-83,314) and
// packed 2 bits per base, MSB first, behind the length word (:224-240).  Paired files are parsed separately and their
// records interleaved (paired_fastx_reader.cpp:7-43).
// Anything else (carriage returns, '+'/'@'-led lines inside FASTA, FASTQ that is not four lines per record, junk before
// the first header) is reported back as status 1: the caller then runs its sequential kseq-compatible parser.
#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

constexpr int kTextChunk = 256;  // bytes per thread in the newline passes

__global__ void k_count_newlines(const char *__restrict__ t, uint64_t n, uint32_t *__restrict__ cnt, uint32_t *__restrict__ has_cr) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = c * kTextChunk;
  if (lo >= n) return;
  const uint64_t hi = lo + kTextChunk < n ? lo + kTextChunk : n;
  uint32_t k = 0, cr = 0;
  for (uint64_t i = lo; i < hi; ++i) {
    k += t[i] == '\n';
    cr |= t[i] == '\r';
  }
  cnt[c] = k;
  if (cr) atomicOr(has_cr, 1u);
}
// line_start[j] = first byte of line j (line 0 starts at 0; a line ends before its '\n' or at n)
__global__ void k_line_starts(const char *__restrict__ t, uint64_t n, const uint64_t *__restrict__ chunk_off, uint64_t *__restrict__ line_start) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = c * kTextChunk;
  if (lo >= n) return;
  const uint64_t hi = lo + kTextChunk < n ? lo + kTextChunk : n;
  uint64_t j = chunk_off[c] + 1;  // newlines before this chunk = index of the line that starts after the next newline, minus 1
  if (c == 0) line_start[0] = 0;
  for (uint64_t i = lo; i < hi; ++i)
    if (t[i] == '\n') line_start[j++] = i + 1;
}

// per line: FASTA -> header flag and number of sequence bytes; anomalies -> *bad
__global__ void k_fasta_lines(const char *__restrict__ t, uint64_t n, const uint64_t *__restrict__ ls, uint64_t n_lines, uint32_t *__restrict__ is_hdr,
                              uint32_t *__restrict__ seq_len, uint32_t *__restrict__ bad) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_lines) return;
  const uint64_t b = ls[j], e = j + 1 < n_lines ? ls[j + 1] - 1 : (n && t[n - 1] == '\n' ? n - 1 : n);
  const uint64_t len = e > b ? e - b : 0;
  const char c0 = len ? t[b] : 0;
  if (c0 == '+' || c0 == '@') atomicOr(bad, 1u);  // kseq would switch to FASTQ mode / start a record here
  is_hdr[j] = c0 == '>';
  seq_len[j] = c0 == '>' ? 0u : (uint32_t)len;
  if (len >> 31) atomicOr(bad, 2u);
}
// FASTA: sequence bytes of every line -> dense buffer; record r = header line hdr_line[r], bytes [rec_off[r], rec_off[r+1])
__global__ void k_fasta_compact(const char *__restrict__ t, const uint64_t *__restrict__ ls, uint64_t n_lines, const uint32_t *__restrict__ is_hdr,
                                const uint32_t *__restrict__ seq_len, const uint64_t *__restrict__ dense_off, const uint64_t *__restrict__ rec_of_line,
                                char *__restrict__ dense, uint64_t *__restrict__ rec_off) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = lane_id();
  if (wave >= n_lines) return;
  if (is_hdr[wave]) {
    if (lane == 0) rec_off[rec_of_line[wave]] = dense_off[wave];  // exclusive header count = this record's index
    return;
  }
  const uint64_t b = ls[wave], d = dense_off[wave];
  const uint32_t len = seq_len[wave];
  for (uint32_t i = lane; i < len; i += kWave) dense[d + i] = t[b + i];
}
// FASTQ (four lines per record): checks + (start, length) of every sequence
__global__ void k_fastq_records(const char *__restrict__ t, uint64_t n, const uint64_t *__restrict__ ls, uint64_t n_lines, uint64_t n_rec,
                                uint64_t *__restrict__ seq_start, uint32_t *__restrict__ seq_len, uint32_t *__restrict__ bad) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  auto line_end = [&](uint64_t j) -> uint64_t { return j + 1 < n_lines ? ls[j + 1] - 1 : (n && t[n - 1] == '\n' ? n - 1 : n); };
  const uint64_t h = 4 * r;
  const uint64_t b0 = ls[h], b1 = ls[h + 1], b2 = ls[h + 2], b3 = ls[h + 3];
  const uint64_t l0 = line_end(h) - b0, l1 = line_end(h + 1) - b1, l2 = line_end(h + 2) - b2, l3 = line_end(h + 3) - b3;
  // kseq: a sequence line led by '>', '+' or '@' would end the sequence; an empty sequence line is skipped (the '+' line
  // would then be taken as sequence...): anything but the plain shape goes to the sequential parser
  if (l0 == 0 || t[b0] != '@' || l2 == 0 || t[b2] != '+' || l1 != l3 || l1 == 0 || (l1 >> 31)) {
    atomicOr(bad, 1u);
    seq_start[r] = 0;  // the kernels that follow are launched regardless of *bad: give them an empty, in-bounds record
    seq_len[r] = 0;    // (the workspace is never zeroed: stale values here would be read as an address and a length)
    return;
  }
  const char c1 = t[b1];
  if (c1 == '>' || c1 == '+' || c1 == '@') atomicOr(bad, 1u);
  seq_start[r] = b1;
  seq_len[r] = (uint32_t)l1;
}
// FastxReader::TrimN (fastx_reader.cpp:56-71): [b, e) = the first maximal stretch without N/n; -> record words
__global__ void k_trim_n(const char *__restrict__ s, const uint64_t *__restrict__ seq_start, const uint32_t *__restrict__ seq_len_in,
                         const uint64_t *__restrict__ rec_off, uint64_t n_rec, uint32_t *__restrict__ tb, uint32_t *__restrict__ tl,
                         uint32_t *__restrict__ words, unsigned long long *__restrict__ totals /* bases, max_len */) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  const uint64_t st = seq_start ? seq_start[r] : rec_off[r];
  const uint32_t len = seq_start ? seq_len_in[r] : (uint32_t)(rec_off[r + 1] - rec_off[r]);
  uint32_t b = len, e = len, i = 0;
  for (; i < len; ++i) {
    const char c = s[st + i];
    if (c == 'N' || c == 'n') {
      if (b < len) break;
    } else if (b == len) {
      b = i;
    }
  }
  e = i;
  uint32_t L = e - b;
  tb[r] = b;
  tl[r] = L;  // 0: the read is faked as one 'A'
  if (L == 0) L = 1;
  words[r] = 1 + (L + 15) / 16;
  atomicAdd(&totals[0], (unsigned long long)L);
  atomicMax(&totals[1], (unsigned long long)L);
}
__device__ __forceinline__ unsigned dna_code(char c) {  // sequence_package.h:78-83,314
  switch (c) {
    case 'C': case 'c': return 1u;
    case 'G': case 'g': case 'N': case 'n': return 2u;
    case 'T': case 't': return 3u;
    default: return 0u;
  }
}
// one record per thread: length word + packed bases (forward orientation, MSB first); interleave: out record = r * step + phase
__global__ void k_pack_records(const char *__restrict__ s, const uint64_t *__restrict__ seq_start, const uint64_t *__restrict__ rec_off,
                               const uint32_t *__restrict__ tb, const uint32_t *__restrict__ tl, const uint64_t *__restrict__ out_off, uint64_t n_rec,
                               uint64_t step, uint64_t phase, uint32_t *__restrict__ out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  const uint64_t st = (seq_start ? seq_start[r] : rec_off[r]) + tb[r];
  const uint32_t L = tl[r];
  uint32_t *o = out + out_off[r * step + phase];
  if (L == 0) {
    o[0] = 1;
    o[1] = 0;  // 'A'
    return;
  }
  o[0] = L;
  for (uint32_t w = 0; w * 16 < L; ++w) {
    uint32_t v = 0;
    const uint32_t m = L - w * 16 < 16 ? L - w * 16 : 16;
    for (uint32_t j = 0; j < m; ++j) v |= dna_code(s[st + w * 16 + j]) << (30 - 2 * j);
    o[1 + w] = v;
  }
}
// record word counts of the two mates interleaved: w[2 i + p] = words_p[i]
__global__ void k_interleave_u32(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n, uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out[2 * i] = a[i];
    out[2 * i + 1] = b[i];
  }
}

// one parsed file, device resident
struct ParsedFastx {
  const char *seq = nullptr;        // text the sequences live in (the file text for FASTQ, the compacted bytes for FASTA)
  const uint64_t *seq_start = nullptr;  // FASTQ: start of every sequence; FASTA: null (rec_off delimits)
  const uint64_t *rec_off = nullptr;
  uint32_t *tb = nullptr, *tl = nullptr, *words = nullptr;
  uint64_t n_rec = 0;
  int status = 0;
};

static ParsedFastx parse_one(mhx_ctx *c, const char *h_text, uint64_t n, const std::string &tag, unsigned long long *totals) {
  hipStream_t st = c->stream;
  ParsedFastx p;
  auto W = [&](const char *name, size_t bytes) -> DevBuf & { return c->ws((tag + name).c_str(), bytes); };
  if (n == 0) return p;
  if (h_text[0] != '>' && h_text[0] != '@') {  // kseq skips junk up to the first header: rare, sequential parser
    p.status = 1;
    return p;
  }
  const bool fastq = h_text[0] == '@';
  char *t = W("text", n + 64).as<char>();
  upload_pinned(c, t, h_text, n);
  const uint64_t n_chunks = div_ceil(n, (uint64_t)kTextChunk);
  uint32_t *cnt = W("nl_cnt", (n_chunks + 1) * 4).as<uint32_t>();
  uint64_t *coff = W("nl_off", (n_chunks + 2) * 8).as<uint64_t>();
  uint32_t *flags = W("flags", 64).as<uint32_t>();  // [0] carriage returns, [1] format anomalies
  MHX_HIP(hipMemsetAsync(flags, 0, 64, st));
  MHX_LAUNCH(c, "fastx_newlines", (double)n,
             hipLaunchKernelGGL(k_count_newlines, dim3((unsigned)div_ceil(n_chunks, 256)), dim3(256), 0, st, t, n, cnt, flags));
  exclusive_scan_u32_u64(c, cnt, coff, n_chunks, coff + n_chunks);
  uint64_t n_nl = 0;
  uint32_t h_flags[2] = {0, 0};
  MHX_HIP(hipMemcpyAsync(&n_nl, coff + n_chunks, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h_flags[0]) {  // '\r': kseq strips it only under conditions that depend on the accumulated string
    p.status = 1;
    return p;
  }
  const uint64_t n_lines = n_nl + (h_text[n - 1] == '\n' ? 0 : 1);
  uint64_t *ls = W("line_start", (n_nl + 2) * 8).as<uint64_t>();
  MHX_LAUNCH(c, "fastx_line_starts", (double)n,
             hipLaunchKernelGGL(k_line_starts, dim3((unsigned)div_ceil(n_chunks, 256)), dim3(256), 0, st, t, n, coff, ls));
  if (fastq) {
    if (n_lines % 4) {
      p.status = 1;
      return p;
    }
    p.n_rec = n_lines / 4;
    uint64_t *ss = W("seq_start", (p.n_rec + 1) * 8).as<uint64_t>();
    uint32_t *sl = W("seq_len", (p.n_rec + 1) * 4).as<uint32_t>();
    MHX_LAUNCH(c, "fastq_records", (double)p.n_rec * 48,
               hipLaunchKernelGGL(k_fastq_records, dim3((unsigned)div_ceil(p.n_rec, 256)), dim3(256), 0, st, t, n, ls, n_lines, p.n_rec, ss, sl, flags + 1));
    p.seq = t;
    p.seq_start = ss;
    p.tb = W("trim_b", (p.n_rec + 1) * 4).as<uint32_t>();
    p.tl = W("trim_l", (p.n_rec + 1) * 4).as<uint32_t>();
    p.words = W("rec_words", (p.n_rec + 1) * 4).as<uint32_t>();
    MHX_LAUNCH(c, "fastx_trim", (double)n / 2,
               hipLaunchKernelGGL(k_trim_n, dim3((unsigned)div_ceil(p.n_rec, 256)), dim3(256), 0, st, t, ss, sl, (const uint64_t *)nullptr, p.n_rec, p.tb,
                                  p.tl, p.words, totals));
  } else {
    uint32_t *is_hdr = W("is_hdr", (n_lines + 1) * 4).as<uint32_t>();
    uint32_t *sl = W("line_seq_len", (n_lines + 1) * 4).as<uint32_t>();
    MHX_LAUNCH(c, "fasta_lines", (double)n_lines * 16,
               hipLaunchKernelGGL(k_fasta_lines, dim3((unsigned)div_ceil(n_lines, 256)), dim3(256), 0, st, t, n, ls, n_lines, is_hdr, sl, flags + 1));
    uint64_t *rec_of_line = W("rec_of_line", (n_lines + 2) * 8).as<uint64_t>();
    uint64_t *dense_off = W("dense_off", (n_lines + 2) * 8).as<uint64_t>();
    exclusive_scan_u32_u64(c, is_hdr, rec_of_line, n_lines, rec_of_line + n_lines);
    exclusive_scan_u32_u64(c, sl, dense_off, n_lines, dense_off + n_lines);
    uint64_t h2[2] = {0, 0};
    MHX_HIP(hipMemcpyAsync(&h2[0], rec_of_line + n_lines, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipMemcpyAsync(&h2[1], dense_off + n_lines, 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    p.n_rec = h2[0];
    char *dense = W("dense", h2[1] + 64).as<char>();
    uint64_t *rec_off = W("rec_off", (p.n_rec + 2) * 8).as<uint64_t>();
    MHX_HIP(hipMemcpyAsync(rec_off + p.n_rec, dense_off + n_lines, 8, hipMemcpyDeviceToDevice, st));
    MHX_LAUNCH(c, "fasta_compact", (double)n * 2,
               hipLaunchKernelGGL(k_fasta_compact, dim3((unsigned)div_ceil(n_lines * kWave, 256)), dim3(256), 0, st, t, ls, n_lines, is_hdr, sl, dense_off,
                                  rec_of_line, dense, rec_off));
    p.seq = dense;
    p.rec_off = rec_off;
    p.tb = W("trim_b", (p.n_rec + 1) * 4).as<uint32_t>();
    p.tl = W("trim_l", (p.n_rec + 1) * 4).as<uint32_t>();
    p.words = W("rec_words", (p.n_rec + 1) * 4).as<uint32_t>();
    if (p.n_rec)
      MHX_LAUNCH(c, "fastx_trim", (double)h2[1],
                 hipLaunchKernelGGL(k_trim_n, dim3((unsigned)div_ceil(p.n_rec, 256)), dim3(256), 0, st, dense, (const uint64_t *)nullptr,
                                    (const uint32_t *)nullptr, rec_off, p.n_rec, p.tb, p.tl, p.words, totals));
  }
  MHX_HIP(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h_flags[1]) p.status = 1;
  return p;
}

int fastx_to_records(mhx_ctx *c, const char *text1, uint64_t n1, const char *text2, uint64_t n2, mhx_fastx_result *out) {
  hipStream_t st = c->stream;
  memset(out, 0, sizeof *out);
  unsigned long long *totals = c->ws("fx_totals", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(totals, 0, 64, st));
  const bool paired = text2 != nullptr;
  ParsedFastx a = parse_one(c, text1, n1, "fx1_", totals);
  ParsedFastx b;
  if (paired) b = parse_one(c, text2, n2, "fx2_", totals);
  if (a.status || b.status) {
    out->status = 1;
    return 0;
  }
  uint64_t n_rec = a.n_rec;
  const uint32_t *words = a.words;
  if (paired) {
    const uint64_t np = std::min(a.n_rec, b.n_rec);  // PairedFastxReader stops when either file ends
    if (a.n_rec != b.n_rec) {  // the totals were accumulated over all records of the longer file: redo the sequential way
      out->status = 1;
      return 0;
    }
    n_rec = 2 * np;
    uint32_t *w = c->ws("fx_words", (n_rec + 1) * 4).as<uint32_t>();
    if (np) hipLaunchKernelGGL(k_interleave_u32, dim3((unsigned)div_ceil(np, 256)), dim3(256), 0, st, a.words, b.words, np, w);
    words = w;
  }
  uint64_t *off = c->ws("fx_off", (n_rec + 2) * 8).as<uint64_t>();
  uint64_t total_words = 0;
  if (n_rec) {
    exclusive_scan_u32_u64(c, words, off, n_rec, off + n_rec);
    MHX_HIP(hipMemcpyAsync(&total_words, off + n_rec, 8, hipMemcpyDeviceToHost, st));
  }
  unsigned long long h_tot[2] = {0, 0};
  MHX_HIP(hipMemcpyAsync(h_tot, totals, 16, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  DevBuf &rec = c->result(MHX_BUF_LIB_RECORDS, total_words * 4 + 8);
  rec.used = total_words * 4;
  if (paired) {
    const uint64_t np = n_rec / 2;
    if (np) {
      MHX_LAUNCH(c, "fastx_pack", (double)total_words * 4,
                 hipLaunchKernelGGL(k_pack_records, dim3((unsigned)div_ceil(np, 256)), dim3(256), 0, st, a.seq, a.seq_start, a.rec_off, a.tb, a.tl, off, np,
                                    (uint64_t)2, (uint64_t)0, rec.as<uint32_t>()));
      MHX_LAUNCH(c, "fastx_pack", (double)total_words * 4,
                 hipLaunchKernelGGL(k_pack_records, dim3((unsigned)div_ceil(np, 256)), dim3(256), 0, st, b.seq, b.seq_start, b.rec_off, b.tb, b.tl, off, np,
                                    (uint64_t)2, (uint64_t)1, rec.as<uint32_t>()));
    }
  } else if (n_rec) {
    MHX_LAUNCH(c, "fastx_pack", (double)total_words * 4,
               hipLaunchKernelGGL(k_pack_records, dim3((unsigned)div_ceil(n_rec, 256)), dim3(256), 0, st, a.seq, a.seq_start, a.rec_off, a.tb, a.tl, off, n_rec,
                                  (uint64_t)1, (uint64_t)0, rec.as<uint32_t>()));
  }
  MHX_HIP(hipStreamSynchronize(st));
  out->n_reads = n_rec;
  out->n_bases = h_tot[0];
  out->max_len = (uint32_t)h_tot[1];
  out->n_words = total_words;
  return 0;
}

}  // namespace mhx
