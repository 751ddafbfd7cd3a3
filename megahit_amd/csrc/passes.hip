// Memory-bounded operation: the reference's lv1 passes (BaseSequenceSortingEngine::Run / AdjustMemory /
// Lv1FindEndBuckets, reference src/sorting/base_engine.cpp:54-141,213-281).
//
// The engines normally materialise every item at once (288 GB of HBM hold ~70 M reads' worth).  For larger inputs
// the caller restricts a call to a set of lv1 buckets (mhx_set_bucket_filter): items are then extracted over
// BATCHES of reads into a small staging buffer and only the items of the kept buckets are appended to the sort
// buffer — the reference's OffsetFiller::IsHandling test (base_engine.h:106-108) applied after extraction instead
// of before it.  Like the reference, every pass rescans all reads.  The bucket histogram that a pass plan needs
// (Lv0CalcBucketSize, e.g. kmer_counter.cpp:114-156) is taken the same way (mhx_bucket_histogram).
#include <algorithm>

#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

// c->seqs temporarily restricted to the reads [r0, r1): extraction kernels index start[]/mult[] relative to the view
// and use absolute base offsets, so they need no change.  The fixed-length shortcuts (read = index / length) do not
// hold inside a view and are switched off.
struct SeqViewGuard {
  mhx_ctx *c;
  void *start_p, *mult_p;
  uint64_t n_seqs;
  uint32_t fixed_len;
  SeqViewGuard(mhx_ctx *ctx, uint64_t r0, uint64_t r1) : c(ctx) {
    SeqSet &s = c->seqs;
    start_p = s.start.p;
    mult_p = s.mult.p;
    n_seqs = s.n_seqs;
    fixed_len = s.fixed_len;
    s.start.p = reinterpret_cast<uint64_t *>(s.start.p) + r0;
    if (s.mult.p) s.mult.p = reinterpret_cast<uint16_t *>(s.mult.p) + r0;
    s.n_seqs = r1 - r0;
    if (r0 != 0 || r1 != n_seqs) s.fixed_len = 0;
  }
  ~SeqViewGuard() {
    SeqSet &s = c->seqs;
    s.start.p = start_p;
    s.mult.p = mult_p;
    s.n_seqs = n_seqs;
    s.fixed_len = fixed_len;
  }
};

__global__ void k_bucket_hist(const uint32_t *__restrict__ items, uint64_t n, int S, unsigned long long *__restrict__ hist) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&hist[items[i * S] >> 16], 1ull);
}

// all items of the current c->seqs for `stage` -> ws("items_a")
static StageItems extract_all(mhx_ctx *c, int stage, uint32_t k, uint32_t m) {
  StageItems r{0, 0, false, true};
  if (stage == MHX_STAGE_S1 || stage == MHX_STAGE_S1_MERCY) {
    const bool compact = s1_compact(c, k, stage == MHX_STAGE_S1_MERCY ? 1 : 0);
    r.n = s1_extract(c, k, compact);
    r.S = s1_stride(k, compact);
  } else if (stage == MHX_STAGE_COUNT) {
    r.n = count_extract(c, k, m);
    r.S = count_stride(k);
  } else if (stage == MHX_STAGE_SEQ2SDBG) {
    r.n = seq2sdbg_extract(c, k);
    r.S = seq2sdbg_stride(k);
  } else if (stage == MHX_STAGE_S2) {
    // every rank / pass must take the same path: the aggregated one needs stage 1 to have run with this (k, m),
    // which the (k <= 22, m >= 2) rule makes a pure function of the arguments
    r.agg = s2_use_aggregated(c, k, m);
    if (r.agg) {
      r.n = s2_agg_extract(c, k);  // stage-1 aggregates + dummies from the bitmap: not a per-read scan
      r.S = 2;
      r.batchable = false;
    } else {
      r.n = s2_extract(c, k, m);
      r.S = s2_stride(k);
    }
  } else throw Error("unknown stage");
  return r;
}

template <class PerBatch>
static StageItems for_each_batch(mhx_ctx *c, int stage, uint32_t k, uint32_t m, PerBatch &&per_batch) {
  SeqSet &s = c->seqs;
  const uint64_t ns = s.n_seqs;
  StageItems first{0, 0, false, true};
  // aggregated stage 2 is not a scan over reads: one "batch"
  if (stage == MHX_STAGE_S2 && s2_use_aggregated(c, k, m)) {
    first = extract_all(c, stage, k, m);
    per_batch(first);
    return first;
  }
  const uint64_t batch_bytes = c->filter_batch_bytes ? c->filter_batch_bytes : (1ull << 30);
  // upper bound of the staged bytes per read: items per read (count/S1: one per position + 4; stage 2: up to 6 per
  // position; seq2sdbg: 2 per position + 4) x item size
  const uint64_t L = s.max_len ? s.max_len : 1;
  uint64_t per_read_bytes;
  if (stage == MHX_STAGE_COUNT) per_read_bytes = (L + 4) * (uint64_t)count_stride(k) * 4;
  else if (stage == MHX_STAGE_S1 || stage == MHX_STAGE_S1_MERCY) per_read_bytes = (L + 4) * (uint64_t)s1_stride(k, false) * 4;
  else if (stage == MHX_STAGE_SEQ2SDBG) per_read_bytes = (2 * L + 4) * (uint64_t)seq2sdbg_stride(k) * 4;
  else per_read_bytes = 6 * L * (uint64_t)s2_stride(k) * 4;
  uint64_t batch_reads = std::max<uint64_t>(1, batch_bytes / per_read_bytes);
  bool have = false;
  for (uint64_t r0 = 0; r0 < ns || !have; r0 += batch_reads) {
    const uint64_t r1 = std::min(ns, r0 + batch_reads);
    SeqViewGuard view(c, r0, r1);
    StageItems b = extract_all(c, stage, k, m);
    if (!have) first = b;
    have = true;
    per_batch(b);
    if (ns == 0) break;
  }
  return first;
}

void bucket_histogram(mhx_ctx *c, int stage, uint32_t k, uint32_t m, uint64_t *h_out) {
  hipStream_t st = c->stream;
  unsigned long long *hist = c->ws("bucket_hist", MHX_NUM_BUCKETS * 8).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(hist, 0, MHX_NUM_BUCKETS * 8, st));
  // (stage 1 without mercy on reads of one length: straight from the packed reads, s1.hip)
  // (count on the stage-1 design, round 6: likewise — otherwise one global atomic per extracted item)
  const bool fast = (stage == MHX_STAGE_S1 && s1_bucket_histogram_fast(c, k, hist)) || (stage == MHX_STAGE_COUNT && count_bucket_histogram_fast(c, k, hist));
  if (!fast) for_each_batch(c, stage, k, m, [&](const StageItems &b) {
    if (b.n)
      hipLaunchKernelGGL(k_bucket_hist, dim3((unsigned)std::min<uint64_t>(div_ceil(b.n, 256), 8192)), dim3(256), 0, st,
                         c->work["items_a"].as<uint32_t>(), b.n, b.S, hist);
  });
  MHX_HIP(hipMemcpyAsync(h_out, hist, MHX_NUM_BUCKETS * 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
}

// items of `stage` -> ws("items_a"), restricted to the kept buckets when a filter is set
StageItems extract_stage(mhx_ctx *c, int stage, uint32_t k, uint32_t m) {
  if (!c->filter_on) return extract_all(c, stage, k, m);
  if (stage == MHX_STAGE_S2 && !s2_use_aggregated(c, k, m) && c->opt("s2_filter_in_extract", 1)) {
    // stage 2 per occurrence: the extraction itself leaves out the items of the dropped buckets (s2.hip s2_kept_mask) — one
    // scan of the reads per pass, no staging batches, no keep/drop split, nothing of the dropped buckets ever written
    c->s2_filter_in_extract = true;
    StageItems r;
    try {
      r = extract_all(c, stage, k, m);
    } catch (...) {
      c->s2_filter_in_extract = false;
      throw;
    }
    c->s2_filter_in_extract = false;
    if (r.n > c->filter_expected) throw Error("bucket filter: more items in the kept buckets than announced");
    c->pre_hist_buf = nullptr;
    return r;
  }
  if (stage == MHX_STAGE_S1 && c->s1_defer_items && s1_filter_in_gen_applies(c, k)) {
    // stage 1 on the fast shape: the first sort pass makes the records and leaves out those of the dropped buckets (s1.hip
    // S1GenT<true>) — one scan of the reads per pass, nothing staged, nothing split; "items_a" is empty until that pass ran
    c->s1_filter_in_gen = true;
    StageItems r{0, 0, false, true};
    try {
      r.n = s1_extract(c, k, true);
    } catch (...) {
      c->s1_filter_in_gen = false;
      throw;
    }
    r.S = s1_stride(k, true);
    return r;
  }
  c->s1_defer_items = false;  // (batches: the items are materialised here)
  hipStream_t st = c->stream;
  const uint8_t *lut = c->work["filter_lut"].as<uint8_t>();
  uint64_t kept = 0;
  DevBuf *keep = nullptr;
  StageItems res = for_each_batch(c, stage, k, m, [&](const StageItems &b) {
    const size_t ib = (size_t)b.S * 4;
    if (!keep) keep = &c->ws("items_keep", c->filter_expected * ib + 64);
    if (!b.n) return;
    uint32_t *part = c->ws("items_part", b.n * ib + 64).as<uint32_t>();
    uint64_t counts[2] = {0, 0};
    partition_by_owner(c, c->work["items_a"].as<uint32_t>(), part, b.n, b.S, lut, 2, counts);  // LUT: 0 keep, 1 drop
    if (kept + counts[0] > c->filter_expected) throw Error("bucket filter: more items in the kept buckets than announced");
    if (counts[0]) MHX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(keep->p) + kept * ib, part, counts[0] * ib, hipMemcpyDeviceToDevice, st));
    kept += counts[0];
  });
  MHX_HIP(hipStreamSynchronize(st));
  if (keep) std::swap(c->work["items_a"], *keep);  // the engines take their input from "items_a"
  c->pre_hist_buf = nullptr;
  res.n = kept;
  return res;
}

}  // namespace mhx
