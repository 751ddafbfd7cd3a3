// Internal host-side declarations of libmhx (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mhx.h"

namespace mhx {

struct OnesweepLaunch;
void set_error(const char *fmt, ...);

struct Error : std::runtime_error {
  explicit Error(const std::string &s) : std::runtime_error(s) {}
};

#define MHX_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      char buf_[512];                                                                          \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
               __LINE__);                                                                      \
      throw mhx::Error(buf_);                                                                  \
    }                                                                                          \
  } while (0)

// A grow-only device allocation that is kept between engine calls.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;   // bytes allocated
  size_t used = 0;  // bytes holding valid data (for result buffers)
  void reserve(size_t bytes);
  void release();
  template <class T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

struct KernelStat {
  uint32_t launches = 0;
  double ms = 0, bytes = 0;
};

struct PendingEvent {
  const char *name;
  hipEvent_t a, b;
  double bytes;
};

// device-resident packed sequence set
struct SeqSet {
  DevBuf words;      // uint32, padded with >= 32 zero words
  DevBuf start;      // uint64[n_seqs+1] (always materialised on device)
  DevBuf mult;       // uint16[n_seqs]
  uint64_t n_seqs = 0, n_bases = 0, n_words = 0;
  uint32_t fixed_len = 0, max_len = 0;
  std::vector<uint64_t> h_start;  // host copy when variable length (small path helpers)
};

}  // namespace mhx

struct mhx_ctx {
  int device = 0;
  int n_cus = 0;  // compute units of the device (0: unknown)
  hipStream_t stream = nullptr;
  mhx::SeqSet seqs;
  std::map<int, mhx::DevBuf> results;      // keyed by enum mhx_buffer
  std::map<std::string, mhx::DevBuf> work; // named scratch workspaces
  uint32_t sorted_item_words = 0;
  // multi-GPU: lv1-bucket partition and this rank's place in the global read set
  int my_part = 0, n_parts = 1;
  std::vector<uint32_t> part_begin;
  uint64_t pos_base = 0, global_bases = 0;
  // stage-2 items aggregated by stage 1 (ws "s2_agg_items"): valid for one (k, m) until the reads or the
  // is_solid bitmap change
  bool agg_valid = false;
  uint32_t agg_k = 0, agg_m = 0;
  uint64_t agg_n = 0;
  // the is_solid bitmap is exactly what stage 1 found for this (k, m) — no mercy edges added, not set by the caller: stage 2 may then take its
  // solid items from a count of the (k+1)-mers instead of from every occurrence (s2.hip s2_agg_from_count; 0: not so)
  uint32_t solid_plain_k = 0, solid_plain_m = 0;
  bool count_edges_only = false;  // count_stream_groups: only the solid edges are wanted (no first_0_out / last_0_in: positions do not matter)
  uint64_t n_route = 0;      // multi-GPU: records in ws("route_records") (count events)
  // digit histograms of the next sort, taken by the extraction kernel (ws "sort_pre_hist"): valid for exactly this buffer
  const void *pre_hist_buf = nullptr;
  uint64_t pre_hist_n = 0;
  int pre_hist_passes = 0;
  uint64_t pre_hist_sig = 0;  // passes_signature() of the plan the histograms were taken for
  // first sort pass whose records are generated instead of loaded (sort_kernels.h OnesweepLaunch; set by s1.hip, consumed by
  // the next radix_sort on exactly this buffer and item count): the buffer then holds NO items yet
  std::function<void(const mhx::OnesweepLaunch &)> gen_first_pass;
  const void *gen_buf = nullptr;
  uint64_t gen_n = 0;      // records the generated pass leaves (= what the rest of the sort handles)
  uint64_t gen_slots = 0;  // item slots it walks (> gen_n when it drops the items of filtered-out lv1 buckets)
  bool s2_filter_in_extract = false;  // passes.hip -> s2_extract: apply ws "filter_lut" while counting / writing the items
  bool s1_filter_in_gen = false;      // passes.hip -> s1_extract: the generating first sort pass applies ws "filter_bits" (s1.hip S1GenT<true>)
  uint32_t filter_kept = 0;           // lv1 buckets the filter keeps
  // stage 1: records per lv1 bucket the ranks of a multi-GPU run agreed on (comm.hip; 0: each call derives it from its own
  // item count, s1.hip s1_density) — every rank must make the same sort plan
  double s1_density = 0;
  bool s1_var_gen = false;            // the last s1_extract armed the variable-length generating pass (S1GenVarT)
  std::string last_s1_plan;           // what the last stage 1 ran as (mhx_last_s1_plan; bench.py prints it)
  bool s1_defer_items = false;  // the caller of extract_stage(S1) will sort right away: s1_extract may defer the items to that sort
  // memory-bounded passes (passes.hip): only items of the kept lv1 buckets are materialised
  bool filter_on = false, accumulate = false;
  uint64_t filter_expected = 0, filter_batch_bytes = 0;
  bool global_marks_inverted = false;  // multi-GPU stage 1 marked the non-solid occurrences (s1.hip)
  uint64_t s1_acc_bits = 0, mercy_acc_n = 0;  // stage-1 state that accumulate continues
  uint32_t s1_acc_k = 0, s1_acc_m = 0;
  uint32_t count_acc_k = 0, count_acc_m = 0;  // (k, m) of the count state that accumulate continues
  uint64_t n_marks = 0;          // multi-GPU, sparse marks: positions in ws("s1_marks") waiting to be routed
  uint64_t dist_local_solid = 0; // multi-GPU, sparse marks: solid occurrences in the local reads after the marks arrived
  bool dist_s2_agg = false;  // the items of the current multi-GPU stage-2 exchange are aggregated ones
  // tuning knobs (mhx_set_option): explicit value, else environment MHX_<NAME>, else the default
  std::map<std::string, long long> options;
  // tuned defaults of this installation: `name = value` lines of mhx_tuning.conf beside libmhx.so (MHX_TUNING_FILE names
  // another file, MHX_NO_TUNING=1 ignores it), read at mhx_create; consulted after explicit options and the environment
  std::map<std::string, long long> tuned;
  long long opt(const char *name, long long dflt) const;
  // pinned staging buffers of upload_pinned (capi.hip)
  void *pinned[2] = {nullptr, nullptr};
  hipEvent_t pinned_free[2] = {nullptr, nullptr};
  // profiling
  bool profiling = false;
  std::vector<mhx::PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  // side streams for kernels that run next to each other (kmsort_emu.hip) and the events that fork / join them
  std::vector<hipStream_t> side_streams;
  std::vector<hipEvent_t> side_events;
  std::map<std::string, mhx::KernelStat> stats;

  mhx::DevBuf &ws(const char *name, size_t bytes) {
    mhx::DevBuf &b = work[name];
    b.reserve(bytes);
    return b;
  }
  mhx::DevBuf &result(int which, size_t bytes) {
    mhx::DevBuf &b = results[which];
    b.reserve(bytes);
    b.used = bytes;
    return b;
  }
  void prof_begin(const char *name, double bytes);
  void prof_end();
  void prof_collect();
};

// Launch wrapper: records a HIP-event pair around the launch when profiling is on.
#define MHX_LAUNCH(ctx, name, bytes, ...) \
  do {                                    \
    (ctx)->prof_begin(name, bytes);       \
    __VA_ARGS__;                          \
    MHX_HIP(hipGetLastError());           \
    (ctx)->prof_end();                    \
  } while (0)

namespace mhx {

// ---- sort.hip ----
struct SortPass {
  int shift;       // bit offset from the LSB of the big-endian key (key_words*32 bits)
  int bits;        // digit = bits [shift, shift+bits) ...
  int shift2 = 0;  // ... optionally continued by bits [shift2, shift2+bits2) as its upper part
  int bits2 = 0;   // bits + bits2 <= 8
  // >= 0: the earlier passes of the plan sorted bits [prev_lo, shift), and the consumer does not look at the order of records
  // that agree on every bit the plan sorts — lets the pass rank with LDS atomics where that cannot matter (sort_kernels.h RANK 2)
  int prev_lo = -1;
};
// Sorts n items of `stride` uint32 words held in buf_a (ping-pong with buf_b) by the digit passes
// (least-significant pass first).  Returns the buffer holding the result.
uint32_t *radix_sort(mhx_ctx *c, uint32_t *buf_a, uint32_t *buf_b, uint64_t n, int stride, int key_words,
                     const std::vector<SortPass> &passes);
// prefix passes + segment finish in LDS when that saves passes (sort.hip); same result as radix_sort with `passes`
uint32_t *sort_whole_key(mhx_ctx *c, uint32_t *buf_a, uint32_t *buf_b, uint64_t n, int stride, int key_words,
                         const std::vector<SortPass> &passes);
std::vector<SortPass> make_passes(int key_words, int lo_bit, int hi_bit);
bool sort_takes_generated_first_pass(const mhx_ctx *c, uint64_t n, int stride, const std::vector<SortPass> &passes);
uint64_t passes_signature(const std::vector<SortPass> &ps);
// kmsort_emu.hip: sort with the reference's exact (unstable) tie order, one GPU thread per lv1 bucket
uint32_t *kmsort_exact(mhx_ctx *c, uint32_t *buf_a, uint32_t *buf_b, uint64_t n, int S, int key_words);

// ---- scan.hip ----
// exclusive scan of n uint32 values into uint64 (in != out); returns total via d_total (device, uint64[1])
void exclusive_scan_u32_u64(mhx_ctx *c, const uint32_t *in, uint64_t *out, uint64_t n, uint64_t *d_total);
void exclusive_scan_u64(mhx_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n, uint64_t *d_total);
// positions i in [0,n) where item i differs from item i-1 in the first `cmp_bits` bits of the key
// (big-endian words), written ascending to heads[]; *d_count receives their number.
void find_group_heads(mhx_ctx *c, const uint32_t *items, uint64_t n, int stride, int cmp_bits,
                      uint64_t *heads, uint64_t *d_count);
uint64_t count_group_heads(mhx_ctx *c, const uint32_t *items, uint64_t n, int stride, int cmp_bits);

struct DigitSpec;
DigitSpec spec_of_pass(const SortPass &ps, int key_words);
std::vector<SortPass> make_passes_ranges(int key_words, const std::vector<std::pair<int, int>> &ranges);
// s2.hip: SdBG records from sorted lv2 items (shared by read2sdbg S2 and seq2sdbg)
void emit_sdbg(mhx_ctx *c, const uint32_t *sorted, uint64_t n_items, int S, int kw, uint32_t k, int is_seq, mhx_sdbg_result *out, int compact_bits = 0);

void partition_by_owner(mhx_ctx *c, const uint32_t *a, uint32_t *b, uint64_t n, int stride, const uint8_t *lut, int n_parts,
                        uint64_t *counts);
uint64_t s1_extract(mhx_ctx *c, uint32_t k, bool compact);
bool s1_compact(const mhx_ctx *c, uint32_t k, int want_mercy);
bool s1_rank_tagged(const mhx_ctx *c, uint32_t k);
int s1_stride(uint32_t k, bool compact);
// stage-1 records pre-sorted by lv1 bucket, in n arrays (multi-GPU: one per sending rank; comm.hip)
struct S1Sources {
  int n = 0;
  std::vector<const uint32_t *> ptr;
  std::vector<uint64_t> count;
  uint32_t *spare = nullptr;  // scratch of >= 12 bytes per record for the group-by's output regions
  int pbits = 16;             // key prefix bits the sources are sorted on (the stream plan's seg_bits)
};
// ---- stage 1 on super-k-mer records (s1_skm.hip) ----
constexpr int kSkmBatch = 8;   // bins per ticket of the group-by
constexpr int kSkmSrcMax = 8;  // sources of the group-by: one array on a single GPU, one per sending rank on several
struct SkmFront {
  int n_src;
  const uint4 *src[kSkmSrcMax];            // the records, ordered by minimizer bin
  const uint64_t *src_bounds[kSkmSrcMax];  // [n_bins + 1] each: where the bins start in the source
  uint32_t *spare;       // the other sort buffer (the workgroups' output regions)
  uint64_t spare_bytes;
  uint64_t n_records, n_windows;
  uint64_t n_items;        // what the reference sorts: L - k + 4 items per read that holds an edge
  uint32_t n_bins, max_bin;
  uint32_t bin_lo, bin_hi;  // the bins of this pass / of this owner
  int bin_bits;
  unsigned long long *hp;   // one GPU: the homopolymer windows counted beside the records ([0..1] counts, [2..3] a position each), else nullptr
};
bool s1_skm_applies(const mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy);
bool s1_skm_dist_applies(const mhx_ctx *c, uint32_t k, uint32_t m);
int s1_skm_passes(const mhx_ctx *c, uint32_t k);
bool s1_skm_front(mhx_ctx *c, uint32_t k, SkmFront *f, int pass, int n_passes, int bin_bits_agreed = 0, bool for_count = false);
void s1_skm_bounds_of(mhx_ctx *c, const uint4 *recs, uint64_t n, uint32_t n_bins, uint64_t *bounds);
void s1_skm_groups_launch(mhx_ctx *c, bool agg, unsigned grid, const SkmFront &f, uint32_t k, uint32_t m, uint8_t *solid_bytes, unsigned long long *hist,
                          uint2 *agg_raw, uint32_t agg_cap, uint32_t *agg_counts, uint32_t *err, unsigned long long *marks_raw, uint32_t marks_cap,
                          uint32_t *marks_counts);
void s1_skm_hp_publish(mhx_ctx *c, const SkmFront &f, uint32_t k, uint32_t m, uint8_t *solid_bytes, unsigned long long *hist, uint2 *agg_items,
                       uint64_t *agg_cursor, bool agg);
// the owner's half on several GPUs (s1.hip): group-by over the received sources, marks as a list, aggregated items.  -> false: a region overflowed
bool s1_skm_owner(mhx_ctx *c, uint32_t k, uint32_t m, const SkmFront &f, mhx_s1_result *out);

bool s1_presort_applies(const mhx_ctx *c, uint32_t k, uint64_t n_local_items);
uint32_t *s1_presort(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, int *pbits);
bool s1_filter_in_gen_applies(const mhx_ctx *c, uint32_t k);
bool s1_bucket_histogram_fast(mhx_ctx *c, uint32_t k, unsigned long long *hist);
uint32_t s1_pos_bits(const mhx_ctx *c);
uint64_t s1_pos_stride(const mhx_ctx *c, uint32_t k);
std::string s1_plan_text(const mhx_ctx *c, uint32_t k, uint64_t n_items);
int s1_process(mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_s1_result *out,
               const S1Sources *pre = nullptr);
__global__ void k_add_u64(unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b, int n);
void sdbg_accumulate(mhx_ctx *c, bool first);
void sdbg_publish_accumulated(mhx_ctx *c);
uint64_t s2_extract(mhx_ctx *c, uint32_t k, uint32_t m);
int s2_process(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_sdbg_result *out);
bool s2_use_aggregated(const mhx_ctx *c, uint32_t k, uint32_t m);
uint64_t s2_agg_extract(mhx_ctx *c, uint32_t k);
int s2_agg_process(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_sdbg_result *out);
struct StageItems {
  uint64_t n;      // items in ws("items_a")
  int S;           // words per item
  bool agg;        // stage 2: aggregated items (s2.hip)
  bool batchable;  // produced by a scan over reads
};
StageItems extract_stage(mhx_ctx *c, int stage, uint32_t k, uint32_t m);
void bucket_histogram(mhx_ctx *c, int stage, uint32_t k, uint32_t m, uint64_t *h_out);
int s2_stride(uint32_t k);
constexpr int MHX_BUF_IS_SOLID_LOCAL = 100;  // internal: this rank's slice of the global bitmap (multi-GPU)
constexpr int MHX_BUF_MERCY_CAND_LOCAL = 101;  // internal: routed mercy candidates of the local reads, local positions
uint64_t count_extract(mhx_ctx *c, uint32_t k, uint32_t m = 0);
int count_stride(uint32_t k);
int count_process(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_count_result *out);
void count_apply_events(mhx_ctx *c, const unsigned long long *ev, uint64_t n);
uint64_t seq2sdbg_extract(mhx_ctx *c, uint32_t k);
int seq2sdbg_stride(uint32_t k);
int seq2sdbg_process(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, mhx_sdbg_result *out);
const uint64_t *sort_u64(mhx_ctx *c, const void *src, uint64_t n, int hi_bit);
void invert_local_marks(mhx_ctx *c, unsigned long long *words, uint64_t n_words);
DevBuf &grow_preserving(mhx_ctx *c, DevBuf &b, size_t bytes, size_t keep);
void stash_route_records(mhx_ctx *c, const void *src, uint64_t n, int hi_bit);
void mercy_adopt_routed(mhx_ctx *c, const long long *recv, uint64_t n);
void s1_apply_marks(mhx_ctx *c, const unsigned long long *recv, uint64_t n);
int sdbg_build_index(mhx_ctx *c, uint32_t k, mhx_sdbg_index_info *out);
int sdbg_remove_tips(mhx_ctx *c, const mhx_sdbg_index_info *info, int max_tip_len, uint64_t *n_removed);
int iterate_edges(mhx_ctx *c, uint32_t k, uint32_t step, const uint32_t *ctg_words, uint64_t ctg_n_words, uint64_t n_ctg, const uint64_t *ctg_start,
                  mhx_iterate_result *out);
int fastx_to_records(mhx_ctx *c, const char *text1, uint64_t n1, const char *text2, uint64_t n2, mhx_fastx_result *out);
int sdbg_load_bytes(mhx_ctx *c, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *off, const uint64_t *items, const uint64_t *tips,
                    const uint64_t *large);

// ---- engines ----
int run_count(mhx_ctx *c, uint32_t k, uint32_t m, mhx_count_result *out);
// count on the bucket streaming of stage 1 (s1.hip: CountGenT, k_s1_stream<COUNT>)
int s2_agg_compact_bits(uint32_t k);
bool s2_agg_from_count_applies(mhx_ctx *c, uint32_t k, uint32_t m);  // s2.hip: stage 2's solid items from a count of the (k+1)-mers  // count bits of a compact aggregated stage-2 item (k = 23..28), 0: none
struct CountStreamOut {
  unsigned grid;      // workgroups = edge regions
  uint32_t cap;       // 8-byte edges a region holds
  uint32_t *counts;   // device: edges per region
  uint32_t *spare;    // the sort buffer the regions live in
  uint32_t *sorted;   // the records, ordered by the plan's prefix
  uint64_t n_items, n_distinct;
  std::string plan;
  unsigned long long *events;  // several GPUs: position << 1 | kind, for the ranks that hold the reads (count.hip k_apply_count_events)
  uint64_t n_events;
  // count on super-k-mer records: what a pass made (the caller sums the passes for the plan line)
  const unsigned long long *skm_hp = nullptr;  // the homopolymer windows k_skm_make counted beside the records (nullptr: none)
  uint64_t skm_records = 0, skm_windows = 0;
  uint32_t skm_max_bin = 0;
  int skm_bin_bits = 0;
};
bool count_stream_applies(const mhx_ctx *c, uint32_t k, uint32_t m);
// `count` on super-k-mer records (s1_skm.hip): one GPU, 19 <= k <= 21, min count <= 2.  *touched: the caller's arrays may hold partial results
bool count_skm_applies(const mhx_ctx *c, uint32_t k, uint32_t m);
bool count_skm_groups(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *first_0_out, uint32_t *last_0_in_p1, unsigned long long *hist, CountStreamOut *o, bool *touched,
                      int pass, int n_passes);
void count_skm_hp_publish(mhx_ctx *c, const unsigned long long *hp, uint32_t k, uint32_t m, unsigned long long *hist, unsigned long long *edges_at,
                          uint64_t *n_edges, uint64_t *n_keys, bool *flagged);
bool count_presort_applies(const mhx_ctx *c, uint32_t k, uint32_t m);
uint32_t *count_presort(mhx_ctx *c, uint32_t k, uint64_t *n_items, uint32_t **other, int *pbits);
int count_process_presorted(mhx_ctx *c, uint32_t k, uint32_t m, const S1Sources &src, mhx_count_result *out);  // count.hip; -1: gave up
bool count_bucket_histogram_fast(mhx_ctx *c, uint32_t k, unsigned long long *hist);  // s1_front.hip
bool count_stream_groups(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *first_0_out, uint32_t *last_0_in_p1, unsigned long long *hist, CountStreamOut *o,
                         const S1Sources *pre = nullptr);
int run_s1(mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy, mhx_s1_result *out);
int run_s1_mercy(mhx_ctx *c, uint32_t k, uint64_t *num_mercy);
int run_s2(mhx_ctx *c, uint32_t k, uint32_t m, mhx_sdbg_result *out);
int run_seq2sdbg(mhx_ctx *c, uint32_t k, mhx_sdbg_result *out);
struct MercyShare {  // multi-GPU seq2sdbg --need_mercy (mercy.hip / comm.hip)
  int my_part = 0, n_parts = 1;
  std::function<void(uint8_t *d_flags, uint64_t n)> reduce_flags;  // bitwise OR of the per-position flags over the ranks
};
int run_gen_mercy(mhx_ctx *c, uint32_t k, const uint32_t *cand_packed, uint64_t cand_words, uint64_t n_cand,
                  const uint64_t *cand_start, uint64_t *n_mercy, const MercyShare *share = nullptr);
void upload_sequences(mhx_ctx *c, const uint32_t *packed, uint64_t n_words, uint64_t n_seqs, uint32_t fixed_len,
                      const uint64_t *start_pos);
void upload_fixed_starts(mhx_ctx *c);
void upload_edges(mhx_ctx *c, const uint32_t *raw, uint64_t n_edges, uint32_t k, uint32_t wpe);
void upload_pinned(mhx_ctx *c, void *d_dst, const void *h_src, size_t bytes);
void append_sequences(mhx_ctx *c, const uint32_t *packed, uint64_t n_words, uint64_t n_new, uint32_t fixed_len, const uint64_t *start_pos,
                      const uint16_t *mult);
void upload_bin_records(mhx_ctx *c, const uint32_t *records, uint64_t n_words, uint64_t n_seqs, int reverse);

inline int round_up2(int x) { return (x + 1) & ~1; }
inline uint64_t div_ceil(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

}  // namespace mhx
