// Stage 1 (read2sdbg) and `count` on its design: what the translation units of the stage share — the record helpers, the argument
// blocks of the group-by kernels, the sort plan, and the launchers each unit exports.
//   s1_front.hip   extraction / digit- and bucket-histogram kernels, the generating first sort pass (s1_gen.h)
//   s1_tile.hip    the tile group-bys: k_tile_groups<S1Op> (classic, mercy) and k_s1_seg (segments)
//   s1_stream.hip  the bucket streaming k_s1_stream (+ giant buckets, + `count`)
//   s1.hip         marks / bitmap kernels, the planner, the stage driver (S1Stage), count's driver
#pragma once
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "dev_prims.h"
#include "mhx_internal.h"
#include "sort_digits.h"
#include "sort_kernels.h"

namespace mhx {

// COMPACT (no mercy requested): the aux part is one word, the absolute position of the (k-1)-mer; that is
// all the group reduction needs to set is_solid, and it makes the record 12 instead of 16 bytes at k <= 29.
// item of slot j (0 .. L-k+3) of the read at base offset st, length L (read_to_sdbg_s1.cpp:228-292, :344-363)
// Positions in compact records: the record's third word holds the low `pos_bits` bits of the (k-1)-mer's global base
// position, the bits above them (the "tag", < 256) ride in key bits that no comparison looks at, between the (k-1)-mer and
// head/tail: bits [6, 14) of the last key word.  pos_bits = 32 unless a test asks for less (s1_pos_bits); read sets below
// 2^pos_bits bases have tag 0 everywhere — the plain 32-bit position.  (Replaces the rank tags of round 2/3: the same bits,
// but a function of the position alone, so one rank may hold more than 2^32 bases: 100 M reads on one GPU.)
__device__ __forceinline__ uint32_t s1_pos_word(uint64_t p, uint32_t pos_bits) { return pos_bits >= 32 ? (uint32_t)p : (uint32_t)p & ((1u << pos_bits) - 1u); }
__device__ __forceinline__ uint32_t s1_pos_tag(uint64_t p, uint32_t pos_bits) { return (uint32_t)(p >> pos_bits) << 6; }

__device__ __forceinline__ uint64_t rc64(uint64_t x, int n) {  // reverse complement of the n chars in the top 2n bits
  uint64_t r = __builtin_bitreverse64(x);
  r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
  return (~r) << (64 - 2 * n);
}

constexpr int kFastPasses = 4;
struct HiDigits {
  unsigned sh[kFastPasses], mk[kFastPasses];
  int n;
};

// ---------------------------------------------------------------------------------------------------------------
// Giant buckets of the bucket streaming (round 5).  A workgroup streams a bucket alone, so ONE bucket of millions of records —
// low-complexity sequence: 1 % poly-A reads put 13 M records of one key into lv1 bucket 0 — held the whole stage up for 15 ms.
// Such a bucket (>= min_records) is cut into slices that many workgroups reduce in parallel (k_s1_giant_reduce: an LDS
// table per slice -> "partial entries" = a key's first record + its count in the slice), the streaming kernel skips it
// (flag[bucket]), and a second launch of the streaming kernel (GIANT) inserts the few partial entries with their counts
// and does the per-key work as for any bucket.  A bucket whose slices do not reduce into their region (many distinct keys)
// clears its flag and is streamed as before.  Everything is found and sized on the device: no host round trip.
struct S1Giant {
  uint8_t *flag;            // [n_buckets] 1: taken by the giant path
  uint32_t *ctr;            // [0] giants found (may exceed gcap)  [2..3] partial entries allotted (64-bit)
  uint32_t *bucket, *sl, *ns, *cap, *cur;  // per giant: bucket, slice length, slices, region capacity, entries written
  unsigned long long *off;  // per giant: first entry of its region in `partial`
  uint4 *partial;           // entries: the three words of a key's first record in the slice + its count there
  uint32_t gcap;            // giants the list holds
  uint32_t min_records;     // a bucket at least this large is a giant
  unsigned long long pcap;  // entries `partial` holds
  // count: the solid keys of a giant that have no in- or out-edge (local key << 2 | flags), listed by the launch on the partial entries;
  // k_count_giant_look then sends MANY workgroups over the giant's records for first_0_out / last_0_in (one workgroup took 43 ms for the
  // 6.6 x 10^7 records of a poly-G bucket).  A list that overflows: the listing workgroup looks at the bucket itself, as for any bucket.
  uint32_t *fl_cnt;            // per giant: keys listed (may exceed fl_cap)
  unsigned long long *fl_key;  // [gcap][fl_cap]
  uint32_t fl_cap;
  unsigned long long *ev;      // several GPUs: the events of k_count_giant_look (one list, a cursor in ev_cur), else nullptr
  uint32_t *ev_cur;
  uint32_t ev_cap;
};
constexpr uint32_t kGiantFlagged = 1024;
constexpr uint32_t kGiantSliceMin = 16384, kGiantEntriesPerSlice = 256;

struct S1SegArgs {
  int k;
  uint32_t m;
  uint32_t pfx_mask;  // bits of key word 0 that form the segment prefix
  uint32_t eq_mask1;  // bits of key word 1 that take part in key equality: (k-1)-mer bits + head/tail (not the rank tag)
  uint8_t *solid_bytes;
  int mark_mode;      // 0: mark solid occurrences, 1: mark the non-solid ones, 2: statistics only
  unsigned long long *hist, *ctr;  // ctr[0] / ctr[2]: solid / head-and-tail occurrences (mark_mode 2)
  // aggregated stage-2 items: every (persistent) workgroup fills a region of its own, agg_raw[blockIdx.x * agg_cap ...],
  // and leaves its item count in agg_counts[blockIdx.x]; k_agg_compact packs the regions afterwards.  (A shared
  // cursor costs one same-address global atomic per wavefront and tile: ~10 ns each, 2.6 M of them at 10 M reads.)
  uint2 *agg_raw;
  uint32_t agg_cap;
  uint32_t *agg_counts;
  // multi-GPU, sparse marks: instead of a store into a byte map of the GLOBAL read set, a mark is the position itself,
  // appended to the workgroup's region marks_raw[blockIdx.x * marks_cap ...] (count in marks_counts[blockIdx.x]); the
  // host packs the regions and routes the positions to the ranks that hold those reads (comm.hip)
  unsigned long long *marks_raw;
  uint32_t marks_cap;
  uint32_t *marks_counts;
  uint64_t pos_stride;
  uint32_t *err;
  int la_chunks;      // look-ahead limit, in chunks of 256 records
  int direct_marks;   // k_s1_stream: the non-solid marks come from the table (one stored position per key), no second read
  S1Giant giant;      // k_s1_stream: buckets handed to the giant path (flag == nullptr: none)
  // k_s1_stream<COUNT>: the reads (first_0_out / last_0_in are per read) and the two arrays (kmer_counter.cpp:307-368)
  const uint64_t *c_start;
  uint64_t c_n_seqs;
  uint32_t c_fixed_len;
  uint32_t *first_0_out, *last_0_in_p1;
  int c_edges_only;  // count: no first_0_out / last_0_in wanted (stage 2's aggregated items from a count of the (k+1)-mers): no second look
  int c_wpe;  // words per packed edge (kmer_counter.cpp:32-52): 2 up to k = 23 — (k+1)-mer and multiplicity in 64 bits —, 3 beyond: 16-byte region entries
};

constexpr unsigned long long kSegEmpty = ~0ull;  // never a key: head/tail bits 63 do not occur (max (4<<3)|4)

constexpr int kSegHist = 512;  // multiplicities counted in LDS

constexpr int kStreamThreads = 1024;  // one workgroup per CU: 8192 slots of key + count + first position = 98 KB of LDS
constexpr uint32_t kStreamEmpty = 0xFFFFFFFFu;  // never a key: head/tail bits 63 do not occur

__global__ void k_bucket_bounds(const uint32_t *__restrict__ items, uint64_t n, int stride, uint64_t *__restrict__ bstart, int pbits);  // kmsort_emu.hip

struct S1StreamGeom {
  int pbits;          // prefix bits the records are sorted on: 2^pbits buckets, bounds[q * (2^pbits + 1) + b]
  int sub0;           // every bucket starts with 2^sub0 sub-rounds
  uint32_t n_buckets; // 1 << pbits
  uint32_t max_fill;  // a round whose table ends up with more keys than this is redone in two halves
};
constexpr int kStreamBatch = 4;     // buckets per ticket
constexpr int kStreamSrcMax = kWave;  // bucket bounds of up to this many sources are staged in LDS (one lane of wave 0 per source)

constexpr int kCountStreamMaxK = 22;          // `count` on this design: the (k+1)-mer, strand, prev / next and an 8-bit tag in two key words
constexpr uint32_t kCountStrandBit = 64u;

// ---- the sort plan (s1.hip) ----
struct S1Plan {
  std::vector<SortPass> passes;
  int seg_bits;
  bool stream;
  int sub0 = 0;            // stream: sub-rounds every bucket starts with (log2)
  double per_bucket = 0;   // the density the plan was made for
};
inline int s1_kw(uint32_t k) { return (int)div_ceil((k - 1) * 2 + 6, 32); }  // read_to_sdbg_s1.cpp:107-108
std::vector<SortPass> s1_sort_passes(uint32_t k);
S1Plan s1_plan(const mhx_ctx *c, uint32_t k, uint64_t n_items, bool compact, int want_mercy, bool allow_stream = true);
bool s1_shape_is_fast(const mhx_ctx *c, uint32_t k, bool compact);
bool s1_shape_is_var_fast(const mhx_ctx *c, uint32_t k, bool compact);

// ---- launchers of the group-by units ----
struct S1StreamLaunch {  // one launch of k_s1_stream (s1_stream.hip)
  bool agg, half, tags, giant, count;
  unsigned grid;
  const uint32_t *items0;
  const uint64_t *bounds;
  S1SegArgs a;
  S1StreamGeom geo;
  uint32_t stride;
  uint32_t *ticket;
  const uint32_t *const *srcs;
  int n_src;
  bool key64 = false;  // local keys wider than 32 bits (k > 22 at a 16-bit prefix)
};
void s1_stream_launch(mhx_ctx *c, const char *name, double bytes, const S1StreamLaunch &l);
void s1_giant_launch(mhx_ctx *c, const uint32_t *items0, const uint32_t *const *srcs, const uint64_t *bounds, int n_src, uint64_t n_buckets, int pbits, int k,
                     const S1Giant &g, bool key64 = false, bool count = false);
// count: first_0_out / last_0_in of the reads that hold a listed key of a giant bucket (s1_stream.hip k_count_giant_look)
void count_giant_look_launch(mhx_ctx *c, const uint32_t *items0, const uint32_t *const *srcs, const uint64_t *bounds, int n_src, uint64_t n_buckets, int pbits,
                             const S1SegArgs &a);
// k_s1_seg (s1_tile.hip): per = 4 | 8 records per thread
void s1_seg_launch(mhx_ctx *c, const char *name, double bytes, int per, bool agg, unsigned grid, const uint32_t *sorted, uint64_t n_items, const S1SegArgs &a,
                   uint64_t n_work, uint32_t stride);
// k_tile_groups<S1Op> on fully sorted records of S words (s1_tile.hip); agg: with the aggregated stage-2 items (S == 3 compact, S == 4 not compact)
void s1_classic_launch(mhx_ctx *c, int S, bool compact, bool agg, const uint32_t *sorted, uint64_t n_items, int KWv, int kmer_bits, uint32_t m, uint8_t *solid_bytes,
                       unsigned long long *is_solid, int mark_atomic, unsigned long long *hist, unsigned long long *ctr, int want_mercy, long long *&mercy, int k,
                       uint2 *agg_items, uint64_t *agg_cursor, int mark_mode);
// the front of `count` on this design (s1_front.hip): digit histograms of the plan's passes from the packed reads, the record buffers, and the
// generating first pass armed for the next radix_sort on *buf_a.  -> false: no read holds an edge
bool count_shape_is_fast(const mhx_ctx *c, uint32_t k);  // the shapes CountGenT / CountGenVarT serve
bool count_bucket_histogram_fast(mhx_ctx *c, uint32_t k, unsigned long long *hist);  // lv1 histogram of count's items from the packed reads
bool count_stream_front(mhx_ctx *c, uint32_t k, const S1Plan &plan, uint32_t **buf_a, uint32_t **buf_b, uint64_t *n_items);

}  // namespace mhx
