/*
 * mhx.h — C ABI of libmhx.so: MI355X-native SdBG construction (MEGAHIT's `count`,
 * `read2sdbg`, `seq2sdbg` hot path) as hand-written HIP kernels for gfx950.
 *
 * The reference (voutcn/megahit v1.2.9) has no FFI for this path; its seams are
 *   B1  the CLI  `megahit_core count|read2sdbg|seq2sdbg`  (src/main_sdbg_build.cpp:35-224),
 *   B2  the engine template method BaseSequenceSortingEngine::Run + six virtuals
 *       (src/sorting/base_engine.h:221-233,256),
 *   B3  the sort functor SelectSortingFunc (src/sorting/kmsort_selector.h:7).
 * Each entry point below names the reference interface it replaces.  The product CLI
 * (megahit_amd/csrc/host/mhx_core.cpp) sits on exactly these symbols and keeps B1's flags
 * and on-disk formats, so `megahit` (the Python orchestrator) and `megahit_core assemble`
 * consume the output unchanged.
 *
 * Conventions: plain C, opaque handle, int status (0 = ok, <0 = error, text via
 * mhx_last_error()), no exceptions cross the boundary, the caller owns every host buffer,
 * the library owns device memory.  One handle drives one GPU and one HIP stream; it is not
 * thread-safe (use one handle per thread).  All integers little-endian native.
 *
 * Packed sequences: uint32 words, base i of the concatenation of all sequences lives in word
 * i/16 at bits 31-2(i%16)..30-2(i%16) (A,C,G,T = 0..3), no padding between sequences — the
 * layout of SequencePackage (reference src/sequence/sequence_package.h:38-320).
 */
#ifndef MHX_H
#define MHX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHX_NUM_BUCKETS 65536 /* reference src/sorting/base_engine.h:21 */
#define MHX_MAX_MUL 65535     /* reference src/sdbg/sdbg_def.h:12 */
#define MHX_MAX_K 255         /* reference src/sdbg/sdbg_def.h:20 */

typedef struct mhx_ctx mhx_ctx;

/* ---- lifetime / errors ---- */
const char *mhx_last_error(void);
const char *mhx_version(void);
/* number of visible HIP devices, <0 on error */
int mhx_device_count(void);
/* device: HIP ordinal.  Fails (returns NULL, message in mhx_last_error) when no GPU / no
 * gfx950 code object is usable — there is no CPU fallback. */
mhx_ctx *mhx_create(int device);
void mhx_destroy(mhx_ctx *);
/* release cached device workspaces (they are otherwise kept between calls) */
int mhx_trim(mhx_ctx *);
/* forget inputs, results, partition, filters and options, keep the device buffers: the next job starts as on a new
   handle without re-allocating (what `mhx_core --serve` does between two sub-programs) */
int mhx_reset(mhx_ctx *);
int mhx_synchronize(mhx_ctx *);
/* Tuning / diagnostic knobs of one handle (never needed for correct results).  A knob that was not set falls back to
 * the environment variable MHX_<NAME IN UPPER CASE>, then to its built-in default.  Known names:
 *   s1_seg (1)        0: stage 1 always sorts fully and reduces with the tile kernel (no segment group-by)
 *   s1_seg_bits (0)   force the prefix width of the partial stage-1 sort (0 = chosen from the item count)
 *   s1_seg_la (3)     look-ahead chunks of the segment group-by before a tile gives up (-> classic path)
 *   s1_seg_per (8)    records per thread and tile of the segment group-by (4 or 8)
 *   s1_stream (1)     0: never the bucket-streaming group-by (k_s1_stream).  Its sort prefix follows the job's density
 *                     (records per lv1 bucket where the group-by runs): s1_stream_max (40000) records per streamed
 *                     bucket at most — 16 prefix bits in two passes up to that, 2^s sub-rounds per bucket up to
 *                     2^s1_stream_sub_max (1) times that, 17..24 bits in three passes beyond (their
 *                     width aims at s1_stream_max3 = s1_stream_max / 2 records per bucket: a third pass costs the same at any width); s1_stream_bits (0) /
 *                     s1_stream_sub0 (-1) force the prefix width / the sub-rounds (tests); s1_stream_fill: keys a
 *                     round's table may end up with before the round is redone in two halves (7/8 of the table); s1_stream_probes (1024; the kernel looks at no more than 128 slots per insert)
 *   s1_filter_in_gen (1)  0: a bucket filter (memory plan) is applied to stage 1 by extraction batches + a keep/drop
 *                     split even where the generating first sort pass could leave the dropped buckets out itself
 *   s1_pos_bits (0)   width of the position word of compact stage-1 records (0 = 32); the position bits above it ride
 *                     as a tag in the key words (s1.hip s1_pos_tag).  Tests use narrow words to exercise the tags.
 *   s1_stream_direct (1)  0: k_s1_stream always reads a bucket twice (second time to mark); 1: when the marks are those of
 *                     the non-solid occurrences and m <= 2 they come from the table and the second read is skipped
 *   count_seg (1), count_seg_bits (0), count_seg_la (3)  the same for count (k_count_seg)
 *   count_extract_fixed (1)  0: count's items always come from the wave-per-read kernel (no fused digit histograms)
 *   dist_sparse_marks (0)  multi-GPU stage 1 emits MHX_ROUTE_S1_MARKS records instead of marking a bitmap of the
 *                     global read set (set by mhx_dist_setup)
 *   kmsort_emu_legacy (0)  1: read2sdbg --need_mercy replays kmsort with one thread per lv1 bucket on whole records
 *                     (round 1) instead of one wave per bucket on tags + indices (kmsort_emu.hip); same output
 *   sort_xcd_units (1)  0: the chained-scan scatter hands its units out from one ticket counter instead of one per
 *                     block-id class b % 8 (sort.hip); same output
 * Returns <0 for a NULL handle/name. */
int mhx_set_option(mhx_ctx *, const char *name, long long value);
/* The value a knob has for this handle: an explicit mhx_set_option, else the environment variable MHX_<NAME>, else the
 * installation's tuned default — a `name = value` line of mhx_tuning.conf beside libmhx.so (MHX_TUNING_FILE names another
 * file, MHX_NO_TUNING=1 ignores it; written by tools/ab_options.py --write-tuning from an A/B on the box; read once, at
 * mhx_create) — else `dflt`.  Knobs only ever choose between code paths with identical results.  Round 3:
 *   sort_unit_runs (1)     0: the chained-scan pass ranks, stages and writes its unit tile by tile (k_radix_onesweep) instead of
 *                          unit-wide runs (k_radix_onesweep_u)
 *   sort_rank_uniform (1)  0: every loading pass ranks with the match-any ballots.  1: a pass whose plan declares the bits
 *                          sorted before it (the prefix plans of stage 1) ranks with one LDS atomic per record wherever all
 *                          records of a wavefront instruction agree on those bits — an order that cannot matter there — and
 *                          with the ballots elsewhere (sort_kernels.h RANK 2); correct on any hardware
 *   s1_gen_blocked (0)     1: the generating first sort pass of stage 1 gives every thread consecutive items and requests the
 *                          window words of a whole unit up front (S1GenBlocked) instead of one window load per item
 *   s1_digit_hist_preload (0) 1: the same in the digit-histogram pre-pass
 * (Round 6: s1_stream_half and s1_marks_list are gone with their code: slower on every box for two rounds.)
 * (Round 4: s1_stream_prefetch / s1_stream_next_bucket / s1_stream_used_list / s1_stream_read_first / s1_stream_unroll are gone:
 *  the bucket streaming always has its next trip's loads in flight, fetches the next bucket's bounds during the current bucket and
 *  walks its table once; setting them changes nothing.)
 * Round 5:
 *   s1_gen_roll (1)        0: the blocked generating pass forms every item's window and reverse complement anew (S1GenBlockedT) instead
 *                          of one 64-bit window + one reverse complement per run of a thread's eight items (S1GenRollT; k <= 23)
 *   s1_digit_hist_roll (1) the same for the digit-histogram pre-pass and the lv1 histogram taken from the packed reads
 *   s1_var_fast (1)        0: a library whose reads are not of one length never takes the generating pass (S1GenVarT / CountGenVarT:
 *                          item slots padded to the longest read's, unfilled slots declined); s1_var_min_fill (50): per cent of the
 *                          slots that have to be real records for that form to be taken
 *   s1_giant (1)           0: a bucket of the bucket streaming is always streamed by one workgroup alone; 1: buckets of at least
 *                          s1_giant_min (262144) records are cut into slices, reduced by many workgroups and finished from their
 *                          partial entries (S1Giant, s1.hip) — low-complexity reads
 *   count_stream (1)       0: `count` always extracts 16-byte items and reduces with the tile kernel (k_count_seg); 1: fixed- or
 *                          variable-length reads on one GPU, k <= 22, min count <= 2 take 12-byte records made by the first sort
 *                          pass and the bucket streaming (k_s1_stream<COUNT>)
 *   sdbg_fast (1)          0: the SdBG records of 8-byte items come from the generic tile kernel (k_tile_groups<SdbgOp>); 1: every run
 *                          head on its own (k_sdbg_fast) for aggregated and seq2sdbg items (short runs), the tile kernel for items per
 *                          occurrence (runs as long as the coverage); 2: the run-head form for every 8-byte item; sdbg_fast_keep (1): the counting launch keeps its findings (4 bytes per item,
 *                          up to sdbg_fast_keep_max_mb = 4096 MB) for the emitting launch; sdbg_fast_halo (128), sdbg_fast_tile (2048):
 *                          records staged either side of a tile / per tile (tests, tuning)
 *   edges_reserve_permille (1000)  room behind the edges mhx_load_edges uploads, for mercy edges (the CLI: 1250, or what
 *                          MEGAHIT_NUM_MERCY_FACTOR says: seq_to_sdbg.cpp:370-378)
 * Older knobs kept for A/Bs and tests (every one only chooses between code paths with identical results):
 *   s1_bucket_hist_fast (1)   0: the lv1 histogram of a memory plan extracts items instead of scanning the packed reads
 *   s1_digit_hist_blocked (1), s1_digit_hist_plain (1)   0: the digit-histogram pre-pass of the generating sort pass in its round-3 forms
 *   s1_gen_any_order (1)      0: the generating first sort pass ranks with ballots (a stable order) instead of LDS atomics
 *   s1_pack_fixed (1)         0: the byte map of the marks becomes the bitmap through the general kernel also for reads of one length
 *   sort_hybrid_margin_ps (6) picoseconds per record the prefix-passes + segment-finish plan has to win by in sort_whole_key's cost model
 * Round 6:
 *   count_stream (1)       now also: pass by pass under a bucket filter (memory plan; the filter sits inside the histogram pre-pass and the
 *                          generating sort pass), on several GPUs (pre-sorted exchange, the first_0_out / last_0_in events routed to the
 *                          read owners), min count 1..15 (3..15: per-char 4-bit counters that stop at m instead of seen-once / seen-twice
 *                          bits); count_stream_wide (1): k = 23..27 on the same design — a window per item (CountGenWideT), 64-bit table
 *                          keys, three-word edges from k = 24 —, reads of one length, read sets below 2^32 bases (no room for position tags)
 *   count_giant (1)        0: `count` streams a giant bucket (>= s1_giant_min records) with one workgroup; 1: slices + partial entries with
 *                          per-char counters + k_count_giant_look for the keys without an in- or out-edge
 *   count_event_share (4)  several GPUs: the event regions of k_s1_stream<COUNT> hold items / count_event_share events in all
 *   s1_stream_wide (1)     0: stage 1 at k = 23..29 takes the segment group-by (k_s1_seg); 1: the bucket streaming with 64-bit table keys
 *   s2_agg_from_count (1)  0: stage 2 always makes its solid items per occurrence where stage 1 left no aggregated ones; 1: min count 1
 *                          (k <= 27) and min count >= 2 at k = 23..27 take them from a count of the (k+1)-mers (s2.hip s2_agg_from_count)
 *   sort_loaded_ut2 (0)    1: the 12-byte passes that load their records take units of two tiles (four workgroups per CU instead of three);
 *                          measured slower (profiles/r06_ab_loaded_ut2.jsonl): the passes are bound per scattered run, not by occupancy
 *   fetch_pinned (1)       0: mhx_fetch is one hipMemcpy into the caller's (pageable) memory; 1: results of >= 16 MB leave through two
 *                          pinned staging buffers, the copy off the device overlapped with the copy into the destination
 *   s1_skm (1)             stage 1 on super-k-mer records (csrc/s1_skm.hip): 0 never; 1 where the shape is served (one GPU, no mercy, no bucket
 *                          filter, 19 <= k <= 22, min count <= 2, fewer than 2^36 bases, reads of one length or at least s1_var_min_fill per cent
 *                          of the padded blocks filled) and the job has s1_skm_min_windows (2^22) windows; 2 whatever the size (tests);
 *                          3 as 1, but a job the path gives up on FAILS instead of taking the prefix plan (mhx_s1_self_planned).
 *                          s1_skm_cap_pct (36): records the arrays hold per 100 windows (0.284 per window in random sequence);
 *                          s1_skm_max_bin (65536): a bin of more records hands the job to the prefix plan (low-complexity reads);
 *                          s1_skm_pass_gb (48): both record arrays of a pass together — larger jobs run in passes over ranges of bins,
 *                          s1_skm_passes (0) forces their number; s1_skm_bin_bits (0 = by density: 16..20) the bins; s1_skm_tags (0) 1: the
 *                          kernel of read sets beyond 2^32 bases on any read set (tests); s1_skm_deal (1) 0: every lane expands its own
 *                          record instead of the wavefront's windows being dealt to the lanes (measured 7 % slower); s1_skm_hp (1) 0: the
 *                          windows of one base (poly-A, poly-G) stay in the records instead of being counted beside them (one GPU)
 *   count_skm (1)          `count` on super-k-mer records (k_skm_make<.., COUNT>, k_count_skm: a record carries one more base either side of its run,
 *                          the table the in / out characters): one GPU, 19 <= k <= 21, min count <= 2; jobs of more than s1_skm_pass_gb of
 *                          records in passes over ranges of bins (mhx_count_self_planned); 0: never; 3: fail instead of falling back;
 *                          count_skm_group (2): chunks of 64 windows per round of compare-and-swaps (4 spills registers: measured slower)
 *   dist_skm (1)           several GPUs: 0: stage 1 never exchanges super-k-mer records by bin (comm.hip dist_s1_skm; the pre-sorted exchange
 *                          of 12-byte records runs); 1: where every rank serves the shape (as s1_skm, at most 8 ranks, no memory plan)
 * The CLI's memory plan (host/mhx_core.cpp plan_ranges): MHX_PLAN_BY_TIME=0 plans by space only; MHX_ALLOC_S_PER_GB=<seconds> sets the
 * hipMalloc rate the time plan assumes (tests).
 * (mhx_tuning.conf of this tree: s1_gen_blocked = 1.) */
long long mhx_get_option(mhx_ctx *, const char *name, long long dflt);
/* What the last stage 1 of this handle ran as: "super-k-mers m13, 2^16 bins (366 M records for 1.29 G windows: 3.52 per record; …)" /
 * "stream p16 sub0 2 passes (20345 records per lv1 bucket)" / "seg p24 3 passes" /
 * "full sort 6 passes".  The string lives until the next stage 1 of the handle; "" before the first. */
const char *mhx_last_s1_plan(const mhx_ctx *);

/* ---- sequence store (replaces SeqPackage held by each engine:
 *      kmer_counter.h:76, read_to_sdbg.h:47-51, seq_to_sdbg.h:79) ---- */
/* Upload a packed sequence set.  start_pos has n_seqs+1 entries (base offsets; start_pos[0]=0)
 * or is NULL when every sequence has fixed_len bases.  For `count`/`read2sdbg` the reads must
 * already be REVERSED as the reference does at load time (binary_reader.h:41-45). */
int mhx_load_sequences(mhx_ctx *, const uint32_t *packed, uint64_t n_words, uint64_t n_seqs,
                       uint32_t fixed_len, const uint64_t *start_pos);
/* Same, but from the forward-orientation `.bin` record stream of a read library
 * (uint32 len + ceil(len/16) words per read, sequence_package.h:224-240): reversal and
 * concatenation happen on the GPU (replaces BinaryReader::Read + AppendReversedCompactSequence). */
int mhx_load_bin_records(mhx_ctx *, const uint32_t *records, uint64_t n_words, uint64_t n_seqs,
                         int reverse);
/* Append more sequences (and their multiplicities, may be NULL -> 0) behind the loaded set, as
 * SeqToSdbg::Initialize does for contigs after edges (seq_to_sdbg.cpp:449-503). */
int mhx_append_sequences(mhx_ctx *, const uint32_t *packed, uint64_t n_words, uint64_t n_seqs,
                         uint32_t fixed_len, const uint64_t *start_pos, const uint16_t *mult);
/* Load packed (k+1)-mer edges exactly as `.edges.<i>` files hold them (words_per_edge words per edge, the multiplicity in
 * the low 16 bits of the last word): sequences and multiplicities are unpacked on the GPU (replaces the per-edge
 * AppendCompactSequence loop of EdgeReader::ReadSorted/ReadUnsorted, edge_reader.h:24-52, seq_to_sdbg.cpp:388-420). */
int mhx_load_edges(mhx_ctx *, const uint32_t *edges, uint64_t n_edges, uint32_t k, uint32_t words_per_edge);
/* per-sequence multiplicities for seq2sdbg (seq_to_sdbg.h:80) */
int mhx_load_multiplicity(mhx_ctx *, const uint16_t *mult, uint64_t n_seqs);
uint64_t mhx_num_sequences(const mhx_ctx *);
/* common length of the loaded sequences, 0 when they differ (SequencePackage's fixed-length fast path,
 * sequence_package.h:131-137) */
uint32_t mhx_fixed_length(const mhx_ctx *);
uint64_t mhx_num_bases(const mhx_ctx *);

/* ---- result buffers (device resident until fetched) ---- */
enum mhx_buffer {
  MHX_BUF_EDGES = 1,        /* uint32[n_edges][words_per_edge], bucket order (edge_writer.h:69-80) */
  MHX_BUF_BUCKET_COUNT = 2, /* uint64[65536]: edges (count) or SdBG items (s2/seq2sdbg) per bucket */
  MHX_BUF_FIRST_0_OUT = 3,  /* uint32[n_reads]   (kmer_counter.h:78) */
  MHX_BUF_LAST_0_IN = 4,    /* uint32[n_reads]   (kmer_counter.h:79) */
  MHX_BUF_MUL_HIST = 5,     /* int64[65536]      (edge_counter.h:14-56) */
  MHX_BUF_IS_SOLID = 6,     /* uint64[ceil(n_bases/64)] (AtomicBitVector, kmbitvector.h:67-88) */
  MHX_BUF_MERCY_CAND = 7,   /* int64[n_mercy_cand], sorted ascending (read_to_sdbg_s1.cpp:466-551) */
  MHX_BUF_SDBG_BYTES = 8,   /* uint8[sdbg_bytes]: bucket byte streams, bucket-id order (sdbg_writer.cpp:38-57) */
  MHX_BUF_BUCKET_OFFSET = 9,/* uint64[65536] starting byte of each bucket in MHX_BUF_SDBG_BYTES */
  MHX_BUF_BUCKET_TIPS = 10, /* uint64[65536] */
  MHX_BUF_BUCKET_LARGE = 11,/* uint64[65536] */
  MHX_BUF_SORTED_ITEMS = 12,/* uint32[n_items][item_words]: the lv2 items of the last engine as its sort left them (tests): a view into the
                              sort workspace, overwritten by the next engine call.  Fully sorted on the tile paths; on the bucket-streaming
                              plans (stage 1 without mercy, `count` at k <= 22: mhx_count_result.item_words == 3) 12-byte records ordered
                              only by the plan's key prefix */
  MHX_BUF_W_COUNT = 13,     /* uint64[9] + ones_in_last: uint64[10] (sdbg_meta.h:41-48) */
  /* device-resident SdBG hand-over, filled by mhx_sdbg_build_index (below); layouts = the reference's in-memory ones */
  MHX_BUF_SDBG_W = 20,          /* uint64[ceil(n/16)]: 4 bits per item, item i at bits 4*(i%16) (sdbg_raw_content.h:22) */
  MHX_BUF_SDBG_LAST = 21,       /* uint64[ceil(n/64)]: bit i%64 of word i/64 */
  MHX_BUF_SDBG_TIP = 22,        /* same */
  MHX_BUF_SDBG_INVALID = 23,    /* tip | (W == 0): SDBG::invalid_ after LoadFromFile (sdbg.h:33-60) */
  MHX_BUF_SDBG_SMALL_MUL = 24,  /* uint8[n], 255 = look in MUL (sdbg_raw_content.cpp:72-83) */
  MHX_BUF_SDBG_MUL = 25,        /* uint16[n]: SDBG::EdgeMultiplicity */
  MHX_BUF_SDBG_TIP_LABELS = 26, /* uint32[n_tips][words_per_tip_label], chars reversed inside each word (:85-91) */
  MHX_BUF_SDBG_PREFIX_LKT = 27, /* int64[65536][2]: first / last item of every bucket (sdbg.h:38-49) */
  MHX_BUF_SDBG_RS_W_L2 = 28,    /* int64[9][num_l2_w]   kmlib::RankAndSelect<4,9> over W (kmrns.h:118-175) */
  MHX_BUF_SDBG_RS_W_L1 = 29,    /* uint16[9][num_l1_w] */
  MHX_BUF_SDBG_RS_W_SEL = 30,   /* uint32[]: select samples of character c at [w_sel_offset[c], w_sel_offset[c+1]) */
  MHX_BUF_SDBG_RS_LAST_L2 = 31, /* int64[num_l2_bits]   RankAndSelect<1,2> over last */
  MHX_BUF_SDBG_RS_LAST_L1 = 32, /* uint16[num_l1_bits] */
  MHX_BUF_SDBG_RS_LAST_SEL = 33,/* uint32[last_sel_count] */
  MHX_BUF_SDBG_RS_TIP_L2 = 34,  /* rank-only structure over tip */
  MHX_BUF_SDBG_RS_TIP_L1 = 35,
  MHX_BUF_LIB_RECORDS = 40      /* uint32[]: read-library records (len + packed words per read) of mhx_fastx_to_records */
};
/* bytes currently held in a result buffer (0 if absent) */
uint64_t mhx_buffer_bytes(const mhx_ctx *, int which);
/* copy [offset, offset+bytes) of a result buffer to host memory */
int mhx_fetch(mhx_ctx *, int which, void *dst, uint64_t offset, uint64_t bytes);

/* ---- engines ---- */
typedef struct {
  uint64_t n_items;        /* items sorted = sum max(0, len-k) */
  uint64_t n_distinct;     /* distinct (k+1)-mers */
  uint64_t n_edges;        /* solid edges emitted */
  uint32_t words_per_edge; /* ceil((2(k+1)+16)/32), edge_writer.h:37-40 */
  uint32_t item_words;     /* stride of MHX_BUF_SORTED_ITEMS */
} mhx_count_result;
/* KmerCounter::Run (kmer_counter.cpp:60-414 under base_engine.cpp:143-211).
 * Fills EDGES, BUCKET_COUNT, FIRST_0_OUT, LAST_0_IN, MUL_HIST. */
int mhx_count(mhx_ctx *, uint32_t k, uint32_t min_count, mhx_count_result *out);

typedef struct {
  uint64_t n_items;      /* (k-1)-mer items sorted */
  uint64_t n_solid;      /* solid (k+1)-mer occurrences marked */
  uint64_t n_mercy_cand; /* 0 unless want_mercy */
  uint32_t item_words;
} mhx_s1_result;
/* Read2SdbgS1::Run (read_to_sdbg_s1.cpp:88-566).  Fills IS_SOLID, MUL_HIST and, when
 * want_mercy != 0, MERCY_CAND.  want_mercy: 0 = no candidates; 1 = candidates with the stable tie order
 * (a group's first item = first in read order); 2 = candidates with the reference's exact tie order
 * (kmlib::kmsort's unstable permutation replayed per lv1 bucket; slower) — see DESIGN.md "H1". */
int mhx_read2sdbg_s1(mhx_ctx *, uint32_t k, uint32_t min_count, int want_mercy, mhx_s1_result *out);
/* 1 when stage 1 of the loaded reads would run on super-k-mer records (csrc/s1_skm.hip: one GPU, no mercy, 19 <= k <= 22, min count <= 2,
 * no bucket filter) — a form that cuts a job too large for one working set into passes over ranges of its own bins BY ITSELF
 * (the lv1 bucket ranges of base_engine.cpp:54-141 would break its records apart).  A caller that plans bucket-range passes asks first:
 * on 1 it sets option s1_skm = 3 and calls mhx_read2sdbg_s1 once, without a filter — with 3 the call fails (instead of quietly taking the
 * prefix plan on the whole job) when the input turns out not to be served, low-complexity reads for one, and the caller plans as before
 * with s1_skm = 0.  host/mhx_core.cpp read2sdbg does exactly that. */
int mhx_s1_self_planned(mhx_ctx *, uint32_t k, uint32_t min_count, int want_mercy);
/* The same question for mhx_count (count on super-k-mer records: one GPU, 19 <= k <= 21, min count <= 2): on 1 the caller sets option
 * count_skm = 3 and calls mhx_count once without a bucket filter; a job the path gives up on fails that call, and the caller plans lv1
 * bucket ranges as before with count_skm = 0. */
int mhx_count_self_planned(mhx_ctx *, uint32_t k, uint32_t min_count);

/* mercy block of Read2SdbgS2::Initialize (read_to_sdbg_s2.cpp:122-266): consumes MERCY_CAND,
 * sets extra IS_SOLID bits on the device copy; *num_mercy receives "Number mercy". */
int mhx_read2sdbg_add_mercy(mhx_ctx *, uint32_t k, uint64_t *num_mercy);
/* replace the device IS_SOLID bitmap (e.g. one produced elsewhere); n_words = ceil(n_bases/64) */
int mhx_set_is_solid(mhx_ctx *, const uint64_t *bits, uint64_t n_words);

typedef struct {
  uint64_t n_items;    /* lv2 items sorted */
  uint64_t n_sdbg;     /* SdBG records emitted */
  uint64_t n_tips;     /* $-tips */
  uint64_t n_large;    /* records with multiplicity > 254 */
  uint64_t sdbg_bytes; /* size of MHX_BUF_SDBG_BYTES */
  uint32_t words_per_tip_label;
  uint32_t item_words;
} mhx_sdbg_result;
/* Read2SdbgS2::Run minus the mercy block (read_to_sdbg_s2.cpp:271-630).  Uses IS_SOLID unless
 * min_count == 1 (for_sure_solid, read_to_sdbg_s2.cpp:295).  Fills SDBG_BYTES, BUCKET_*. */
int mhx_read2sdbg_s2(mhx_ctx *, uint32_t k, uint32_t min_count, mhx_sdbg_result *out);
/* SeqToSdbg::Run minus input parsing (seq_to_sdbg.cpp:530-807) on the loaded sequences +
 * multiplicities.  Fills SDBG_BYTES, BUCKET_*. */
int mhx_seq2sdbg(mhx_ctx *, uint32_t k, mhx_sdbg_result *out);

/* SeqToSdbg::GenMercyEdges (seq_to_sdbg.cpp:171-357): the loaded sequences must be the sorted
 * (k+1)-mer edges; cand = candidate reads (packed, start_pos as in mhx_load_sequences).  Appends
 * the mercy edges (multiplicity 1) to the loaded set; *n_mercy receives their number. */
int mhx_gen_mercy_edges(mhx_ctx *, uint32_t k, const uint32_t *cand_packed, uint64_t cand_words,
                        uint64_t n_cand, const uint64_t *cand_start, uint64_t *n_mercy);

/* ---- device-resident SdBG hand-over.  Builds, on the GPU and from the SdBG the handle holds
 * (MHX_BUF_SDBG_BYTES + MHX_BUF_BUCKET_* of the last stage-2 / seq2sdbg call, or a stream installed with
 * mhx_sdbg_load_bytes), everything SDBG::LoadFromFile builds on one CPU thread: the W / last / tip / multiplicity /
 * tip-label arrays of LoadSdbgRawContent (sdbg_raw_content.cpp:18-96) and the rank/select tables, prefix table, f and
 * rank_f of sdbg.h:26-61 over kmlib/kmrns.h:118-175 — in the reference's own layouts (MHX_BUF_SDBG_*), so a
 * downstream stage can adopt them without reading .sdbg files back. ---- */
typedef struct {
  uint64_t n_items, n_tips, n_large;
  uint32_t k, words_per_tip_label;
  int use_full_mul;          /* what the reference would choose: n_large >= 0.08 n (sdbg_raw_content.cpp:29-30) */
  uint64_t num_l1_w, num_l2_w;       /* table lengths per character of the W structure */
  uint64_t num_l1_bits, num_l2_bits; /* ... of the last / tip structures */
  uint64_t w_char_count[9];          /* kmrns char_count_ (character 0 includes the zero padding of the last word) */
  uint64_t w_sel_offset[10];
  uint64_t ones_in_last, ones_in_tip, last_sel_count;
  long long f[6], rank_f[6];         /* sdbg.h:482-483 */
} mhx_sdbg_index_info;
int mhx_sdbg_build_index(mhx_ctx *, uint32_t k, mhx_sdbg_index_info *out);
/* SdBG-level tip trimming on the device-resident graph — sdbg_pruning::RemoveTips
 * (assembly/sdbg_pruning.cpp:61-179: rounds of Trim with len = 2, 4, ... < max_tip_len, then max_tip_len) over the
 * succinct graph's Forward/Backward navigation (sdbg.h:106-121,240-330), one thread per edge, on the buffers that
 * mhx_sdbg_build_index left in HBM.  Updates MHX_BUF_SDBG_INVALID in place; *n_removed = tips removed (the number the
 * reference logs).  `assemble` calls it with max_tip_len = 2k by default (main_assemble.cpp:143-156). */
int mhx_sdbg_remove_tips(mhx_ctx *, const mhx_sdbg_index_info *info, int max_tip_len, uint64_t *n_removed);
/* install an SdBG produced elsewhere (e.g. read back from .sdbg.* files: bucket byte ranges back to back) as the handle's
 * current SdBG; the four tables have 65536 entries (starting byte, items, tips, large multiplicities per bucket) */
int mhx_sdbg_load_bytes(mhx_ctx *, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *bucket_offset, const uint64_t *bucket_items,
                        const uint64_t *bucket_tips, const uint64_t *bucket_large);

/* ---- buildlib on the GPU.  FASTA / FASTQ text (already inflated) -> the record stream of a read
 * library (`<lib>.bin`: per read uint32 length + ceil(len/16) words, 2 bits per base MSB first, forward orientation;
 * sequence_package.h:224-240) in MHX_BUF_LIB_RECORDS, with the reference's N-trimming (fastx_reader.cpp:56-71) and
 * character mapping (sequence_package.h:78-83).  text2 != NULL: a paired library, records interleaved r1, r2, r1, ...
 * (paired_fastx_reader.cpp:7-43).  Replaces SequenceLibCollection::Build's parse + pack (sequence_lib.cpp:8-91).
 * out->status 1 = the text is not plain FASTA / four-line FASTQ (carriage returns, multi-line FASTQ, junk before the
 * first header, mates of different counts): nothing was produced, run a sequential kseq-compatible parser instead. ---- */
typedef struct {
  uint64_t n_reads, n_bases, n_words; /* n_bases counts an all-N / empty read as 1 (it is stored as one 'A') */
  uint32_t max_len;
  int status;
} mhx_fastx_result;
int mhx_fastx_to_records(mhx_ctx *, const char *text1, uint64_t n1, const char *text2, uint64_t n2, mhx_fastx_result *out);

/* ---- `iterate` — the (k+step+1)-mer edges that the loaded reads (FORWARD orientation:
 * mhx_load_bin_records(..., reverse = 0)) support next to the contigs of round k.  Replaces ContigFlankIndex +
 * KmerCollector (iterate/contig_flank_index.h:16-219, iterate/kmer_collector.h:49-69, main_iterate.cpp:117-145): flank
 * records sorted and searched, one thread per read, the k-mer set = sort + unique.  contigs: packed as
 * mhx_load_sequences takes them, forward, loop / standalone contigs already dropped by the caller
 * (async_sequence_reader.h:80).  Fills MHX_BUF_EDGES with n_edges records of words_per_edge words, multiplicity 0 as
 * the reference writes them, in sorted order (the reference's order is its hash set's). ---- */
typedef struct {
  uint64_t n_flanks, n_kmers, n_edges;
  uint32_t words_per_edge;
} mhx_iterate_result;
int mhx_iterate(mhx_ctx *, uint32_t k, uint32_t step, const uint32_t *contig_packed, uint64_t contig_words, uint64_t n_contigs,
                const uint64_t *contig_start, mhx_iterate_result *out);

/* B3: sort n fixed-width records in place on the GPU, ascending by the first key_words words
 * (lexicographic on uint32, as Substr::operator<, kmsort_selector.cpp:18-27); aux words ride
 * along.  Replaces SelectSortingFunc(key_words, aux_words) (kmsort_selector.cpp:61-63).  Stable. */
int mhx_sort_records(mhx_ctx *, uint32_t *host_items, uint64_t n, uint32_t key_words, uint32_t aux_words);

/* ---- multi-GPU: one handle per GPU/process.  The 65536 lv1 buckets are split into
 * contiguous owner ranges; every rank extracts the items of ITS reads, partitions them by owner
 * (one stable multisplit pass), the caller moves them with an all-to-all on its own communication
 * backend (or lets mhx_comm / mhx_dist_* below do it over RCCL), and every rank sorts + reduces the
 * buckets it owns — the reference's OffsetFiller::IsHandling bucket filter (base_engine.h:106-108)
 * turned into an exchange.  All three sub-programs are supported (BASELINE configs[2..4]). ---- */
int mhx_set_partition(mhx_ctx *, int my_part, int n_parts, const uint32_t *bucket_begin /* n_parts+1 */);
/* This rank's reads sit at base offset pos_base of a global read set of global_bases bases
 * (is_solid then spans the global set).  (0, 0) switches the global layout off. */
int mhx_set_global_layout(mhx_ctx *, uint64_t pos_base, uint64_t global_bases);
enum mhx_stage {
  MHX_STAGE_S1 = 1,        /* read2sdbg stage 1, no mercy: compact items */
  MHX_STAGE_S2 = 2,        /* read2sdbg stage 2 */
  MHX_STAGE_COUNT = 3,     /* count */
  MHX_STAGE_SEQ2SDBG = 4,  /* seq2sdbg */
  MHX_STAGE_S1_MERCY = 5   /* read2sdbg stage 1 with mercy candidates: full items (prev/next, 64-bit position) */
};
typedef struct {
  void *d_items;       /* device pointer: items grouped by owner, owners ascending */
  uint64_t n_items;
  uint32_t item_bytes;
} mhx_dist_items;
/* extract + partition; counts[p] = items destined to owner p (n_parts entries) */
int mhx_dist_extract(mhx_ctx *, int stage, uint32_t k, uint32_t min_count, mhx_dist_items *out, uint64_t *counts);
/* library-owned device buffer to receive n_items items into */
void *mhx_dist_recv_buffer(mhx_ctx *, uint64_t n_items, uint32_t item_bytes);
/* sort + reduce the received items: S1 sets bits of the GLOBAL bitmap MHX_BUF_IS_SOLID — by convention the bits of the
 * NON-solid (k+1)-mer occurrences of the owned buckets (fewer marks on typical inputs; summing over ranks still means
 * OR); mhx_adopt_is_solid_slice derives is_solid = "a (k+1)-mer starts here and it is not marked" for the local reads.
 * S2 emits the SdBG records of the owned buckets */
int mhx_dist_process_s1(mhx_ctx *, uint32_t k, uint32_t min_count, int want_mercy, uint64_t n_items, mhx_s1_result *out);
int mhx_dist_process_s2(mhx_ctx *, uint32_t k, uint64_t n_items, mhx_sdbg_result *out);
/* count: solid edges + bucket counts + histogram of the owned buckets; first_0_out / last_0_in of the LOCAL reads are
 * complete only after the read events have been routed (below).  seq2sdbg: SdBG records of the owned buckets (items
 * carry no positions, so the sequences may be spread over the ranks in any way; no mercy edges in this mode). */
int mhx_dist_process_count(mhx_ctx *, uint32_t k, uint32_t min_count, uint64_t n_items, mhx_count_result *out);
int mhx_dist_process_seq2sdbg(mhx_ctx *, uint32_t k, uint64_t n_items, mhx_sdbg_result *out);
/* Records keyed by a position in the GLOBAL read set, produced by the bucket owners and consumed by the rank that
 * holds the read (SURVEY §8e "secondary reductions"): the first_0_out/last_0_in updates of count
 * (kmer_counter.cpp:307-368) and the mercy candidates of read2sdbg stage 1 (read_to_sdbg_s1.cpp:466-551).
 * route: 8-byte records sorted by position + counts[p] = records for rank p, where rank p holds positions
 * [p*stride_bases, (p+1)*stride_bases); the caller moves them with an all-to-all into mhx_dist_recv_buffer;
 * apply: count -> finishes MHX_BUF_FIRST_0_OUT / MHX_BUF_LAST_0_IN; mercy -> installs the local candidate list used by
 * mhx_read2sdbg_add_mercy (which, with a global layout set, updates the adopted is_solid slice). */
enum mhx_route {
  MHX_ROUTE_COUNT_EVENTS = 1,
  MHX_ROUTE_MERCY_CAND = 2,
  /* stage 1 with the knob dist_sparse_marks set (mhx_dist_setup sets it): the global positions of the NON-solid
   * (k+1)-mer occurrences of the owned buckets; apply: the read owner derives its local is_solid from them — nothing of
   * the size of the global read set is allocated or reduced */
  MHX_ROUTE_S1_MARKS = 3
};
int mhx_dist_route_records(mhx_ctx *, int which, uint64_t stride_bases, mhx_dist_items *out, uint64_t *counts);
int mhx_dist_apply_routed(mhx_ctx *, int which, uint64_t n_records);
/* raw device pointer of a result buffer (for collectives on it); NULL if absent */
void *mhx_device_pointer(mhx_ctx *, int which);
/* after the bitmap reduction: install this rank's slice (device pointer, n_words uint64) as the
 * local is_solid used by stage 2 */
int mhx_adopt_is_solid_slice(mhx_ctx *, const void *d_words, uint64_t n_words);

/* ---- multi-GPU behind the C ABI: communicator + collective drivers (megahit_amd/csrc/comm.hip).  One rank per GPU;
 * ranks are threads of one process (mhx_core --gpus N) or separate processes (bench.py --gpus N).  The drivers are
 * COLLECTIVE: every rank calls them with the same arguments.  They replace, for N GPUs, what BaseSequenceSortingEngine::Run
 * does for N OpenMP threads (base_engine.cpp:143-211,318-363): lv1 buckets -> owners, one item all-to-all per stage
 * (RCCL ncclSend/ncclRecv over xGMI, <= 256 MiB per message), position-keyed records routed back to the read owners. ---- */
typedef struct mhx_comm mhx_comm;
#define MHX_COMM_ID_BYTES 128
/* RCCL transport: one rank creates the id and ships its 128 bytes to the others by any means (the launcher's store,
 * a file, a broadcast), then every rank calls mhx_comm_init_rank with the handle of ITS GPU.  n_ranks == 1 with
 * id == NULL gives a trivial communicator without RCCL.  Returns NULL on error. */
int mhx_comm_unique_id(void *id /* MHX_COMM_ID_BYTES */);
mhx_comm *mhx_comm_init_rank(mhx_ctx *, const void *id, int rank, int n_ranks);
/* A communicator whose bytes the CALLER moves, through host memory — e.g. torch.distributed over gloo when the ranks are
   processes without a shared RCCL world (several processes on one GPU; tests/test_gpu_multiprocess.py).  libmhx stages the
   per-peer segments in host memory and calls back; the callbacks return 0 on success.
     all_reduce_u64    in-place element-wise sum (is_max = 0) or max (is_max = 1) of n 64-bit values over all ranks; the max
                       may be computed on the values read as SIGNED 64-bit (libmhx only mixes values of one sign per element)
     all_to_all_bytes  send / recv: the segments for / from rank 0, 1, ... back to back, send_bytes[p] / recv_bytes[p] bytes
                       (0 for the caller's own rank) */
typedef struct mhx_host_transport {
  int (*all_reduce_u64)(void *user, uint64_t *values, uint64_t n, int is_max);
  int (*all_to_all_bytes)(void *user, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes);
  void *user;
} mhx_host_transport;
mhx_comm *mhx_comm_init_hosted(mhx_ctx *, int rank, int n_ranks, const mhx_host_transport *t);
/* in-process transport without RCCL: the n ranks are threads of this process and may share GPUs (tests); exchanges are
 * device-to-device copies between the ranks' buffers.  out receives n handles, out[r] bound to ctxs[r]. */
int mhx_comm_local_group(int n_ranks, mhx_ctx *const *ctxs, mhx_comm **out);
void mhx_comm_destroy(mhx_comm *);
int mhx_comm_rank(const mhx_comm *);
int mhx_comm_size(const mhx_comm *);
int mhx_comm_barrier(mhx_comm *);
/* payload bytes this rank has handed to OTHER ranks through the item / record exchanges since the last reset (what crosses xGMI) */
uint64_t mhx_comm_bytes_sent(mhx_comm *, int reset);
int mhx_comm_all_reduce_u64(mhx_comm *, uint64_t *values, uint64_t n, int is_max /* else sum */);
/* agree on the bucket partition (balance_stage = 0: equal ranges; else an enum mhx_stage whose all-reduced lv1 bucket
 * histogram balances the ranges) and on the global read layout (rank r's bases at r * stride); sets dist_sparse_marks */
int mhx_dist_setup(mhx_ctx *, mhx_comm *, int balance_stage, uint32_t k, uint32_t min_count);
/* read2sdbg over all ranks: afterwards every handle holds the SdBG records of ITS bucket range (MHX_BUF_SDBG_BYTES,
 * MHX_BUF_BUCKET_*: one file of the reference's multi-file .sdbg.<i> format) and MHX_BUF_MUL_HIST of its buckets.
 * need_mercy as want_mercy of mhx_read2sdbg_s1. */
int mhx_dist_read2sdbg(mhx_ctx *, mhx_comm *, uint32_t k, uint32_t min_count, int need_mercy, mhx_s1_result *out1,
                       mhx_sdbg_result *out2, uint64_t *num_mercy);
/* count: edges / bucket counts / histogram of the owned buckets, first_0_out / last_0_in of the local reads */
int mhx_dist_count(mhx_ctx *, mhx_comm *, uint32_t k, uint32_t min_count, mhx_count_result *out);
/* seq2sdbg: the loaded sequences (+ multiplicities) may be spread over the ranks in any way */
/* collective GenMercyEdges (seq_to_sdbg.cpp:171-357) for `seq2sdbg --need_mercy` on several GPUs: the (k+1)-mer edges are
   sharded over the ranks (mhx_load_edges of a contiguous slice each; a rank may hold none), EVERY rank passes ALL candidate
   reads; has_in / has_out are OR-ed over the ranks, each mercy edge is appended on exactly one rank.  *num_mercy = all. */
int mhx_dist_gen_mercy_edges(mhx_ctx *, mhx_comm *, uint32_t k, const uint32_t *cand_packed, uint64_t cand_words, uint64_t n_cand,
                             const uint64_t *cand_start, uint64_t *num_mercy);
int mhx_dist_seq2sdbg(mhx_ctx *, mhx_comm *, uint32_t k, mhx_sdbg_result *out);

/* ---- memory-bounded operation: the reference's lv1 passes (base_engine.cpp:54-141,213-281) ----
 * By default an engine call materialises all its items at once.  For inputs whose items exceed HBM, run the call
 * once per set of lv1 buckets: with a filter set, items are extracted over batches of reads (batch_bytes of staging,
 * 0 = 1 GiB) and only those of the kept buckets are stored, so the working set is 2 x expected_items x item size.
 * keep: 65536 flags (NULL switches the filter off); expected_items: an upper bound of the items in the kept buckets
 * (sum of mhx_bucket_histogram over them).  The outputs of a filtered call cover the kept buckets only (edges / SdBG
 * records and the per-bucket tables); with accumulate != 0 the state that spans buckets continues from the previous
 * call instead of being reset: is_solid marks, multiplicity histogram, mercy candidates and aggregated stage-2 items
 * of read2sdbg stage 1; first_0_out / last_0_in and the histogram of count.  Like the reference, every pass rescans
 * all reads.  Works with the single-GPU calls and with mhx_dist_extract. */
uint64_t mhx_device_free_bytes(mhx_ctx *);  /* free HBM on the handle's device right now (workspaces of this handle included in "used") */
int mhx_bucket_histogram(mhx_ctx *, int stage /* enum mhx_stage */, uint32_t k, uint32_t min_count, uint64_t *hist /* 65536 */);
/* Device bytes a bucket-range pass of `stage` over n_items kept items needs for its items (sort buffers, staging, status
 * words), the stage's fixed state aside — for the sequences loaded now (stage 1 on fixed-length reads makes its records in
 * the first sort pass and needs two 12-byte buffers; the general path three item buffers).  0 on error. */
uint64_t mhx_stage_pass_bytes(mhx_ctx *, int stage, uint32_t k, uint32_t min_count, uint64_t n_items);
int mhx_set_bucket_filter(mhx_ctx *, const uint8_t *keep, uint64_t expected_items, uint64_t batch_bytes, int accumulate);

/* ---- measurement ---- */
typedef struct {
  char name[48];
  uint32_t launches;
  double total_ms;     /* HIP-event time on the engine's stream */
  double algo_bytes;   /* algorithmic bytes moved by those launches */
} mhx_kernel_stat;
/* host seconds this process spent in hipMalloc / hipFree for libmhx's device buffers, bytes and calls of hipMalloc
   (process-wide; any pointer may be NULL).  A process started right behind another GPU process waits there. */
void mhx_alloc_stats(double *malloc_s, double *free_s, uint64_t *bytes, uint64_t *calls);
int mhx_profile_enable(mhx_ctx *, int on);
int mhx_profile_reset(mhx_ctx *);
/* returns number of distinct kernels; fills up to cap entries */
int mhx_profile_get(mhx_ctx *, mhx_kernel_stat *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
