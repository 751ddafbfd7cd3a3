/*
 * oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's SdBG-construction hot path
 * (MEGAHIT v1.2.9, the src/sorting directory).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or run anything in oracle/.
 *
 * Parity status: PINNED — every engine below is checked bit-for-bit (on the
 * bucket-ordered canonical stream) against oracle/_ref/ref_core,
 * the reference's own sources compiled in place by oracle/Makefile
 * (tests/test_oracle_vs_ref.py; fixtures in tests/golden/ were generated from
 * ref_core by tools/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#ifndef MHX_ORACLE_H
#define MHX_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_BUCKETS 65536 /* sorting/base_engine.h:21 */
#define ORC_SENTINEL 4u       /* '$' char, kmer_counter.h:48 */
#define ORC_MAX_MUL 65535     /* sdbg/sdbg_def.h:12 */

/* Packed sequence store, sequence/sequence_package.h:38-320:
 * base i of the concatenation is bits 31-2j..30-2j of word i/16, j=i%16. */
typedef struct {
  uint32_t *words;
  uint64_t n_words_cap;
  uint64_t *start; /* n_seqs+1 */
  uint64_t n_seqs, seq_cap;
} orc_pkg;

void orc_pkg_init(orc_pkg *p);
void orc_pkg_free(orc_pkg *p);
/* append `len` bases from a 2-bit packed source (MSB first); reversed if rev.
 * len==0 fakes a 1-base 'A' sequence (sequence_package.h:275-281). */
void orc_pkg_append_packed(orc_pkg *p, const uint32_t *src, uint32_t len, int rev);
void orc_pkg_append_string(orc_pkg *p, const char *s, uint32_t len, int rev);
static inline unsigned orc_base(const orc_pkg *p, uint64_t i) {
  return (p->words[i >> 4] >> (30 - 2 * (i & 15))) & 3u;
}
/* <prefix>.lib_info + <prefix>.bin  (sequence/io/sequence_lib.cpp:93-118,
 * binary_reader.h:23-53).  Returns 0 on success. */
int orc_load_read_lib(const char *prefix, int reverse, orc_pkg *out);

/* A growable array of fixed-width uint32 records. */
typedef struct {
  uint32_t *d;
  uint64_t n, cap;
  int w;
} orc_vec;

/* how ties between equal keys are ordered inside a bucket */
enum { ORC_TIE_STABLE = 0, ORC_TIE_KMSORT = 1 };

/* ---- count (sorting/kmer_counter.cpp) ---- */
typedef struct {
  int words_per_edge;
  orc_vec edges;                       /* sorted, words_per_edge each */
  int64_t bucket_count[ORC_NUM_BUCKETS]; /* solid edges per bucket */
  uint32_t *first_0_out, *last_0_in;   /* per read */
  int64_t hist[ORC_MAX_MUL + 1];
  int64_t n_items;
} orc_count_out;
int orc_count(const orc_pkg *reads, int k, int m, orc_count_out *out);
void orc_count_free(orc_count_out *o);
/* the halves of orc_count, exposed so that tests can exchange items and read events between ranks */
void orc_count_items(const orc_pkg *reads, int k, uint64_t pos_base, orc_vec *items);
void orc_count_reduce(orc_vec *items /* consumed */, int k, int m, orc_count_out *out /* zeroed; first/last untouched */,
                      uint64_t **events, uint64_t *n_events);
void orc_count_apply_events(const orc_pkg *reads, uint64_t pos_base, const uint64_t *ev, uint64_t n_ev, uint32_t *first_0_out,
                            uint32_t *last_0_in);

/* ---- read2sdbg stage 1 (sorting/read_to_sdbg_s1.cpp) ---- */
typedef struct {
  uint64_t *is_solid; /* ceil(n_bases/64) words, bit i = word i/64 bit i%64 */
  uint64_t n_bits;
  int64_t hist[ORC_MAX_MUL + 1];
  int64_t *mercy; /* packed (abs_offset<<2)|flag, sorted ascending */
  uint64_t n_mercy, mercy_cap;
  int64_t n_items;
} orc_s1_out;
int orc_s1(const orc_pkg *reads, int k, int m, int tie_mode, orc_s1_out *out);
void orc_s1_free(orc_s1_out *o);

void orc_s1_reduce_ex(const orc_pkg *reads, const orc_vec *items, int k, int m, int tie_mode, int mercy_without_reads,
                      orc_s1_out *out);
/* the two halves of orc_s1 / orc_s2, exposed so that tests can exchange items between ranks */
void orc_s1_items(const orc_pkg *reads, int k, uint64_t pos_base, orc_vec *items);
void orc_s1_reduce(const orc_pkg *reads /* may be NULL: no mercy */, const orc_vec *items, int k, int m, int tie_mode,
                   orc_s1_out *out /* is_solid preallocated, zeroed */);

/* mercy block of Read2SdbgS2::Initialize (read_to_sdbg_s2.cpp:122-266);
 * returns "Number mercy". cands must be sorted. */
int64_t orc_s2_add_mercy(const orc_pkg *reads, int k, uint64_t *is_solid,
                         const int64_t *cands, uint64_t n_cands);

/* ---- SdBG output shared by S2 and seq2sdbg ---- */
typedef struct {
  int k, words_per_tip_label;
  uint8_t *bytes; /* concatenation of the buckets' byte streams in bucket-id order */
  uint64_t n_bytes, cap;
  uint64_t bucket_off[ORC_NUM_BUCKETS]; /* starting byte of each bucket */
  uint64_t bucket_items[ORC_NUM_BUCKETS], bucket_tips[ORC_NUM_BUCKETS],
      bucket_large[ORC_NUM_BUCKETS];
  uint64_t w_count[9], ones_in_last;
  int64_t n_sort_items;
} orc_sdbg_out;
void orc_sdbg_free(orc_sdbg_out *o);

/* read_to_sdbg_s2.cpp:271-614; is_solid may be NULL iff m==1 (for_sure_solid) */
int orc_s2(const orc_pkg *reads, int k, int m, const uint64_t *is_solid, orc_sdbg_out *out);

void orc_s2_items(const orc_pkg *reads, int k, int m, const uint64_t *is_solid, orc_vec *items);
/* seq_to_sdbg.cpp:530-789; mult has one entry per sequence */
int orc_seq2sdbg(const orc_pkg *seqs, const uint16_t *mult, int k, orc_sdbg_out *out);
void orc_seq2sdbg_items(const orc_pkg *seqs, const uint16_t *mult, int k, orc_vec *items);

/* sort + postprocess + SdbgWriter of arbitrary lv2 items (frees items->d) */
void orc_sdbg_from_items(orc_vec *items, int k, int is_seq2sdbg, orc_sdbg_out *out);

/* seq_to_sdbg.cpp:100-357 (GenMercyEdges): edges = sorted (k+1)-mer package,
 * cand = candidate reads.  Appends mercy edges to `edges` (and mult 1 to *mult). */
int64_t orc_gen_mercy_edges(orc_pkg *edges, uint16_t **mult, uint64_t *n_mult,
                            const orc_pkg *cand, int k);

/* ---- file formats ---- */
int orc_write_edges(const char *prefix, int k, const orc_count_out *c); /* .edges.0 + .edges.info */
int orc_write_cand(const char *prefix, const orc_pkg *reads, const orc_count_out *c);
int orc_write_counting(const char *prefix, const int64_t *hist);
int orc_write_sdbg(const char *prefix, const orc_sdbg_out *s); /* .sdbg.0 + .sdbg_info */
int orc_write_mercy_cand(const char *prefix, const orc_pkg *reads, const orc_s1_out *s);
/* edge_reader.h / edge_io_meta.h; sorted (any #files) or unsorted */
int orc_read_edges(const char *prefix, orc_pkg *pkg, uint16_t **mult, uint64_t *n_mult, int *k_out);
/* contig_reader.h:52-119 */
int orc_read_contigs(const char *fasta, orc_pkg *pkg, uint16_t **mult, uint64_t *n_mult,
                     unsigned min_len, unsigned k_from, unsigned k_to, int reverse);
/* .bin-format reader used for <prefix>.cand (binary_reader.h) */
int orc_read_bin(const char *file, int reverse, orc_pkg *out);

/* the kmsort restatement (kmlib/kmsort.h:23-122 via kmsort_selector.cpp:13-33);
 * exported for tests */
void orc_sort_items(uint32_t *items, int64_t n, int words, int key_words, int tie_mode);
void orc_kmsort_u64(uint64_t *a, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
