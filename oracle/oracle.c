/*
 * oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Plain-C restatement of MEGAHIT's CX1 SdBG construction (the src/sorting directory).
 * Structure: every engine (1) enumerates its sortable items in the reference's
 * global emission order, (2) stable-partitions them into the 65536 lv1 buckets
 * (top 16 bits of word 0), (3) sorts every bucket with a restatement of
 * kmlib::kmsort (or a stable sort), (4) runs the engine's Lv2Postprocess on the
 * bucket.  This is semantically what BaseSequenceSortingEngine::Run does
 * (sorting/base_engine.cpp:143-211,329-351) without its memory planning.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DIVCEIL(a, b) (((a) + (b)-1) / (b))

/* ------------------------------------------------------------------ */
/* package                                                             */
/* ------------------------------------------------------------------ */
void orc_pkg_init(orc_pkg *p) {
  memset(p, 0, sizeof(*p));
  p->seq_cap = 16;
  p->start = (uint64_t *)calloc(p->seq_cap + 1, sizeof(uint64_t));
  p->n_words_cap = 16;
  p->words = (uint32_t *)calloc(p->n_words_cap, sizeof(uint32_t));
}
void orc_pkg_free(orc_pkg *p) {
  free(p->words);
  free(p->start);
  memset(p, 0, sizeof(*p));
}
static void pkg_reserve(orc_pkg *p, uint64_t more_bases) {
  uint64_t need_words = (p->start[p->n_seqs] + more_bases + 15) / 16 + 2;
  if (need_words > p->n_words_cap) {
    uint64_t nc = p->n_words_cap * 2;
    if (nc < need_words) nc = need_words;
    p->words = (uint32_t *)realloc(p->words, nc * sizeof(uint32_t));
    memset(p->words + p->n_words_cap, 0, (nc - p->n_words_cap) * sizeof(uint32_t));
    p->n_words_cap = nc;
  }
  if (p->n_seqs + 1 > p->seq_cap) {
    p->seq_cap *= 2;
    p->start = (uint64_t *)realloc(p->start, (p->seq_cap + 1) * sizeof(uint64_t));
  }
}
static inline void pkg_push_base(orc_pkg *p, uint64_t pos, unsigned c) {
  p->words[pos >> 4] |= (uint32_t)(c & 3u) << (30 - 2 * (pos & 15));
}
void orc_pkg_append_packed(orc_pkg *p, const uint32_t *src, uint32_t len, int rev) {
  if (len == 0) { /* sequence_package.h:275-281 */
    uint32_t fake = 0;
    orc_pkg_append_packed(p, &fake, 1, 0);
    return;
  }
  pkg_reserve(p, len);
  uint64_t pos = p->start[p->n_seqs];
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t j = rev ? len - 1 - i : i;
    pkg_push_base(p, pos + i, (src[j >> 4] >> (30 - 2 * (j & 15))) & 3u);
  }
  p->start[++p->n_seqs] = pos + len;
}
void orc_pkg_append_string(orc_pkg *p, const char *s, uint32_t len, int rev) {
  if (len == 0) { /* sequence_package.h:262-267 */
    orc_pkg_append_string(p, "A", 1, 0);
    return;
  }
  pkg_reserve(p, len);
  uint64_t pos = p->start[p->n_seqs];
  for (uint32_t i = 0; i < len; ++i) {
    char ch = s[rev ? len - 1 - i : i];
    unsigned c; /* "ACGTNacgtn" -> "0123201232", sequence_package.h:80-82; others 0 */
    switch (ch) {
      case 'C': case 'c': c = 1; break;
      case 'G': case 'g': case 'N': case 'n': c = 2; break;
      case 'T': case 't': c = 3; break;
      default: c = 0;
    }
    pkg_push_base(p, pos + i, c);
  }
  p->start[++p->n_seqs] = pos + len;
}

int orc_read_bin(const char *file, int reverse, orc_pkg *out) {
  FILE *f = fopen(file, "rb");
  if (!f) return -1;
  orc_pkg_init(out);
  uint32_t len, cap = 64, *buf = (uint32_t *)malloc(cap * 4);
  while (fread(&len, 4, 1, f) == 1) {
    uint32_t nw = DIVCEIL(len, 16u);
    if (nw > cap) {
      cap = nw * 2;
      buf = (uint32_t *)realloc(buf, cap * 4);
    }
    if (fread(buf, 4, nw, f) != nw) {
      free(buf);
      fclose(f);
      return -2;
    }
    orc_pkg_append_packed(out, buf, len, reverse);
  }
  free(buf);
  fclose(f);
  return 0;
}

int orc_load_read_lib(const char *prefix, int reverse, orc_pkg *out) {
  char path[4096];
  snprintf(path, sizeof path, "%s.bin", prefix);
  return orc_read_bin(path, reverse, out);
}

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */
static void vec_init(orc_vec *v, int w) {
  v->w = w;
  v->n = 0;
  v->cap = 1024;
  v->d = (uint32_t *)malloc(v->cap * w * sizeof(uint32_t));
}
static uint32_t *vec_push(orc_vec *v) {
  if (v->n == v->cap) {
    v->cap *= 2;
    v->d = (uint32_t *)realloc(v->d, v->cap * v->w * sizeof(uint32_t));
  }
  return v->d + (v->n++) * v->w;
}

/* chars [pos, pos+n) of the package (or their reverse complement) packed MSB
 * first into nw zeroed words: the net effect of CopySubstring / CopySubstringRC
 * (sequence/copy_substr.h:53-101,115-176) including their tail masking. */
static void get_chars(const orc_pkg *p, uint64_t pos, unsigned n, int rc, uint32_t *out, int nw) {
  for (int i = 0; i < nw; ++i) out[i] = 0;
  for (unsigned j = 0; j < n; ++j) {
    unsigned c = rc ? 3u - orc_base(p, pos + n - 1 - j) : orc_base(p, pos + j);
    out[j >> 4] |= c << (30 - 2 * (j & 15));
  }
}
static int cmp_words(const uint32_t *a, const uint32_t *b, int n) {
  for (int i = 0; i < n; ++i) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
static inline unsigned comp_or_sentinel(unsigned c) { return c == ORC_SENTINEL ? ORC_SENTINEL : 3u - c; }

/* id of the sequence holding absolute base offset `off` (GetSeqID, sequence_package.h:144-164) */
static uint64_t seq_of_offset(const orc_pkg *p, uint64_t off) {
  uint64_t lo = 0, hi = p->n_seqs; /* start[lo] <= off < start[hi] */
  while (hi - lo > 1) {
    uint64_t mid = (lo + hi) / 2;
    if (p->start[mid] <= off) lo = mid; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------ */
/* kmsort restatement                                                  */
/* ------------------------------------------------------------------ */
typedef struct {
  int words, key_words, n_bytes;
} sort_ctx;

/* Substr::kth_byte, kmsort_selector.cpp:28-32 */
static inline int item_byte(const sort_ctx *c, const uint32_t *it, int b) {
  return (it[c->key_words - 1 - b / 4] >> ((b % 4) * 8)) & 0xFF;
}
static inline int item_less(const sort_ctx *c, const uint32_t *a, const uint32_t *b) {
  return cmp_words(a, b, c->key_words) < 0; /* Substr::operator<, kmsort_selector.cpp:18-27 */
}
/* insert_sort_core, kmsort.h:23-35 (stable) */
static void km_insertion(const sort_ctx *c, uint32_t *s, int64_t n) {
  uint32_t tmp[64];
  int w = c->words;
  for (int64_t i = 1; i < n; ++i) {
    if (item_less(c, s + i * w, s + (i - 1) * w)) {
      memcpy(tmp, s + i * w, w * 4);
      int64_t j = i;
      do {
        memcpy(s + j * w, s + (j - 1) * w, w * 4);
        --j;
      } while (j > 0 && item_less(c, tmp, s + (j - 1) * w));
      memcpy(s + j * w, tmp, w * 4);
    }
  }
}
/* radix_sort_core, kmsort.h:45-106: one American-flag pass on byte `b`, then
 * recursion on b-1 for bins of more than 64 items, insertion sort for 2..64. */
static void km_radix(const sort_ctx *c, uint32_t *s, int64_t n, int b) {
  int w = c->words;
  int64_t count[256] = {0}, head[257], bin_end;
  int64_t *cursor = head + 1; /* cursor[-1] is valid, as in the reference's last_[] */
  uint32_t hold[64], t2[64];
  for (int64_t i = 0; i < n; ++i) count[item_byte(c, s + i * w, b)]++;
  head[0] = head[1] = 0;
  for (int i = 1; i < 256; ++i) cursor[i] = cursor[i - 1] + count[i - 1];
  for (int i = 0; i < 256; ++i) {
    bin_end = cursor[i - 1] + count[i];
    if (bin_end == n) {
      cursor[i] = n;
      break;
    }
    while (cursor[i] != bin_end) {
      memcpy(hold, s + cursor[i] * w, w * 4);
      int tag = item_byte(c, hold, b);
      if (tag != i) {
        do {
          uint32_t *dst = s + (cursor[tag]++) * w;
          memcpy(t2, dst, w * 4);
          memcpy(dst, hold, w * 4);
          memcpy(hold, t2, w * 4);
        } while ((tag = item_byte(c, hold, b)) != i);
        memcpy(s + cursor[i] * w, hold, w * 4);
      }
      ++cursor[i];
    }
  }
  if (b > 0) {
    for (int i = 0; i < 256; ++i) {
      if (count[i] > 64) km_radix(c, s + cursor[i - 1] * w, cursor[i] - cursor[i - 1], b - 1);
      else if (count[i] > 1) km_insertion(c, s + cursor[i - 1] * w, cursor[i] - cursor[i - 1]);
    }
  }
}
static void stable_merge_sort(const sort_ctx *c, uint32_t *a, uint32_t *tmp, int64_t n) {
  if (n <= 16) {
    km_insertion(c, a, n);
    return;
  }
  int64_t h = n / 2;
  int w = c->words;
  stable_merge_sort(c, a, tmp, h);
  stable_merge_sort(c, a + h * w, tmp, n - h);
  int64_t i = 0, j = h, o = 0;
  while (i < h && j < n) {
    if (item_less(c, a + j * w, a + i * w)) memcpy(tmp + (o++) * w, a + (j++) * w, w * 4);
    else memcpy(tmp + (o++) * w, a + (i++) * w, w * 4);
  }
  while (i < h) memcpy(tmp + (o++) * w, a + (i++) * w, w * 4);
  while (j < n) memcpy(tmp + (o++) * w, a + (j++) * w, w * 4);
  memcpy(a, tmp, n * w * 4);
}
void orc_sort_items(uint32_t *items, int64_t n, int words, int key_words, int tie_mode) {
  sort_ctx c = {words, key_words, 4 * key_words - 2}; /* n_bytes, kmsort_selector.cpp:16-17 */
  assert(words <= 64);
  if (n <= 1) return;
  if (tie_mode == ORC_TIE_KMSORT) { /* radix_sort_entry, kmsort.h:109-122 */
    if (n <= 64) km_insertion(&c, items, n);
    else km_radix(&c, items, n, c.n_bytes - 1);
  } else {
    uint32_t *tmp = (uint32_t *)malloc((size_t)n * words * 4);
    stable_merge_sort(&c, items, tmp, n);
    free(tmp);
  }
}
static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}
void orc_kmsort_u64(uint64_t *a, int64_t n) { qsort(a, (size_t)n, 8, cmp_u64); }

/* stable partition of items by bucket = word0 >> 16; returns bucket starts[65537] */
static uint32_t *bucketize(const orc_vec *v, int64_t *starts) {
  int w = v->w;
  memset(starts, 0, (ORC_NUM_BUCKETS + 1) * sizeof(int64_t));
  for (uint64_t i = 0; i < v->n; ++i) starts[(v->d[i * w] >> 16) + 1]++;
  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) starts[b + 1] += starts[b];
  int64_t *cur = (int64_t *)malloc(ORC_NUM_BUCKETS * sizeof(int64_t));
  memcpy(cur, starts, ORC_NUM_BUCKETS * sizeof(int64_t));
  uint32_t *out = (uint32_t *)malloc((v->n ? v->n : 1) * w * sizeof(uint32_t));
  for (uint64_t i = 0; i < v->n; ++i) {
    int b = v->d[i * w] >> 16;
    memcpy(out + (cur[b]++) * w, v->d + i * w, w * 4);
  }
  free(cur);
  return out;
}

/* ------------------------------------------------------------------ */
/* count  (sorting/kmer_counter.cpp)                                   */
/* ------------------------------------------------------------------ */
/* item enumeration: Lv1FillOffsets (kmer_counter.cpp:158-206) + Lv2ExtractSubString (:208-252).
 * pos_base shifts the recorded positions (multi-rank tests: position in the global read set). */
void orc_count_items(const orc_pkg *reads, int k, uint64_t pos_base, orc_vec *items) {
  const int W = DIVCEIL((k + 1) * 2, 32); /* kmer_counter.cpp:78-79 */
  vec_init(items, W + 2);
  uint32_t e[20], r[20];
  for (uint64_t rid = 0; rid < reads->n_seqs; ++rid) {
    uint64_t st = reads->start[rid];
    uint32_t L = (uint32_t)(reads->start[rid + 1] - st);
    if (L < (uint32_t)k + 1) continue;
    for (uint32_t p = 0; p + k + 1 <= L; ++p) {
      get_chars(reads, st + p, k + 1, 0, e, W);
      get_chars(reads, st + p, k + 1, 1, r, W);
      int strand = cmp_words(r, e, W) < 0; /* rev_edge.cmp(edge) < 0 -> strand 1, :179 */
      unsigned prev = p > 0 ? orc_base(reads, st + p - 1) : ORC_SENTINEL;
      unsigned next = p + k + 1 < L ? orc_base(reads, st + p + k + 1) : ORC_SENTINEL;
      uint64_t full = ((pos_base + st + p) << 1) | (uint64_t)strand;
      uint64_t info;
      uint32_t *it = vec_push(items);
      if (!strand) {
        memcpy(it, e, W * 4);
        info = (full << 6) | (prev << 3) | next;
      } else {
        memcpy(it, r, W * 4);
        info = (full << 6) | (comp_or_sentinel(next) << 3) | comp_or_sentinel(prev);
      }
      it[W] = (uint32_t)(info >> 32); /* DecomposeUint64, utils/utils.h:59-62 */
      it[W + 1] = (uint32_t)info;
    }
  }
}

/* per bucket: sort + Lv2Postprocess (kmer_counter.cpp:254-381) of the given items (consumed).  The
 * first_0_out / last_0_in updates are returned as events ((position << 1) | which; which 0: last_0_in =
 * max(offset), 1: first_0_out = min(offset + 1)) so that a multi-rank test can route them to the rank that
 * holds the read; orc_count applies them directly. */
void orc_count_reduce(orc_vec *items, int k, int m, orc_count_out *out, uint64_t **events, uint64_t *n_events) {
  const int W = DIVCEIL((k + 1) * 2, 32);
  const int IW = W + 2;
  out->words_per_edge = DIVCEIL((k + 1) * 2 + 16, 32); /* kmer_counter.cpp:80-81 */
  vec_init(&out->edges, out->words_per_edge);
  out->n_items = (int64_t)items->n;
  uint64_t ev_cap = 1024, ev_n = 0;
  uint64_t *ev = (uint64_t *)malloc(ev_cap * 8);
  int64_t *starts = (int64_t *)malloc((ORC_NUM_BUCKETS + 1) * sizeof(int64_t));
  uint32_t *sorted = bucketize(items, starts);
  free(items->d);
  items->d = NULL;
  items->n = 0;
  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) {
    int64_t n = starts[b + 1] - starts[b];
    if (!n) continue;
    uint32_t *s = sorted + starts[b] * IW;
    orc_sort_items(s, n, IW, W, ORC_TIE_KMSORT);
    for (int64_t i = 0, to; i < n; i = to) {
      to = i + 1;
      while (to < n && cmp_words(s + i * IW, s + to * IW, W) == 0) ++to;
      int64_t count = to - i, cp[8] = {0}, cn[8] = {0};
      for (int64_t j = i; j < to; ++j) {
        unsigned pn = s[j * IW + W + 1] & 63u;
        cp[pn >> 3]++;
        cn[pn & 7]++;
      }
      int has_in = 0, has_out = 0;
      for (int j = 0; j < 4; ++j) {
        if (cp[j] >= m) has_in = 1;
        if (cn[j] >= m) has_out = 1;
      }
      for (int pass = 0; pass < 2; ++pass) {
        /* pass 0: !has_in (kmer_counter.cpp:307-337); pass 1: !has_out (:339-368) */
        if (count < m || (pass == 0 ? has_in : has_out)) continue;
        for (int64_t j = i; j < to; ++j) {
          uint64_t info = (((uint64_t)s[j * IW + W] << 32) | s[j * IW + W + 1]) >> 6;
          int strand = (int)(info & 1);
          int update_last = (pass == 0) ? (strand == 0) : (strand == 1);
          if (ev_n == ev_cap) ev = (uint64_t *)realloc(ev, (ev_cap *= 2) * 8);
          ev[ev_n++] = ((info >> 1) << 1) | (update_last ? 0u : 1u);
        }
      }
      out->hist[count > ORC_MAX_MUL ? ORC_MAX_MUL : count]++; /* edge_counter.h:30-32 */
      if (count >= m) { /* PackEdge, kmer_counter.cpp:32-52 */
        uint32_t *ed = vec_push(&out->edges);
        for (int x = 0; x < out->words_per_edge; ++x) ed[x] = x < W ? s[i * IW + x] : 0;
        ed[out->words_per_edge - 1] |= (uint32_t)(count > ORC_MAX_MUL ? ORC_MAX_MUL : count);
        out->bucket_count[b]++;
      }
    }
  }
  free(sorted);
  free(starts);
  *events = ev;
  *n_events = ev_n;
}

/* first_0_out / last_0_in of the local reads from events whose positions are relative to pos_base */
void orc_count_apply_events(const orc_pkg *reads, uint64_t pos_base, const uint64_t *ev, uint64_t n_ev, uint32_t *first_0_out,
                            uint32_t *last_0_in) {
  for (uint64_t i = 0; i < n_ev; ++i) {
    uint64_t abs = (ev[i] >> 1) - pos_base;
    uint64_t rid = seq_of_offset(reads, abs);
    uint32_t off = (uint32_t)(abs - reads->start[rid]);
    if (!(ev[i] & 1)) {
      uint32_t old = last_0_in[rid];
      if (old == 0xFFFFFFFFu || old < off) last_0_in[rid] = off;
    } else {
      if (first_0_out[rid] > off + 1) first_0_out[rid] = off + 1;
    }
  }
}

int orc_count(const orc_pkg *reads, int k, int m, orc_count_out *out) {
  memset(out, 0, sizeof(*out));
  uint64_t nr = reads->n_seqs;
  out->first_0_out = (uint32_t *)malloc((nr ? nr : 1) * 4);
  out->last_0_in = (uint32_t *)malloc((nr ? nr : 1) * 4);
  memset(out->first_0_out, 0xFF, (nr ? nr : 1) * 4); /* kmer_counter.cpp:86-89 */
  memset(out->last_0_in, 0xFF, (nr ? nr : 1) * 4);
  orc_vec items;
  orc_count_items(reads, k, 0, &items);
  uint64_t *ev, n_ev;
  orc_count_reduce(&items, k, m, out, &ev, &n_ev);
  orc_count_apply_events(reads, 0, ev, n_ev, out->first_0_out, out->last_0_in);
  free(ev);
  return 0;
}
void orc_count_free(orc_count_out *o) {
  free(o->edges.d);
  free(o->first_0_out);
  free(o->last_0_in);
  memset(o, 0, sizeof(*o));
}

/* ------------------------------------------------------------------ */
/* read2sdbg stage 1  (sorting/read_to_sdbg_s1.cpp)                    */
/* ------------------------------------------------------------------ */
static void s1_push_mercy(orc_s1_out *o, int64_t v) {
  if (o->n_mercy == o->mercy_cap) {
    o->mercy_cap = o->mercy_cap ? o->mercy_cap * 2 : 1024;
    o->mercy = (int64_t *)realloc(o->mercy, o->mercy_cap * 8);
  }
  o->mercy[o->n_mercy++] = v;
}
static void s1_emit(const orc_pkg *reads, orc_vec *items, int W, int k, uint64_t st, uint32_t L,
                    uint32_t q, int strand) {
  /* Lv2ExtractSubString, read_to_sdbg_s1.cpp:298-366 */
  unsigned head = q >= 1 ? orc_base(reads, st + q - 1) : ORC_SENTINEL;
  unsigned prev = q >= 2 ? orc_base(reads, st + q - 2) : ORC_SENTINEL;
  unsigned tail = q + k - 1 < L ? orc_base(reads, st + q + k - 1) : ORC_SENTINEL;
  unsigned next = q + k < L ? orc_base(reads, st + q + k) : ORC_SENTINEL;
  uint64_t full = ((st + q) << 1) | (uint64_t)strand, info;
  uint32_t *it = vec_push(items);
  get_chars(reads, st + q, k - 1, strand, it, W);
  if (!strand) {
    it[W - 1] |= (head << 3) | tail;
    info = (full << 6) | (prev << 3) | next;
  } else {
    it[W - 1] |= (comp_or_sentinel(tail) << 3) | comp_or_sentinel(head);
    info = (full << 6) | (comp_or_sentinel(next) << 3) | comp_or_sentinel(prev);
  }
  it[W] = (uint32_t)(info >> 32);
  it[W + 1] = (uint32_t)info;
}
/* IsDiffKMinusOneMer, read_to_sdbg_s1.cpp:41-64 (same in s2.cpp:44-67, seq_to_sdbg.cpp:45-68) */
static int diff_km1(const uint32_t *a, const uint32_t *b, int k) {
  int chars_last = (k - 1) % 16, full = (k - 1) / 16;
  if (chars_last > 0 && (a[full] >> (16 - chars_last) * 2) != (b[full] >> (16 - chars_last) * 2)) return 1;
  for (int i = full - 1; i >= 0; --i)
    if (a[i] != b[i]) return 1;
  return 0;
}

/* item enumeration of stage 1 (Lv1FillOffsets, read_to_sdbg_s1.cpp:208-296); pos_base is added to
 * every absolute position (0 for the single-process engines; a rank offset in the multi-GPU tests) */
void orc_s1_items(const orc_pkg *reads, int k, uint64_t pos_base, orc_vec *items) {
  const int W = DIVCEIL((k - 1) * 2 + 6, 32); /* read_to_sdbg_s1.cpp:107-108 */
  vec_init(items, W + 2);
  uint32_t f[20], r[20];
  for (uint64_t rid = 0; rid < reads->n_seqs; ++rid) {
    uint64_t st = reads->start[rid];
    uint32_t L = (uint32_t)(reads->start[rid + 1] - st);
    if (L < (uint32_t)k + 1) continue;
    uint64_t first = items->n;
    s1_emit(reads, items, W, k, st, L, 0, 0); /* first (k-1)-mer: both strands, :239-245 */
    s1_emit(reads, items, W, k, st, L, 0, 1);
    for (uint32_t q = 1; q + k - 1 < L; ++q) { /* middles: q = 1 .. L-k */
      get_chars(reads, st + q, k - 1, 0, f, W);
      get_chars(reads, st + q, k - 1, 1, r, W);
      int c = cmp_words(f, r, W);
      int strand;
      if (c > 0) strand = 1;
      else if (c < 0) strand = 0;
      else { /* palindrome, :264-279 */
        unsigned pv = orc_base(reads, st + q - 1), nx = orc_base(reads, st + q + k - 1);
        strand = pv <= 3 - nx ? 0 : 1;
      }
      s1_emit(reads, items, W, k, st, L, q, strand);
    }
    s1_emit(reads, items, W, k, st, L, L - k + 1, 0); /* last one: both strands, :286-292 */
    s1_emit(reads, items, W, k, st, L, L - k + 1, 1);
    if (pos_base) /* shift the positions stored in the aux words */
      for (uint64_t i = first; i < items->n; ++i) {
        uint32_t *it = items->d + i * (W + 2);
        uint64_t info = ((uint64_t)it[W] << 32) | it[W + 1];
        info += pos_base << 7;
        it[W] = (uint32_t)(info >> 32);
        it[W + 1] = (uint32_t)info;
      }
  }
}

/* bucket sort + Lv2Postprocess (read_to_sdbg_s1.cpp:368-555) of arbitrary stage-1 items.  reads may be
 * NULL (then no mercy candidates are produced: positions need not belong to a local package). */
void orc_s1_reduce(const orc_pkg *reads, const orc_vec *items_in, int k, int m, int tie_mode, orc_s1_out *out) {
  orc_s1_reduce_ex(reads, items_in, k, m, tie_mode, 0, out);
}
/* mercy_without_reads: multi-rank tests reduce items whose reads live on another rank; a candidate is
 * ((pkg_offset + offset) << 2) | flag and pkg_offset + offset = position - 1, so the read need not be known. */
void orc_s1_reduce_ex(const orc_pkg *reads, const orc_vec *items_in, int k, int m, int tie_mode, int mercy_without_reads,
                      orc_s1_out *out) {
  const int W = DIVCEIL((k - 1) * 2 + 6, 32);
  const int IW = W + 2;
  out->n_items = (int64_t)items_in->n;
  int64_t *starts = (int64_t *)malloc((ORC_NUM_BUCKETS + 1) * sizeof(int64_t));
  uint32_t *sorted = bucketize(items_in, starts);

  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) {
    int64_t n = starts[b + 1] - starts[b];
    if (!n) continue;
    uint32_t *s = sorted + starts[b] * IW;
    orc_sort_items(s, n, IW, W, tie_mode);
    /* Lv2Postprocess, read_to_sdbg_s1.cpp:368-555 */
    int64_t end_idx;
    for (int64_t i = 0; i < n; i = end_idx) {
      int64_t cph[5][5], ctn[5][5], cht[64];
      memset(cph, 0, sizeof cph);
      memset(ctn, 0, sizeof ctn);
      memset(cht, 0, sizeof cht);
      const uint32_t *first = s + i * IW;
      /* NOTE (H1): prev/next always come from the FIRST item of the group, :399 */
      unsigned pn_first = first[W + 1] & 63u;
      end_idx = i;
      while (end_idx < n && (end_idx == i || !diff_km1(first, s + end_idx * IW, k))) {
        unsigned ht = s[end_idx * IW + W - 1] & 63u;
        cph[pn_first >> 3][ht >> 3]++;
        ctn[ht & 7][pn_first & 7]++;
        cht[ht]++;
        ++end_idx;
      }
      int has_in = 0, has_out = 0, l_has_out = 0, r_has_in = 0;
      for (int j = 0; j < 4; ++j) {
        for (int x = 0; x < 4; ++x)
          if (cph[x][j] >= m) { has_in |= 1 << j; break; }
        for (int x = 0; x < 4; ++x)
          if (ctn[j][x] >= m) { has_out |= 1 << j; break; }
      }
      for (int j = 0; j < 4; ++j)
        for (int x = 0; x < 4; ++x)
          if (cht[(j << 3) | x] >= m) { l_has_out |= 1 << j; r_has_in |= 1 << x; }

      int64_t idx = i;
      while (idx < end_idx) {
        unsigned ht = s[idx * IW + W - 1] & 63u, head = ht >> 3, tail = ht & 7;
        int both = head != ORC_SENTINEL && tail != ORC_SENTINEL;
        if (both) out->hist[cht[ht] > ORC_MAX_MUL ? ORC_MAX_MUL : cht[ht]]++;
        int solid = both && cht[ht] >= m;
        for (int64_t j = 0; j < cht[ht]; ++j, ++idx) {
          uint64_t info = (((uint64_t)s[idx * IW + W] << 32) | s[idx * IW + W + 1]) >> 6;
          uint64_t abs = info >> 1;
          int strand = (int)(info & 1);
          if (solid) out->is_solid[(abs - 1) >> 6] |= 1ull << ((abs - 1) & 63); /* :464 */
          if (!reads && !mercy_without_reads) continue;
          int64_t base = 0;
          if (reads) base = (int64_t)reads->start[seq_of_offset(reads, abs)];
          int64_t off = (int64_t)abs - base - 1;
          int64_t l_off = strand == 0 ? off : off + 1, r_off = strand == 0 ? off + 1 : off;
          if (solid) {
            if (!(has_in & (1 << head))) s1_push_mercy(out, ((base + l_off) << 2) | (1 + strand));
            if (!(has_out & (1 << tail))) s1_push_mercy(out, ((base + r_off) << 2) | (2 - strand));
          } else { /* :485-551 */
            if (l_has_out & (1 << head)) {
              if (has_in & (1 << head)) s1_push_mercy(out, ((base + l_off) << 2) | 0);
              else s1_push_mercy(out, ((base + l_off) << 2) | (1 + strand));
            } else if (has_in & (1 << head)) {
              s1_push_mercy(out, ((base + l_off) << 2) | (2 - strand));
            }
            if (r_has_in & (1 << tail)) {
              if (has_out & (1 << tail)) s1_push_mercy(out, ((base + r_off) << 2) | 0);
              else s1_push_mercy(out, ((base + r_off) << 2) | (2 - strand));
            } else if (has_out & (1 << tail)) {
              s1_push_mercy(out, ((base + r_off) << 2) | (1 + strand));
            }
          }
        }
      }
    }
  }
  free(sorted);
  free(starts);
  orc_kmsort_u64((uint64_t *)out->mercy, (int64_t)out->n_mercy);
}

int orc_s1(const orc_pkg *reads, int k, int m, int tie_mode, orc_s1_out *out) {
  memset(out, 0, sizeof(*out));
  out->n_bits = reads->start[reads->n_seqs];
  out->is_solid = (uint64_t *)calloc(DIVCEIL(out->n_bits, 64) + 1, 8);
  orc_vec items;
  orc_s1_items(reads, k, 0, &items);
  orc_s1_reduce(reads, &items, k, m, tie_mode, out);
  free(items.d);
  return 0;
}
void orc_s1_free(orc_s1_out *o) {
  free(o->is_solid);
  free(o->mercy);
  memset(o, 0, sizeof(*o));
}

/* read_to_sdbg_s2.cpp:122-266 */
int64_t orc_s2_add_mercy(const orc_pkg *reads, int k, uint64_t *is_solid, const int64_t *cands,
                         uint64_t n) {
  int64_t num_mercy = 0;
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < reads->n_seqs; ++i) {
    uint32_t L = (uint32_t)(reads->start[i + 1] - reads->start[i]);
    if (L > max_len) max_len = L;
  }
  unsigned char *no_in = (unsigned char *)malloc(max_len + 2), *no_out = (unsigned char *)malloc(max_len + 2),
                *has_solid = (unsigned char *)malloc(max_len + 2);
  uint64_t idx = 0;
  while (idx < n) {
    uint64_t rid = seq_of_offset(reads, (uint64_t)cands[idx] >> 2);
    uint64_t base = reads->start[rid];
    uint32_t L = (uint32_t)(reads->start[rid + 1] - base);
    int first_0_out = (int)max_len + 1, last_0_in = -1;
    memset(no_in, 0, max_len + 2);
    memset(no_out, 0, max_len + 2);
    memset(has_solid, 0, max_len + 2);
    while (idx < n && ((uint64_t)cands[idx] >> 2) < reads->start[rid + 1]) {
      int off = (int)(((uint64_t)cands[idx] >> 2) - base);
      if ((cands[idx] & 3) == 2) {
        no_out[off] = 1;
        if (off < first_0_out) first_0_out = off;
      } else if ((cands[idx] & 3) == 1) {
        no_in[off] = 1;
        if (off > last_0_in) last_0_in = off;
      }
      has_solid[off] = 1;
      ++idx;
    }
    if (last_0_in < first_0_out) continue;
    int last_no_out = -1;
    for (uint32_t i = 0; i + k < L; ++i)
      if ((is_solid[(base + i) >> 6] >> ((base + i) & 63)) & 1) has_solid[i] = has_solid[i + 1] = 1;
    for (uint32_t i = 0; i + k <= L; ++i) {
      if (no_in[i] && last_no_out != -1) {
        for (uint32_t j = (uint32_t)last_no_out; j < i; ++j) is_solid[(base + j) >> 6] |= 1ull << ((base + j) & 63);
        num_mercy += i - last_no_out;
      }
      if (has_solid[i]) last_no_out = -1;
      if (no_out[i]) last_no_out = (int)i;
    }
  }
  free(no_in);
  free(no_out);
  free(has_solid);
  return num_mercy;
}

/* ------------------------------------------------------------------ */
/* SdBG emission shared by S2 and seq2sdbg                             */
/* ------------------------------------------------------------------ */
static void sdbg_put(orc_sdbg_out *o, const void *src, size_t nbytes) {
  if (o->n_bytes + nbytes > o->cap) {
    o->cap = o->cap ? o->cap * 2 : 4096;
    if (o->cap < o->n_bytes + nbytes) o->cap = o->n_bytes + nbytes;
    o->bytes = (uint8_t *)realloc(o->bytes, o->cap);
  }
  memcpy(o->bytes + o->n_bytes, src, nbytes);
  o->n_bytes += nbytes;
}
/* SdbgWriter::Write, sdbg/sdbg_writer.cpp:25-58; SdbgItem, sdbg/sdbg_item.h:14-24 */
static void sdbg_write(orc_sdbg_out *o, int bucket, int w, int last, int tip, int mul, const uint32_t *label) {
  uint8_t rec[2];
  rec[0] = (uint8_t)(w | (last << 4) | (tip << 5));
  rec[1] = (uint8_t)(mul > 255 ? 255 : mul); /* min(mul, kSmallMulSentinel) */
  sdbg_put(o, rec, 2);
  o->bucket_items[bucket]++;
  o->w_count[w]++;
  o->ones_in_last += last;
  if (mul > 254) { /* > kMaxSmallMul */
    uint16_t m16 = (uint16_t)mul;
    sdbg_put(o, &m16, 2);
    o->bucket_large[bucket]++;
  }
  if (tip) {
    sdbg_put(o, label, 4 * o->words_per_tip_label);
    o->bucket_tips[bucket]++;
  }
}
void orc_sdbg_free(orc_sdbg_out *o) {
  free(o->bytes);
  memset(o, 0, sizeof(*o));
}

/* Lv2Postprocess of S2 (read_to_sdbg_s2.cpp:521-614) and SeqToSdbg (seq_to_sdbg.cpp:702-789).
 * flag_shift = 3 for S2, 19 for seq2sdbg (Extract_a / Extract_b / ExtractCounting). */
static void sdbg_postprocess(orc_sdbg_out *o, int bucket, uint32_t *s, int64_t n, int W, int k, int is_seq2sdbg) {
  const int bshift = is_seq2sdbg ? 16 : 0, fshift = bshift + 3;
  const int aw = (k - 1) / 16, ai = (k - 1) % 16;
  int64_t end_idx;
#define EX_A(it) ((((it)[W - 1] >> fshift) & 1) ? (int)(((it)[aw] >> (15 - ai) * 2) & 3) : 4)
#define EX_B(it) ((int)(((it)[W - 1] >> bshift) & 7))
  for (int64_t start = 0; start < n; start = end_idx) {
    end_idx = start + 1;
    while (end_idx < n && !diff_km1(s + start * W, s + end_idx * W, k)) ++end_idx;
    int has_solid_a = 0, has_solid_b = 0, outputed_b = 0;
    int64_t last_a[4] = {-1, -1, -1, -1};
    for (int64_t i = start; i < end_idx; ++i) {
      int a = EX_A(s + i * W), b = EX_B(s + i * W);
      if (a != 4 && b != 4) {
        has_solid_a |= 1 << a;
        has_solid_b |= 1 << b;
      }
      if (a != 4 && (b != 4 || !(has_solid_a & (1 << a)))) last_a[a] = i;
    }
    for (int64_t i = start, j; i < end_idx; i = j) {
      const uint32_t *cur = s + i * W;
      int a = EX_A(cur), b = EX_B(cur);
      j = i + 1;
      while (j < end_idx && EX_A(s + j * W) == a && EX_B(s + j * W) == b) ++j;
      int is_dollar = 0;
      if (a == 4) {
        if (has_solid_b & (1 << b)) continue;
        is_dollar = 1;
      }
      if (b == 4) {
        if (has_solid_a & (1 << a)) continue;
      }
      int w = b == 4 ? 0 : ((outputed_b & (1 << b)) ? b + 5 : b + 1);
      int last = a == 4 ? 0 : (last_a[a] == j - 1);
      outputed_b |= 1 << b;
      int mul;
      if (is_seq2sdbg) mul = ORC_MAX_MUL - (int)(cur[W - 1] & 0xFFFF); /* seq_to_sdbg.cpp:782-785 */
      else mul = (int)(j - i > ORC_MAX_MUL ? ORC_MAX_MUL : j - i);     /* read_to_sdbg_s2.cpp:579 */
      sdbg_write(o, bucket, w, last, is_dollar, mul, cur);
    }
  }
#undef EX_A
#undef EX_B
}

static void sdbg_run_buckets(orc_sdbg_out *out, orc_vec *items, int W, int k, int is_seq2sdbg) {
  int64_t *starts = (int64_t *)malloc((ORC_NUM_BUCKETS + 1) * sizeof(int64_t));
  uint32_t *sorted = bucketize(items, starts);
  out->n_sort_items = (int64_t)items->n;
  free(items->d);
  items->d = NULL;
  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) {
    int64_t n = starts[b + 1] - starts[b];
    out->bucket_off[b] = out->n_bytes;
    if (!n) continue;
    uint32_t *s = sorted + starts[b] * W;
    orc_sort_items(s, n, W, W, ORC_TIE_KMSORT);
    sdbg_postprocess(out, b, s, n, W, k, is_seq2sdbg);
  }
  free(sorted);
  free(starts);
}

/* ------------------------------------------------------------------ */
/* read2sdbg stage 2  (sorting/read_to_sdbg_s2.cpp)                    */
/* ------------------------------------------------------------------ */
static void s2_emit(const orc_pkg *reads, orc_vec *items, int W, int k, uint64_t st, uint32_t p, int strand, int type) {
  /* Lv2ExtractSubString, read_to_sdbg_s2.cpp:442-519 */
  uint32_t *it = vec_push(items);
  unsigned n = k, prev = ORC_SENTINEL;
  uint64_t off = st + p;
  if (!strand) {
    if (type == 1) { prev = orc_base(reads, off); off += 1; }
    else if (type == 2) { prev = orc_base(reads, off + 1); off += 2; n--; }
  } else {
    if (type == 0) { n--; prev = 3 - orc_base(reads, off + k - 1); }
    else if (type == 1) { prev = 3 - orc_base(reads, off + k); }
    else off += 1;
  }
  get_chars(reads, off, n, strand, it, W);
  it[W - 1] |= (uint32_t)(n == (unsigned)k) << 3;
  it[W - 1] |= prev;
}
/* Lv1FillOffsets, read_to_sdbg_s2.cpp:347-440 */
void orc_s2_items(const orc_pkg *reads, int k, int m, const uint64_t *is_solid, orc_vec *items) {
  const int W = DIVCEIL(k * 2 + 4, 32); /* read_to_sdbg_s2.cpp:98-99 */
  const int EW = DIVCEIL((k + 1) * 2, 32);
  const int sure = (m == 1);
  vec_init(items, W);
  uint32_t e[20], r[20];
#define SOLID(x) (sure || ((is_solid[(x) >> 6] >> ((x)&63)) & 1))
  for (uint64_t rid = 0; rid < reads->n_seqs; ++rid) {
    uint64_t st = reads->start[rid];
    uint32_t L = (uint32_t)(reads->start[rid + 1] - st);
    if (L < (uint32_t)k + 1) continue;
    for (uint32_t p = 0; p + k + 1 <= L; ++p) {
      uint64_t fo = st + p;
      if (!SOLID(fo)) continue;
      get_chars(reads, fo, k + 1, 0, e, EW);
      get_chars(reads, fo, k + 1, 1, r, EW);
      int pal = cmp_words(r, e, EW) == 0;
      if (p == 0 || !SOLID(fo - 1)) {
        s2_emit(reads, items, W, k, st, p, 0, 0);
        if (!pal) s2_emit(reads, items, W, k, st, p, 1, 0);
      }
      s2_emit(reads, items, W, k, st, p, 0, 1);
      if (!pal) s2_emit(reads, items, W, k, st, p, 1, 1);
      if (p + k + 1 == L || !SOLID(fo + 1)) {
        s2_emit(reads, items, W, k, st, p, 0, 2);
        if (!pal) s2_emit(reads, items, W, k, st, p, 1, 2);
      }
    }
  }
#undef SOLID
}
/* bucket sort + Lv2Postprocess + SdbgWriter of arbitrary lv2 items (consumes items->d) */
void orc_sdbg_from_items(orc_vec *items, int k, int is_seq2sdbg, orc_sdbg_out *out) {
  memset(out, 0, sizeof(*out));
  out->k = k;
  out->words_per_tip_label = DIVCEIL(k, 16);
  sdbg_run_buckets(out, items, items->w, k, is_seq2sdbg);
}
int orc_s2(const orc_pkg *reads, int k, int m, const uint64_t *is_solid, orc_sdbg_out *out) {
  orc_vec items;
  orc_s2_items(reads, k, m, is_solid, &items);
  orc_sdbg_from_items(&items, k, 0, out);
  return 0;
}

/* ------------------------------------------------------------------ */
/* seq2sdbg  (sorting/seq_to_sdbg.cpp)                                 */
/* ------------------------------------------------------------------ */
void orc_seq2sdbg_items(const orc_pkg *seqs, const uint16_t *mult, int k, orc_vec *items_out) {
  const int W = DIVCEIL(k * 2 + 3 + 1 + 16, 32); /* seq_to_sdbg.cpp:511-513 */
  orc_vec items;
  vec_init(&items, W);
  /* Lv1FillOffsets (seq_to_sdbg.cpp:579-628) + Lv2ExtractSubString (:630-700) */
  for (uint64_t sid = 0; sid < seqs->n_seqs; ++sid) {
    uint64_t st = seqs->start[sid];
    int64_t n = (int64_t)(seqs->start[sid + 1] - st);
    if (n < k + 1) continue;
    for (int64_t o = 0; o + k - 1 <= n; ++o) {
      for (int strand = 0; strand < 2; ++strand) {
        uint32_t *it = vec_push(&items);
        unsigned nc = (unsigned)k - (o + k > n);
        int counting = (o > 0 && o + k <= n) ? mult[sid] : 0;
        unsigned prev;
        if (!strand) {
          prev = o == 0 ? ORC_SENTINEL : orc_base(seqs, st + o - 1);
          get_chars(seqs, st + o, nc, 0, it, W);
        } else {
          prev = o == 0 ? ORC_SENTINEL : 3 - orc_base(seqs, st + n - o);
          int64_t off = n - 1 - o - (k - 1);
          if (off < 0) off = 0;
          get_chars(seqs, st + off, nc, 1, it, W);
        }
        it[W - 1] |= (uint32_t)(nc == (unsigned)k) << 19;
        it[W - 1] |= prev << 16;
        it[W - 1] |= (uint32_t)(ORC_MAX_MUL - counting > 0 ? ORC_MAX_MUL - counting : 0);
      }
    }
  }
  *items_out = items;
}

int orc_seq2sdbg(const orc_pkg *seqs, const uint16_t *mult, int k, orc_sdbg_out *out) {
  memset(out, 0, sizeof(*out));
  out->k = k;
  out->words_per_tip_label = DIVCEIL(k, 16);
  const int W = DIVCEIL(k * 2 + 3 + 1 + 16, 32);
  orc_vec items;
  orc_seq2sdbg_items(seqs, mult, k, &items);
  sdbg_run_buckets(out, &items, W, k, 1);
  return 0;
}

/* ------------------------------------------------------------------ */
/* mercy edges for seq2sdbg  (seq_to_sdbg.cpp:100-357)                 */
/* ------------------------------------------------------------------ */
/* chars [0,n) of sequence `sid` compared with query q (n chars, MSB-first words) */
static int cmp_edge_prefix(const orc_pkg *e, uint64_t sid, const uint32_t *q, int n) {
  uint64_t st = e->start[sid];
  for (int j = 0; j < n; ++j) {
    unsigned a = (q[j >> 4] >> (30 - 2 * (j & 15))) & 3u, b = orc_base(e, st + j);
    if (a != b) return a < b ? -1 : 1;
  }
  return 0;
}
/* BinarySearchKmer (seq_to_sdbg.cpp:132-161) without the 12-base lookup table
 * (the table only narrows [l, r]); searches the first n_sorted sequences. */
static int64_t find_edge(const orc_pkg *e, int64_t n_sorted, const uint32_t *q, int n) {
  int64_t l = 0, r = n_sorted - 1;
  while (l <= r) {
    int64_t mid = (l + r) / 2;
    int c = cmp_edge_prefix(e, (uint64_t)mid, q, n);
    if (c > 0) l = mid + 1;
    else if (c < 0) r = mid - 1;
    else return mid;
  }
  return -1;
}
static void set_char(uint32_t *w, int idx, unsigned c) {
  w[idx >> 4] = (w[idx >> 4] & ~(3u << (30 - 2 * (idx & 15)))) | (c << (30 - 2 * (idx & 15)));
}
/* The reference prunes its (k+1)-mer probes with comparisons against the
 * reverse complement (seq_to_sdbg.cpp:233-247,270-297); on a sorted list of
 * canonical edges (the only input GenMercyEdges ever sees: `count` output) the
 * pruned probes cannot hit, so the outcome equals the un-pruned statement
 *   has_in[i]  <=> some (k+1)-mer X.kmer_i or its rc is in the list,
 *   has_out[i] <=> some (k+1)-mer kmer_i.Y or its rc is in the list. */
int64_t orc_gen_mercy_edges(orc_pkg *edges, uint16_t **mult, uint64_t *n_mult, const orc_pkg *cand, int k) {
  int64_t n_sorted = (int64_t)edges->n_seqs, num_mercy = 0;
  const int W = DIVCEIL(k + 1, 16) + 1;
  uint32_t q[20], rq[20];
  for (uint64_t rid = 0; rid < cand->n_seqs; ++rid) {
    uint64_t st = cand->start[rid];
    uint32_t L = (uint32_t)(cand->start[rid + 1] - st);
    if (L < (uint32_t)k + 2) continue;
    unsigned char *has_in = (unsigned char *)calloc(L + 2, 1), *has_out = (unsigned char *)calloc(L + 2, 1);
    for (uint32_t i = 0; i + k <= L; ++i) {
      /* incoming: rc(kmer) as a k-prefix, or c.kmer as a (k+1)-mer */
      get_chars(cand, st + i, k, 1, rq, W);
      if (find_edge(edges, n_sorted, rq, k) != -1) has_in[i] = 1;
      else {
        get_chars(cand, st + i, k, 0, q, W);
        for (int j = k; j > 0; --j) set_char(q, j, (q[(j - 1) >> 4] >> (30 - 2 * ((j - 1) & 15))) & 3u);
        for (unsigned c = 0; c < 4 && !has_in[i]; ++c) {
          set_char(q, 0, c);
          if (find_edge(edges, n_sorted, q, k + 1) != -1) has_in[i] = 1;
        }
      }
      /* outgoing: kmer as a k-prefix, or c.rc(kmer) as a (k+1)-mer */
      get_chars(cand, st + i, k, 0, q, W);
      if (find_edge(edges, n_sorted, q, k) != -1) has_out[i] = 1;
      else {
        for (int j = k; j > 0; --j) set_char(rq, j, (rq[(j - 1) >> 4] >> (30 - 2 * ((j - 1) & 15))) & 3u);
        for (unsigned c = 0; c < 4 && !has_out[i]; ++c) {
          set_char(rq, 0, c);
          if (find_edge(edges, n_sorted, rq, k + 1) != -1) has_out[i] = 1;
        }
      }
    }
    int last_no_out = -1; /* seq_to_sdbg.cpp:310-345 */
    for (uint32_t i = 0; i + k <= L; ++i) {
      int state = has_in[i] | (has_out[i] << 1);
      if (state == 1) last_no_out = (int)i;
      else if (state == 2) {
        if (last_no_out >= 0) {
          for (uint32_t j = (uint32_t)last_no_out; j < i; ++j) {
            get_chars(cand, st + j, k + 1, 0, q, W);
            orc_pkg_append_packed(edges, q, k + 1, 0);
          }
          num_mercy += i - last_no_out;
        }
        last_no_out = -1;
      } else if (state == 3) last_no_out = -1;
    }
    free(has_in);
    free(has_out);
  }
  *mult = (uint16_t *)realloc(*mult, (*n_mult + num_mercy + 1) * 2);
  for (int64_t i = 0; i < num_mercy; ++i) (*mult)[(*n_mult)++] = 1; /* :353 */
  return num_mercy;
}

/* ------------------------------------------------------------------ */
/* file formats                                                        */
/* ------------------------------------------------------------------ */
/* EdgeWriter (sequence/io/edge/edge_writer.h:17-111) + EdgeIoMetadata::Serialize
 * (edge_io_meta.h:25-44), one file, buckets in id order. */
int orc_write_edges(const char *prefix, int k, const orc_count_out *c) {
  char path[4096];
  snprintf(path, sizeof path, "%s.edges.0", prefix);
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  fwrite(c->edges.d, 4, c->edges.n * c->words_per_edge, f);
  fclose(f);
  snprintf(path, sizeof path, "%s.edges.info", prefix);
  f = fopen(path, "w");
  if (!f) return -1;
  fprintf(f, "kmer_size %d\nwords_per_edge %d\nnum_files 1\nnum_buckets %d\nnum_edges %lld\nis_sorted 1\n", k,
          c->words_per_edge, ORC_NUM_BUCKETS, (long long)c->edges.n);
  int64_t off = 0;
  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) {
    if (c->bucket_count[b]) fprintf(f, "%d 0 %lld %lld\n", b, (long long)off, (long long)c->bucket_count[b]);
    else fprintf(f, "%d -1 0 0\n", b);
    off += c->bucket_count[b];
  }
  fclose(f);
  return 0;
}
/* KmerCounter::Lv0Postprocess, kmer_counter.cpp:383-403 + SeqPackage::WriteSequences */
int orc_write_cand(const char *prefix, const orc_pkg *reads, const orc_count_out *c) {
  char path[4096];
  snprintf(path, sizeof path, "%s.cand", prefix);
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  for (uint64_t i = 0; i < reads->n_seqs; ++i) {
    uint32_t first = c->first_0_out[i], last = c->last_0_in[i];
    if (first != 0xFFFFFFFFu && last != 0xFFFFFFFFu && last > first) {
      uint32_t len = (uint32_t)(reads->start[i + 1] - reads->start[i]), nw = DIVCEIL(len, 16u);
      uint32_t *buf = (uint32_t *)calloc(nw + 1, 4);
      get_chars(reads, reads->start[i], len, 0, buf, (int)nw);
      fwrite(&len, 4, 1, f);
      fwrite(buf, 4, nw, f);
      free(buf);
    }
  }
  fclose(f);
  return 0;
}
/* EdgeMultiplicityRecorder::DumpStat, edge_counter.h:44-52 */
int orc_write_counting(const char *prefix, const int64_t *hist) {
  char path[4096];
  snprintf(path, sizeof path, "%s.counting", prefix);
  FILE *f = fopen(path, "w");
  if (!f) return -1;
  for (int i = 1; i <= ORC_MAX_MUL; ++i) fprintf(f, "%d %lld\n", i, (long long)hist[i]);
  fclose(f);
  return 0;
}
/* SdbgMeta::Serialize, sdbg/sdbg_meta.cpp:51-61; real buckets first (sorted by
 * file, offset), null buckets after (sdbg_meta.cpp:44-49) */
int orc_write_sdbg(const char *prefix, const orc_sdbg_out *s) {
  char path[4096];
  snprintf(path, sizeof path, "%s.sdbg.0", prefix);
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  fwrite(s->bytes, 1, s->n_bytes, f);
  fclose(f);
  snprintf(path, sizeof path, "%s.sdbg_info", prefix);
  f = fopen(path, "w");
  if (!f) return -1;
  int any = 0, n_null = 0;
  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) any |= s->bucket_items[b] != 0;
  fprintf(f, "k %d\nwords_per_tip_label %d\nnum_buckets %d\nnum_files %d\n", s->k, s->words_per_tip_label,
          ORC_NUM_BUCKETS, any ? 1 : 0);
  for (int b = 0; b < ORC_NUM_BUCKETS; ++b) {
    if (!s->bucket_items[b]) { n_null++; continue; }
    fprintf(f, "%d 0 %llu %llu %llu %llu\n", b, (unsigned long long)s->bucket_off[b],
            (unsigned long long)s->bucket_items[b], (unsigned long long)s->bucket_tips[b],
            (unsigned long long)s->bucket_large[b]);
  }
  for (int i = 0; i < n_null; ++i) fprintf(f, "18446744073709551615 18446744073709551615 0 0 0 0\n");
  fclose(f);
  return 0;
}
/* read_to_sdbg_s1.cpp:116-124,466-551: file = read_id & (n_files-1) */
int orc_write_mercy_cand(const char *prefix, const orc_pkg *reads, const orc_s1_out *s) {
  int nf = 1;
  while (nf * 10485760LL < (int64_t)reads->n_seqs && nf < 64) nf <<= 1;
  FILE *fs[64];
  char path[4096];
  for (int i = 0; i < nf; ++i) {
    snprintf(path, sizeof path, "%s.mercy_cand.%d", prefix, i);
    fs[i] = fopen(path, "wb");
    if (!fs[i]) return -1;
  }
  for (uint64_t i = 0; i < s->n_mercy; ++i) {
    uint64_t rid = seq_of_offset(reads, (uint64_t)s->mercy[i] >> 2);
    fwrite(&s->mercy[i], 8, 1, fs[rid & (uint64_t)(nf - 1)]);
  }
  for (int i = 0; i < nf; ++i) fclose(fs[i]);
  return nf;
}

static int scan_field(FILE *f, const char *name, long long *v) {
  char buf[256];
  if (fscanf(f, "%255s %lld", buf, v) != 2 || strcmp(buf, name)) return -1;
  return 0;
}
/* EdgeReader (sequence/io/edge/edge_reader.h:105-158), EdgeIoMetadata::Deserialize
 * (edge_io_meta.h:46-66): sorted files are read in bucket-id order */
int orc_read_edges(const char *prefix, orc_pkg *pkg, uint16_t **mult, uint64_t *n_mult, int *k_out) {
  char path[4096];
  snprintf(path, sizeof path, "%s.edges.info", prefix);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  long long k, wpe, nfiles, nb, nedges, sorted;
  if (scan_field(f, "kmer_size", &k) || scan_field(f, "words_per_edge", &wpe) || scan_field(f, "num_files", &nfiles) ||
      scan_field(f, "num_buckets", &nb) || scan_field(f, "num_edges", &nedges) || scan_field(f, "is_sorted", &sorted)) {
    fclose(f);
    return -2;
  }
  *k_out = (int)k;
  FILE **fs = (FILE **)calloc((size_t)nfiles + 1, sizeof(FILE *));
  for (long long i = 0; i < nfiles; ++i) {
    snprintf(path, sizeof path, "%s.edges.%lld", prefix, i);
    fs[i] = fopen(path, "rb");
    if (!fs[i]) return -3;
  }
  *mult = (uint16_t *)realloc(*mult, (*n_mult + (uint64_t)nedges + 1) * 2);
  uint32_t *buf = (uint32_t *)malloc((size_t)wpe * 4);
  if (sorted) {
    for (long long b = 0; b < nb; ++b) {
      long long id, fid, off, cnt;
      if (fscanf(f, "%lld %lld %lld %lld", &id, &fid, &off, &cnt) != 4 || id != b) return -4;
      if (fid < 0) continue;
      fseek(fs[fid], off * wpe * 4, SEEK_SET);
      for (long long i = 0; i < cnt; ++i) {
        if (fread(buf, 4, (size_t)wpe, fs[fid]) != (size_t)wpe) return -5;
        orc_pkg_append_packed(pkg, buf, (uint32_t)k + 1, 0);
        (*mult)[(*n_mult)++] = (uint16_t)(buf[wpe - 1] & 0xFFFF);
      }
    }
  } else {
    for (long long i = 0; i < nedges; ++i) {
      if (fread(buf, 4, (size_t)wpe, fs[0]) != (size_t)wpe) return -5;
      orc_pkg_append_packed(pkg, buf, (uint32_t)k + 1, 0);
      (*mult)[(*n_mult)++] = (uint16_t)(buf[wpe - 1] & 0xFFFF);
    }
  }
  free(buf);
  for (long long i = 0; i < nfiles; ++i) fclose(fs[i]);
  free(fs);
  fclose(f);
  return 0;
}

/* ContigReader::ReadWithMultiplicity (sequence/io/contig/contig_reader.h:52-119)
 * over a plain FASTA file: header ">name comment", comment = "flag=F multi=M.MMMM ...".
 * (kseq semantics: name = up to first whitespace, comment = rest of the line.) */
int orc_read_contigs(const char *fasta, orc_pkg *pkg, uint16_t **mult, uint64_t *n_mult, unsigned min_len,
                     unsigned k_from, unsigned k_to, int reverse) {
  FILE *f = fopen(fasta, "r");
  if (!f) return -1;
  int extend_loop = k_from < k_to;
  size_t cap = 1 << 16, len = 0, lcap = 0;
  char *seq = (char *)malloc(cap), *line = NULL, comment[512] = "";
  ssize_t n;
  int have = 0, eof = 0;
  int64_t n_read = 0;
  while (!eof) {
    n = getline(&line, &lcap, f);
    if (n < 0) eof = 1;
    if (eof || line[0] == '>') {
      if (have && len >= min_len) {
        unsigned flag = (unsigned)(comment[5] - '0');
        int skip = 0;
        if (extend_loop && (flag & 2u)) { /* contig_flag::kLoop */
          if (len < k_to + 1u) skip = 1;
          else {
            if (len + (k_to - k_from) + 1 > cap) { cap = (len + k_to) * 2; seq = (char *)realloc(seq, cap); }
            for (unsigned i = k_from; i < k_to; ++i) seq[len++] = seq[i];
          }
        }
        if (!skip) {
          orc_pkg_append_string(pkg, seq, (uint32_t)len, reverse);
          *mult = (uint16_t *)realloc(*mult, (*n_mult + 2) * 2);
          (*mult)[(*n_mult)++] = (uint16_t)(atof(comment + 13) + .5); /* GetMultiplicity, :111-119 */
          ++n_read;
        }
      }
      if (eof) break;
      have = 1;
      len = 0;
      char *sp = line + 1;
      while (*sp && *sp != ' ' && *sp != '\t' && *sp != '\n') ++sp;
      while (*sp == ' ' || *sp == '\t') ++sp;
      strncpy(comment, sp, sizeof comment - 1);
      comment[sizeof comment - 1] = 0;
    } else if (have) {
      while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
      if (len + (size_t)n + 1 > cap) { cap = (len + n) * 2; seq = (char *)realloc(seq, cap); }
      memcpy(seq + len, line, (size_t)n);
      len += (size_t)n;
    }
  }
  free(seq);
  free(line);
  fclose(f);
  return (int)n_read;
}
