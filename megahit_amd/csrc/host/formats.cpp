#include "formats.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace mhxio {

bool g_fatal_throws = false;
void fatal(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "FATAL ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  if (g_fatal_throws) throw Fatal{};  // a resident server (mhx_core --serve) fails the request, not the process
  exit(1);
}
void info(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "INFO  ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
}

void PackedSeqs::append_packed(const uint32_t *src, uint32_t len, bool rev) {
  if (len == 0) {  // fake 1-base sequence, sequence_package.h:276-281
    uint32_t fake = 0;
    append_packed(&fake, 1, false);
    return;
  }
  uint64_t pos = start.back();
  words.resize((pos + len + 15) / 16, 0);
  if (!rev && (pos & 15) == 0) {
    memcpy(&words[pos >> 4], src, ((len + 15) / 16) * 4);
    if (len & 15) words[(pos + len) >> 4] &= 0xFFFFFFFFu << (32 - 2 * (len & 15));
  } else {
    for (uint32_t i = 0; i < len; ++i) {
      uint32_t j = rev ? len - 1 - i : i;
      uint32_t c = (src[j >> 4] >> (30 - 2 * (j & 15))) & 3u;
      words[(pos + i) >> 4] |= c << (30 - 2 * ((pos + i) & 15));
    }
  }
  start.push_back(pos + len);
}

void PackedSeqs::append_string(const char *s, uint32_t len, bool rev) {
  if (len == 0) {  // sequence_package.h:262-267
    append_string("A", 1, false);
    return;
  }
  // "ACGTNacgtn" -> 0123201232, anything else 0 (sequence_package.h:80-82); a table look-up per base and one store per 16 bases (the
  // per-base read-modify-write of round 1 took 0.3 s for the 70 M contig bases of a k-list step: most of seq2sdbg's wall time)
  static const struct Lut {
    uint8_t c[256];
    Lut() {
      memset(c, 0, sizeof c);
      c[(unsigned char)'C'] = c[(unsigned char)'c'] = 1;
      c[(unsigned char)'G'] = c[(unsigned char)'g'] = c[(unsigned char)'N'] = c[(unsigned char)'n'] = 2;
      c[(unsigned char)'T'] = c[(unsigned char)'t'] = 3;
    }
  } lut;
  uint64_t pos = start.back();
  if (words.capacity() < (pos + len + 15) / 16) words.reserve(std::max<size_t>((pos + len + 15) / 16, words.capacity() * 2));
  words.resize((pos + len + 15) / 16, 0);
  uint32_t *w = words.data();
  const unsigned char *u = reinterpret_cast<const unsigned char *>(s);
  uint32_t i = 0;
  uint64_t p = pos;
  auto code = [&](uint32_t j) -> uint32_t { return lut.c[u[rev ? len - 1 - j : j]]; };
  for (; i < len && (p & 15); ++i, ++p) w[p >> 4] |= code(i) << (30 - 2 * (p & 15));
  if (rev) {
    for (; i + 16 <= len; i += 16, p += 16) {
      const unsigned char *q = u + (len - 1 - i);  // bases i .. i + 15 are q[0], q[-1], ..., q[-15]
      uint32_t x = 0;
      for (int j = 0; j < 16; ++j) x = (x << 2) | lut.c[q[-j]];
      w[p >> 4] = x;
    }
  } else {
    for (; i + 16 <= len; i += 16, p += 16) {
      const unsigned char *q = u + i;
      uint32_t x = 0;
      for (int j = 0; j < 16; ++j) x = (x << 2) | lut.c[q[j]];
      w[p >> 4] = x;
    }
  }
  for (; i < len; ++i, ++p) w[p >> 4] |= code(i) << (30 - 2 * (p & 15));
  start.push_back(pos + len);
}

void read_lib_info(const std::string &prefix, int64_t *total_bases, int64_t *total_reads) {
  std::ifstream f(prefix + ".lib_info");
  if (!f || !(f >> *total_bases >> *total_reads)) fatal("cannot read %s.lib_info", prefix.c_str());
}

std::vector<uint32_t> read_bin_file(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) fatal("Cannot open %s", path.c_str());
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint32_t> v((size_t)sz / 4);
  if (sz && fread(v.data(), 4, v.size(), f) != v.size()) fatal("short read on %s", path.c_str());
  fclose(f);
  return v;
}

std::vector<char> read_text_file(const std::string &path) {
  gzFile f = path == "-" ? gzdopen(fileno(stdin), "r") : gzopen(path.c_str(), "r");
  if (!f) fatal("Cannot open file %s", path.c_str());
  gzbuffer(f, 1 << 20);
  std::vector<char> v;
  size_t cap = 1 << 24, len = 0;
  v.resize(cap);
  for (;;) {
    if (cap - len < (1u << 22)) {
      cap *= 2;
      v.resize(cap);
    }
    const int got = gzread(f, v.data() + len, (unsigned)std::min<size_t>(cap - len, 1u << 30));
    if (got < 0) fatal("read error on %s", path.c_str());
    if (got == 0) break;
    len += (size_t)got;
  }
  gzclose(f);
  v.resize(len);
  return v;
}

TextFile open_text_file(const std::string &path) {
  TextFile tf;
  if (path != "-") {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) fatal("Cannot open file %s", path.c_str());
    struct stat st;
    unsigned char magic[2] = {0, 0};
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
      void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
        tf.map = m;
        tf.data = static_cast<const char *>(m);
        tf.size = (size_t)st.st_size;
        ::close(fd);
        return tf;
      }
    }
    ::close(fd);
  }
  tf.owned = read_text_file(path);  // gzip, pipes, stdin: through zlib
  tf.data = tf.owned.data();
  tf.size = tf.owned.size();
  return tf;
}
void TextFile::close() {
  if (map) munmap(map, size);
  map = nullptr;
  data = nullptr;
  owned.clear();
  owned.shrink_to_fit();
}

void parse_fastx_sequential(const char *t, size_t n, const std::function<void(const char *, size_t)> &on_seq) {
  size_t i = 0;
  int last_char = 0;
  std::string seq, line;
  // one line starting at i (without its '\n'); advances i past the newline
  auto take_line = [&](std::string *dst, bool append) {
    size_t e = i;
    while (e < n && t[e] != '\n') ++e;
    if (!append) dst->clear();
    dst->append(t + i, e - i);
    i = e < n ? e + 1 : n;
    if (dst->size() > 1 && dst->back() == '\r') dst->pop_back();  // ks_getuntil2: on the accumulated string
  };
  for (;;) {
    if (last_char == 0) {  // jump to the next header char, anywhere
      while (i < n && t[i] != '>' && t[i] != '@') ++i;
      if (i >= n) return;
      last_char = t[i++];
    }
    if (i >= n) return;  // a header char at the very end: ks_getuntil finds nothing -> EOF
    // name = up to the first whitespace, then the rest of the header line is the comment
    {
      size_t e = i;
      while (e < n && !isspace((unsigned char)t[e])) ++e;
      const bool at_newline = e < n && t[e] == '\n';
      i = e < n ? e + 1 : n;
      if (!at_newline && e < n) take_line(&line, false);
    }
    seq.clear();
    int c = -1;
    while (i < n) {
      c = (unsigned char)t[i++];
      if (c == '>' || c == '+' || c == '@') break;
      if (c == '\n') {
        c = -1;
        continue;
      }
      seq.push_back((char)c);
      take_line(&seq, true);
      c = -1;
    }
    if (c == '>' || c == '@') last_char = c;
    else last_char = 0;
    if (c != '+') {  // FASTA record (or the last record of the stream)
      on_seq(seq.data(), seq.size());
      if (c == -1) return;  // end of text
      continue;
    }
    // FASTQ: skip the rest of the '+' line, then read quality lines until they cover the sequence
    while (i < n && t[i] != '\n') ++i;
    if (i >= n) return;  // no quality string: error -> the reader stops
    ++i;
    std::string qual;
    do take_line(&qual, true);  // (kseq reads at least one quality line, even for an empty sequence)
    while (qual.size() < seq.size() && i < n);
    last_char = 0;
    if (qual.size() != seq.size()) return;  // malformed: the reader stops here
    on_seq(seq.data(), seq.size());
  }
}

uint32_t append_bin_record(std::vector<uint32_t> *out, const char *s, size_t len) {
  size_t b = len, e = len, i = 0;
  for (; i < len; ++i) {
    if (s[i] == 'N' || s[i] == 'n') {
      if (b < len) break;
    } else if (b == len) {
      b = i;
    }
  }
  e = i;
  const char *p = s + b;
  size_t L = e - b;
  static const char fake[] = "A";
  if (L == 0) {
    p = fake;
    L = 1;
  }
  out->push_back((uint32_t)L);
  for (size_t w = 0; w * 16 < L; ++w) {
    uint32_t v = 0;
    for (size_t j = 0; j < 16 && w * 16 + j < L; ++j) {
      unsigned code = 0;
      switch (p[w * 16 + j]) {
        case 'C': case 'c': code = 1; break;
        case 'G': case 'g': case 'N': case 'n': code = 2; break;
        case 'T': case 't': code = 3; break;
        default: code = 0;
      }
      v |= code << (30 - 2 * j);
    }
    out->push_back(v);
  }
  return (uint32_t)L;
}

std::vector<uint64_t> index_bin_records(const uint32_t *rec, uint64_t n_words) {
  std::vector<uint64_t> off;
  uint64_t pos = 0;
  while (pos < n_words) {
    off.push_back(pos);
    pos += 1 + (rec[pos] + 15) / 16;
  }
  if (pos != n_words) fatal("truncated .bin record stream");
  return off;
}
std::vector<uint64_t> index_bin_records(const std::vector<uint32_t> &rec) { return index_bin_records(rec.data(), rec.size()); }

BinFile open_bin_file(const std::string &path) {
  BinFile b;
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) fatal("Cannot open %s", path.c_str());
  struct stat st;
  if (fstat(fd, &st) != 0) fatal("Cannot stat %s", path.c_str());
  b.n_words = (uint64_t)st.st_size / 4;
  if (st.st_size > 0) {
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
      (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
      (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
      b.map_ = m;
      b.map_bytes_ = (uint64_t)st.st_size;
      b.data = static_cast<const uint32_t *>(m);
    } else {  // no mmap on this file system: read it
      b.owned_.resize(b.n_words);
      size_t got = 0;
      while (got < b.n_words * 4) {
        ssize_t r = pread(fd, reinterpret_cast<char *>(b.owned_.data()) + got, b.n_words * 4 - got, (off_t)got);
        if (r <= 0) fatal("short read on %s", path.c_str());
        got += (size_t)r;
      }
      b.data = b.owned_.data();
    }
  }
  ::close(fd);
  // candidate for the arithmetic index: the first record's size divides the stream (the caller confirms that every
  // record really has that length before relying on it)
  if (b.n_words) {
    const uint32_t rw = 1 + (b.data[0] + 15) / 16;
    if (b.data[0] && b.n_words % rw == 0) {
      b.fixed_rw = rw;
      b.n_reads = b.n_words / rw;
    }
  }
  if (!b.fixed_rw) b.build_index();
  return b;
}
void BinFile::build_index() {
  fixed_rw = 0;
  off = index_bin_records(data, n_words);
  n_reads = off.size();
}
void BinFile::close() {
  if (map_) munmap(map_, map_bytes_);
  map_ = nullptr;
  data = nullptr;
  owned_.clear();
}

// buckets -> files: contiguous bucket ranges with roughly equal payload
static std::vector<int> split_buckets(const uint64_t *weight, int n_buckets, int n_files) {
  std::vector<int> file_of(n_buckets, 0);
  long double total = 0;
  for (int b = 0; b < n_buckets; ++b) total += weight[b];
  long double acc = 0;
  for (int b = 0; b < n_buckets; ++b) {
    int f = total > 0 ? (int)(acc * n_files / total) : 0;
    file_of[b] = std::min(f, n_files - 1);
    acc += weight[b];
  }
  return file_of;
}

void write_edges(const std::string &prefix, uint32_t k, uint32_t wpe, const uint32_t *edges, uint64_t n_edges,
                 const uint64_t *bucket_count, int n_files) {
  const int NB = 65536;
  if (n_files < 1) n_files = 1;
  std::vector<int> file_of = split_buckets(bucket_count, NB, n_files);
  std::vector<FILE *> fs(n_files);
  for (int i = 0; i < n_files; ++i) {
    std::string p = prefix + ".edges." + std::to_string(i);
    fs[i] = fopen(p.c_str(), "wb");
    if (!fs[i]) fatal("Cannot open %s", p.c_str());
  }
  std::vector<int64_t> in_file(n_files, 0);
  std::ofstream meta(prefix + ".edges.info");
  meta << "kmer_size " << k << '\n' << "words_per_edge " << wpe << '\n' << "num_files " << n_files << '\n'
       << "num_buckets " << NB << '\n' << "num_edges " << n_edges << '\n' << "is_sorted " << 1 << '\n';
  uint64_t pos = 0;
  for (int b = 0; b < NB; ++b) {
    if (bucket_count[b] == 0) {
      meta << b << " -1 0 0\n";  // EdgeIoBucketInfo defaults, edge_io_meta.h:11-15
      continue;
    }
    int f = file_of[b];
    if (fwrite(edges + pos * wpe, 4, bucket_count[b] * wpe, fs[f]) != bucket_count[b] * wpe) fatal("write error on %s.edges.%d", prefix.c_str(), f);
    meta << b << ' ' << f << ' ' << in_file[f] << ' ' << bucket_count[b] << '\n';
    in_file[f] += (int64_t)bucket_count[b];
    pos += bucket_count[b];
  }
  if (pos != n_edges) fatal("write_edges: bucket counts do not add up");
  for (FILE *f : fs)
    if (fclose(f) != 0) fatal("write error on %s.edges.*", prefix.c_str());
}

void write_cand(const std::string &prefix, const BinFile &bin, const uint32_t *first_0_out, const uint32_t *last_0_in, int64_t *n_cand,
                int64_t *n_has_tips) {
  FILE *f = fopen((prefix + ".cand").c_str(), "wb");
  if (!f) fatal("Cannot open %s.cand", prefix.c_str());
  *n_cand = *n_has_tips = 0;
  PackedSeqs one;
  const uint32_t *rec = bin.data;
  for (uint64_t i = 0; i < bin.n_reads; ++i) {
    uint32_t first = first_0_out[i], last = last_0_in[i];
    if (first != 0xFFFFFFFFu && last != 0xFFFFFFFFu) {
      ++*n_has_tips;
      if (last > first) {
        ++*n_cand;
        const uint64_t ro = bin.offset_of(i);
        uint32_t len = rec[ro];
        one.words.clear();
        one.start.assign(1, 0);
        one.append_packed(&rec[ro + 1], len, true);  // the engine's package holds reversed reads
        uint32_t out_len = (uint32_t)one.n_bases();
        if (fwrite(&out_len, 4, 1, f) != 1 || fwrite(one.words.data(), 4, (out_len + 15) / 16, f) != (out_len + 15) / 16)
          fatal("write error on %s.cand", prefix.c_str());
      }
    }
  }
  if (fclose(f) != 0) fatal("write error on %s.cand", prefix.c_str());
}

void write_counting(const std::string &prefix, const int64_t *hist) {
  FILE *f = fopen((prefix + ".counting").c_str(), "w");
  if (!f) fatal("Cannot open %s.counting", prefix.c_str());
  for (int i = 1; i <= 65535; ++i) fprintf(f, "%d %lld\n", i, (long long)hist[i]);
  if (fclose(f) != 0) fatal("write error on %s.counting", prefix.c_str());
}

void write_sdbg(const std::string &prefix, uint32_t k, uint32_t wpt, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *bucket_off,
                const uint64_t *b_items, const uint64_t *b_tips, const uint64_t *b_large, int n_files) {
  const int NB = 65536;
  if (n_files < 1) n_files = 1;
  std::vector<uint64_t> weight(NB);
  for (int b = 0; b < NB; ++b) weight[b] = 2 * (b_items[b] + b_large[b]) + 4ull * wpt * b_tips[b];
  std::vector<int> file_of = split_buckets(weight.data(), NB, n_files);
  std::vector<FILE *> fs(n_files);
  for (int i = 0; i < n_files; ++i) {
    std::string p = prefix + ".sdbg." + std::to_string(i);
    fs[i] = fopen(p.c_str(), "wb");
    if (!fs[i]) fatal("Cannot open %s", p.c_str());
  }
  std::vector<uint64_t> in_file(n_files, 0);
  // bucket records sorted by (file_id, starting_offset), null buckets last (sdbg_meta.cpp:44-61)
  std::ostringstream lines;
  int used_files = 0, n_null = 0;
  uint64_t total = 0;
  for (int b = 0; b < NB; ++b) {
    if (b_items[b] == 0) {
      ++n_null;
      continue;
    }
    int f = file_of[b];
    if (weight[b] && fwrite(bytes + bucket_off[b], 1, weight[b], fs[f]) != weight[b]) fatal("write error on %s.sdbg.%d", prefix.c_str(), f);
    lines << b << ' ' << f << ' ' << in_file[f] << ' ' << b_items[b] << ' ' << b_tips[b] << ' ' << b_large[b] << '\n';
    in_file[f] += weight[b];
    total += weight[b];
    used_files = std::max(used_files, f + 1);
  }
  if (total != n_bytes) fatal("write_sdbg: bucket sizes do not add up (%llu vs %llu)", (unsigned long long)total, (unsigned long long)n_bytes);
  for (FILE *f : fs)
    if (fclose(f) != 0) fatal("write error on %s.sdbg.*", prefix.c_str());
  std::ofstream meta(prefix + ".sdbg_info");
  meta << "k " << k << "\n" << "words_per_tip_label " << wpt << "\n" << "num_buckets " << NB << "\n" << "num_files " << used_files << "\n";
  meta << lines.str();
  for (int i = 0; i < n_null; ++i) meta << "18446744073709551615 18446744073709551615 0 0 0 0\n";
  meta.close();
  if (!meta) fatal("write error on %s.sdbg_info", prefix.c_str());
}

static bool scan_field(std::istream &is, const char *name, long long *v) {
  std::string s;
  return static_cast<bool>(is >> s >> *v) && s == name;
}

EdgeSet read_edges(const std::string &prefix) {
  std::ifstream meta(prefix + ".edges.info");
  if (!meta) fatal("Cannot open %s.edges.info", prefix.c_str());
  long long k, wpe, nfiles, nb, nedges, sorted;
  if (!scan_field(meta, "kmer_size", &k) || !scan_field(meta, "words_per_edge", &wpe) || !scan_field(meta, "num_files", &nfiles) ||
      !scan_field(meta, "num_buckets", &nb) || !scan_field(meta, "num_edges", &nedges) || !scan_field(meta, "is_sorted", &sorted))
    fatal("Invalid format: %s.edges.info", prefix.c_str());
  EdgeSet es;
  es.k = (uint32_t)k;
  es.words_per_edge = (uint32_t)wpe;
  es.sorted = sorted != 0;
  // one file whose buckets lie one after the other (what EdgeWriter leaves with one thread, and every file of ours): map it,
  // the upload reads the page cache directly; anything else is gathered into a copy
  std::vector<long long> b_fid, b_off, b_cnt;
  bool in_order = nfiles == 1;
  uint64_t pos = 0;
  if (es.sorted) {
    for (long long b = 0; b < nb; ++b) {
      long long id, fid, off, cnt;
      if (!(meta >> id >> fid >> off >> cnt) || id != b) fatal("Invalid format: bucket id not matched!");
      if (fid >= nfiles) fatal("Record ID %lld is greater than number of files %lld", fid, nfiles);
      if (fid < 0) continue;
      if (off != (long long)pos) in_order = false;
      b_fid.push_back(fid);
      b_off.push_back(off);
      b_cnt.push_back(cnt);
      pos += (uint64_t)cnt;
    }
  } else {
    pos = (uint64_t)nedges;
  }
  if (pos != (uint64_t)nedges) fatal("edge count mismatch in %s", prefix.c_str());
  es.n_words = (uint64_t)nedges * wpe;
  if (in_order && nedges) {
    const std::string p = prefix + ".edges.0";
    int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) fatal("Cannot open %s", p.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < es.n_words * 4) fatal("short read on edges file");
    const size_t bytes = (size_t)es.n_words * 4;
    void *m = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m != MAP_FAILED) {
      (void)madvise(m, bytes, MADV_SEQUENTIAL);
      (void)madvise(m, bytes, MADV_WILLNEED);
      es.map = std::shared_ptr<void>(m, [bytes](void *q) { munmap(q, bytes); });
      es.data = static_cast<const uint32_t *>(m);
      return es;
    }
  }
  es.raw.resize((size_t)nedges * wpe);
  es.data = es.raw.data();
  std::vector<FILE *> fs((size_t)nfiles);
  for (long long i = 0; i < nfiles; ++i) {
    std::string p = prefix + ".edges." + std::to_string(i);
    fs[i] = fopen(p.c_str(), "rb");
    if (!fs[i]) fatal("Cannot open %s", p.c_str());
  }
  pos = 0;
  if (es.sorted) {
    for (size_t b = 0; b < b_fid.size(); ++b) {
      fseek(fs[b_fid[b]], b_off[b] * wpe * 4, SEEK_SET);
      if (fread(&es.raw[pos * wpe], 4, (size_t)(b_cnt[b] * wpe), fs[b_fid[b]]) != (size_t)(b_cnt[b] * wpe)) fatal("short read on edges file");
      pos += (uint64_t)b_cnt[b];
    }
  } else if (nedges) {
    if (fread(es.raw.data(), 4, (size_t)(nedges * wpe), fs[0]) != (size_t)(nedges * wpe)) fatal("short read on edges file");
  }
  for (FILE *f : fs) fclose(f);
  return es;
}

void write_edges_unsorted(const std::string &prefix, uint32_t k, uint32_t wpe, const uint32_t *edges, uint64_t n_edges) {
  FILE *f = fopen((prefix + ".edges.0").c_str(), "wb");
  if (!f) fatal("Cannot open %s.edges.0", prefix.c_str());
  if (n_edges && fwrite(edges, 4, (size_t)n_edges * wpe, f) != (size_t)n_edges * wpe) fatal("write error on %s.edges.0", prefix.c_str());
  if (fclose(f) != 0) fatal("write error on %s.edges.0", prefix.c_str());
  std::ofstream meta(prefix + ".edges.info");
  meta << "kmer_size " << k << '\n' << "words_per_edge " << wpe << '\n' << "num_files " << 1 << '\n'
       << "num_buckets " << 0 << '\n' << "num_edges " << n_edges << '\n' << "is_sorted " << 0 << '\n';
  meta.close();
  if (!meta) fatal("write error on %s.edges.info", prefix.c_str());
}

// a plain (not gzip'ed) contig file, mapped: header lines and sequence lines found with memchr, one-line sequences packed straight out
// of the mapping (contig_reader.h:52-119 semantics, as the stream parser below)
static int64_t read_contigs_mapped(const TextFile &tf, PackedSeqs *pkg, std::vector<uint16_t> *mult, unsigned min_len, unsigned k_from, unsigned k_to,
                                   bool reverse, unsigned discard_flags) {
  const bool extend_loop = k_from < k_to;
  const char *p = tf.data, *end = tf.data + tf.size;
  int64_t n_read = 0;
  std::string seq;
  auto line_end = [&](const char *q) -> const char * {
    const char *e = static_cast<const char *>(memchr(q, '\n', (size_t)(end - q)));
    return e ? e : end;
  };
  // skip to the first header
  while (p < end && *p != '>') p = std::min(end, line_end(p) + 1);
  while (p < end) {
    const char *he = line_end(p);  // header: [p + 1, he)
    const char *hs = p + 1, *hz = he;
    while (hz > hs && (hz[-1] == '\r')) --hz;
    const char *sp = hs;
    while (sp < hz && *sp != ' ' && *sp != '\t') ++sp;
    while (sp < hz && (*sp == ' ' || *sp == '\t')) ++sp;
    const char *comment = sp;
    const size_t clen = (size_t)(hz - sp);
    // sequence lines up to the next header
    const char *q = std::min(end, he + 1);
    const char *s0 = nullptr;
    size_t slen = 0;
    bool multi = false;
    while (q < end && *q != '>') {
      const char *le = line_end(q);
      size_t n = (size_t)(le - q);
      while (n && q[n - 1] == '\r') --n;
      if (n) {
        if (!s0) {
          s0 = q;
          slen = n;
        } else {
          if (!multi) {
            seq.assign(s0, slen);
            multi = true;
          }
          seq.append(q, n);
        }
      }
      q = std::min(end, le + 1);
    }
    if (multi) {
      s0 = seq.data();
      slen = seq.size();
    }
    p = q;
    if (slen < min_len) continue;
    const unsigned flag = clen > 5 ? (unsigned)(comment[5] - '0') : 0;  // "flag=x multi=..." (contig_reader.h:66)
    if (discard_flags & flag) continue;                                // contig_reader.h:67-70
    if (extend_loop && (flag & 2u)) {                                  // contig_flag::kLoop
      if (slen < k_to + 1u) continue;
      if (!multi) seq.assign(s0, slen);
      for (unsigned i = k_from; i < k_to; ++i) seq.push_back(seq[i]);
      s0 = seq.data();
      slen = seq.size();
    }
    pkg->append_string(s0, (uint32_t)slen, reverse);
    double m = 0;
    if (clen > 13) {
      char buf[48];
      const size_t n = std::min(clen - 13, sizeof buf - 1);
      memcpy(buf, comment + 13, n);
      buf[n] = 0;
      m = atof(buf);
    }
    mult->push_back((uint16_t)(m + .5));  // GetMultiplicity<mul_t>, contig_reader.h:111-119
    ++n_read;
  }
  return n_read;
}

int64_t read_contigs(const std::string &fasta, PackedSeqs *pkg, std::vector<uint16_t> *mult, unsigned min_len, unsigned k_from,
                     unsigned k_to, bool reverse, unsigned discard_flags) {
  if (!getenv("MHX_CONTIG_STREAM_PARSER")) {
    TextFile tf = open_text_file(fasta);
    if (tf.map) {  // (a gzip'ed file comes inflated into tf.owned: parsed the same way)
      const int64_t n = read_contigs_mapped(tf, pkg, mult, min_len, k_from, k_to, reverse, discard_flags);
      tf.close();
      return n;
    }
    if (!tf.owned.empty() || tf.size == 0) {
      const int64_t n = read_contigs_mapped(tf, pkg, mult, min_len, k_from, k_to, reverse, discard_flags);
      tf.close();
      return n;
    }
    tf.close();
  }
  gzFile f = gzopen(fasta.c_str(), "r");
  if (!f) fatal("Cannot open %s", fasta.c_str());
  const bool extend_loop = k_from < k_to;
  std::string seq, comment;
  std::vector<char> line(1 << 20);
  bool have = false, eof = false;
  int64_t n_read = 0;
  auto flush = [&]() {
    if (!have || seq.size() < min_len) return;
    unsigned flag = comment.size() > 5 ? (unsigned)(comment[5] - '0') : 0;  // "flag=x multi=..." (contig_reader.h:66)
    if (discard_flags & flag) return;                                       // contig_reader.h:67-70
    if (extend_loop && (flag & 2u)) {                                       // contig_flag::kLoop
      if (seq.size() < k_to + 1u) return;
      for (unsigned i = k_from; i < k_to; ++i) seq.push_back(seq[i]);
    }
    pkg->append_string(seq.data(), (uint32_t)seq.size(), reverse);
    double m = comment.size() > 13 ? atof(comment.c_str() + 13) : 0;
    mult->push_back((uint16_t)(m + .5));  // GetMultiplicity<mul_t>, contig_reader.h:111-119
    ++n_read;
  };
  while (!eof) {
    char *r = gzgets(f, line.data(), (int)line.size());
    if (!r) eof = true;
    if (eof || line[0] == '>') {
      flush();
      if (eof) break;
      have = true;
      seq.clear();
      std::string hdr(line.data() + 1);
      while (!hdr.empty() && (hdr.back() == '\n' || hdr.back() == '\r')) hdr.pop_back();
      size_t sp = hdr.find_first_of(" \t");
      comment = sp == std::string::npos ? "" : hdr.substr(hdr.find_first_not_of(" \t", sp));
    } else if (have) {
      size_t n = strlen(line.data());
      while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
      seq.append(line.data(), n);
    }
  }
  gzclose(f);
  return n_read;
}

}  // namespace mhxio
