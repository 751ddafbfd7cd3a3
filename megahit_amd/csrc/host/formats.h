// Host-side readers/writers of the on-disk formats around the SdBG-construction path.
// Byte-for-byte the reference's formats; each function cites the reference code
// whose output/input it must match.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace mhxio {

struct Fatal {};                   // thrown by fatal() instead of exit(1) when g_fatal_throws is set (message already printed)
extern bool g_fatal_throws;
[[noreturn]] void fatal(const char *fmt, ...);
void info(const char *fmt, ...);

// 2-bit packed, gap-free sequence set on the host (SequencePackage layout).
struct PackedSeqs {
  std::vector<uint32_t> words;
  std::vector<uint64_t> start{0};
  uint64_t n_seqs() const { return start.size() - 1; }
  uint64_t n_bases() const { return start.back(); }
  void append_packed(const uint32_t *src, uint32_t len, bool rev);  // sequence_package.h:275-306
  void append_string(const char *s, uint32_t len, bool rev);        // sequence_package.h:261-273
  unsigned base(uint64_t i) const { return (words[i >> 4] >> (30 - 2 * (i & 15))) & 3u; }
};

// <prefix>.lib_info: total_bases total_reads (sequence_lib.cpp:84-90,93-99)
void read_lib_info(const std::string &prefix, int64_t *total_bases, int64_t *total_reads);
// whole <prefix>.bin record stream: per read uint32 len + ceil(len/16) words (sequence_package.h:224-240)
std::vector<uint32_t> read_bin_file(const std::string &path);
// offsets (in words) of each record in a .bin stream
std::vector<uint64_t> index_bin_records(const std::vector<uint32_t> &rec);
std::vector<uint64_t> index_bin_records(const uint32_t *rec, uint64_t n_words);
// The record stream of a read library mapped read-only (no copy: the pages come straight from the page cache), with
// the record offsets either arithmetic (every read has the same length: `fixed_rw` words per record, confirmed by the
// caller) or indexed on demand.
struct BinFile {
  const uint32_t *data = nullptr;
  uint64_t n_words = 0;
  uint32_t fixed_rw = 0;          // != 0: record i starts at word i * fixed_rw
  std::vector<uint64_t> off;      // else: offsets of all records
  uint64_t n_reads = 0;
  uint64_t offset_of(uint64_t i) const { return fixed_rw ? i * fixed_rw : off[i]; }
  uint64_t end_offset(uint64_t i) const { return i >= n_reads ? n_words : offset_of(i); }
  void build_index();             // variable lengths: scan the record headers
  void close();
 private:
  void *map_ = nullptr;
  uint64_t map_bytes_ = 0;
  std::vector<uint32_t> owned_;
  friend BinFile open_bin_file(const std::string &path);
};
BinFile open_bin_file(const std::string &path);

// whole (possibly gzip'ed) file -> memory; "-" = stdin (FastxReader opens its files through zlib the same way)
std::vector<char> read_text_file(const std::string &path);
// the same without a copy when the file is a plain (not gzip'ed) regular file: mapped read-only
struct TextFile {
  const char *data = nullptr;
  size_t size = 0;
  std::vector<char> owned;
  void *map = nullptr;
  void close();
};
TextFile open_text_file(const std::string &path);
// Sequential FASTA/FASTQ parser with kseq's semantics (kseq.h:193-247): multi-line records, junk before the first header,
// '\r' stripped at line ends as ks_getuntil2 does, the stream ends at the first malformed FASTQ record.  calls
// on_seq(ptr, len) per record.  The fallback of `buildlib` for texts the GPU parser declines.
void parse_fastx_sequential(const char *text, size_t n, const std::function<void(const char *, size_t)> &on_seq);
// TrimN + character map + 2-bit packing of one sequence, appended as a .bin record (fastx_reader.cpp:56-71,
// sequence_package.h:78-83,224-240,262-267); returns the stored length
uint32_t append_bin_record(std::vector<uint32_t> *out, const char *s, size_t len);

// EdgeWriter + EdgeIoMetadata::Serialize (edge_writer.h:17-111, edge_io_meta.h:25-44); edges are in
// bucket order; they are split over n_files files at bucket boundaries.
void write_edges(const std::string &prefix, uint32_t k, uint32_t words_per_edge, const uint32_t *edges, uint64_t n_edges,
                 const uint64_t *bucket_count, int n_files);
// KmerCounter::Lv0Postprocess (kmer_counter.cpp:383-403): reversed reads with first_0_out < last_0_in
void write_cand(const std::string &prefix, const BinFile &bin, const uint32_t *first_0_out, const uint32_t *last_0_in, int64_t *n_cand,
                int64_t *n_has_tips);
// EdgeMultiplicityRecorder::DumpStat (edge_counter.h:44-52)
void write_counting(const std::string &prefix, const int64_t *hist);
// SdbgWriter::Finalize + SdbgMeta::Serialize (sdbg_writer.cpp:68-79, sdbg_meta.cpp:12-61)
void write_sdbg(const std::string &prefix, uint32_t k, uint32_t words_per_tip_label, const uint8_t *bytes, uint64_t n_bytes,
                const uint64_t *bucket_off, const uint64_t *bucket_items, const uint64_t *bucket_tips, const uint64_t *bucket_large,
                int n_files);

// EdgeReader (edge_reader.h:13-158): sorted (bucket-id order over any number of files) or unsorted
struct EdgeSet {
  uint32_t k = 0, words_per_edge = 0;
  bool sorted = true;
  const uint32_t *data = nullptr;  // n_edges * words_per_edge, in reading order: the mapped file when its buckets lie in order
  uint64_t n_words = 0;
  std::vector<uint32_t> raw;       // ... else a copy
  std::shared_ptr<void> map;       // keeps the mapping alive
  uint64_t n_edges() const { return words_per_edge ? n_words / words_per_edge : 0; }
};
EdgeSet read_edges(const std::string &prefix);

// ContigReader::ReadAllWithMultiplicity (contig_reader.h:52-119) on a FASTA file
int64_t read_contigs(const std::string &fasta, PackedSeqs *pkg, std::vector<uint16_t> *mult, unsigned min_len, unsigned k_from,
                     unsigned k_to, bool reverse, unsigned discard_flags = 0);
// unsorted edge file as KmerCollector / EdgeWriter::WriteUnordered leave it (edge_writer.h:48-53,94-110): one file, no
// bucket lines, is_sorted 0
void write_edges_unsorted(const std::string &prefix, uint32_t k, uint32_t words_per_edge, const uint32_t *edges, uint64_t n_edges);

}  // namespace mhxio
