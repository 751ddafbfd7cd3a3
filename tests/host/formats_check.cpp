// Host-only checks of the edge-file reader / writers of megahit_amd/csrc/host/formats.cpp (no GPU, no libmhx):
//   formats_check <tmp dir>
// write_edges with 1 and 3 files -> read_edges gives the records back in bucket order (mapped when one file holds the
// buckets in order, gathered otherwise); a hand-made .edges.info whose buckets lie in another order in the file and over
// two files (what EdgeWriter leaves with several threads: edge_writer.h:56-92); the unsorted variant of iterate.
// Prints "ok" and exits 0, or a message and exits 1.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "formats.h"

static void fail(const char *what) {
  fprintf(stderr, "formats_check: %s\n", what);
  exit(1);
}

int main(int argc, char **argv) {
  if (argc != 2) fail("usage: formats_check <dir>");
  const std::string d = argv[1];
  const uint32_t k = 21, wpe = 2;
  // edges in bucket order: bucket = word0 >> 16
  std::vector<uint32_t> edges;
  std::vector<uint64_t> bcount(65536, 0);
  uint32_t x = 12345;
  for (uint32_t b = 0; b < 65536; b += 97) {
    const uint32_t n = 1 + b % 5;
    for (uint32_t i = 0; i < n; ++i) {
      x = x * 1664525u + 1013904223u;
      edges.push_back((b << 16) | (x & 0xFFFFu));
      edges.push_back((x & 0xFFFF0000u) | (1 + i));
    }
    bcount[b] = n;
  }
  const uint64_t n_edges = edges.size() / wpe;
  for (int n_files : {1, 3}) {
    const std::string p = d + "/e" + std::to_string(n_files);
    mhxio::write_edges(p, k, wpe, edges.data(), n_edges, bcount.data(), n_files);
    mhxio::EdgeSet es = mhxio::read_edges(p);
    if (es.k != k || es.words_per_edge != wpe || !es.sorted || es.n_edges() != n_edges) fail("header of a written edge set");
    if (memcmp(es.data, edges.data(), edges.size() * 4) != 0) fail("records of a written edge set");
    if (n_files == 1 && !es.map) fail("a single in-order file should be mapped");
    if (n_files == 3 && es.raw.empty()) fail("several files should be gathered");
  }
  {  // buckets out of order inside the files, two files
    const std::string p = d + "/shuffled";
    // three buckets: 5 (2 edges), 9 (1 edge), 70 (3 edges); file 0 holds [70, 5], file 1 holds [9]
    const uint32_t e5[] = {(5u << 16) | 1, 11, (5u << 16) | 2, 12}, e9[] = {(9u << 16) | 3, 13},
                   e70[] = {(70u << 16) | 4, 14, (70u << 16) | 5, 15, (70u << 16) | 6, 16};
    FILE *f0 = fopen((p + ".edges.0").c_str(), "wb"), *f1 = fopen((p + ".edges.1").c_str(), "wb");
    if (!f0 || !f1) fail("cannot create files");
    fwrite(e70, 4, 6, f0);
    fwrite(e5, 4, 4, f0);
    fwrite(e9, 4, 2, f1);
    fclose(f0);
    fclose(f1);
    std::ofstream meta(p + ".edges.info");
    meta << "kmer_size 21\nwords_per_edge 2\nnum_files 2\nnum_buckets 65536\nnum_edges 6\nis_sorted 1\n";
    for (int b = 0; b < 65536; ++b) {
      if (b == 5) meta << "5 0 3 2\n";
      else if (b == 9) meta << "9 1 0 1\n";
      else if (b == 70) meta << "70 0 0 3\n";
      else meta << b << " -1 0 0\n";
    }
    meta.close();
    mhxio::EdgeSet es = mhxio::read_edges(p);
    std::vector<uint32_t> want(e5, e5 + 4);
    want.insert(want.end(), e9, e9 + 2);
    want.insert(want.end(), e70, e70 + 6);
    if (es.n_edges() != 6 || memcmp(es.data, want.data(), want.size() * 4) != 0) fail("buckets out of order over two files");
  }
  {  // one file, buckets out of order: must not be mapped as it lies
    const std::string p = d + "/one_shuffled";
    const uint32_t a[] = {(7u << 16) | 1, 21}, b[] = {(3u << 16) | 2, 22};
    FILE *f0 = fopen((p + ".edges.0").c_str(), "wb");
    fwrite(a, 4, 2, f0);
    fwrite(b, 4, 2, f0);
    fclose(f0);
    std::ofstream meta(p + ".edges.info");
    meta << "kmer_size 21\nwords_per_edge 2\nnum_files 1\nnum_buckets 65536\nnum_edges 2\nis_sorted 1\n";
    for (int q = 0; q < 65536; ++q) {
      if (q == 3) meta << "3 0 1 1\n";
      else if (q == 7) meta << "7 0 0 1\n";
      else meta << q << " -1 0 0\n";
    }
    meta.close();
    mhxio::EdgeSet es = mhxio::read_edges(p);
    const uint32_t want[] = {(3u << 16) | 2, 22, (7u << 16) | 1, 21};
    if (es.n_edges() != 2 || memcmp(es.data, want, sizeof want) != 0) fail("one file, buckets out of order");
  }
  {  // unsorted (iterate)
    const std::string p = d + "/unsorted";
    const uint32_t u[] = {9, 8, 7, 6, 5, 4};
    mhxio::write_edges_unsorted(p, 29, 3, u, 2);
    mhxio::EdgeSet es = mhxio::read_edges(p);
    if (es.sorted || es.k != 29 || es.words_per_edge != 3 || es.n_edges() != 2 || memcmp(es.data, u, sizeof u) != 0) fail("unsorted edge set");
    mhxio::write_edges_unsorted(d + "/empty", 29, 3, u, 0);
    if (mhxio::read_edges(d + "/empty").n_edges() != 0) fail("empty unsorted edge set");
  }
  {  // contigs: the discard flags of iterate (kStandalone | kLoop), multiplicities, the reverse flag
    const std::string fa = d + "/c.fa";
    std::ofstream f(fa);
    f << ">k21_0 flag=0 multi=3.5000 len=30\nACGTACGTACGTACGTACGTACGTACGTAC\n"
      << ">k21_1 flag=1 multi=2.0000 len=30\nTTTTACGTACGTACGTACGTACGTACGTAC\n"
      << ">k21_2 flag=2 multi=9.0000 len=30\nGGGGACGTACGTACGTACGTACGTACGTAC\n"
      << ">k21_3 flag=0 multi=65534.6000 len=25\nCCCCACGTACGTACGTACGTACGTA\n";
    f.close();
    mhxio::PackedSeqs all, kept;
    std::vector<uint16_t> m_all, m_kept;
    if (mhxio::read_contigs(fa, &all, &m_all, 0, 0, 0, false) != 4 || all.n_seqs() != 4) fail("read_contigs: all");
    if (mhxio::read_contigs(fa, &kept, &m_kept, 0, 0, 0, false, 1u | 2u) != 2 || kept.n_seqs() != 2) fail("read_contigs: discard flags");
    if (kept.start[1] != 30 || kept.start[2] != 55 || kept.base(0) != 0 || kept.base(30) != 1) fail("read_contigs: kept sequences");
    if (m_all.size() != 4 || m_all[0] != 4 || m_all[1] != 2 || m_all[3] != 65535) fail("read_contigs: multiplicities (GetMultiplicity rounds m + .5, contig_reader.h:111-119)");
  }
  printf("ok\n");
  return 0;
}
