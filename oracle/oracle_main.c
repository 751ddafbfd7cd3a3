/*
 * oracle_main.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * File-level front end of the C restatement with the reference's flag names
 * (main_sdbg_build.cpp:35-224), so tests can diff <prefix>.* outputs of
 * ref_core / oracle_core / the product CLI through one canonicaliser.
 * Extra flag: --tie stable|kmsort (S1 tie order; default kmsort).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static const char *opt(int argc, char **argv, const char *a, const char *b, const char *def) {
  for (int i = 2; i + 1 < argc; ++i)
    if (!strcmp(argv[i], a) || (b && !strcmp(argv[i], b))) return argv[i + 1];
  return def;
}
static int flag(int argc, char **argv, const char *a) {
  for (int i = 2; i < argc; ++i)
    if (!strcmp(argv[i], a)) return 1;
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const char *out = opt(argc, argv, "--output_prefix", "-o", "out");
  if (!strcmp(argv[1], "count") || !strcmp(argv[1], "read2sdbg")) {
    int k = atoi(opt(argc, argv, "--kmer_k", "-k", "21"));
    int m = atoi(opt(argc, argv, "--min_kmer_frequency", "-m", "2"));
    const char *lib = opt(argc, argv, "--read_lib_file", NULL, NULL);
    if (!lib) { fprintf(stderr, "No read library configuration file!\n"); return 1; }
    orc_pkg reads;
    if (orc_load_read_lib(lib, 1, &reads)) { fprintf(stderr, "cannot read %s.bin\n", lib); return 1; }
    if (!strcmp(argv[1], "count")) {
      orc_count_out c;
      orc_count(&reads, k, m, &c);
      orc_write_edges(out, k, &c);
      orc_write_cand(out, &reads, &c);
      orc_write_counting(out, c.hist);
      fprintf(stderr, "oracle count: %lld items, %llu solid edges\n", (long long)c.n_items, (unsigned long long)c.edges.n);
      orc_count_free(&c);
    } else {
      int tie = !strcmp(opt(argc, argv, "--tie", NULL, "kmsort"), "stable") ? ORC_TIE_STABLE : ORC_TIE_KMSORT;
      orc_s1_out s1;
      memset(&s1, 0, sizeof s1);
      if (m > 1) { /* main_sdbg_build.cpp:139-147 */
        orc_s1(&reads, k, m, tie, &s1);
        orc_write_counting(out, s1.hist);
        orc_write_mercy_cand(out, &reads, &s1);
        if (flag(argc, argv, "--need_mercy")) {
          long long nm = orc_s2_add_mercy(&reads, k, s1.is_solid, s1.mercy, s1.n_mercy);
          fprintf(stderr, "oracle Number mercy: %lld\n", nm);
        }
      }
      orc_sdbg_out sd;
      orc_s2(&reads, k, m, m > 1 ? s1.is_solid : NULL, &sd);
      orc_write_sdbg(out, &sd);
      fprintf(stderr, "oracle read2sdbg: %lld sort items, %llu bytes\n", (long long)sd.n_sort_items, (unsigned long long)sd.n_bytes);
      orc_sdbg_free(&sd);
      if (m > 1) orc_s1_free(&s1);
    }
    orc_pkg_free(&reads);
    return 0;
  }
  if (!strcmp(argv[1], "seq2sdbg")) {
    int k = atoi(opt(argc, argv, "--kmer_size", "-k", "0"));
    int k_from = atoi(opt(argc, argv, "--kmer_from", NULL, "0"));
    const char *in = opt(argc, argv, "--input_prefix", NULL, NULL);
    const char *contig = opt(argc, argv, "--contig", NULL, NULL), *bubble = opt(argc, argv, "--bubble", NULL, NULL);
    const char *addi = opt(argc, argv, "--addi_contig", NULL, NULL), *local = opt(argc, argv, "--local_contig", NULL, NULL);
    if (k < 9) { fprintf(stderr, "kmer size must be >= 9!\n"); return 1; }
    orc_pkg pkg;
    orc_pkg_init(&pkg);
    uint16_t *mult = NULL;
    uint64_t n_mult = 0;
    if (in) {
      int kin;
      if (orc_read_edges(in, &pkg, &mult, &n_mult, &kin)) { fprintf(stderr, "cannot read edges %s\n", in); return 1; }
      if (flag(argc, argv, "--need_mercy")) { /* seq_to_sdbg.cpp:433-447 */
        char path[4096];
        snprintf(path, sizeof path, "%s.cand", in);
        orc_pkg cand;
        if (orc_read_bin(path, 0, &cand)) { fprintf(stderr, "cannot read %s\n", path); return 1; }
        long long nm = orc_gen_mercy_edges(&pkg, &mult, &n_mult, &cand, k);
        fprintf(stderr, "oracle Number of mercy edges: %lld\n", nm);
        orc_pkg_free(&cand);
      }
    }
    /* order: contig (+loop extension), bubble, addi, local — seq_to_sdbg.cpp:449-503 */
    if (contig) {
      orc_read_contigs(contig, &pkg, &mult, &n_mult, k + 1, k_from, k, 1);
      if (bubble) orc_read_contigs(bubble, &pkg, &mult, &n_mult, k + 1, 0, 0, 1);
    }
    if (addi) orc_read_contigs(addi, &pkg, &mult, &n_mult, k + 1, 0, 0, 1);
    if (local) orc_read_contigs(local, &pkg, &mult, &n_mult, k + 1, 0, 0, 1);
    orc_sdbg_out sd;
    orc_seq2sdbg(&pkg, mult, k, &sd);
    orc_write_sdbg(out, &sd);
    fprintf(stderr, "oracle seq2sdbg: %llu seqs, %lld sort items, %llu bytes\n", (unsigned long long)pkg.n_seqs,
            (long long)sd.n_sort_items, (unsigned long long)sd.n_bytes);
    orc_sdbg_free(&sd);
    free(mult);
    orc_pkg_free(&pkg);
    return 0;
  }
  fprintf(stderr, "unknown sub-program\n");
  return 1;
}
