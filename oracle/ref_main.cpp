// TEST INFRASTRUCTURE — not product code.
// Tiny dispatcher linked against the UNMODIFIED reference sources that lie
// under /root/reference/src (compiled in place by oracle/Makefile, objects and
// the binary go to oracle/_ref/, nothing is copied into this repo).
// It exposes only the sub-programs of the SdBG-construction hot path
// (reference: src/main.cpp:68-110 dispatches these same entry points).
#include <cstdio>
#include <cstring>

int main_build_lib(int argc, char **argv);   // reference src/main_buildlib.cpp
int main_kmer_count(int argc, char **argv);  // reference src/main_sdbg_build.cpp:35
int main_read2sdbg(int argc, char **argv);   // reference src/main_sdbg_build.cpp:88
int main_seq2sdbg(int argc, char **argv);    // reference src/main_sdbg_build.cpp:158

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s buildlib|count|read2sdbg|seq2sdbg ...\n", argv[0]);
    return 1;
  }
  if (!strcmp(argv[1], "buildlib")) return main_build_lib(argc - 1, argv + 1);
  if (!strcmp(argv[1], "count")) return main_kmer_count(argc - 1, argv + 1);
  if (!strcmp(argv[1], "read2sdbg")) return main_read2sdbg(argc - 1, argv + 1);
  if (!strcmp(argv[1], "seq2sdbg")) return main_seq2sdbg(argc - 1, argv + 1);
  fprintf(stderr, "unknown sub-program %s\n", argv[1]);
  return 1;
}
