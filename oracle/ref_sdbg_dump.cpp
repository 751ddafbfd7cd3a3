// TEST INFRASTRUCTURE — not product code.
// Loads an SdBG with the REFERENCE's own loader (SDBG::LoadFromFile = LoadSdbgRawContent + the rank/select
// construction of kmlib::RankAndSelect; /root/reference/src/sdbg/sdbg.h:26-61, sdbg_raw_content.cpp:18-96,
// kmlib/kmrns.h:118-175), compiled from the sources where they lie, and dumps every array it built, so that
// tests/test_gpu_sdbg_index.py can compare the device-resident hand-over of libmhx word for word.
// The index structures are private members of the reference's classes: this dumper (and only it) opens them up.
#include <chrono>
#define private public
#define protected public
#include "sdbg/sdbg.h"
#include "assembly/sdbg_pruning.h"
#undef private
#undef protected

#include <cstdio>
#include <string>
#include <vector>

static FILE *g_out;
static void section(const char *name, const void *p, size_t elem, size_t n) {
  char nm[32] = {0};
  snprintf(nm, sizeof nm, "%s", name);
  uint64_t hdr[2] = {elem, n};
  fwrite(nm, 1, 32, g_out);
  fwrite(hdr, 8, 2, g_out);
  if (n) fwrite(p, elem, n, g_out);
}
template <class RS>
static void dump_rs(const char *prefix, const RS &rs, unsigned c_lo, unsigned c_hi, bool with_select) {
  std::vector<int64_t> cc;
  for (unsigned c = c_lo; c < c_hi; ++c) {
    char nm[32];
    snprintf(nm, sizeof nm, "%s_l2_%u", prefix, c);
    section(nm, rs.l2_occ_[c].data(), 8, rs.l2_occ_[c].size());
    snprintf(nm, sizeof nm, "%s_l1_%u", prefix, c);
    section(nm, rs.l1_occ_[c].data(), 2, rs.l1_occ_[c].size());
    if (with_select) {
      snprintf(nm, sizeof nm, "%s_sel_%u", prefix, c);
      section(nm, rs.rank2itv_[c].data(), 4, rs.rank2itv_[c].size());
    }
    cc.push_back(rs.char_count_[c]);
  }
  char nm[32];
  snprintf(nm, sizeof nm, "%s_count", prefix);
  section(nm, cc.data(), 8, cc.size());
}

int main(int argc, char **argv) {
  if (argc != 3 && argc != 4) {
    fprintf(stderr, "usage: %s <sdbg prefix> <dump file> [max_tip_len: also run RemoveTips]\n", argv[0]);
    return 1;
  }
  SDBG g;
  const auto t_load0 = std::chrono::steady_clock::now();
  g.LoadFromFile(argv[1]);
  fprintf(stderr, "ref_sdbg_dump: LoadFromFile %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load0).count());
  g_out = fopen(argv[2], "wb");
  if (!g_out) return 1;
  const auto &ct = g.content_;
  const uint64_t n = ct.meta.item_count();
  uint64_t meta[6] = {n, ct.meta.tip_count(), ct.meta.large_mul_count(), g.k_, ct.meta.words_per_tip_label(), ct.full_mul.empty() ? 0u : 1u};
  section("meta", meta, 8, 6);
  section("w", ct.w.data(), 8, ct.w.word_count());
  section("last", ct.last.data(), 8, ct.last.word_count());
  section("tip", ct.tip.data(), 8, ct.tip.word_count());
  section("invalid", g.invalid_.data_array_.data(), 8, g.invalid_.data_array_.size());
  std::vector<uint16_t> mul(n);
  std::vector<uint8_t> small(n);
  for (uint64_t i = 0; i < n; ++i) {
    mul[i] = g.EdgeMultiplicity(i);
    small[i] = mul[i] < kMaxSmallMul ? (uint8_t)mul[i] : kSmallMulSentinel;  // sdbg_raw_content.cpp:76-81
  }
  section("mul", mul.data(), 2, n);
  section("small_mul", small.data(), 1, n);
  section("tip_labels", ct.tip_lables.data(), 4, ct.tip_lables.size());
  std::vector<int64_t> lkt;
  for (auto &p : g.prefix_look_up_) {
    lkt.push_back(p.first);
    lkt.push_back(p.second);
  }
  section("prefix_lkt", lkt.data(), 8, lkt.size());
  section("f", g.f_, 8, kAlphabetSize + 2);
  section("rank_f", g.rank_f_, 8, kAlphabetSize + 2);
  dump_rs("rsw", g.rs_w_, 0, kWAlphabetSize, true);
  dump_rs("rslast", g.rs_last_, 1, 2, true);
  dump_rs("rstip", g.rs_is_tip_, 1, 2, false);
  // a functional sample on top of the tables: Forward/Backward of spread edges (sdbg.h:106-121)
  std::vector<int64_t> fb;
  for (uint64_t i = 0; i < n; i += (n / 997) + 1) {
    fb.push_back((int64_t)i);
    fb.push_back(g.IsValidEdge(i) ? (int64_t)g.Forward(i) : -2);
    fb.push_back((int64_t)g.Backward(i));
  }
  section("fwd_bwd", fb.data(), 8, fb.size());
  if (argc >= 4) {  // the reference's SdBG-level tip trimming (assembly/sdbg_pruning.cpp:147-179)
    const int max_tip_len = atoi(argv[3]);
    const auto t_tips0 = std::chrono::steady_clock::now();
    uint64_t n_tips = sdbg_pruning::RemoveTips(g, max_tip_len);
    fprintf(stderr, "ref_sdbg_dump: RemoveTips %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tips0).count());
    section("tips_removed", &n_tips, 8, 1);
    section("invalid_after_tips", g.invalid_.data_array_.data(), 8, g.invalid_.data_array_.size());
  }
  fclose(g_out);
  return 0;
}
