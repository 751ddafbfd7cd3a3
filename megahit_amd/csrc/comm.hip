// Multi-GPU SdBG construction behind the C ABI: communicator + the distributed drivers of the three sub-programs.
//
// One rank per GPU; ranks are threads of one process (mhx_core --gpus N) or separate processes (bench.py under
// torch.distributed.run).  The 65536 lv1 buckets are split into contiguous owner ranges; per stage every rank extracts
// the items of ITS reads, partitions them by owner, the items cross the node in one all-to-all, and every rank sorts
// and reduces the buckets it owns (the reference shards buckets over OpenMP threads instead:
// src/sorting/base_engine.cpp:213-223,318-327).  Records keyed by a read position — non-solid marks of stage 1, mercy
// candidates, first_0_out/last_0_in events of count — travel back to the rank that holds the read with a second
// all-to-all; nothing of the size of the global read set is ever allocated or reduced.
//
// Transports (function table mhx_transport):
//   rccl   RCCL called directly (dlopen of librccl; grouped ncclSend/ncclRecv in <= 256 MiB messages per pair and round
//          for the all-to-all, ncclAllGather / ncclAllReduce for counts and plans), on the engine's own HIP stream
//   local  ranks = threads of this process, any devices (also several ranks on ONE GPU: the tests): exchanges are
//          device-to-device copies out of the peers' send buffers between two barriers
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include "mhx_internal.h"

namespace mhx {

// ---- RCCL through dlopen: libmhx has no link-time dependency on it, and a process that already maps a librccl
// (PyTorch bundles one) keeps using that instance ----
struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> cand;
    if (const char *e = getenv("MHX_RCCL_LIB")) cand.push_back(e);
    {  // an instance that is already mapped into this process
      std::ifstream maps("/proc/self/maps");
      std::string line;
      while (std::getline(maps, line)) {
        const size_t p = line.find('/');
        if (p != std::string::npos && line.find("librccl.so", p) != std::string::npos) {
          cand.push_back(line.substr(p));
          break;
        }
      }
    }
    cand.push_back("librccl.so.1");
    cand.push_back("/opt/rocm/lib/librccl.so.1");
    cand.push_back("librccl.so");
    for (const std::string &c : cand) {
      api.h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.h) break;
    }
    if (!api.h) return;
#define MHX_SYM(name) api.name = reinterpret_cast<decltype(api.name)>(dlsym(api.h, "nccl" #name))
    MHX_SYM(GetUniqueId);
    MHX_SYM(CommInitRank);
    MHX_SYM(CommDestroy);
    MHX_SYM(AllReduce);
    MHX_SYM(AllGather);
    MHX_SYM(Send);
    MHX_SYM(Recv);
    MHX_SYM(GroupStart);
    MHX_SYM(GroupEnd);
    MHX_SYM(GetErrorString);
#undef MHX_SYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.AllGather || !api.Send || !api.Recv || !api.GroupStart ||
        !api.GroupEnd)
      api.h = nullptr;
  });
  if (!api.h) throw Error("RCCL is not available (librccl.so could not be loaded; set MHX_RCCL_LIB)");
  return api;
}
#define MHX_NCCL(expr)                                                                                  \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) {                                                                            \
      char buf_[512];                                                                                   \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr,                                       \
               mhx::rccl().GetErrorString ? mhx::rccl().GetErrorString(r_) : "rccl error", __FILE__, __LINE__); \
      throw mhx::Error(buf_);                                                                           \
    }                                                                                                   \
  } while (0)

// in-process group: the ranks are threads
struct LocalGroup {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  int refs = 0;
  std::vector<std::vector<uint64_t>> vals;      // per rank: a host vector being reduced / gathered
  std::vector<const char *> send_ptr;            // per rank: device send buffer of the running all-to-all
  std::vector<std::vector<uint64_t>> send_off;   // per rank: byte offsets of the per-peer segments (n + 1)
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

}  // namespace mhx

struct mhx_comm {
  int rank = 0, n = 1;
  mhx_ctx *ctx = nullptr;
  // rccl transport
  ncclComm_t nccl = nullptr;
  // local transport
  mhx::LocalGroup *grp = nullptr;
  // hosted transport: the caller moves the bytes (host memory) with whatever it has — e.g. torch.distributed over gloo
  mhx_host_transport hosted{};
  bool is_hosted = false;
  std::vector<char> h_send, h_recv;
  uint64_t max_msg_bytes = 1ull << 28;  // 256 MiB per message (one 16 GB message hung RCCL 2.26 on MI355X)
  // agreed layout
  uint64_t stride_bases = 0;
  // stages that were found to need no bucket-range passes since the last mhx_dist_setup (key: stage, k, m): the check costs
  // an all-reduce and a hipMemGetInfo per call; every rank sees the same calls, so the caches agree
  std::vector<uint64_t> one_pass_ok;
  // payload bytes this rank handed to OTHER ranks in all_to_all_v since the last reset (mhx_comm_bytes_sent: the tests bound the
  // bytes per foreign record with it, the bench prints it per step)
  uint64_t bytes_sent = 0;

  // ---- collectives on small host vectors ----
  void all_reduce(std::vector<uint64_t> &v, bool is_max) {
    if (n == 1) return;
    hipStream_t st = ctx->stream;
    if (is_hosted) {
      if (hosted.all_reduce_u64(hosted.user, v.data(), v.size(), is_max ? 1 : 0) != 0) throw mhx::Error("hosted transport: all_reduce failed");
      return;
    }
    if (nccl) {
      unsigned long long *d = ctx->ws("comm_small", v.size() * 8 + 64).as<unsigned long long>();
      MHX_HIP(hipMemcpyAsync(d, v.data(), v.size() * 8, hipMemcpyHostToDevice, st));
      MHX_NCCL(mhx::rccl().AllReduce(d, d, v.size(), ncclUint64, is_max ? ncclMax : ncclSum, nccl, st));
      MHX_HIP(hipMemcpyAsync(v.data(), d, v.size() * 8, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      return;
    }
    grp->vals[rank] = v;
    grp->barrier();
    for (size_t i = 0; i < v.size(); ++i) {
      uint64_t acc = is_max ? 0 : 0;
      for (int r = 0; r < n; ++r) acc = is_max ? std::max(acc, grp->vals[r][i]) : acc + grp->vals[r][i];
      v[i] = acc;
    }
    grp->barrier();
  }
  // send[p] = what this rank has for p  ->  recv[p] = what p has for this rank
  void all_to_all_counts(const std::vector<uint64_t> &send, std::vector<uint64_t> &recv) {
    recv.assign(n, 0);
    if (n == 1) {
      recv[0] = send[0];
      return;
    }
    hipStream_t st = ctx->stream;
    if (is_hosted) {  // the n x n count matrix as a sum of one-row matrices
      std::vector<uint64_t> m((size_t)n * n, 0);
      for (int p = 0; p < n; ++p) m[(size_t)rank * n + p] = send[p];
      all_reduce(m, false);
      for (int p = 0; p < n; ++p) recv[p] = m[(size_t)p * n + rank];
      return;
    }
    if (nccl) {
      unsigned long long *d = ctx->ws("comm_small", (size_t)(n + 1) * n * 8 + 64).as<unsigned long long>();
      MHX_HIP(hipMemcpyAsync(d, send.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
      MHX_NCCL(mhx::rccl().AllGather(d, d + n, (size_t)n, ncclUint64, nccl, st));  // row r = counts of rank r
      std::vector<uint64_t> m((size_t)n * n);
      MHX_HIP(hipMemcpyAsync(m.data(), d + n, (size_t)n * n * 8, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      for (int p = 0; p < n; ++p) recv[p] = m[(size_t)p * n + rank];
      return;
    }
    grp->vals[rank] = send;
    grp->barrier();
    for (int p = 0; p < n; ++p) recv[p] = grp->vals[p][rank];
    grp->barrier();
  }
  // device buffers with the per-peer segments back to back (peers ascending); counts in items
  // skip_self: this rank's own segment stays where it is (the receive buffer has no room for it: recv_counts[rank] is ignored)
  void all_to_all_v(const void *d_send, const std::vector<uint64_t> &send_counts, void *d_recv, const std::vector<uint64_t> &recv_counts,
                    uint32_t item_bytes, bool skip_self = false) {
    hipStream_t st = ctx->stream;
    std::vector<uint64_t> so(n + 1, 0), ro(n + 1, 0);
    for (int p = 0; p < n; ++p) {
      so[p + 1] = so[p] + send_counts[p] * item_bytes;
      ro[p + 1] = ro[p] + (skip_self && p == rank ? 0 : recv_counts[p]) * item_bytes;
    }
    const char *s = static_cast<const char *>(d_send);
    char *r = static_cast<char *>(d_recv);
    for (int p = 0; p < n; ++p)
      if (p != rank) bytes_sent += so[p + 1] - so[p];
    if (is_hosted && n > 1) {
      // own segment: a device copy; everything else through host memory and the caller's byte mover
      if (!skip_self && so[rank + 1] > so[rank])
        MHX_HIP(hipMemcpyAsync(r + ro[rank], s + so[rank], so[rank + 1] - so[rank], hipMemcpyDeviceToDevice, st));
      std::vector<uint64_t> sb(n), rb(n);
      uint64_t s_tot = 0, r_tot = 0;
      for (int p = 0; p < n; ++p) {
        sb[p] = p == rank ? 0 : so[p + 1] - so[p];
        rb[p] = p == rank ? 0 : ro[p + 1] - ro[p];
        s_tot += sb[p];
        r_tot += rb[p];
      }
      h_send.resize(s_tot);
      h_recv.resize(r_tot);
      uint64_t at = 0;
      for (int p = 0; p < n; ++p)
        if (sb[p]) {
          MHX_HIP(hipMemcpyAsync(h_send.data() + at, s + so[p], sb[p], hipMemcpyDeviceToHost, st));
          at += sb[p];
        }
      MHX_HIP(hipStreamSynchronize(st));
      if (hosted.all_to_all_bytes(hosted.user, h_send.data(), sb.data(), h_recv.data(), rb.data()) != 0) throw mhx::Error("hosted transport: all_to_all failed");
      at = 0;
      for (int p = 0; p < n; ++p)
        if (rb[p]) {
          MHX_HIP(hipMemcpyAsync(r + ro[p], h_recv.data() + at, rb[p], hipMemcpyHostToDevice, st));
          at += rb[p];
        }
      MHX_HIP(hipStreamSynchronize(st));
      return;
    }
    if (nccl || n == 1) {
      // own segment: a device copy; every other segment: point-to-point messages of at most max_msg_bytes, all pairs of a
      // round in one group (direct xGMI sends, not a ring).  Sender and receiver derive the same chunking from the counts.
      if (!skip_self && so[rank + 1] > so[rank])
        MHX_HIP(hipMemcpyAsync(r + ro[rank], s + so[rank], so[rank + 1] - so[rank], hipMemcpyDeviceToDevice, st));
      const uint64_t chunk = std::max<uint64_t>(item_bytes, max_msg_bytes / item_bytes * item_bytes);
      uint64_t rounds = 0;
      for (int p = 0; p < n; ++p)
        if (p != rank) rounds = std::max({rounds, mhx::div_ceil(so[p + 1] - so[p], chunk), mhx::div_ceil(ro[p + 1] - ro[p], chunk)});
      for (uint64_t c = 0; c < rounds; ++c) {
        MHX_NCCL(mhx::rccl().GroupStart());
        for (int d = 1; d < n; ++d) {
          const int to = (rank + d) % n, from = (rank - d + n) % n;
          uint64_t lo = so[to] + c * chunk, hi = std::min(so[to + 1], so[to] + (c + 1) * chunk);
          if (lo < hi) MHX_NCCL(mhx::rccl().Send(s + lo, hi - lo, ncclChar, to, nccl, st));
          lo = ro[from] + c * chunk;
          hi = std::min(ro[from + 1], ro[from] + (c + 1) * chunk);
          if (lo < hi) MHX_NCCL(mhx::rccl().Recv(r + lo, hi - lo, ncclChar, from, nccl, st));
        }
        MHX_NCCL(mhx::rccl().GroupEnd());
      }
      MHX_HIP(hipStreamSynchronize(st));
      return;
    }
    // local: publish the send buffer (complete: the stream is drained first), pull the segments addressed to this rank
    MHX_HIP(hipStreamSynchronize(st));
    grp->send_ptr[rank] = s;
    grp->send_off[rank] = so;
    grp->barrier();
    for (int p = 0; p < n; ++p) {
      const uint64_t bytes = ro[p + 1] - ro[p];
      if (bytes) MHX_HIP(hipMemcpyAsync(r + ro[p], grp->send_ptr[p] + grp->send_off[p][rank], bytes, hipMemcpyDeviceToDevice, st));
    }  // (skip_self: ro[rank + 1] == ro[rank], nothing is copied for this rank's own segment)
    MHX_HIP(hipStreamSynchronize(st));
    grp->barrier();  // nobody reuses a send buffer before every peer has read it
  }
  void barrier() {
    if (n == 1) return;
    if (nccl || is_hosted) {
      std::vector<uint64_t> one(1, 1);
      all_reduce(one, false);
    } else {
      grp->barrier();
    }
  }
};

namespace mhx {

// ---- the distributed drivers ----
static void move_items(mhx_ctx *c, mhx_comm *cm, const mhx_dist_items &it, const std::vector<uint64_t> &counts, uint64_t *n_recv) {
  std::vector<uint64_t> rc;
  cm->all_to_all_counts(counts, rc);
  uint64_t n = 0;
  for (uint64_t v : rc) n += v;
  void *recv = mhx_dist_recv_buffer(c, n, it.item_bytes);
  if (!recv) throw Error(mhx_last_error());
  cm->all_to_all_v(it.d_items, counts, recv, rc, it.item_bytes);
  *n_recv = n;
}
#define MHX_CK(call)                                  \
  do {                                                \
    if ((call) != 0) throw mhx::Error(mhx_last_error()); \
  } while (0)

static uint64_t exchange_stage(mhx_ctx *c, mhx_comm *cm, int stage, uint32_t k, uint32_t m) {
  mhx_dist_items it{};
  std::vector<uint64_t> counts(cm->n, 0);
  MHX_CK(mhx_dist_extract(c, stage, k, m, &it, counts.data()));
  uint64_t n = 0;
  move_items(c, cm, it, counts, &n);
  return n;
}
__global__ void k_bucket_bounds(const uint32_t *__restrict__ items, uint64_t n, int stride, uint64_t *__restrict__ bstart, int pbits);  // kmsort_emu.hip

// Stage 1 without the owner multisplit (no mercy candidates, 12-byte records, the bucket-streaming plan): every rank runs
// the SAME LSD passes as a single GPU does — two over the 16 bits of the lv1 bucket at 10 M reads per GPU, three over up to
// 24 prefix bits when the job's buckets are larger (the plan follows the density the ranks agreed on, mhx_ctx::s1_density) —
// they order its records by lv1 bucket, so the records of each owner's contiguous bucket range are contiguous already —
// sends those slices as they are, keeps its own slice in place, and the owner's group-by kernel reads a bucket of the
// plan's prefix as one sub-range per sender (k_s1_stream's sources) instead of sorting the received records again.  Replaces: owner histogram + owner scatter + a second histogram + two more passes over the
// received records (round 2: 91 ms against 53 ms for a single rank).  Returns false (nothing exchanged yet, the extracted
// items are in ws "items_a") when the plan does not apply on every rank.
static bool dist_s1_presorted(mhx_ctx *c, mhx_comm *cm, uint32_t k, uint32_t m, const StageItems &it, mhx_s1_result *r1) {
  const int n = cm->n, rank = cm->rank;
  std::vector<uint64_t> v{it.n, it.S == 3 ? 0ull : 1ull};
  cm->all_reduce(v, true);  // the largest local item count decides for everybody; any rank with other records vetoes
  if (v[1] || !c->opt("dist_presort", 1) || !s1_presort_applies(c, k, v[0])) return false;
  hipStream_t st = c->stream;
  uint32_t *a = c->work["items_a"].as<uint32_t>();
  uint32_t *b = c->ws("items_b", it.n * 12 + 64).as<uint32_t>();
  // (deferred items: the histograms taken at extraction belong to the plan of the agreed density, the one s1_presort makes)
  int pbits = 16;
  uint32_t *sorted = s1_presort(c, k, a, b, it.n, &pbits);
  uint64_t *d_bounds = c->ws("dist_bounds", (MHX_NUM_BUCKETS + 2) * 8).as<uint64_t>();
  std::vector<uint64_t> bounds(MHX_NUM_BUCKETS + 1, 0);
  if (it.n) {
    hipLaunchKernelGGL(k_bucket_bounds, dim3(MHX_NUM_BUCKETS / 256 + 1), dim3(256), 0, st, sorted, it.n, 3, d_bounds, 16);
    MHX_HIP(hipMemcpyAsync(bounds.data(), d_bounds, (MHX_NUM_BUCKETS + 1) * 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  std::vector<uint64_t> counts(n), rc;
  for (int p = 0; p < n; ++p) counts[p] = bounds[c->part_begin[p + 1]] - bounds[c->part_begin[p]];
  cm->all_to_all_counts(counts, rc);
  uint64_t n_recv = 0, total = 0;
  for (int p = 0; p < n; ++p) {
    if (p != rank) n_recv += rc[p];
    total += rc[p];
  }
  uint32_t *recv = c->ws("items_recv", n_recv * 12 + 64).as<uint32_t>();
  cm->all_to_all_v(sorted, counts, recv, rc, 12, true);
  S1Sources src;
  src.n = n;
  src.pbits = pbits;
  uint64_t at = 0;
  for (int p = 0; p < n; ++p) {
    if (p == rank) src.ptr.push_back(sorted + bounds[c->part_begin[rank]] * 3);
    else {
      src.ptr.push_back(recv + at * 3);
      at += rc[p];
    }
    src.count.push_back(rc[p]);
  }
  // the group-by's output regions: the local ping-pong buffer that does not hold the sorted records, if it is large enough
  uint32_t *other = sorted == a ? b : a;
  src.spare = total <= it.n ? other : c->ws("s1_spare", total * 12 + 64).as<uint32_t>();
  s1_process(c, k, m, 0, nullptr, nullptr, total, r1, &src);
  c->last_s1_plan += " [pre-sorted exchange]";
  return true;
}

// Stage 1 on super-k-mer records over several GPUs (round 6, s1_skm.hip): every rank makes the 16-byte records of ITS reads — the windows of a
// read that share a minimizer are one record: 4.6 bytes per window, where the pre-sorted exchange above moves a 12-byte record per window —
// orders them by minimizer bin (the passes of a single GPU), sends each owner of a contiguous range of BINS its slice as it is, and the
// owner's group-by (k_s1_skm) reads a bin as one sub-range per sender.  Marks leave the owners as lists of global positions for the read
// owners, aggregated stage-2 items stay for stage 2's exchange, as before.  The ranks agree on the number of bins (from the size of the
// whole job), on taking this form at all (any rank may veto: shape, a memory plan, more than kSkmSrcMax ranks) and on giving it up (a
// record array or an output region that overflowed, a bin of low-complexity reads): -> false, nothing of the pass published, the
// pre-sorted exchange of the prefix plan runs.
static bool dist_s1_skm(mhx_ctx *c, mhx_comm *cm, uint32_t k, uint32_t m, mhx_s1_result *r1, std::string *why) {
  const int n = cm->n, rank = cm->rank;
  const SeqSet &sq = c->seqs;
  const uint64_t n_win_local = sq.n_bases > sq.n_seqs * (uint64_t)k ? sq.n_bases - sq.n_seqs * (uint64_t)k : 0;
  std::vector<uint64_t> v{s1_skm_dist_applies(c, k, m) ? 0ull : 1ull, n_win_local};
  cm->all_reduce(v, true);  // any rank that cannot vetoes; the largest shard sizes the bins
  const long long knob = c->opt("s1_skm", 1);
  if (v[0] || (knob < 2 && v[1] * (uint64_t)n < (uint64_t)c->opt("s1_skm_min_windows", 1 << 22))) return false;
  int bin_bits = 16;
  while (bin_bits < 20 && (double)v[1] * (double)n * 0.29 / (double)(1ull << bin_bits) > 8192.0) ++bin_bits;
  if (const long long fb = c->opt("s1_skm_bin_bits", 0)) bin_bits = (int)std::min<long long>(20, std::max<long long>(8, fb));
  hipStream_t st = c->stream;
  SkmFront f{};
  c->gen_first_pass = nullptr;
  c->pre_hist_buf = nullptr;
  std::vector<uint64_t> fail{s1_skm_front(c, k, &f, 0, 1, bin_bits) ? 0ull : 1ull};
  cm->all_reduce(fail, true);
  if (fail[0]) {
    *why = f.n_records ? "a rank's bin of low-complexity reads" : "a rank's record array overflowed";
    return false;
  }
  // the owners' bin ranges and this rank's slice for each
  const uint32_t n_bins = f.n_bins;
  std::vector<uint64_t> cut(n + 1, 0), at_cut(n + 1, 0);
  for (int p = 0; p <= n; ++p) cut[p] = (uint64_t)n_bins * p / n;
  for (int p = 0; p <= n; ++p) MHX_HIP(hipMemcpyAsync(&at_cut[p], f.src_bounds[0] + cut[p], 8, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  std::vector<uint64_t> counts(n), rc;
  for (int p = 0; p < n; ++p) counts[p] = at_cut[p + 1] - at_cut[p];
  cm->all_to_all_counts(counts, rc);
  uint64_t n_recv = 0, total = 0;
  for (int p = 0; p < n; ++p) {
    if (p != rank) n_recv += rc[p];
    total += rc[p];
  }
  uint4 *recv = c->ws("items_recv", n_recv * 16 + 64).as<uint4>();
  cm->all_to_all_v(f.src[0], counts, recv, rc, 16, true);
  // the sources of this owner's group-by: its own slice where it lies, the others in the receive buffer; where the bins start in each
  SkmFront o = f;
  o.n_src = n;
  uint64_t *sb = c->ws("skm_src_bounds", (size_t)n * ((size_t)n_bins + 1) * 8 + 64).as<uint64_t>();
  uint64_t at = 0;
  for (int p = 0; p < n; ++p) {
    const uint4 *ptr;
    if (p == rank) ptr = f.src[0] + at_cut[rank];
    else {
      ptr = recv + at;
      at += rc[p];
    }
    o.src[p] = ptr;
    o.src_bounds[p] = sb + (size_t)p * ((size_t)n_bins + 1);
    s1_skm_bounds_of(c, ptr, rc[p], n_bins, sb + (size_t)p * ((size_t)n_bins + 1));
  }
  o.bin_lo = (uint32_t)cut[rank];
  o.bin_hi = (uint32_t)cut[rank + 1];
  o.n_records = total;
  mhx_s1_result rp{};
  std::vector<uint64_t> fail2{s1_skm_owner(c, k, m, o, &rp) ? 0ull : 1ull};
  cm->all_reduce(fail2, true);
  if (fail2[0]) {
    *why = "an owner's output region overflowed";
    return false;
  }
  *r1 = rp;
  return true;
}

// `count` the same way (round 6): every rank makes its 12-byte count records in the first pass of the single-GPU sort plan (CountGenT, also
// under the bucket filter of a pass), orders them by the plan's prefix, sends each owner its contiguous slice, and the owner's bucket
// streaming (k_s1_stream<COUNT>) reads a bucket as one sub-range per sender; the events that move first_0_out / last_0_in of reads held
// by other ranks leave the kernel as a list and are routed like the tile path's (MHX_ROUTE_COUNT_EVENTS).  -> false: the ranks do not all
// take this form (nothing exchanged), or an owner's streaming gave up (its state is as before the pass): the classic exchange runs.
// Reference: KmerCounter over lv1 bucket ranges, kmer_counter.cpp:158-381 + base_engine.cpp:176-201.
static bool dist_count_presorted(mhx_ctx *c, mhx_comm *cm, uint32_t k, uint32_t m, mhx_count_result *out) {
  const int n = cm->n, rank = cm->rank;
  std::vector<uint64_t> v{count_presort_applies(c, k, m) ? 0ull : 1ull};
  cm->all_reduce(v, true);  // any rank that cannot vetoes
  if (v[0]) return false;
  hipStream_t st = c->stream;
  uint64_t n_local = 0;
  uint32_t *other = nullptr;
  int pbits = 16;
  uint32_t *sorted = count_presort(c, k, &n_local, &other, &pbits);
  uint64_t *d_bounds = c->ws("dist_bounds", (MHX_NUM_BUCKETS + 2) * 8).as<uint64_t>();
  std::vector<uint64_t> bounds(MHX_NUM_BUCKETS + 1, 0);
  if (n_local) {
    hipLaunchKernelGGL(k_bucket_bounds, dim3(MHX_NUM_BUCKETS / 256 + 1), dim3(256), 0, st, sorted, n_local, 3, d_bounds, 16);
    MHX_HIP(hipMemcpyAsync(bounds.data(), d_bounds, (MHX_NUM_BUCKETS + 1) * 8, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
  }
  std::vector<uint64_t> counts(n), rc;
  for (int p = 0; p < n; ++p) counts[p] = bounds[c->part_begin[p + 1]] - bounds[c->part_begin[p]];
  cm->all_to_all_counts(counts, rc);
  uint64_t n_recv = 0, total = 0;
  for (int p = 0; p < n; ++p) {
    if (p != rank) n_recv += rc[p];
    total += rc[p];
  }
  uint32_t *recv = c->ws("items_recv", n_recv * 12 + 64).as<uint32_t>();
  cm->all_to_all_v(sorted, counts, recv, rc, 12, true);
  S1Sources src;
  src.n = n;
  src.pbits = pbits;
  uint64_t at = 0;
  for (int p = 0; p < n; ++p) {
    if (p == rank) src.ptr.push_back(sorted + bounds[c->part_begin[rank]] * 3);
    else {
      src.ptr.push_back(recv + at * 3);
      at += rc[p];
    }
    src.count.push_back(rc[p]);
  }
  // the edge regions: the local ping-pong buffer that does not hold the sorted records, if it is large enough
  src.spare = total <= n_local ? other : c->ws("s1_spare", total * 12 + 64).as<uint32_t>();
  mhx_count_result r{};
  std::vector<uint64_t> fail{count_process_presorted(c, k, m, src, &r) == 0 ? 0ull : 1ull};
  cm->all_reduce(fail, true);
  if (fail[0]) return false;  // (every rank's first_0_out / last_0_in / histogram are as before this pass: count_run_stream restores them)
  *out = r;
  return true;
}

static uint64_t route(mhx_ctx *c, mhx_comm *cm, int which) {
  mhx_dist_items it{};
  std::vector<uint64_t> counts(cm->n, 0);
  MHX_CK(mhx_dist_route_records(c, which, cm->stride_bases, &it, counts.data()));
  uint64_t n = 0;
  move_items(c, cm, it, counts, &n);
  MHX_CK(mhx_dist_apply_routed(c, which, n));
  return n;
}

// ---- bucket-range passes on several GPUs (the reference's lv1 passes, base_engine.cpp:54-141,254-281, per rank) ----
// When the items a rank would own (or extract) in one go do not fit its free memory, the stage runs P times; pass i
// handles, of EVERY owner's bucket range, the i-th of P sub-ranges of about equal weight (global lv1 histogram), so that
// all ranks stay busy in every pass.  Every rank extracts only the kept buckets of its reads (mhx_set_bucket_filter).
struct DistPasses {
  int n = 1;
  std::vector<std::vector<uint8_t>> keep;   // [pass][bucket]
  std::vector<uint64_t> expected;           // [pass]: local items in the kept buckets
};
static DistPasses plan_dist_passes(mhx_ctx *c, mhx_comm *cm, int stage, uint32_t k, uint32_t m, size_t item_bytes) {
  DistPasses dp;
  // bytes per item a pass needs on a rank: the extracted copy + the send copy / sorted copy + receive buffer + its sort twin
  const double per_item = 4.0 * (double)item_bytes + 1.0;
  uint64_t max_items = (uint64_t)c->opt("dist_max_items", 0);
  if (const char *e = getenv("MHX_MAX_ITEMS")) max_items = strtoull(e, nullptr, 10);
  const uint64_t plan_key = ((uint64_t)stage << 48) | ((uint64_t)k << 16) | (uint64_t)(m & 0xFFFFu);
  if (!max_items && !getenv("MHX_FREE_BYTES")) {
    if (std::find(cm->one_pass_ok.begin(), cm->one_pass_ok.end(), plan_key) != cm->one_pass_ok.end()) return dp;
    // the usual case decided without a scan of the reads: an upper bound of the items of the whole job (stage 1 / count:
    // one per base + 4 per read; stage 2 per occurrence, seq2sdbg: ~2 per base), twice the fair share per rank
    const mhx::SeqSet &s = c->seqs;
    // (stage 2 after a stage 1 with k <= 22 sorts the aggregated items: one or two per DISTINCT solid (k+1)-mer)
    const double per_base = stage == MHX_STAGE_S2 ? (m > 1 && k <= 22 ? 0.5 : 2.2) : (stage == MHX_STAGE_SEQ2SDBG ? 2.2 : 1.0);
    std::vector<uint64_t> v{(uint64_t)(per_base * (double)s.n_bases) + 4 * s.n_seqs, ~0ull - mhx_device_free_bytes(c)};
    std::vector<uint64_t> mx = v;
    cm->all_reduce(mx, true);
    const double fit = (double)(~0ull - mx[1]) * 0.8 / per_item;
    if (2.0 * (double)mx[0] <= fit) {  // (the largest local bound stands in for every rank's share)
      cm->one_pass_ok.push_back(plan_key);
      return dp;
    }
  }
  std::vector<uint64_t> hist(MHX_NUM_BUCKETS, 0);
  const uint64_t saved = c->global_bases;
  MHX_CK(mhx_bucket_histogram(c, stage, k, m, hist.data()));
  c->global_bases = saved;
  const std::vector<uint64_t> local = hist;
  cm->all_reduce(hist, false);
  if (!max_items) {
    std::vector<uint64_t> fr{~0ull - mhx_device_free_bytes(c)};
    cm->all_reduce(fr, true);  // the rank with the least free memory decides
    double free_bytes = (double)(~0ull - fr[0]);
    if (const char *e = getenv("MHX_FREE_BYTES")) free_bytes = atof(e);
    max_items = (uint64_t)std::max(1.0, free_bytes * 0.8 / per_item);
  }
  uint64_t worst = 0, local_total = 0;
  for (int p = 0; p < cm->n; ++p) {
    uint64_t owned = 0;
    for (uint32_t b = c->part_begin[p]; b < c->part_begin[p + 1]; ++b) owned += hist[b];
    worst = std::max(worst, owned);
  }
  for (uint64_t v : local) local_total += v;
  std::vector<uint64_t> w{std::max(worst, local_total)};
  cm->all_reduce(w, true);
  const uint64_t P = std::min<uint64_t>(4096, (w[0] + max_items - 1) / max_items);
  if (P <= 1) {
    if (!c->opt("dist_max_items", 0) && !getenv("MHX_MAX_ITEMS") && !getenv("MHX_FREE_BYTES")) cm->one_pass_ok.push_back(plan_key);
    return dp;
  }
  dp.n = (int)P;
  dp.keep.assign(P, std::vector<uint8_t>(MHX_NUM_BUCKETS, 0));
  dp.expected.assign(P, 0);
  for (int p = 0; p < cm->n; ++p) {
    long double owned = 0, acc = 0;
    for (uint32_t b = c->part_begin[p]; b < c->part_begin[p + 1]; ++b) owned += hist[b];
    uint64_t i = 0;
    for (uint32_t b = c->part_begin[p]; b < c->part_begin[p + 1]; ++b) {
      while (i + 1 < P && acc >= owned * (long double)(i + 1) / (long double)P) ++i;
      dp.keep[i][b] = 1;
      dp.expected[i] += local[b];
      acc += hist[b];
    }
  }
  return dp;
}
static void set_pass(mhx_ctx *c, const DistPasses &dp, int i, bool accumulate) {
  if (dp.n <= 1) return;
  MHX_CK(mhx_set_bucket_filter(c, dp.keep[i].data(), dp.expected[i], 0, accumulate && i > 0 ? 1 : 0));
}
static void clear_pass(mhx_ctx *c, const DistPasses &dp) {
  if (dp.n > 1) MHX_CK(mhx_set_bucket_filter(c, nullptr, 0, 0, 0));
}
static void add_result(mhx_sdbg_result &a, const mhx_sdbg_result &b) {
  a.n_items += b.n_items;
  a.n_sdbg += b.n_sdbg;
  a.n_tips += b.n_tips;
  a.n_large += b.n_large;
  a.sdbg_bytes += b.sdbg_bytes;
  a.words_per_tip_label = b.words_per_tip_label;
  a.item_words = b.item_words;
}

}  // namespace mhx

extern "C" {

#define MHX_TRYC(body)                 \
  try {                                \
    body;                              \
    return 0;                          \
  } catch (const std::exception &e) {  \
    mhx::set_error("%s", e.what());    \
    return -1;                         \
  }

int mhx_comm_unique_id(void *id) {
  MHX_TRYC({
    static_assert(sizeof(ncclUniqueId) <= MHX_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    MHX_NCCL(mhx::rccl().GetUniqueId(&u));
    memset(id, 0, MHX_COMM_ID_BYTES);
    memcpy(id, &u, sizeof u);
  })
}

mhx_comm *mhx_comm_init_rank(mhx_ctx *c, const void *id, int rank, int n_ranks) {
  try {
    if (!c || n_ranks < 1 || n_ranks > 256 || rank < 0 || rank >= n_ranks) throw mhx::Error("comm_init_rank: bad arguments");
    MHX_HIP(hipSetDevice(c->device));
    mhx_comm *cm = new mhx_comm;
    cm->rank = rank;
    cm->n = n_ranks;
    cm->ctx = c;
    if (const char *e = getenv("MHX_COMM_MAX_MSG")) cm->max_msg_bytes = std::max<uint64_t>(4096, strtoull(e, nullptr, 10));
    if (n_ranks > 1 || id) {
      if (!id) {
        delete cm;
        throw mhx::Error("comm_init_rank: a unique id is needed for more than one rank");
      }
      ncclUniqueId u;
      memcpy(&u, id, sizeof u);
      MHX_NCCL(mhx::rccl().CommInitRank(&cm->nccl, n_ranks, u, rank));
    }
    return cm;
  } catch (const std::exception &e) {
    mhx::set_error("%s", e.what());
    return nullptr;
  }
}

mhx_comm *mhx_comm_init_hosted(mhx_ctx *c, int rank, int n_ranks, const mhx_host_transport *t) {
  try {
    if (!c || !t || !t->all_reduce_u64 || !t->all_to_all_bytes || n_ranks < 1 || n_ranks > 256 || rank < 0 || rank >= n_ranks)
      throw mhx::Error("comm_init_hosted: bad arguments");
    mhx_comm *cm = new mhx_comm;
    cm->rank = rank;
    cm->n = n_ranks;
    cm->ctx = c;
    cm->hosted = *t;
    cm->is_hosted = true;
    return cm;
  } catch (const std::exception &e) {
    mhx::set_error("%s", e.what());
    return nullptr;
  }
}

int mhx_comm_local_group(int n_ranks, mhx_ctx *const *ctxs, mhx_comm **out) {
  MHX_TRYC({
    if (n_ranks < 1 || n_ranks > 256 || !ctxs || !out) throw mhx::Error("comm_local_group: bad arguments");
    mhx::LocalGroup *g = new mhx::LocalGroup;
    g->n = n_ranks;
    g->refs = n_ranks;
    g->vals.resize(n_ranks);
    g->send_ptr.assign(n_ranks, nullptr);
    g->send_off.resize(n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
      mhx_comm *cm = new mhx_comm;
      cm->rank = r;
      cm->n = n_ranks;
      cm->ctx = ctxs[r];
      cm->grp = g;
      out[r] = cm;
    }
  })
}

void mhx_comm_destroy(mhx_comm *cm) {
  if (!cm) return;
  if (cm->nccl && mhx::rccl().CommDestroy) mhx::rccl().CommDestroy(cm->nccl);
  if (cm->grp) {
    bool last;
    {
      std::lock_guard<std::mutex> lk(cm->grp->mu);
      last = --cm->grp->refs == 0;
    }
    if (last) delete cm->grp;
  }
  delete cm;
}
int mhx_comm_rank(const mhx_comm *cm) { return cm ? cm->rank : -1; }
int mhx_comm_size(const mhx_comm *cm) { return cm ? cm->n : -1; }
int mhx_comm_barrier(mhx_comm *cm) { MHX_TRYC(cm->barrier()) }
uint64_t mhx_comm_bytes_sent(mhx_comm *cm, int reset) {
  if (!cm) return 0;
  const uint64_t v = cm->bytes_sent;
  if (reset) cm->bytes_sent = 0;
  return v;
}
int mhx_comm_all_reduce_u64(mhx_comm *cm, uint64_t *values, uint64_t n, int is_max) {
  MHX_TRYC({
    std::vector<uint64_t> v(values, values + n);
    cm->all_reduce(v, is_max != 0);
    memcpy(values, v.data(), n * 8);
  })
}

int mhx_dist_setup(mhx_ctx *c, mhx_comm *cm, int balance_stage, uint32_t k, uint32_t min_count) {
  MHX_TRYC({
    MHX_HIP(hipSetDevice(c->device));
    const int n = cm->n;
    std::vector<uint32_t> begin(n + 1, 0);
    for (int r = 0; r <= n; ++r) begin[r] = (uint32_t)(((uint64_t)MHX_NUM_BUCKETS * r) / n);
    if (balance_stage && n > 1) {
      // contiguous bucket ranges of ~equal weight from the all-reduced lv1 bucket histogram (the reference balances its
      // size-sorted bucket list the same way: base_engine.cpp:231-252)
      std::vector<uint64_t> hist(MHX_NUM_BUCKETS, 0);
      c->global_bases = 0;  // the histogram is a local scan
      MHX_CK(mhx_bucket_histogram(c, balance_stage, k, min_count, hist.data()));
      cm->all_reduce(hist, false);
      long double total = 0;
      for (uint64_t v : hist) total += v;
      long double acc = 0;
      int r = 1;
      for (uint32_t b = 0; b < MHX_NUM_BUCKETS && r < n; ++b) {
        while (r < n && acc >= total * r / n) begin[r++] = b;
        acc += hist[b];
      }
      while (r < n) begin[r++] = MHX_NUM_BUCKETS;
      begin[n] = MHX_NUM_BUCKETS;
      for (int i = 1; i <= n; ++i) begin[i] = std::max(begin[i], begin[i - 1]);
    }
    MHX_CK(mhx_set_partition(c, cm->rank, n, begin.data()));
    // global read layout: rank r's bases start at r * stride, stride = the largest local read set rounded up to 64
    std::vector<uint64_t> nb(1, c->seqs.n_bases);
    cm->all_reduce(nb, true);
    cm->stride_bases = (nb[0] + 63) / 64 * 64;
    if (!cm->stride_bases) cm->stride_bases = 64;
    {  // a multiple of 2^j with n * stride <= 2^(j+8): 8 position bits then tell the ranks apart (capi.hip, MHX_ROUTE_S1_MARKS)
      int j = 6;
      auto up = [&](int jj) { return (cm->stride_bases + (1ull << jj) - 1) >> jj << jj; };
      while ((uint64_t)n * up(j) > (1ull << (j + 8))) ++j;
      cm->stride_bases = up(j);
    }
    MHX_CK(mhx_set_global_layout(c, (uint64_t)cm->rank * cm->stride_bases, (uint64_t)n * cm->stride_bases));
    c->options["dist_sparse_marks"] = 1;
    cm->one_pass_ok.clear();
  })
}

int mhx_dist_read2sdbg(mhx_ctx *c, mhx_comm *cm, uint32_t k, uint32_t min_count, int need_mercy, mhx_s1_result *out1,
                       mhx_sdbg_result *out2, uint64_t *num_mercy) {
  MHX_TRYC({
    MHX_HIP(hipSetDevice(c->device));
    if (!c->global_bases || !cm->stride_bases) throw mhx::Error("dist_read2sdbg: call mhx_dist_setup first");
    mhx_s1_result r1{};
    if (num_mercy) *num_mercy = 0;
    if (min_count > 1) {  // stage 1 is skipped when every edge is solid (main_sdbg_build.cpp:139-147)
      if (c->work.find("owner_lut") == c->work.end()) throw mhx::Error("dist_read2sdbg: call mhx_dist_setup first");
      const int st1 = need_mercy ? MHX_STAGE_S1_MERCY : MHX_STAGE_S1;
      const mhx::DistPasses dp = mhx::plan_dist_passes(c, cm, st1, k, min_count, (size_t)mhx::s1_stride(k, mhx::s1_compact(c, k, need_mercy)) * 4);
      uint64_t n_items_all = 0;
      {  // the records per lv1 bucket of the whole job, agreed on by the ranks: every rank makes the same stage-1 sort plan from it
        const mhx::SeqSet &sq = c->seqs;
        const uint64_t local = sq.fixed_len >= k + 1 ? sq.n_seqs * (uint64_t)(sq.fixed_len - k + 4) : sq.n_bases + 4 * sq.n_seqs;
        std::vector<uint64_t> mx{local};
        cm->all_reduce(mx, true);
        c->s1_density = std::max(1.0, (double)mx[0] * (double)cm->n / (double)MHX_NUM_BUCKETS);
      }
      for (int pass = 0; pass < dp.n; ++pass) {
        mhx::set_pass(c, dp, pass, true);
        mhx_s1_result rp{};
        bool done = false;
        std::string skm_why;
        if (!need_mercy && dp.n == 1) done = mhx::dist_s1_skm(c, cm, k, min_count, &rp, &skm_why);  // super-k-mer records, exchanged by bin
        if (!need_mercy && !done) {
          // (the pre-sort's first pass may make the records itself — and drop those of the buckets a pass leaves out: then
          // "items_a" holds nothing yet — s1.hip, S1GenT)
          c->gen_first_pass = nullptr;
          c->s1_defer_items = c->opt("dist_presort", 1) != 0;
          mhx::StageItems it = mhx::extract_stage(c, MHX_STAGE_S1, k, min_count);
          c->s1_defer_items = false;
          done = mhx::dist_s1_presorted(c, cm, k, min_count, it, &rp);
          if (!done) {  // the classic exchange of the items extracted above: owner multisplit, all-to-all, sort at the owner
            if (c->gen_first_pass) {  // ... which were deferred to a sort that will not happen: make them now
              c->gen_first_pass = nullptr;
              it = mhx::extract_stage(c, MHX_STAGE_S1, k, min_count);
            }
            c->pre_hist_buf = nullptr;
            mhx_dist_items di{};
            std::vector<uint64_t> counts(cm->n, 0);
            uint32_t *send = c->ws("items_send", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
            mhx::partition_by_owner(c, c->work["items_a"].as<uint32_t>(), send, it.n, it.S, c->work["owner_lut"].as<uint8_t>(), c->n_parts, counts.data());
            di.d_items = send;
            di.n_items = it.n;
            di.item_bytes = (uint32_t)it.S * 4;
            uint64_t n1 = 0;
            mhx::move_items(c, cm, di, counts, &n1);
            MHX_CK(mhx_dist_process_s1(c, k, min_count, 0, n1, &rp));
          }
        } else if (need_mercy) {
          const uint64_t n1 = mhx::exchange_stage(c, cm, MHX_STAGE_S1_MERCY, k, min_count);
          MHX_CK(mhx_dist_process_s1(c, k, min_count, need_mercy, n1, &rp));
        }
        if (!skm_why.empty()) c->last_s1_plan += " [super-k-mer records given up: " + skm_why + "]";
        n_items_all += rp.n_items;
        r1 = rp;
      }
      mhx::clear_pass(c, dp);
      c->s1_density = 0;
      r1.n_items = n_items_all;
      // the marks of the NON-solid (k+1)-mer occurrences of the owned buckets -> the ranks that hold those reads, which
      // derive is_solid = "a (k+1)-mer starts here and it is not marked" for their reads
      mhx::route(c, cm, MHX_ROUTE_S1_MARKS);
      r1.n_solid = c->dist_local_solid;
      if (need_mercy) {  // candidates -> read owners; the mercy block of Read2SdbgS2::Initialize runs there
        mhx::route(c, cm, MHX_ROUTE_MERCY_CAND);
        uint64_t nm = 0;
        MHX_CK(mhx_read2sdbg_add_mercy(c, k, &nm));
        if (num_mercy) *num_mercy = nm;
      }
    }
    mhx_sdbg_result r2{};
    {
      const size_t ib2 = (min_count > 1 && k <= 22) ? 8 : (size_t)mhx::s2_stride(k) * 4;
      const mhx::DistPasses dp = mhx::plan_dist_passes(c, cm, MHX_STAGE_S2, k, min_count, ib2);
      for (int pass = 0; pass < dp.n; ++pass) {
        mhx::set_pass(c, dp, pass, false);
        const uint64_t n2 = mhx::exchange_stage(c, cm, MHX_STAGE_S2, k, min_count);
        mhx_sdbg_result rp{};
        MHX_CK(mhx_dist_process_s2(c, k, n2, &rp));
        if (dp.n > 1) mhx::sdbg_accumulate(c, pass == 0);
        mhx::add_result(r2, rp);
      }
      mhx::clear_pass(c, dp);
      if (dp.n > 1) mhx::sdbg_publish_accumulated(c);
    }
    if (out1) *out1 = r1;
    if (out2) *out2 = r2;
  })
}

int mhx_dist_count(mhx_ctx *c, mhx_comm *cm, uint32_t k, uint32_t min_count, mhx_count_result *out) {
  MHX_TRYC({
    MHX_HIP(hipSetDevice(c->device));
    if (!c->global_bases || !cm->stride_bases) throw mhx::Error("dist_count: call mhx_dist_setup first");
    const mhx::DistPasses dp = mhx::plan_dist_passes(c, cm, MHX_STAGE_COUNT, k, min_count, (size_t)mhx::count_stride(k) * 4);
    mhx_count_result r{};
    hipStream_t st = c->stream;
    uint64_t acc_edge_bytes = 0;
    {  // the records per lv1 bucket of the whole job, agreed on by the ranks: every rank makes the same sort plan from it
      const mhx::SeqSet &sq = c->seqs;
      const uint64_t local = sq.fixed_len >= k + 1 ? sq.n_seqs * (uint64_t)(sq.fixed_len - k) : sq.n_bases;
      std::vector<uint64_t> mx{local};
      cm->all_reduce(mx, true);
      c->s1_density = std::max(1.0, (double)mx[0] * (double)cm->n / (double)MHX_NUM_BUCKETS);
    }
    for (int pass = 0; pass < dp.n; ++pass) {
      mhx::set_pass(c, dp, pass, true);
      mhx_count_result rp{};
      if (!mhx::dist_count_presorted(c, cm, k, min_count, &rp)) {  // the classic exchange: 16-byte items, owner multisplit, tile path at the owner
        c->gen_first_pass = nullptr;
        c->pre_hist_buf = nullptr;
        const uint64_t n = mhx::exchange_stage(c, cm, MHX_STAGE_COUNT, k, min_count);
        MHX_CK(mhx_dist_process_count(c, k, min_count, n, &rp));
      }
      mhx::route(c, cm, MHX_ROUTE_COUNT_EVENTS);
      if (dp.n > 1) {  // the solid edges and their per-bucket numbers of every pass, kept on the device (a bucket belongs to one pass)
        mhx::DevBuf &src = c->results[MHX_BUF_EDGES];
        mhx::DevBuf &dst = mhx::grow_preserving(c, c->work["acc_edges"], acc_edge_bytes + src.used + 64, acc_edge_bytes);
        if (src.used) MHX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(dst.p) + acc_edge_bytes, src.p, src.used, hipMemcpyDeviceToDevice, st));
        acc_edge_bytes += src.used;
        unsigned long long *ab = c->ws("acc_edge_buckets", MHX_NUM_BUCKETS * 8).as<unsigned long long>();
        if (pass == 0) MHX_HIP(hipMemsetAsync(ab, 0, MHX_NUM_BUCKETS * 8, st));
        hipLaunchKernelGGL(mhx::k_add_u64, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, ab, c->results[MHX_BUF_BUCKET_COUNT].as<unsigned long long>(),
                           MHX_NUM_BUCKETS);
        MHX_HIP(hipStreamSynchronize(st));
      }
      r.n_items += rp.n_items;
      r.n_distinct += rp.n_distinct;
      r.n_edges += rp.n_edges;
      r.words_per_edge = rp.words_per_edge;
    }
    mhx::clear_pass(c, dp);
    c->s1_density = 0;
    if (dp.n > 1) {
      std::swap(c->results[MHX_BUF_EDGES].p, c->work["acc_edges"].p);
      std::swap(c->results[MHX_BUF_EDGES].cap, c->work["acc_edges"].cap);
      c->results[MHX_BUF_EDGES].used = acc_edge_bytes;
      std::swap(c->results[MHX_BUF_BUCKET_COUNT].p, c->work["acc_edge_buckets"].p);
      std::swap(c->results[MHX_BUF_BUCKET_COUNT].cap, c->work["acc_edge_buckets"].cap);
      c->results[MHX_BUF_BUCKET_COUNT].used = MHX_NUM_BUCKETS * 8;
    }
    if (out) *out = r;
  })
}

int mhx_dist_gen_mercy_edges(mhx_ctx *c, mhx_comm *cm, uint32_t k, const uint32_t *cand_packed, uint64_t cand_words, uint64_t n_cand,
                             const uint64_t *cand_start, uint64_t *num_mercy) {
  MHX_TRYC({
    MHX_HIP(hipSetDevice(c->device));
    mhx::MercyShare share;
    share.my_part = cm->rank;
    share.n_parts = cm->n;
    share.reduce_flags = [&](uint8_t *d_flags, uint64_t nf) {
      // flags are has_in | has_out << 1 per base position; OR over the ranks = the MAXIMUM of 0/1 fields — exact for any number
      // of ranks (a sum of 8-bit fields wraps at 256 ranks): four positions per 64-bit word, 8 bits per flag bit; the maximum of
      // the packed words is taken field by field below, from one reduction per field pair
      std::vector<uint8_t> h(nf);
      MHX_HIP(hipMemcpyAsync(h.data(), d_flags, nf, hipMemcpyDeviceToHost, c->stream));
      MHX_HIP(hipStreamSynchronize(c->stream));
      std::vector<uint64_t> v((nf + 3) / 4, 0);
      for (uint64_t i = 0; i < nf; ++i) v[i / 4] |= (uint64_t)((h[i] & 1u) | ((h[i] & 2u) << 7)) << (16 * (i % 4));
      if (cm->n <= 255) {
        cm->all_reduce(v, false);  // (no field can wrap: a sum is one reduction)
      } else {  // every field on its own, as a maximum
        std::vector<uint64_t> acc(v.size());
        std::vector<uint64_t> one(v.size());
        for (int f = 0; f < 8; ++f) {
          for (size_t i = 0; i < v.size(); ++i) one[i] = (v[i] >> (8 * f)) & 0xFFull;
          cm->all_reduce(one, true);
          for (size_t i = 0; i < v.size(); ++i) acc[i] |= one[i] << (8 * f);
        }
        v.swap(acc);
      }
      for (uint64_t i = 0; i < nf; ++i) {
        const uint64_t x = v[i / 4] >> (16 * (i % 4));
        h[i] = (uint8_t)(((x & 0xFFu) ? 1u : 0u) | (((x >> 8) & 0xFFu) ? 2u : 0u));
      }
      MHX_HIP(hipMemcpyAsync(d_flags, h.data(), nf, hipMemcpyHostToDevice, c->stream));
      MHX_HIP(hipStreamSynchronize(c->stream));
    };
    uint64_t nm = 0;
    mhx::run_gen_mercy(c, k, cand_packed, cand_words, n_cand, cand_start, &nm, &share);
    if (num_mercy) *num_mercy = nm;
  })
}

int mhx_dist_seq2sdbg(mhx_ctx *c, mhx_comm *cm, uint32_t k, mhx_sdbg_result *out) {
  MHX_TRYC({
    MHX_HIP(hipSetDevice(c->device));
    if (c->part_begin.empty()) throw mhx::Error("dist_seq2sdbg: call mhx_dist_setup first");
    const uint64_t saved = c->global_bases;  // items carry no positions
    const mhx::DistPasses dp = mhx::plan_dist_passes(c, cm, MHX_STAGE_SEQ2SDBG, k, 0, (size_t)mhx::seq2sdbg_stride(k) * 4);
    mhx_sdbg_result r{};
    for (int pass = 0; pass < dp.n; ++pass) {
      mhx::set_pass(c, dp, pass, false);
      const uint64_t n = mhx::exchange_stage(c, cm, MHX_STAGE_SEQ2SDBG, k, 0);
      mhx_sdbg_result rp{};
      MHX_CK(mhx_dist_process_seq2sdbg(c, k, n, &rp));
      if (dp.n > 1) mhx::sdbg_accumulate(c, pass == 0);
      mhx::add_result(r, rp);
    }
    mhx::clear_pass(c, dp);
    if (dp.n > 1) mhx::sdbg_publish_accumulated(c);
    c->global_bases = saved;
    if (out) *out = r;
  })
}

}  // extern "C"
