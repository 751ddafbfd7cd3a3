// read2sdbg stage 1: replaces Read2SdbgS1 (reference src/sorting/read_to_sdbg_s1.cpp) on the GPU.
//
//   extract   one item per canonical (k-1)-mer occurrence (+ both strands at the read ends) with
//             (head,tail) in the low 6 key bits and (prev,next,position) as aux
//             (Lv1FillOffsets :208-296 + Lv2ExtractSubString :298-366 fused)
//   sort      by (k-1)-mer then (head,tail)                                   (sort.hip)
//   groups    heads of equal-(k-1)-mer groups                                 (scan.hip)
//   reduce    per group: (head,tail) run lengths -> is_solid bits (atomicOr), multiplicity histogram,
//             optional mercy candidates                                       (Lv2Postprocess :368-555)
//
// Tie order: the sort is stable and items are emitted in the reference's global order, so the
// "first item" of a group (whose prev/next the reference re-uses for the whole group, :399) is the
// first in read order.  kmlib::kmsort is unstable for buckets > 64 items, so mercy candidates can
// differ from the reference there; is_solid and the histogram never depend on it.
#include "s1_shared.h"

namespace mhx {

// regions of k_s1_seg -> one dense array: block (r, j) copies slice j of region r behind the items of the regions before it
__global__ __launch_bounds__(256) void k_agg_compact(const uint2 *__restrict__ raw, uint32_t cap, const uint32_t *__restrict__ counts,
                                                    uint2 *__restrict__ dense, int from_back) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint32_t r = blockIdx.x;
  uint64_t part = 0;
  for (uint32_t i = threadIdx.x; i < r; i += 256) part += counts[i];
  uint64_t off;
  block_exclusive_sum<uint64_t, 256>(part, sm, &off);
  const uint32_t n = counts[r];
  const uint2 *src = raw + (size_t)r * cap;
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) dense[off + i] = from_back ? src[cap - 1 - i] : src[i];
}

// multi-GPU, sparse marks, classic path: positions of the set bytes of the (global) byte map, appended in any order
__global__ __launch_bounds__(256) void k_collect_marks(const uint8_t *__restrict__ bytes, uint64_t n_bytes, unsigned long long *__restrict__ out,
                                                      unsigned long long *__restrict__ cursor) {
  __shared__ uint32_t s_n;
  __shared__ unsigned long long s_base;
  const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  uint32_t mask = 0;
  if (p0 < n_bytes) {
    const uint4 v = *reinterpret_cast<const uint4 *>(bytes + p0);  // the map is padded to a multiple of 64 bytes
    const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (p0 + t < n_bytes && ((xs[t >> 2] >> ((t & 3) * 8)) & 1u)) mask |= 1u << t;
  }
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t cnt = __builtin_popcount(mask);
  uint32_t at = cnt ? atomicAdd(&s_n, cnt) : 0u;
  __syncthreads();
  if (threadIdx.x == 0 && s_n) s_base = atomicAdd(cursor, (unsigned long long)s_n);
  __syncthreads();
  for (uint32_t mm = mask; mm; mm &= mm - 1) out[s_base + at++] = p0 + (uint64_t)__builtin_ctz(mm);
}
// routed marks (global positions of non-solid occurrences in the local reads) -> local byte map
__global__ void k_apply_marks(const unsigned long long *__restrict__ pos, uint64_t n, uint64_t pos_base, uint64_t n_local, uint8_t *__restrict__ bytes,
                              uint32_t *__restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = pos[i] - pos_base;
  if (p < n_local) bytes[p] = 1;
  else atomicOr(bad, 1u);
}

// byte map -> AtomicBitVector layout (bit i = word i/64, bit i%64; kmbitvector.h:67-88) + popcount.
// A thread takes 16 bytes (one coalesced 16-byte load per lane: a wavefront reads 1 KB in one instruction), four
// neighbouring lanes put their 16 bits together.  (One thread per 64 bytes made every load instruction touch 64 lines.)
__device__ __forceinline__ uint32_t low_bits_of_16_bytes(const uint4 x) {
  const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
  uint32_t m = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t b = xs[t] & 0x01010101u;  // bytes are 0/1: gather the low bit of each of the 4 bytes
    m |= ((b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xFu) << (4 * t);
  }
  return m;
}
// m16 of lanes 4j..4j+3 -> bits of word j (returned on lane 4j)
__device__ __forceinline__ unsigned long long join_4_lanes(uint32_t m16) {
  return (unsigned long long)m16 | ((unsigned long long)__shfl_down(m16, 1, kWave) << 16) | ((unsigned long long)__shfl_down(m16, 2, kWave) << 32) |
         ((unsigned long long)__shfl_down(m16, 3, kWave) << 48);
}
// Persistent: a thread walks over chunks g, g + T, g + 2T, ... (T = all threads, a multiple of 4) and adds up its popcounts;
// one block reduction and one atomic per workgroup at the end.
__global__ __launch_bounds__(256) void k_pack_solid(const uint8_t *__restrict__ bytes, uint64_t n_bits, unsigned long long *__restrict__ words,
                                                    uint64_t n_words, unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x, n_chunks = n_words * 4;  // the byte map is padded to whole words
  uint32_t pop = 0;
  for (uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x; g0 < n_chunks; g0 += T) {  // workgroup-uniform trip count (shuffles inside)
    const uint64_t g = g0 + threadIdx.x, p0 = g * 16;
    uint32_t m16 = 0;
    if (g < n_chunks && p0 < n_bits) {
      m16 = low_bits_of_16_bytes(reinterpret_cast<const uint4 *>(bytes)[g]);
      if (p0 + 16 > n_bits) m16 &= (1u << (n_bits - p0)) - 1u;
    }
    const unsigned long long v = join_4_lanes(m16);
    if ((threadIdx.x & 3) == 0 && g < n_chunks) words[g >> 2] = v;
    pop += (uint32_t)__builtin_popcount(m16);
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>((uint64_t)pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

__global__ __launch_bounds__(256) void k_count_solid(const unsigned long long *__restrict__ words, uint64_t n_words,
                                                     unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  uint64_t pop = 0;  // one atomic per workgroup of a bounded grid: 10^5 atomics on one word cost ~1 ms by themselves
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x)
    pop += (uint64_t)__builtin_popcountll(words[w]);
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>(pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

// mark_mode 1: bytes mark the NON-solid occurrences; a position is solid iff a (k+1)-mer starts there
// (offset + k + 1 <= read length) and it is not marked.  One thread per 64 positions.
__global__ __launch_bounds__(256) void k_pack_solid_inv(const uint8_t *__restrict__ bytes, uint64_t n_bits, const uint64_t *__restrict__ start,
                                                        uint64_t n_seqs, uint32_t fixed_len, int k, unsigned long long *__restrict__ words,
                                                        uint64_t n_words, unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  uint64_t pop = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long v = 0;
    const uint64_t p0 = w * 64;
    uint64_t rid = p0 < n_bits ? seq_of_offset(start, n_seqs, fixed_len, p0) : 0;
    uint64_t re = start[rid + 1];
    const uint4 *pb = reinterpret_cast<const uint4 *>(bytes + p0);
    for (int q = 0; q < 4; ++q) {
      const uint4 x = pb[q];
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
      for (int t = 0; t < 16; ++t) {
        const uint64_t p = p0 + q * 16 + t;
        if (p >= n_bits) break;
        while (p >= re) re = start[++rid + 1];
        const bool valid = p + (uint64_t)k + 1 <= re;  // a (k+1)-mer of this read starts at p
        const bool marked = (xs[t >> 2] >> ((t & 3) * 8)) & 1u;
        if (valid && !marked) v |= 1ull << (q * 16 + t);
      }
    }
    words[w] = v;
    pop += (uint64_t)__builtin_popcountll(v);
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>(pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

// The same for reads of one length L with L - k >= 16: the valid positions of 16 consecutive ones follow from the offset
// of the first in its read (valid: offset <= L - k - 1), so a thread needs one 16-byte load, one remainder and a few masks
// instead of a 64-step walk over the read boundaries.
__global__ __launch_bounds__(256) void k_pack_solid_inv_fixed(const uint8_t *__restrict__ bytes, uint64_t n_bits, uint32_t L, int k,
                                                              unsigned long long *__restrict__ words, uint64_t n_words,
                                                              unsigned long long *__restrict__ n_solid) {
  __shared__ uint64_t sm[256 / kWave + 1];
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x, n_chunks = n_words * 4;
  const uint32_t step = (uint32_t)((T * 16) % L);  // the offset in the read advances by this much (mod L) per trip: one
  uint32_t off = (uint32_t)((((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16) % L);  // 64-bit remainder per thread, not per chunk
  uint32_t pop = 0;
  for (uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x; g0 < n_chunks; g0 += T) {
    const uint64_t g = g0 + threadIdx.x, p0 = g * 16;
    uint32_t m16 = 0;
    if (g < n_chunks && p0 < n_bits) {
      const uint32_t marked = low_bits_of_16_bytes(reinterpret_cast<const uint4 *>(bytes)[g]);
      const int t1 = min(max((int)L - k - (int)off, 0), 16);  // positions [0, t1): a (k+1)-mer of this read starts there
      const int t2 = min((int)L - (int)off, 16);              // positions [t2, 16): the next read, offsets < 16 <= L - k
      uint32_t valid = ((1u << t1) - 1u) | (0xFFFFu & ~((1u << t2) - 1u));
      if (p0 + 16 > n_bits) valid &= (1u << (n_bits - p0)) - 1u;
      m16 = valid & ~marked & 0xFFFFu;
    }
    const unsigned long long v = join_4_lanes(m16);
    if ((threadIdx.x & 3) == 0 && g < n_chunks) words[g >> 2] = v;
    pop += (uint32_t)__builtin_popcount(m16);
    off += step;
    if (off >= L) off -= L;
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, 256>((uint64_t)pop, sm, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

// multi-GPU: the adopted slice holds the summed NON-solid marks of the local reads -> is_solid = valid & ~marked
__global__ __launch_bounds__(256) void k_invert_marks(unsigned long long *__restrict__ words, uint64_t n_words, uint64_t n_bits,
                                                      const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t fixed_len, int k) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const uint64_t p0 = w * 64;
  unsigned long long valid = 0;
  if (p0 < n_bits) {
    uint64_t rid = seq_of_offset(start, n_seqs, fixed_len, p0);
    uint64_t re = start[rid + 1];
    for (int t = 0; t < 64; ++t) {
      const uint64_t p = p0 + t;
      if (p >= n_bits) break;
      while (p >= re) re = start[++rid + 1];
      if (p + (uint64_t)k + 1 <= re) valid |= 1ull << t;
    }
  }
  words[w] = valid & ~words[w];
}
void invert_local_marks(mhx_ctx *c, unsigned long long *words, uint64_t n_words) {
  SeqSet &s = c->seqs;
  if (n_words)
    MHX_LAUNCH(c, "invert_marks", (double)n_words * 16,
               hipLaunchKernelGGL(k_invert_marks, dim3((unsigned)div_ceil(n_words, 256)), dim3(256), 0, c->stream, words, n_words, s.n_bases,
                                  s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, (int)c->s1_acc_k));
}

// int64 <-> (hi,lo) word pairs so that the big-endian record sort orders them numerically
__global__ void k_swap_words(uint32_t *__restrict__ v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint2 x = reinterpret_cast<uint2 *>(v)[i];
    reinterpret_cast<uint2 *>(v)[i] = make_uint2(x.y, x.x);
  }
}

// make `b` at least `bytes` large, preserving its first `keep` bytes
DevBuf &grow_preserving(mhx_ctx *c, DevBuf &b, size_t bytes, size_t keep) {
  if (b.cap >= bytes) return b;
  DevBuf nb;
  nb.reserve(bytes + bytes / 4);
  if (keep && b.p) MHX_HIP(hipMemcpyAsync(nb.p, b.p, keep, hipMemcpyDeviceToDevice, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  b.release();
  b = nb;
  return b;
}

// numeric sort of n 64-bit records on the device: swap to (hi,lo) words, record sort with 2 key words, swap back.
// Returns a pointer into the workspace ("u64_sort_a" / "u64_sort_b") holding the sorted copy.
const uint64_t *sort_u64(mhx_ctx *c, const void *src, uint64_t n, int hi_bit) {
  hipStream_t st = c->stream;
  uint32_t *ma = c->ws("u64_sort_a", n * 8 + 64).as<uint32_t>();
  uint32_t *mb = c->ws("u64_sort_b", n * 8 + 64).as<uint32_t>();
  if (!n) return reinterpret_cast<const uint64_t *>(ma);
  MHX_HIP(hipMemcpyAsync(ma, src, n * 8, hipMemcpyDeviceToDevice, st));
  const unsigned g2 = (unsigned)div_ceil(n, 256);
  hipLaunchKernelGGL(k_swap_words, dim3(g2), dim3(256), 0, st, ma, n);
  uint32_t *ms = radix_sort(c, ma, mb, n, 2, 2, make_passes(2, 0, hi_bit));
  hipLaunchKernelGGL(k_swap_words, dim3(g2), dim3(256), 0, st, ms, n);
  return reinterpret_cast<const uint64_t *>(ms);
}
// multi-GPU: position-keyed records (count events, mercy candidates) waiting to be routed to the ranks that hold the reads
void stash_route_records(mhx_ctx *c, const void *src, uint64_t n, int hi_bit) {
  const uint64_t *sorted = sort_u64(c, src, n, hi_bit);
  DevBuf &r = c->ws("route_records", n * 8 + 64);
  if (n) MHX_HIP(hipMemcpyAsync(r.p, sorted, n * 8, hipMemcpyDeviceToDevice, c->stream));
  c->n_route = n;
}

// ---- host driver, in two halves so that the multi-GPU path can exchange items in between ----
// LSD passes of the stage-1 sort: the 6 head/tail bits, then the (k-1)-mer
std::vector<SortPass> s1_sort_passes(uint32_t k) {
  const int KWv = s1_kw(k), kmer_bits = (int)(k - 1) * 2;
  return make_passes_ranges(KWv, {{0, 6}, {KWv * 32 - kmer_bits, KWv * 32}});
}
// Stage-1 sort plan.  seg_bits > 0: only the top seg_bits of the (k-1)-mer are sorted and an LDS group-by counts the equal
// keys of each segment (no-mercy 12-byte records).
//   stream: one workgroup streams one bucket of the seg_bits-bit prefix (k_s1_stream).  The width follows the DENSITY of the
//   job — records per lv1 bucket where the group-by runs — so that a streamed bucket holds at most s1_stream_max records
//   (a third of which are distinct keys at 60x coverage: what the 8192-slot table takes with room to spare) whatever the job
//   size: 16 bits and two LSD passes up to s1_stream_max records per lv1 bucket (10 M reads per GPU), the same two passes
//   with 2^sub0 sub-rounds per bucket up to 2^s1_stream_sub_max times that (a second read of the bucket is cheaper than a
//   third pass over all records), beyond that 17..24 bits in three passes.  A bucket that overflows anyway splits itself
//   (k_s1_stream): no job size and no single bucket sends the stage anywhere else.
//   otherwise: k_s1_seg on tiles, the width chosen so that a segment holds ~100 records.
// records per lv1 bucket at the group-by: the items of the whole job over the buckets in play.  A rank of a multi-GPU run
// owns 1/n_parts of the key space; under a bucket filter (memory plan) the announced item count of the kept buckets stands
// for the call's own (the plan is made before the kept items are counted, and every later look must give the same plan);
// comm.hip sets the figure the ranks agreed on (s1_density).
static double s1_density(const mhx_ctx *c, uint64_t n_items) {
  if (c->s1_density > 0) return c->s1_density;
  const double n = c->filter_on ? (double)c->filter_expected : (double)n_items;
  const double buckets = c->filter_on && c->filter_kept ? (double)c->filter_kept : (double)MHX_NUM_BUCKETS;
  return n * (double)(c->n_parts > 1 ? c->n_parts : 1) / buckets;
}
// LSD passes over the top `pbits` key bits in digits of about equal width (<= 8 bits); every pass after the first declares
// the bits sorted before it: the consumers of these plans count equal keys, the order of records equal in all sorted bits is
// free (SortPass::prev_lo)
static std::vector<SortPass> s1_prefix_passes(int pbits) {
  const int np = (pbits + 7) / 8, lo = 64 - pbits;
  std::vector<SortPass> p;
  int at = lo;
  for (int i = 0; i < np; ++i) {
    const int w = pbits / np + (i < pbits % np ? 1 : 0);
    SortPass sp{at, w, 0, 0};
    sp.prev_lo = i ? lo : -1;
    p.push_back(sp);
    at += w;
  }
  return p;
}
S1Plan s1_plan(const mhx_ctx *c, uint32_t k, uint64_t n_items, bool compact, int want_mercy, bool allow_stream) {
  const int force_bits = (int)c->opt("s1_seg_bits", 0);
  const int kmer_bits = (int)(k - 1) * 2;
  S1Plan p{s1_sort_passes(k), 0, false};
  if (!c->opt("s1_seg", 1) || !compact || want_mercy || s1_kw(k) != 2 || s1_stride(k, compact) != 3 || !n_items) return p;
  const double per_bucket = s1_density(c, n_items);
  p.per_bucket = per_bucket;
  // (k <= 22: the local key — the (k-1)-mer below a 16-bit prefix + head/tail — fits 32 bits; up to k = 29, the widest (k-1)-mer the
  //  12-byte record holds, the table keys are 64 bits wide: s1_stream_wide, round 6)
  if (allow_stream && c->opt("s1_stream", 1) && !force_bits && k >= 10 && (k <= 22 || (k <= 29 && c->opt("s1_stream_wide", 1)))) {
    const double cap = (double)std::max<long long>(1, c->opt("s1_stream_max", 40000));
    const int sub_max = (int)std::min<long long>(std::max<long long>(c->opt("s1_stream_sub_max", 1), 0), 6);
    int need = 0;  // the buckets have to be 2^need times finer than the lv1 buckets
    while (need < 30 && per_bucket > cap * (double)(1ull << need)) ++need;
    int pbits = 16, sub0 = 0;
    if (need <= sub_max) sub0 = need;
    else {
      // a third pass costs the same for 17 or 24 prefix bits: aim at buckets that fill a third of the table — the inserts of a
      // table at two thirds take twice as long (tools/micro/insert_probe.hip), and the canonical (k-1)-mers make the low lv1
      // buckets twice as full as the average — but not at so many buckets that their fixed cost (a walk over 8192 slots) shows
      const double cap3 = (double)std::max<long long>(1, c->opt("s1_stream_max3", std::max<long long>(1, (long long)cap / 2)));
      need = 0;
      while (need < 30 && per_bucket > cap3 * (double)(1ull << need)) ++need;
      pbits = std::min(16 + std::max(need, 1), 24);
      sub0 = std::min(16 + need - pbits, 6);  // (past 24 bits: the rest as sub-rounds; the kernel splits further if it has to)
    }
    if (const long long f = c->opt("s1_stream_bits", 0)) pbits = (int)std::min<long long>(std::max<long long>(f, 9), 24);
    const long long fs = c->opt("s1_stream_sub0", -1);
    if (fs >= 0) sub0 = (int)std::min<long long>(fs, 6);
    pbits = std::min(pbits, kmer_bits);
    p.seg_bits = pbits;
    p.stream = true;
    p.sub0 = sub0;
    p.passes = s1_prefix_passes(pbits);
    return p;
  }
  const double n_eff = per_bucket * (double)MHX_NUM_BUCKETS;
  int bits = 8;
  while (bits < 32 && n_eff / 96.0 > (double)(1ull << bits)) bits += 8;
  if (force_bits) bits = force_bits;
  bits = std::max(1, std::min(bits, std::min(32, kmer_bits)));
  p.seg_bits = bits;
  p.passes = make_passes(2, 64 - bits, 64);
  return p;
}
// what a caller may print: "stream p16 s0 2 passes" / "seg 24" / "full sort"
std::string s1_plan_text(const mhx_ctx *c, uint32_t k, uint64_t n_items) {
  const S1Plan p = s1_plan(c, k, n_items, s1_compact(c, k, 0), 0);
  char buf[128];
  if (p.stream) snprintf(buf, sizeof buf, "stream p%d sub%d %zu passes (%.0f records per lv1 bucket)", p.seg_bits, p.sub0, p.passes.size(), p.per_bucket);
  else if (p.seg_bits) snprintf(buf, sizeof buf, "seg p%d %zu passes", p.seg_bits, p.passes.size());
  else snprintf(buf, sizeof buf, "full sort %zu passes", p.passes.size());
  return buf;
}
// Compact records (12 bytes at k <= 29: key + one position word) whenever no mercy candidates are wanted and the positions
// fit: below 2^pos_bits bases as they are, beyond that with the upper position bits as a tag inside the key words
// (s1_pos_tag: 8 spare bits between the (k-1)-mer and head/tail, i.e. up to 2^(pos_bits + 8) bases: 7 G reads of 150 bp).
uint32_t s1_pos_bits(const mhx_ctx *c) {
  const bool force = getenv("MHX_S1_FORCE_TAGGED") != nullptr;  // tests: tags at small sizes too
  const long long f = c->opt("s1_pos_bits", 0);
  if (f > 0) return (uint32_t)std::min<long long>(std::max<long long>(f, 4), 32);
  if (force) {  // the narrowest position word that keeps the tags below 128
    const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
    uint32_t b = 4;
    while (b < 32 && (n_bits >> b) >= 128) ++b;
    return b;
  }
  return 32;
}
bool s1_rank_tagged(const mhx_ctx *c, uint32_t k) {
  const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
  const uint32_t pb = s1_pos_bits(c);
  if ((n_bits >> pb) == 0) return false;  // every tag is 0
  const int spare = 32 * s1_kw(k) - (int)(k - 1) * 2 - 6;
  return spare >= 8 && (n_bits >> pb) < 256;
}
// global position = position word + tag * s1_pos_stride (0: no tags in this read set)
uint64_t s1_pos_stride(const mhx_ctx *c, uint32_t k) { return s1_rank_tagged(c, k) ? 1ull << s1_pos_bits(c) : 0ull; }
bool s1_compact(const mhx_ctx *c, uint32_t k, int want_mercy) {
  if (want_mercy) return false;
  if (s1_kw(k) < 2) return false;  // k <= 14: a one-word key; the 16-byte records serve (8-byte compact records have no kernels)
  const uint64_t n_bits = c->global_bases ? c->global_bases : c->seqs.n_bases;
  return (n_bits >> s1_pos_bits(c)) == 0 || s1_rank_tagged(c, k);
}
int s1_stride(uint32_t k, bool compact) {
  const int kw = s1_kw(k);
  if (!compact) return round_up2(kw + 2);
  return kw + 1 == 3 ? 3 : round_up2(kw + 1);  // 12-byte records are supported natively, other odd widths are padded
}


// sort + group reduction of n_items items held in buf_a (buf_b = ping-pong space of the same size)
bool s1_presort_applies(const mhx_ctx *c, uint32_t k, uint64_t n_local_items) {
  // (k_s1_stream keeps the bounds and the arrays of up to kStreamSrcMax senders in LDS; more ranks take the classic exchange)
  return c->n_parts <= kStreamSrcMax && s1_plan(c, k, n_local_items, s1_compact(c, k, 0), 0).stream;
}
// the LSD passes that order this rank's stage-1 records by the plan's prefix (the first half of s1_process on the stream
// plan; the ranks agreed on the density the plan follows: mhx_ctx::s1_density)
uint32_t *s1_presort(mhx_ctx *c, uint32_t k, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items, int *pbits) {
  const S1Plan plan = s1_plan(c, k, std::max<uint64_t>(n_items, 1), true, 0);
  if (!plan.stream) throw Error("s1_presort: the bucket-streaming plan does not apply");
  *pbits = plan.seg_bits;
  uint32_t *sorted = n_items ? radix_sort(c, buf_a, buf_b, n_items, 3, s1_kw(k), plan.passes) : buf_a;
  c->pre_hist_buf = nullptr;
  return sorted;
}

// ---- stage 1 behind the extraction, in steps (Read2SdbgS1::Lv2Postprocess and what it needs, read_to_sdbg_s1.cpp:368-555) ----
//   sort_records   the plan's passes (or the reference-exact order for want_mercy == 2)
//   open_outputs   is_solid / histogram / marks / aggregated stage-2 items, continued or fresh (bucket-range passes accumulate)
//   polarity       mark the solid or the non-solid occurrences (a 1/64 sample decides on a single GPU)
//   group_partial  the LDS group-bys on partially sorted records (bucket streaming, segments), with their ways back
//   group_classic  full sort + k_tile_groups (mercy, wide keys, every give-up)
//   publish        bitmap, counters, mercy candidates, result buffers
namespace {
struct S1Stage {
  mhx_ctx *c;
  uint32_t k, m;
  int want_mercy;
  uint32_t *buf_a, *buf_b;
  uint64_t n_items;
  const S1Sources *pre;
  SeqSet &s;
  hipStream_t st;
  bool compact, global, tagged_keys;
  int KWv, S, kmer_bits;
  size_t item_bytes;
  uint64_t pos_stride;
  S1Plan plan;
  uint32_t *sorted = nullptr, *spare = nullptr;
  // outputs
  uint64_t n_bits = 0, n_words64 = 0;
  int mark_atomic = 0;
  const char *mark_env = nullptr;
  bool sparse = false, acc = false;
  bool list_local = false;  // the last launch_partial left its marks in the workgroups' regions (one GPU: applied by run_partial)
  unsigned long long *is_solid = nullptr, *hist = nullptr, *ctr = nullptr;
  uint8_t *solid_bytes = nullptr;
  uint64_t marks_prev = 0, seg_marks = 0;
  const uint32_t *giant_ctr = nullptr;  // device counters of the giant path of the last launch_partial (nullptr: not taken)
  bool classic_ran = false;
  long long *mercy = nullptr;
  bool agg = false;
  uint2 *agg_items = nullptr;
  uint64_t *agg_cursor = nullptr;
  uint64_t agg_prev = 0, agg_bound = 0;
  uint32_t seg_grid = 0, seg_cap = 0, seg_mcap = 0;
  uint32_t *seg_err = nullptr;
  int mark_mode = 0, mark_mode_used = 0;

  S1Stage(mhx_ctx *c_, uint32_t k_, uint32_t m_, int wm, uint32_t *a, uint32_t *b, uint64_t n, const S1Sources *pre_)
      : c(c_), k(k_), m(m_), want_mercy(wm), buf_a(a), buf_b(b), n_items(n), pre(pre_), s(c_->seqs), st(c_->stream) {
    compact = s1_compact(c, k, want_mercy);
    KWv = s1_kw(k);
    S = s1_stride(k, compact);
    item_bytes = (size_t)S * 4;
    global = c->global_bases != 0;  // multi-GPU: positions index the global read set
    kmer_bits = (int)(k - 1) * 2;
    // (records that carry position bits between the (k-1)-mer and head/tail must not be ordered by whole key words)
    tagged_keys = compact && s1_rank_tagged(c, k);
    pos_stride = compact ? s1_pos_stride(c, k) : 0;
    plan = s1_plan(c, k, n_items, compact, want_mercy);
    // pre: the records come pre-sorted by the plan's prefix in several arrays (multi-GPU: one per sending rank, comm.hip);
    // n_items is their total.  Only the bucket-streaming group-by reads them in place; if it gives up, they are gathered and
    // the stage continues as if they had arrived unsorted.
    if (pre && (!plan.stream || want_mercy || !compact)) throw Error("s1_process: pre-sorted sources need the bucket-streaming plan");
    if (pre && pre->pbits != plan.seg_bits) throw Error("s1_process: the sources were sorted for another plan than the one this rank makes");
  }

  void sort_records() {
    // want_mercy == 2: records with equal keys in exactly the order the reference's kmsort leaves them (H1)
    sorted = pre ? nullptr
             : want_mercy == 2
                 ? kmsort_exact(c, buf_a, buf_b, n_items, S, KWv)
                 : (plan.seg_bits || tagged_keys ? radix_sort(c, buf_a, buf_b, n_items, S, KWv, plan.passes)
                                                 : sort_whole_key(c, buf_a, buf_b, n_items, S, KWv, plan.passes));
    c->pre_hist_buf = nullptr;
    set_spare(pre ? pre->spare : (sorted == buf_a ? buf_b : buf_a));
  }
  void set_spare(uint32_t *sp) {
    spare = sp;
    mercy = reinterpret_cast<long long *>(spare);  // mercy candidates (<= 2 per item, 8 B each): S*4 >= 16 bytes per item
  }
  void ensure_byte_map() {  // (sparse: only when the classic kernel runs; always zeroed, its marks are collected afterwards)
    if (solid_bytes) return;
    const uint64_t nw = div_ceil(n_bits, 64);
    solid_bytes = c->ws("solid_bytes", (nw + 1) * 64).as<uint8_t>();
    if (!acc || sparse) MHX_HIP(hipMemsetAsync(solid_bytes, 0, (nw + 1) * 64, st));
  }
  void open_outputs() {
    n_bits = global ? c->global_bases : s.n_bases;
    // MHX_S1_MARK: atomic (atomicOr into the bitmap) | solid | nonsolid (force the byte-map polarity) | unset = auto
    mark_env = getenv("MHX_S1_MARK");
    mark_atomic = mark_env && !strcmp(mark_env, "atomic") ? 1 : 0;
    // multi-GPU with sparse marks (comm.hip): the marks of the non-solid occurrences leave this stage as a list of
    // global positions (ws "s1_marks", c->n_marks) to be routed to the read owners; no bitmap / byte map of the GLOBAL read
    // set exists unless the classic tile kernel has to run (then its byte map is converted to the list)
    sparse = global && !mark_atomic && c->opt("dist_sparse_marks", 0) != 0;
    n_words64 = sparse ? 0 : div_ceil(n_bits, 64);
    is_solid = c->result(MHX_BUF_IS_SOLID, (n_words64 + 1) * 8).as<unsigned long long>();
    c->results[MHX_BUF_IS_SOLID].used = n_words64 * 8;
    hist = c->result(MHX_BUF_MUL_HIST, (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
    // accumulate (bucket-range passes after the first, passes.hip): marks, histogram, aggregated stage-2 items and
    // mercy candidates of the earlier passes are kept and the published results are cumulative
    acc = c->accumulate && c->s1_acc_bits == n_bits && c->s1_acc_k == k && c->s1_acc_m == m;
    marks_prev = sparse && acc ? c->n_marks : 0;  // marks of the earlier bucket-range passes stay in front
    if (!sparse || !acc) c->n_marks = 0;
    if (mark_atomic) {
      if (!acc) MHX_HIP(hipMemsetAsync(is_solid, 0, (n_words64 + 1) * 8, st));
    } else {
      if (!sparse) ensure_byte_map();
      MHX_HIP(hipMemsetAsync(is_solid + n_words64, 0, 8, st));
    }
    if (!acc) MHX_HIP(hipMemsetAsync(hist, 0, (MHX_MAX_MUL + 1) * 8, st));
    c->s1_acc_bits = n_bits;
    c->s1_acc_k = k;
    c->s1_acc_m = m;
    ctr = c->ws("s1_counters", 64).as<unsigned long long>();
    MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
    // aggregated stage-2 items (k <= 22, m >= 2): at most 2 per solid run, a solid run has >= m records
    const bool agg_off = getenv("MHX_S2_PER_OCCURRENCE") != nullptr;  // (read per call: a resident server answers requests with different environments)
    agg = !agg_off && k <= 22 && m >= 2 && KWv == 2;
    const bool agg_continues = acc && c->agg_valid && c->agg_k == k && c->agg_m == m;
    c->agg_valid = false;
    agg_cursor = c->ws("s2_agg_cursor", 64).as<uint64_t>();
    agg_prev = agg_continues ? c->agg_n : 0;  // items of the earlier passes stay in front
    agg_bound = (n_items / m + 16) * 2;
    seg_err = c->ws("s1_seg_err", 64).as<uint32_t>();
  }
  void agg_prepare_classic() {  // k_tile_groups appends to the dense array through a global cursor
    if (!agg) return;
    agg_items = grow_preserving(c, c->work["s2_agg_items"], (agg_prev + agg_bound) * 8, agg_prev * 8).as<uint2>();
    MHX_HIP(hipMemsetAsync(agg_cursor, 0, 24, st));
    if (agg_prev) MHX_HIP(hipMemcpyAsync(agg_cursor, &agg_prev, 8, hipMemcpyHostToDevice, st));
  }

  // the LDS group-bys on the partially sorted records: k_s1_stream (one bucket of the plan's prefix per workgroup at a time) or
  // k_s1_seg (tiles).  mode = mark_mode (2: statistics on a 1/64 sample)
  void launch_partial(int mode) {
    const int per = (int)c->opt("s1_seg_per", 8);
    const int la = (int)std::min<long long>(std::max<long long>(c->opt("s1_seg_la", 3), 0), 200);  // 16-bit tile counters
    const int T = 256 * (per == 4 ? 4 : 8);
    const uint64_t n_tiles = div_ceil(n_items, (uint64_t)T);
    const uint32_t stride = mode == 2 ? 64u : 1u;
    const uint64_t n_buckets = plan.stream ? 1ull << plan.seg_bits : 0;
    const uint64_t n_work = plan.stream ? div_ceil(n_buckets, stride) : div_ceil(n_tiles, stride);
    const uint32_t pfx_mask = plan.seg_bits >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> plan.seg_bits);
    const uint32_t eq_mask1 = (kmer_bits > 32 ? ~(0xFFFFFFFFu >> (kmer_bits - 32)) : 0u) | 63u;
    const bool agg_on = agg && mode != 2;
    // (round 6: s1_stream_half — two 512-thread workgroups with 4096-slot tables per CU — and s1_marks_list — the marks as a list applied by a
    //  kernel of their own — lost on every box for two rounds and are gone with their code: profiles/r03_ab_*, r05_ab_marks_list.jsonl)
    const bool half = false;
    const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_work, plan.stream ? (half ? 2 * cus : cus) : (per == 4 ? 256 * 6 : 256 * 3));
    // per-workgroup output regions in the spare sort buffer (S*4 >= 12 bytes per record, outputs are 8-byte entries)
    const uint32_t region = (uint32_t)std::min<uint64_t>(n_items * (uint64_t)S * 4 / 8 / grid, 0xFFFFFFF0u);
    uint2 *raw = nullptr;
    uint32_t *counts = nullptr;
    if (agg_on) {
      seg_grid = grid;
      seg_cap = region;
      raw = reinterpret_cast<uint2 *>(spare);
      counts = c->ws("s2_agg_counts", (size_t)grid * 4).as<uint32_t>();
    }
    // sparse marks go to the spare sort buffer (>= 12 bytes per record, a record yields at most one 8-byte mark): a
    // workgroup's region holds every record it can meet
    unsigned long long *mraw = nullptr;
    uint32_t *mcounts = nullptr;
    uint32_t mcap = 0;
    list_local = false;
    if ((sparse || list_local) && mode != 2) {
      mraw = reinterpret_cast<unsigned long long *>(spare);
      mcap = region;
      mcounts = c->ws("s1_mark_counts", (size_t)grid * 4).as<uint32_t>();
      seg_grid = grid;
      seg_mcap = mcap;
    }
    // the stream kernel marks the non-solid occurrences from its table when each of them is its key's only record
    const int direct = plan.stream && mode == 1 && m <= 2 && (mraw || solid_bytes) && c->opt("s1_stream_direct", 1) ? 1 : 0;
    S1SegArgs a{(int)k, m, pfx_mask, eq_mask1, solid_bytes, mode, hist, ctr, raw, seg_cap, counts, mraw, mcap, mcounts, pos_stride, seg_err,
                plan.stream ? (int)c->opt("s1_stream_probes", 1024) : la, direct};
    MHX_HIP(hipMemsetAsync(seg_err, 0, 4, st));
    const char *nm = mode == 2 ? "s1_sample" : "s1_groups";
    const double bytes = plan.stream ? (double)n_items * 12 / stride * (double)(1u << plan.sub0) : (double)n_work * T * 12;
    if (plan.stream) {
      const int n_src = pre && sorted == nullptr ? pre->n : 1;
      uint64_t *bounds = c->ws("s1_bucket_bounds", (size_t)n_src * (n_buckets + 1) * 8 + 64).as<uint64_t>();
      uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
      MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
      const unsigned bgrid = (unsigned)((n_buckets + 1 + 255) / 256);
      const uint32_t *const *srcs = nullptr;
      if (n_src > kStreamSrcMax) throw Error("s1: more pre-sorted sources than the bucket streaming takes");
      if (n_src > 1 || (pre && sorted == nullptr)) {
        DevBuf &sp = c->ws("s1_src_ptrs", (size_t)n_src * 8 + 64);
        MHX_HIP(hipMemcpyAsync(sp.p, pre->ptr.data(), (size_t)n_src * 8, hipMemcpyHostToDevice, st));
        srcs = sp.as<const uint32_t *>();
        for (int q = 0; q < n_src; ++q)
          hipLaunchKernelGGL(k_bucket_bounds, dim3(bgrid), dim3(256), 0, st, pre->ptr[q], pre->count[q], 3, bounds + (size_t)q * (n_buckets + 1),
                             plan.seg_bits);
      } else {
        MHX_LAUNCH(c, "bucket_bounds", (double)n_buckets * 8 * 30,
                   hipLaunchKernelGGL(k_bucket_bounds, dim3(bgrid), dim3(256), 0, st, sorted, n_items, 3, bounds, plan.seg_bits));
      }
      const uint32_t *items0 = pre && sorted == nullptr ? pre->ptr[0] : sorted;
      const uint32_t nslot = half ? 4096u : 8192u;
      const S1StreamGeom geo{plan.seg_bits, plan.sub0, (uint32_t)n_buckets,
                             (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", nslot * 7 / 8), 1), nslot)};
      const bool tags = pos_stride != 0;
      const bool key64 = kmer_bits - plan.seg_bits + 6 > 32;  // the local key: the (k-1)-mer below the prefix + head/tail
      // giant buckets (S1Giant): found, cut into slices and reduced on the device before the streaming launch, finished by a second
      // launch of the streaming kernel over the same grid (direct marks only: a key's first record stands for its slice)
      const bool giant_on = direct && !half && mode == 1 && c->opt("s1_giant", 1) != 0;
      uint32_t *ticket2 = nullptr;
      if (giant_on) {
        S1Giant &g = a.giant;
        g.gcap = 4096;
        g.min_records = (uint32_t)std::max<long long>(1, c->opt("s1_giant_min", 262144));
        g.pcap = std::max<uint64_t>(2u << 20, n_items / 64);
        g.flag = c->ws("s1_giant_flag", n_buckets + 64).as<uint8_t>();
        uint32_t *lists = c->ws("s1_giant_lists", 64 + (size_t)g.gcap * (5 * 4 + 8)).as<uint32_t>();
        g.ctr = lists;
        ticket2 = lists + 8;
        g.off = reinterpret_cast<unsigned long long *>(lists + 16);
        g.bucket = lists + 16 + 2 * g.gcap;
        g.sl = g.bucket + g.gcap;
        g.ns = g.sl + g.gcap;
        g.cap = g.ns + g.gcap;
        g.cur = g.cap + g.gcap;
        g.partial = c->ws("s1_giant_partial", g.pcap * 16 + 64).as<uint4>();
        giant_ctr = lists;
        MHX_HIP(hipMemsetAsync(g.flag, 0, n_buckets, st));
        MHX_HIP(hipMemsetAsync(lists, 0, 64, st));
        s1_giant_launch(c, items0, srcs, bounds, n_src, n_buckets, plan.seg_bits, (int)k, g, key64);
      }
      S1StreamLaunch sl{agg_on, half, tags, false, false, grid, items0, bounds, a, geo, stride, ticket, srcs, n_src};
      sl.key64 = key64;
      s1_stream_launch(c, nm, bytes, sl);
      if (giant_on) {  // the giants' partial entries -> the same per-key work, the same per-workgroup output regions
        S1StreamLaunch gl{agg_on, false, tags, true, false, grid, items0, bounds, a, geo, 1u, ticket2, nullptr, 1};
        gl.key64 = key64;
        s1_stream_launch(c, "s1_giant_groups", 0.0, gl);
      }
      return;
    }
    s1_seg_launch(c, nm, bytes, per, agg_on, grid, sorted, n_items, a, n_work, stride);
  }

  void launch_classic(bool with_agg) {  // k_tile_groups<S1Op> on the fully sorted records
    s1_classic_launch(c, S, compact, with_agg, sorted, n_items, KWv, kmer_bits, m, solid_bytes, is_solid, mark_atomic, hist, ctr, want_mercy, mercy, (int)k,
                      agg_items, agg_cursor, mark_mode);
  }
  void launch_classic_plain() { launch_classic(false); }

  // marking polarity from a 1/64 sample: when most occurrences are solid it is cheaper to mark the non-solid ones (each
  // mark is a partial HBM write).  Single GPU only: ranks must agree on the meaning.
  void polarity() {
    const bool can_invert = !global && !mark_atomic && !c->filter_on && !c->accumulate;  // passes must agree on the meaning
    // multi-GPU: every rank marks the NON-solid occurrences of the buckets it owns (a fixed convention, so that the
    // summed bitmaps mean the same on all ranks; typical inputs are mostly solid);
    // mhx_adopt_is_solid_slice turns "valid position and not marked" into the local is_solid
    if (global && !mark_atomic) mark_mode = 1;
    // bucket-range passes (memory plan): the same fixed convention, so that the passes' marks add up — and the bucket
    // streaming takes them straight from its table (direct_marks)
    else if ((c->filter_on || c->accumulate) && !mark_atomic && !mark_env) mark_mode = 1;
    else if (can_invert && mark_env && !strcmp(mark_env, "nonsolid")) mark_mode = 1;
    else if (can_invert && !mark_env && n_items > (1u << 16)) {
      mark_mode = 2;
      if (plan.seg_bits) launch_partial(2);
      else launch_classic_plain();
      unsigned long long hs[3] = {0, 0, 0};
      MHX_HIP(hipMemcpyAsync(hs, ctr, 24, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      mark_mode = hs[2] > 0 && hs[0] * 2 > hs[2] ? 1 : 0;
      MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
    }
  }

  // one run of the partial group-by: launch, read its error word and region counts back, pack the regions.  -> error word
  uint32_t run_partial() {
    giant_ctr = nullptr;
    launch_partial(mark_mode);
    uint32_t e = 0;
    std::vector<uint32_t> h_counts(agg ? seg_grid : 0), h_mcounts(sparse ? seg_grid : 0);
    uint32_t h_giant[4] = {0, 0, 0, 0};  // giants found, -, partial entries allotted (64 bits)
    if (giant_ctr) MHX_HIP(hipMemcpyAsync(h_giant, giant_ctr, 16, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipMemcpyAsync(&e, seg_err, 4, hipMemcpyDeviceToHost, st));
    if (agg) MHX_HIP(hipMemcpyAsync(h_counts.data(), c->work["s2_agg_counts"].p, (size_t)seg_grid * 4, hipMemcpyDeviceToHost, st));
    if (sparse) MHX_HIP(hipMemcpyAsync(h_mcounts.data(), c->work["s1_mark_counts"].p, (size_t)seg_grid * 4, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    if (e) return e;
    if (h_giant[0]) c->last_s1_plan += " [" + std::to_string(h_giant[0]) + " giant buckets in slices]";
    if (sparse) {  // pack the workgroups' mark regions (they live in the spare sort buffer) behind the earlier passes' marks
      for (uint32_t v : h_mcounts) seg_marks += v;
      unsigned long long *dense = grow_preserving(c, c->work["s1_marks"], (marks_prev + seg_marks) * 8 + 64, marks_prev * 8).as<unsigned long long>();
      if (seg_marks)
        MHX_LAUNCH(c, "marks_compact", (double)seg_marks * 16,
                   hipLaunchKernelGGL(k_agg_compact, dim3(seg_grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(spare), seg_mcap,
                                      c->work["s1_mark_counts"].as<uint32_t>(), reinterpret_cast<uint2 *>(dense + marks_prev), 0));
      c->n_marks = marks_prev + seg_marks;
    }
    if (agg) {  // pack the workgroups' regions behind the items of the earlier passes
      uint64_t total = 0;
      for (uint32_t v : h_counts) total += v;
      uint2 *dense = grow_preserving(c, c->work["s2_agg_items"], (agg_prev + total) * 8 + 64, agg_prev * 8).as<uint2>();
      if (total)
        MHX_LAUNCH(c, "agg_compact", (double)total * 16,
                   hipLaunchKernelGGL(k_agg_compact, dim3(seg_grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(spare), seg_cap,
                                      c->work["s2_agg_counts"].as<uint32_t>(), dense + agg_prev, 1));
      const uint64_t agg_n = agg_prev + total;
      MHX_HIP(hipMemcpyAsync(agg_cursor, &agg_n, 8, hipMemcpyHostToDevice, st));
      MHX_HIP(hipStreamSynchronize(st));  // agg_n is a stack variable
    }
    return 0;
  }
  // the sources of a pre-sorted stage become one unsorted array (a way back only)
  void gather_sources() {
    buf_a = c->ws("s1_gather_a", n_items * item_bytes + 64).as<uint32_t>();
    buf_b = c->ws("s1_gather_b", n_items * item_bytes + 64).as<uint32_t>();
    uint64_t at = 0;
    for (int q = 0; q < pre->n; ++q) {
      if (pre->count[q]) MHX_HIP(hipMemcpyAsync(buf_a + at * S, pre->ptr[q], pre->count[q] * item_bytes, hipMemcpyDeviceToDevice, st));
      at += pre->count[q];
    }
    sorted = buf_a;
  }
  // -> true when the partial group-by did the job; false: the records are fully sorted now and the classic kernel has to run.
  // The ways back (marks are idempotent, the histogram is restored): streaming -> segments when an output REGION of the
  // streaming overflowed (a bucket whose keys overflow the table is split inside the kernel, it never comes here);
  // segments -> full sort + classic when a segment outgrows the look-ahead or a tile's table.
  bool group_partial() {
    unsigned long long *hist_save = c->ws("s1_hist_save", (MHX_MAX_MUL + 1) * 8).as<unsigned long long>();
    MHX_HIP(hipMemcpyAsync(hist_save, hist, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
    for (;;) {
      const uint32_t e = run_partial();
      if (!e) return true;
      MHX_HIP(hipMemcpyAsync(hist, hist_save, (MHX_MAX_MUL + 1) * 8, hipMemcpyDeviceToDevice, st));
      MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
      seg_marks = 0;
      uint32_t *other;
      if (plan.stream) {
        if (pre && sorted == nullptr) gather_sources();
        plan = s1_plan(c, k, n_items, compact, want_mercy, false);
        other = sorted == buf_a ? buf_b : buf_a;
        sorted = radix_sort(c, sorted, other, n_items, S, KWv, plan.passes);
        set_spare(sorted == buf_a ? buf_b : buf_a);
        continue;
      }
      other = sorted == buf_a ? buf_b : buf_a;
      sorted = tagged_keys ? radix_sort(c, sorted, other, n_items, S, KWv, s1_sort_passes(k)) : sort_whole_key(c, sorted, other, n_items, S, KWv, s1_sort_passes(k));
      set_spare(sorted == buf_a ? buf_b : buf_a);
      return false;
    }
  }
  // stage 1 on super-k-mer records (s1_skm.hip): the group-by per minimizer bin, its aggregated items packed as run_partial packs them.
  // -> false: an output region overflowed (nothing published; the caller takes the prefix plan from scratch)
  bool run_skm(const SkmFront &f) {
    mark_mode = 1;  // the marks of the non-solid occurrences, from the table
    if (!sparse) ensure_byte_map();
    const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
    // (a workgroup takes tickets of kSkmBatch bins: no more workgroups than tickets, so that a small job's regions are not cut thinner than its work)
    const unsigned grid = (unsigned)std::min<uint64_t>(cus, std::max<uint64_t>(1, div_ceil((uint64_t)(f.bin_hi - f.bin_lo), (uint64_t)kSkmBatch)));
    // per-workgroup output regions in the spare sort buffer: aggregated items in its first half, (several GPUs) marks in its second
    const uint64_t region8 = f.spare_bytes / 8 / 2 / std::max(1u, grid);
    uint2 *raw = nullptr;
    uint32_t *counts = nullptr, *mcounts = nullptr;
    unsigned long long *mraw = nullptr;
    seg_grid = grid;
    if (agg) {
      seg_cap = (uint32_t)std::min<uint64_t>(region8, 0xFFFFFFF0u);
      raw = reinterpret_cast<uint2 *>(spare);
      counts = c->ws("s2_agg_counts", (size_t)grid * 4).as<uint32_t>();
    }
    if (sparse) {
      seg_mcap = (uint32_t)std::min<uint64_t>(region8, 0xFFFFFFF0u);
      mraw = reinterpret_cast<unsigned long long *>(spare) + f.spare_bytes / 8 / 2;
      mcounts = c->ws("s1_mark_counts", (size_t)grid * 4).as<uint32_t>();
    }
    MHX_HIP(hipMemsetAsync(seg_err, 0, 4, st));
    s1_skm_groups_launch(c, agg, grid, f, k, m, solid_bytes, hist, raw, seg_cap, counts, seg_err, mraw, seg_mcap, mcounts);
    uint32_t e = 0;
    std::vector<uint32_t> h_counts(agg ? seg_grid : 0), h_mcounts(sparse ? seg_grid : 0);
    MHX_HIP(hipMemcpyAsync(&e, seg_err, 4, hipMemcpyDeviceToHost, st));
    if (agg) MHX_HIP(hipMemcpyAsync(h_counts.data(), counts, (size_t)seg_grid * 4, hipMemcpyDeviceToHost, st));
    if (sparse) MHX_HIP(hipMemcpyAsync(h_mcounts.data(), mcounts, (size_t)seg_grid * 4, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    if (e) return false;
    if (sparse) {  // pack the workgroups' mark regions behind the earlier passes' marks (run_partial's way)
      for (uint32_t v : h_mcounts) seg_marks += v;
      unsigned long long *dense = grow_preserving(c, c->work["s1_marks"], (marks_prev + seg_marks) * 8 + 64, marks_prev * 8).as<unsigned long long>();
      if (seg_marks)
        MHX_LAUNCH(c, "marks_compact", (double)seg_marks * 16,
                   hipLaunchKernelGGL(k_agg_compact, dim3(seg_grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(mraw), seg_mcap, mcounts,
                                      reinterpret_cast<uint2 *>(dense + marks_prev), 0));
      c->n_marks = marks_prev + seg_marks;
    }
    if (agg) {
      uint64_t total = 0;
      for (uint32_t v : h_counts) total += v;
      // (a pass of several: the array is sized for all of them after the first — the bins fill evenly — instead of growing pass by pass)
      const uint64_t expect = !sparse && f.bin_hi > f.bin_lo ? (uint64_t)((double)(agg_prev + total) * (double)f.n_bins / (double)f.bin_hi * 1.05) : agg_prev + total;
      uint2 *dense = grow_preserving(c, c->work["s2_agg_items"], std::max(agg_prev + total, expect) * 8 + 64, agg_prev * 8).as<uint2>();
      if (total)
        MHX_LAUNCH(c, "agg_compact", (double)total * 16,
                   hipLaunchKernelGGL(k_agg_compact, dim3(seg_grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(spare), seg_cap, counts, dense + agg_prev, 1));
      const uint64_t agg_n = agg_prev + total;
      MHX_HIP(hipMemcpyAsync(agg_cursor, &agg_n, 8, hipMemcpyHostToDevice, st));
      MHX_HIP(hipStreamSynchronize(st));  // agg_n is a stack variable
      agg_prev = agg_n;  // (the next pass over a range of bins appends)
    }
    return true;
  }
  // the keys of the homopolymer windows, which k_skm_make counted beside the records: behind the last pass
  void publish_skm_hp(const SkmFront &f) {
    if (!f.hp) return;
    uint2 *dense = nullptr;
    if (agg) dense = grow_preserving(c, c->work["s2_agg_items"], (agg_prev + 4) * 8 + 64, agg_prev * 8).as<uint2>();  // (room for two keys' items)
    s1_skm_hp_publish(c, f, k, m, solid_bytes, hist, dense, agg_cursor, agg);
  }
  void group_classic() {
    agg_prepare_classic();
    ensure_byte_map();
    classic_ran = true;
    launch_classic(agg && ((S == 3 && compact) || (S == 4 && !compact)));
  }

  void publish(mhx_s1_result *out) {
    if (agg && (S == 3 || (S == 4 && !compact))) {  // also with zero local items: every rank takes the same stage-2 path
      c->agg_n = 0;
      MHX_HIP(hipMemcpyAsync(&c->agg_n, agg_cursor, 8, hipMemcpyDeviceToHost, st));
      c->agg_valid = true;
      c->agg_k = k;
      c->agg_m = m;
    }
    if (sparse && classic_ran) {  // the classic kernel marked a byte map of the global read set: turn it into the list
      const uint64_t n_bytes = div_ceil(n_bits, 64) * 64;
      unsigned long long *cur = c->ws("s1_mark_cursor", 64).as<unsigned long long>();
      MHX_HIP(hipMemsetAsync(cur, 0, 8, st));
      // upper bound of the marks: every record
      unsigned long long *dense = grow_preserving(c, c->work["s1_marks"], (marks_prev + n_items) * 8 + 64, marks_prev * 8).as<unsigned long long>();
      MHX_LAUNCH(c, "collect_marks", (double)n_bytes,
                 hipLaunchKernelGGL(k_collect_marks, dim3((unsigned)div_ceil(n_bytes, 256 * 16)), dim3(256), 0, st, solid_bytes, n_bytes,
                                    dense + marks_prev, cur));
      uint64_t got = 0;
      MHX_HIP(hipMemcpyAsync(&got, cur, 8, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      c->n_marks = marks_prev + got;
    }
    if (n_words64 && mark_atomic)
      MHX_LAUNCH(c, "count_solid", (double)n_words64 * 8,
                 hipLaunchKernelGGL(k_count_solid, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64, 256), 4096)), dim3(256), 0, st, is_solid, n_words64, ctr));
    c->global_marks_inverted = global && !mark_atomic;
    if (n_words64 && !mark_atomic && mark_mode_used == 1 && !global) {
      if (s.fixed_len >= k + 16 && c->opt("s1_pack_fixed", 1))
        MHX_LAUNCH(c, "pack_solid", (double)n_words64 * 72,
                   hipLaunchKernelGGL(k_pack_solid_inv_fixed, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64 * 4, 256), 4096)), dim3(256), 0, st, solid_bytes, n_bits,
                                      s.fixed_len, (int)k, is_solid, n_words64, ctr));
      else
        MHX_LAUNCH(c, "pack_solid", (double)n_words64 * 72,
                   hipLaunchKernelGGL(k_pack_solid_inv, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64, 256), 4096)), dim3(256), 0, st, solid_bytes, n_bits,
                                      s.start.as<uint64_t>(), s.n_seqs, s.fixed_len, (int)k, is_solid, n_words64, ctr));
    }
    if (n_words64 && !mark_atomic && (mark_mode_used != 1 || global))  // global: the marks themselves (see polarity)
      MHX_LAUNCH(c, "pack_solid", (double)n_words64 * 72,
                 hipLaunchKernelGGL(k_pack_solid, dim3((unsigned)std::min<uint64_t>(div_ceil(n_words64 * 4, 256), 4096)), dim3(256), 0, st, solid_bytes, n_bits, is_solid,
                                    n_words64, ctr));
    uint64_t n_solid = 0, n_mercy = 0;
    {
      unsigned long long h[2];
      MHX_HIP(hipMemcpyAsync(h, ctr, 16, hipMemcpyDeviceToHost, st));
      MHX_HIP(hipStreamSynchronize(st));
      n_solid = h[0];
      n_mercy = h[1];
      if (want_mercy && (c->accumulate || c->filter_on)) {  // keep the candidates of every pass; publish their sorted union
        const uint64_t prev = acc ? c->mercy_acc_n : 0;
        DevBuf &ma = grow_preserving(c, c->work["mercy_acc"], (prev + n_mercy) * 8 + 8, prev * 8);
        if (n_mercy) MHX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(ma.p) + prev * 8, mercy, n_mercy * 8, hipMemcpyDeviceToDevice, st));
        c->mercy_acc_n = n_mercy = prev + n_mercy;
        mercy = ma.as<long long>();
      }
      if (want_mercy && n_mercy) {
        int hi_bit = 3;
        while (hi_bit < 64 && ((n_bits << 2) >> hi_bit)) ++hi_bit;
        const uint64_t *ms = sort_u64(c, mercy, n_mercy, hi_bit);
        DevBuf &res = c->result(MHX_BUF_MERCY_CAND, n_mercy * 8);
        MHX_HIP(hipMemcpyAsync(res.p, ms, n_mercy * 8, hipMemcpyDeviceToDevice, st));
      }
    }
    if (!want_mercy || !n_mercy) {
      c->result(MHX_BUF_MERCY_CAND, 8);
      c->results[MHX_BUF_MERCY_CAND].used = 0;
    }
    c->solid_plain_k = global ? 0 : k;  // (the bitmap is stage 1's own for this (k, m) until mercy edges or the caller change it)
    c->solid_plain_m = m;
    c->results[MHX_BUF_SORTED_ITEMS].release();
    c->results[MHX_BUF_SORTED_ITEMS].p = sorted;
    c->results[MHX_BUF_SORTED_ITEMS].cap = 0;
    c->results[MHX_BUF_SORTED_ITEMS].used = sorted ? n_items * item_bytes : 0;  // (pre-sorted sources stay where they are)
    c->sorted_item_words = S;
    MHX_HIP(hipStreamSynchronize(st));
    if (out) {
      out->n_items = n_items;
      out->n_solid = n_solid;
      out->n_mercy_cand = want_mercy ? n_mercy : 0;
      out->item_words = S;
    }
  }
};
}  // namespace

int s1_process(mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy, uint32_t *buf_a, uint32_t *buf_b, uint64_t n_items,
               mhx_s1_result *out, const S1Sources *pre) {
  S1Stage stage(c, k, m, want_mercy, buf_a, buf_b, n_items, pre);
  c->last_s1_plan = s1_plan_text(c, k, n_items);
  if (c->s1_var_gen) c->last_s1_plan += " [reads of several lengths: " + std::to_string(c->seqs.max_len - k + 4) + " item slots per read on the generating pass]";
  c->s1_var_gen = false;
  stage.sort_records();
  stage.open_outputs();
  if (n_items) {
    stage.polarity();
    const bool done = stage.plan.seg_bits && stage.group_partial();
    if (!done) stage.group_classic();
    stage.mark_mode_used = stage.mark_mode;
  } else {
    stage.agg_prepare_classic();  // no launch at all: the cursor still has to hold the earlier passes' count
  }
  stage.publish(out);
  return 0;
}

// multi-GPU, sparse marks: the routed marks of the local reads (global positions of their non-solid (k+1)-mer occurrences)
// -> local is_solid = "a (k+1)-mer starts here and it is not marked" (MHX_BUF_IS_SOLID_LOCAL, read by stage 2)
void s1_apply_marks(mhx_ctx *c, const unsigned long long *recv, uint64_t n) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const uint64_t need = div_ceil(s.n_bases, 64);
  uint8_t *bytes = c->ws("solid_bytes_local", (need + 1) * 64).as<uint8_t>();
  MHX_HIP(hipMemsetAsync(bytes, 0, (need + 1) * 64, st));
  unsigned long long *ctr = c->ws("s1_counters", 64).as<unsigned long long>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  uint32_t *bad = reinterpret_cast<uint32_t *>(ctr + 4);
  if (n)
    MHX_LAUNCH(c, "apply_marks", (double)n * 40,
               hipLaunchKernelGGL(k_apply_marks, dim3((unsigned)div_ceil(n, 256)), dim3(256), 0, st, recv, n, c->pos_base, s.n_bases, bytes, bad));
  DevBuf &b = c->result(MHX_BUF_IS_SOLID_LOCAL, (need + 1) * 8);
  b.used = need * 8;
  if (need && s.fixed_len >= c->s1_acc_k + 16 && c->opt("s1_pack_fixed", 1))
    MHX_LAUNCH(c, "pack_solid", (double)need * 72,
               hipLaunchKernelGGL(k_pack_solid_inv_fixed, dim3((unsigned)std::min<uint64_t>(div_ceil(need * 4, 256), 4096)), dim3(256), 0, st, bytes, s.n_bases, s.fixed_len,
                                  (int)c->s1_acc_k, b.as<unsigned long long>(), need, ctr));
  else if (need)
    MHX_LAUNCH(c, "pack_solid", (double)need * 72,
               hipLaunchKernelGGL(k_pack_solid_inv, dim3((unsigned)std::min<uint64_t>(div_ceil(need, 256), 4096)), dim3(256), 0, st, bytes, s.n_bases, s.start.as<uint64_t>(),
                                  s.n_seqs, s.fixed_len, (int)c->s1_acc_k, b.as<unsigned long long>(), need, ctr));
  unsigned long long h[5] = {0, 0, 0, 0, 0};
  MHX_HIP(hipMemcpyAsync(h, ctr, 40, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h[4] & 0xFFFFFFFFull) throw Error("dist_apply_routed: a mark outside this rank's reads (ranks disagree on the global layout)");
  c->dist_local_solid = h[0];
  c->global_marks_inverted = false;
}

// ---- `count` on the bucket streaming (k_s1_stream<COUNT>): k <= 22, min count <= 2; one GPU (also pass by pass under the bucket filter
// of a memory plan) or, on several GPUs, at the bucket owners behind the pre-sorted exchange (comm.hip dist_count_presorted) ----
static uint64_t count_plan_items(const mhx_ctx *c, uint32_t k) {  // the item count the plan is made for (var: an estimate)
  const SeqSet &s = c->seqs;
  return s.fixed_len ? s.n_seqs * (uint64_t)(s.fixed_len - k) : s.n_bases - s.n_seqs * (uint64_t)k;
}
// what both forms need: the shape the generators serve, the plan's digits in the first key word, a 32-bit table key
static bool count_stream_shape(const mhx_ctx *c, uint32_t k, uint32_t m, S1Plan *plan_out) {
  const SeqSet &s = c->seqs;
  if (!c->opt("count_stream", 1)) return false;
  if (c->filter_on && !c->opt("s1_filter_in_gen", 1)) return false;
  // (a caller that asks for a particular form of the tile path gets the tile path)
  if (!c->opt("count_seg", 1) || c->opt("count_seg_bits", 0) || !c->opt("count_extract_fixed", 1)) return false;
  // (min count 1, 2: seen-once / seen-twice bits per prev / next char in the table slot; 3..15: 4-bit counters that stop at m)
  if (!s.n_seqs || k < 9 || m < 1 || m > 15) return false;  // (k: count_shape_is_fast — up to 22 with a shared window per run, up to 27 with one per item)
  if (!count_shape_is_fast(c, k)) return false;  // (reads of several lengths: item slots padded to the longest read's, CountGenVarT)
  if (!c->opt("s1_fused_first_pass", 1) || !c->opt("sort_unit_runs", 1) || !c->opt("s1_gen_any_order", 1)) return false;
  const uint64_t n_bits = c->count_edges_only ? 0 : (c->global_bases ? c->global_bases : s.n_bases);  // (edges only: nobody reads the positions)
  if ((n_bits >> s1_pos_bits(c)) >= 256) return false;  // (positions beyond the tags)
  const uint64_t n_items = count_plan_items(c, k);
  const S1Plan plan = s1_plan(c, k, n_items, true, 0);
  if (!plan.stream || plan.passes.empty() || (int)plan.passes.size() > kFastPasses) return false;
  for (const SortPass &ps : plan.passes)
    if (ps.bits2 || ps.shift < 32) return false;  // (digits: bit fields of the first key word)
  // the table key is the (k+1)-mer below the prefix: 32 bits (the all-ones word stands for an empty slot) up to k = 22 at a 16-bit
  // prefix, the 64-bit form beyond (k = 23..27, or a forced narrow prefix — s1_stream_bits), which carries no position tags
  if (2 * ((int)k + 1) - plan.seg_bits > 31 && (n_bits >> s1_pos_bits(c)) != 0) return false;
  if (plan_out) *plan_out = plan;
  return sort_takes_generated_first_pass(c, c->filter_on ? std::max<uint64_t>(c->filter_expected, 1) : n_items, 3, plan.passes);
}
bool count_stream_applies(const mhx_ctx *c, uint32_t k, uint32_t m) {
  // (under a bucket filter — a pass of the memory plan — the generating pass keeps only the records of the kept lv1 buckets and the
  //  plan follows the density of that bucket range, as in stage 1; first_0_out / last_0_in and the histogram accumulate over the passes)
  if (c->global_bases || c->n_parts > 1) return false;
  return count_stream_shape(c, k, m, nullptr);
}
// several GPUs: this rank's say (the ranks decide together, comm.hip); the plan follows the density they agreed on (mhx_ctx::s1_density)
bool count_presort_applies(const mhx_ctx *c, uint32_t k, uint32_t m) {
  if (!c->global_bases || c->n_parts > kStreamSrcMax || !c->opt("dist_presort", 1)) return false;
  return count_stream_shape(c, k, m, nullptr);
}
// the front half on a rank of a multi-GPU run: this rank's records, made by the first sort pass and ordered by the plan's prefix
// (n_items == 0: a rank or pass without an edge — nothing made, `sorted` is an empty buffer)
uint32_t *count_presort(mhx_ctx *c, uint32_t k, uint64_t *n_items, uint32_t **other, int *pbits) {
  const S1Plan plan = s1_plan(c, k, std::max<uint64_t>(count_plan_items(c, k), 1), true, 0);
  if (!plan.stream) throw Error("count_presort: the bucket-streaming plan does not apply");
  *pbits = plan.seg_bits;
  uint32_t *buf_a = nullptr, *buf_b = nullptr;
  *n_items = 0;
  c->gen_first_pass = nullptr;
  if (!count_stream_front(c, k, plan, &buf_a, &buf_b, n_items)) {
    *n_items = 0;
    c->gen_first_pass = nullptr;
    c->pre_hist_buf = nullptr;
    buf_a = c->ws("items_a", 64).as<uint32_t>();
    *other = c->ws("items_b", 64).as<uint32_t>();
    return buf_a;
  }
  uint32_t *sorted = radix_sort(c, buf_a, buf_b, *n_items, 3, 2, plan.passes);
  c->pre_hist_buf = nullptr;
  *other = sorted == buf_a ? buf_b : buf_a;
  return sorted;
}
// records made by the first sort pass, prefix passes, bucket streaming.  -> false: gave up (an output region too small): nothing
// published, the caller runs the extraction + tile path; true: the solid edges lie in the per-workgroup regions of *spare.
// pre: the records lie pre-sorted by the plan's prefix in several arrays (several GPUs: one per sending rank); the events that move
// first_0_out / last_0_in then leave as a list (o->events) for the ranks that hold the reads instead of being applied here.
bool count_stream_groups(mhx_ctx *c, uint32_t k, uint32_t m, uint32_t *first_0_out, uint32_t *last_0_in_p1, unsigned long long *hist,
                         CountStreamOut *o, const S1Sources *pre) {
  SeqSet &s = c->seqs;
  hipStream_t st = c->stream;
  const bool global = c->global_bases != 0;
  if (global != (pre != nullptr)) throw Error("count_stream_groups: pre-sorted sources belong to the multi-GPU layout and the other way round");
  const S1Plan plan = s1_plan(c, k, std::max<uint64_t>(count_plan_items(c, k), 1), true, 0);
  const int KWv = 2;
  uint32_t *sorted = nullptr, *spare = nullptr;
  uint64_t n_items = 0;
  if (pre) {
    if (pre->pbits != plan.seg_bits) throw Error("count: the sources were sorted for another plan than the one this rank makes");
    if (pre->n > kStreamSrcMax) throw Error("count: more pre-sorted sources than the bucket streaming takes");
    for (uint64_t v : pre->count) n_items += v;
    spare = pre->spare;
  } else {
    uint32_t *buf_a = nullptr, *buf_b = nullptr;
    if (!count_stream_front(c, k, plan, &buf_a, &buf_b, &n_items)) return false;  // (no read holds an edge: the general path knows what to publish)
    sorted = radix_sort(c, buf_a, buf_b, n_items, 3, KWv, plan.passes);
    c->pre_hist_buf = nullptr;
    spare = sorted == buf_a ? buf_b : buf_a;
  }
  const uint64_t n_bits = c->count_edges_only ? 0 : (global ? c->global_bases : s.n_bases);
  const uint64_t pos_stride = (n_bits >> s1_pos_bits(c)) ? 1ull << s1_pos_bits(c) : 0ull;
  // bucket streaming
  const uint64_t n_buckets = 1ull << plan.seg_bits;
  const uint64_t cus = c->n_cus > 0 ? (uint64_t)c->n_cus : 256;
  // every workgroup's edge region is its share of the spare sort buffer (1.5 entries per record of the job): a small job takes fewer
  // workgroups, so that a region holds at least ~6000 edges — with one workgroup, every record's (a region that overflows is
  // found only after the whole pass ran, and the tile path then repeats the work)
  const unsigned grid = (unsigned)std::min<uint64_t>(std::min<uint64_t>(n_buckets, cus), std::max<uint64_t>(n_items / 4096, 1));
  const uint32_t region = (uint32_t)std::min<uint64_t>(n_items * 12 / 8 / grid, 0xFFFFFFF0u) & ~1u;  // (even: 16-byte entries at k >= 24)
  uint32_t *counts = c->ws("cs_edge_counts", (size_t)grid * 4).as<uint32_t>();
  unsigned long long *ctr = c->ws("s1_counters", 64).as<unsigned long long>();
  uint32_t *seg_err = c->ws("s1_seg_err", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ctr, 0, 64, st));
  MHX_HIP(hipMemsetAsync(seg_err, 0, 4, st));
  const int n_src = pre ? pre->n : 1;
  uint64_t *bounds = c->ws("s1_bucket_bounds", (size_t)n_src * (n_buckets + 1) * 8 + 64).as<uint64_t>();
  uint32_t *ticket = c->ws("s1_stream_ticket", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ticket, 0, 4, st));
  const unsigned bgrid = (unsigned)((n_buckets + 1 + 255) / 256);
  const uint32_t *const *srcs = nullptr;
  if (pre) {
    DevBuf &sp = c->ws("s1_src_ptrs", (size_t)n_src * 8 + 64);
    MHX_HIP(hipMemcpyAsync(sp.p, pre->ptr.data(), (size_t)n_src * 8, hipMemcpyHostToDevice, st));
    srcs = sp.as<const uint32_t *>();
    for (int q = 0; q < n_src; ++q)
      hipLaunchKernelGGL(k_bucket_bounds, dim3(bgrid), dim3(256), 0, st, pre->ptr[q], pre->count[q], 3, bounds + (size_t)q * (n_buckets + 1), plan.seg_bits);
  } else {
    MHX_LAUNCH(c, "bucket_bounds", (double)n_buckets * 8 * 30,
               hipLaunchKernelGGL(k_bucket_bounds, dim3(bgrid), dim3(256), 0, st, sorted, n_items, 3, bounds, plan.seg_bits));
  }
  S1SegArgs a{};
  a.k = (int)k;
  a.m = m;
  a.mark_mode = 1;
  a.hist = hist;
  a.ctr = ctr;
  a.agg_raw = reinterpret_cast<uint2 *>(spare);
  a.agg_cap = region;
  a.agg_counts = counts;
  a.pos_stride = pos_stride;
  a.err = seg_err;
  a.la_chunks = (int)c->opt("s1_stream_probes", 1024);
  a.c_start = s.start.as<uint64_t>();
  a.c_n_seqs = s.n_seqs;
  a.c_fixed_len = s.fixed_len;
  a.first_0_out = first_0_out;
  a.last_0_in_p1 = last_0_in_p1;
  a.c_wpe = c->count_edges_only ? 3 : (int)div_ceil((k + 1) * 2 + 16, 32);  // (edges only: always the 16-byte entries, s2.hip reads them)
  a.c_edges_only = c->count_edges_only ? 1 : 0;
  uint32_t ecap = 0;
  uint32_t *ecounts = nullptr;
  if (global) {
    // events (8 bytes each; at most two per record of a solid key without an in- or out-edge: read ends, tips — a few per thousand
    // records) in per-workgroup regions of a buffer of their own: the spare sort buffer holds the edge regions
    const uint64_t total = std::max<uint64_t>(n_items / (uint64_t)std::max<long long>(c->opt("count_event_share", 4), 1), (uint64_t)grid * 4096);
    ecap = (uint32_t)std::min<uint64_t>(total / grid, 0xFFFFFFF0u);
    a.marks_raw = c->ws("cs_events", (size_t)grid * ecap * 8 + 64).as<unsigned long long>();
    a.marks_cap = ecap;
    a.marks_counts = ecounts = c->ws("cs_event_counts", (size_t)grid * 4).as<uint32_t>();
  }
  const S1StreamGeom geo{plan.seg_bits, plan.sub0, (uint32_t)n_buckets,
                         (uint32_t)std::min<long long>(std::max<long long>(c->opt("s1_stream_fill", 8192 * 7 / 8), 1), 8192)};
  const double bytes = (double)n_items * 12 * (double)(1u << plan.sub0);
  const uint32_t *items0 = pre ? pre->ptr[0] : sorted;
  const bool key64 = 2 * ((int)k + 1) - plan.seg_bits > 31;  // the (k+1)-mer below the prefix: wider than a 32-bit table key
  // giant buckets (round 6; S1Giant as in stage 1): a bucket of >= s1_giant_min records — 5 % poly-G reads put 6.6 x 10^7 records of
  // one key into one — is cut into slices that many workgroups reduce into partial entries (a key's count and per-char counters in
  // the slice); the streaming launch skips it, a second launch over the same grid inserts the partial entries and does the per-key
  // work; a solid key without an in- or out-edge in such a bucket sends that launch's workgroup over the bucket's records once
  const bool giant_on = c->opt("s1_giant", 1) != 0 && c->opt("count_giant", 1) != 0;
  uint32_t *ticket2 = nullptr;
  const uint32_t *giant_ctr = nullptr;
  if (giant_on) {
    S1Giant &g = a.giant;
    g.gcap = 4096;
    g.min_records = (uint32_t)std::max<long long>(1, c->opt("s1_giant_min", 262144));
    g.pcap = std::max<uint64_t>(2u << 20, n_items / 64);
    g.flag = c->ws("s1_giant_flag", n_buckets + 64).as<uint8_t>();
    uint32_t *lists = c->ws("s1_giant_lists", 64 + (size_t)g.gcap * (5 * 4 + 8)).as<uint32_t>();
    g.ctr = lists;
    ticket2 = lists + 8;
    g.off = reinterpret_cast<unsigned long long *>(lists + 16);
    g.bucket = lists + 16 + 2 * g.gcap;
    g.sl = g.bucket + g.gcap;
    g.ns = g.sl + g.gcap;
    g.cap = g.ns + g.gcap;
    g.cur = g.cap + g.gcap;
    g.partial = c->ws("s1_giant_partial", g.pcap * 16 + 64).as<uint4>();
    giant_ctr = lists;
    // the listed keys of every giant (k_count_giant_look) and, on several GPUs, that kernel's events
    g.fl_cap = kGiantFlagged;
    g.fl_cnt = c->ws("cs_giant_fl_cnt", (size_t)g.gcap * 4 + 64).as<uint32_t>();
    g.fl_key = c->ws("cs_giant_fl_key", (size_t)g.gcap * g.fl_cap * 8 + 64).as<unsigned long long>();
    MHX_HIP(hipMemsetAsync(g.fl_cnt, 0, (size_t)g.gcap * 4 + 64, st));
    if (global) {
      g.ev_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_items / 8, 1u << 20), 0xFFFFFFF0u);
      g.ev = c->ws("cs_giant_events", (size_t)g.ev_cap * 8 + 64).as<unsigned long long>();
      g.ev_cur = g.fl_cnt + g.gcap;  // (one of the spare words behind the counters, zeroed with them)
    }
    MHX_HIP(hipMemsetAsync(g.flag, 0, n_buckets, st));
    MHX_HIP(hipMemsetAsync(lists, 0, 64, st));
    s1_giant_launch(c, items0, srcs, bounds, n_src, n_buckets, plan.seg_bits, (int)k, g, key64, true);
  }
  S1StreamLaunch sl{true, false, pos_stride != 0, false, true, grid, items0, bounds, a, geo, 1u, ticket, srcs, n_src};
  sl.key64 = key64;
  s1_stream_launch(c, "count_groups", bytes, sl);
  if (giant_on) {
    S1StreamLaunch gl{true, false, pos_stride != 0, true, true, grid, items0, bounds, a, geo, 1u, ticket2, srcs, n_src};
    gl.key64 = key64;
    s1_stream_launch(c, "count_giant_groups", 0.0, gl);
    count_giant_look_launch(c, items0, srcs, bounds, n_src, n_buckets, plan.seg_bits, a);
  }
  uint32_t e = 0;
  unsigned long long h_ctr[8] = {0};
  std::vector<uint32_t> h_ec(global ? grid : 0);
  MHX_HIP(hipMemcpyAsync(&e, seg_err, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipMemcpyAsync(h_ctr, ctr, 64, hipMemcpyDeviceToHost, st));
  if (global) MHX_HIP(hipMemcpyAsync(h_ec.data(), ecounts, (size_t)grid * 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  o->grid = grid;
  o->cap = region;
  o->counts = counts;
  o->spare = spare;
  o->sorted = sorted;
  o->n_items = n_items;
  o->n_distinct = h_ctr[4];
  o->plan = s1_plan_text(c, k, std::max<uint64_t>(count_plan_items(c, k), 1));
  if (giant_ctr) {
    uint32_t h_giant[4] = {0, 0, 0, 0};
    MHX_HIP(hipMemcpy(h_giant, giant_ctr, 16, hipMemcpyDeviceToHost));
    if (h_giant[0]) o->plan += " [" + std::to_string(h_giant[0]) + " giant buckets in slices]";
  }
  o->events = nullptr;
  o->n_events = 0;
  if (e == 0 && global) {  // the workgroups' event regions (+ the events of the look at the giants) -> one list
    for (uint32_t v : h_ec) o->n_events += v;
    uint32_t n_look = 0;
    if (giant_on) {
      MHX_HIP(hipMemcpy(&n_look, a.giant.ev_cur, 4, hipMemcpyDeviceToHost));
      n_look = std::min(n_look, a.giant.ev_cap);
    }
    unsigned long long *dense = c->ws("cs_events_dense", (o->n_events + n_look) * 8 + 64).as<unsigned long long>();
    if (o->n_events)
      MHX_LAUNCH(c, "events_compact", (double)o->n_events * 16,
                 hipLaunchKernelGGL(k_agg_compact, dim3(grid, 8), dim3(256), 0, st, reinterpret_cast<const uint2 *>(a.marks_raw), ecap, ecounts,
                                    reinterpret_cast<uint2 *>(dense), 0));
    if (n_look) MHX_HIP(hipMemcpyAsync(dense + o->n_events, a.giant.ev, (size_t)n_look * 8, hipMemcpyDeviceToDevice, st));
    o->n_events += n_look;
    o->events = dense;
  }
  return e == 0;
}

// stage 1 on super-k-mer records.  -> false: the shape is served but this input is not (more records than windows / 2, a bin of
// low-complexity reads, an output region that overflowed): nothing published, the prefix plan runs from scratch
static bool s1_skm_try(mhx_ctx *c, uint32_t k, uint32_t m, mhx_s1_result *out, std::string *why) {
  // (a job whose record arrays would take more than s1_skm_pass_gb runs in passes over ranges of bins: marks, histogram and aggregated
  //  items add up; every pass scans the reads once more)
  const int n_passes = s1_skm_passes(c, k);
  SkmFront f{};
  S1Stage stage(c, k, m, 0, nullptr, nullptr, 0, nullptr);
  stage.sorted = nullptr;
  uint64_t n_records = 0;
  uint32_t max_bin = 0;
  for (int p = 0; p < n_passes; ++p) {
    if (!s1_skm_front(c, k, &f, p, n_passes)) {
      *why = f.n_records ? "a bin of " + std::to_string(f.max_bin) + " records" : "more records than the array holds";
      return false;
    }
    stage.set_spare(f.spare);
    if (p == 0) stage.open_outputs();
    if (!stage.run_skm(f)) {
      *why = "an output region overflowed";
      return false;
    }
    n_records += f.n_records;
    max_bin = std::max(max_bin, f.max_bin);
  }
  stage.publish_skm_hp(f);
  stage.n_items = f.n_items;  // what the reference sorts (read_to_sdbg_s1.cpp:344-363)
  stage.mark_mode_used = 1;
  char txt[320];
  snprintf(txt, sizeof txt, "super-k-mers m%u, 2^%d bins (%llu records for %llu windows: %.2f per record; largest bin %u)%s", k + 1 - 9, f.bin_bits,
           (unsigned long long)n_records, (unsigned long long)f.n_windows, n_records ? (double)f.n_windows / (double)n_records : 0.0, max_bin,
           c->seqs.fixed_len ? "" : " [reads of several lengths]");
  c->last_s1_plan = txt;
  if (n_passes > 1) c->last_s1_plan += " [" + std::to_string(n_passes) + " passes over ranges of bins]";
  stage.publish(out);
  return true;
}

// several GPUs: the owner's half of stage 1 on super-k-mer records — the sources are the record slices the ranks sent for this rank's bins
// (comm.hip dist_s1_skm); marks leave as a list of global positions for the read owners, aggregated items stay for stage 2's exchange
bool s1_skm_owner(mhx_ctx *c, uint32_t k, uint32_t m, const SkmFront &f, mhx_s1_result *out) {
  S1Stage stage(c, k, m, 0, nullptr, nullptr, f.n_items, nullptr);
  stage.sorted = nullptr;
  stage.set_spare(f.spare);
  stage.open_outputs();
  if (!stage.sparse) throw Error("s1_skm_owner: the marks of a global layout leave as a list (dist_sparse_marks)");
  if (!stage.run_skm(f)) return false;
  stage.mark_mode_used = 1;
  char txt[320];
  snprintf(txt, sizeof txt, "super-k-mers m%u, 2^%d bins (%llu records at this owner, bins %u..%u) [records exchanged by bin]", k + 1 - 9, f.bin_bits,
           (unsigned long long)f.n_records, f.bin_lo, f.bin_hi);
  c->last_s1_plan = txt;
  stage.publish(out);
  return true;
}

int run_s1(mhx_ctx *c, uint32_t k, uint32_t m, int want_mercy, mhx_s1_result *out) {
  if (c->global_bases) throw Error("read2sdbg_s1: a global layout is set; use the mhx_dist_* entry points");
  std::string skm_why;
  if (s1_skm_applies(c, k, m, want_mercy)) {
    c->gen_first_pass = nullptr;
    c->pre_hist_buf = nullptr;
    if (s1_skm_try(c, k, m, out, &skm_why)) return 0;
    // s1_skm = 3: a caller that left the memory plan to this path (mhx_s1_self_planned) hears that it did not serve — the prefix plan
    // of a whole job that was never cut into lv1 bucket ranges may not fit
    if (c->opt("s1_skm", 1) == 3) throw Error("read2sdbg_s1: super-k-mer records given up (" + skm_why + ")");
  }
  c->s1_defer_items = !want_mercy;  // s1_process sorts "items_a" first thing: its first pass may make the records (and apply a bucket filter)
  c->gen_first_pass = nullptr;
  const StageItems it = extract_stage(c, want_mercy ? MHX_STAGE_S1_MERCY : MHX_STAGE_S1, k, m);
  c->s1_defer_items = false;
  uint32_t *buf_a = c->work["items_a"].as<uint32_t>();
  uint32_t *buf_b = c->ws("items_b", it.n * (size_t)it.S * 4 + 64).as<uint32_t>();
  const int rc = s1_process(c, k, m, want_mercy, buf_a, buf_b, it.n, out);
  if (!skm_why.empty()) c->last_s1_plan += " [super-k-mer records given up: " + skm_why + "]";
  return rc;
}

}  // namespace mhx
