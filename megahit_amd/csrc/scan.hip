// Prefix sums and group-head compaction (reduce -> scan of block sums -> downsweep).
// HBM-bound streaming kernels; each element is read twice and written once.
#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

template <class TIn>
__global__ __launch_bounds__(kScanThreads) void k_block_sums(const TIn *__restrict__ in, uint64_t n, uint64_t *__restrict__ block_sums) {
  __shared__ uint64_t sm[kScanThreads / kWave + 1];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    uint64_t idx = base + (uint64_t)i * kScanThreads + threadIdx.x;
    if (idx < n) s += in[idx];
  }
  uint64_t tot;
  block_exclusive_sum<uint64_t, kScanThreads>(s, sm, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of m values in place, total to *d_total
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint64_t *__restrict__ v, uint64_t m, uint64_t *__restrict__ d_total) {
  __shared__ uint64_t sm[1024 / kWave + 1];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < m; base += 1024) {
    uint64_t idx = base + threadIdx.x;
    uint64_t x = idx < m ? v[idx] : 0, tot;
    uint64_t ex = block_exclusive_sum<uint64_t, 1024>(x, sm, &tot);
    if (idx < m) v[idx] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}

template <class TIn>
__global__ __launch_bounds__(kScanThreads) void k_downsweep(const TIn *__restrict__ in, uint64_t *__restrict__ out, uint64_t n,
                                                            const uint64_t *__restrict__ block_offs) {
  __shared__ uint64_t sm[kScanThreads / kWave + 1];
  // blocked arrangement: thread t owns kScanItems consecutive elements
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint64_t vals[kScanItems], s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    vals[i] = (base + i < n) ? (uint64_t)in[base + i] : 0;
    s += vals[i];
  }
  uint64_t ex = block_exclusive_sum<uint64_t, kScanThreads>(s, sm, nullptr) + block_offs[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += vals[i];
  }
}

template <class TIn>
static void exclusive_scan_impl(mhx_ctx *c, const TIn *in, uint64_t *out, uint64_t n, uint64_t *d_total) {
  if (n == 0) {
    if (d_total) MHX_HIP(hipMemsetAsync(d_total, 0, 8, c->stream));
    return;
  }
  uint64_t nb = div_ceil(n, kScanTile);
  uint64_t *bs = c->ws("scan_block_sums", nb * 8).as<uint64_t>();
  MHX_LAUNCH(c, "scan_block_sums", (double)n * sizeof(TIn),
             hipLaunchKernelGGL(k_block_sums<TIn>, dim3((unsigned)nb), dim3(kScanThreads), 0, c->stream, in, n, bs));
  MHX_LAUNCH(c, "scan_top", (double)nb * 16,
             hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, c->stream, bs, nb, d_total));
  MHX_LAUNCH(c, "scan_downsweep", (double)n * (sizeof(TIn) + 8),
             hipLaunchKernelGGL(k_downsweep<TIn>, dim3((unsigned)nb), dim3(kScanThreads), 0, c->stream, in, out, n, bs));
}

void exclusive_scan_u32_u64(mhx_ctx *c, const uint32_t *in, uint64_t *out, uint64_t n, uint64_t *d_total) {
  exclusive_scan_impl<uint32_t>(c, in, out, n, d_total);
}
void exclusive_scan_u64(mhx_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n, uint64_t *d_total) {
  exclusive_scan_impl<uint64_t>(c, in, out, n, d_total);
}

// ---- group heads ---------------------------------------------------------
// item i is a head iff i == 0 or the first cmp_bits bits of its key differ from item i-1.
__device__ __forceinline__ bool is_head(const uint32_t *__restrict__ items, uint64_t i, int stride, int full_words, uint32_t last_mask) {
  if (i == 0) return true;
  const uint32_t *a = items + i * stride, *b = a - stride;
  for (int w = 0; w < full_words; ++w)
    if (a[w] != b[w]) return true;
  if (last_mask && ((a[full_words] ^ b[full_words]) & last_mask)) return true;
  return false;
}

constexpr int kHeadThreads = 256;
constexpr int kHeadItems = 8;
constexpr int kHeadTile = kHeadThreads * kHeadItems;

__global__ __launch_bounds__(kHeadThreads) void k_count_heads(const uint32_t *__restrict__ items, uint64_t n, int stride, int full_words,
                                                             uint32_t last_mask, uint64_t *__restrict__ block_counts) {
  __shared__ uint64_t sm[kHeadThreads / kWave + 1];
  const uint64_t base = (uint64_t)blockIdx.x * kHeadTile + (uint64_t)threadIdx.x * kHeadItems;
  uint64_t cnt = 0;
#pragma unroll
  for (int i = 0; i < kHeadItems; ++i)
    if (base + i < n && is_head(items, base + i, stride, full_words, last_mask)) ++cnt;
  uint64_t tot;
  block_exclusive_sum<uint64_t, kHeadThreads>(cnt, sm, &tot);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kHeadThreads) void k_write_heads(const uint32_t *__restrict__ items, uint64_t n, int stride, int full_words,
                                                             uint32_t last_mask, const uint64_t *__restrict__ block_offs,
                                                             uint64_t *__restrict__ heads) {
  __shared__ uint64_t sm[kHeadThreads / kWave + 1];
  const uint64_t base = (uint64_t)blockIdx.x * kHeadTile + (uint64_t)threadIdx.x * kHeadItems;
  unsigned flags = 0;
  uint64_t cnt = 0;
#pragma unroll
  for (int i = 0; i < kHeadItems; ++i)
    if (base + i < n && is_head(items, base + i, stride, full_words, last_mask)) {
      flags |= 1u << i;
      ++cnt;
    }
  uint64_t pos = block_exclusive_sum<uint64_t, kHeadThreads>(cnt, sm, nullptr) + block_offs[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kHeadItems; ++i)
    if (flags & (1u << i)) heads[pos++] = base + i;
}

uint64_t count_group_heads(mhx_ctx *c, const uint32_t *items, uint64_t n, int stride, int cmp_bits) {
  if (n == 0) return 0;
  const int full_words = cmp_bits / 32, rem = cmp_bits % 32;
  const uint32_t last_mask = rem ? 0xFFFFFFFFu << (32 - rem) : 0;
  uint64_t nb = div_ceil(n, kHeadTile);
  uint64_t *bc = c->ws("head_block_counts", (nb + 1) * 8).as<uint64_t>();
  uint64_t *d_total = bc + nb;
  MHX_LAUNCH(c, "heads_count", (double)n * stride * 4,
             hipLaunchKernelGGL(k_count_heads, dim3((unsigned)nb), dim3(kHeadThreads), 0, c->stream, items, n, stride, full_words,
                                last_mask, bc));
  MHX_LAUNCH(c, "scan_top", (double)nb * 16, hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, c->stream, bc, nb, d_total));
  uint64_t total = 0;
  MHX_HIP(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  MHX_HIP(hipStreamSynchronize(c->stream));
  return total;
}

// must follow count_group_heads() with the same arguments (reuses its scanned block counts)
void find_group_heads(mhx_ctx *c, const uint32_t *items, uint64_t n, int stride, int cmp_bits, uint64_t *heads, uint64_t *d_count) {
  (void)d_count;
  if (n == 0) return;
  const int full_words = cmp_bits / 32, rem = cmp_bits % 32;
  const uint32_t last_mask = rem ? 0xFFFFFFFFu << (32 - rem) : 0;
  uint64_t nb = div_ceil(n, kHeadTile);
  uint64_t *bc = c->ws("head_block_counts", (nb + 1) * 8).as<uint64_t>();
  MHX_LAUNCH(c, "heads_write", (double)n * stride * 4,
             hipLaunchKernelGGL(k_write_heads, dim3((unsigned)nb), dim3(kHeadThreads), 0, c->stream, items, n, stride, full_words,
                                last_mask, bc, heads));
}

}  // namespace mhx
