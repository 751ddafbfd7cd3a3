// Micro-benchmark (not part of the product): how fast ONE workgroup per CU can read 12-byte records the way k_s1_stream does —
// per thread UNR records NT records apart (dwordx3), the next trip requested before the current one is used — against wider
// forms: the same bytes as dwordx4 loads of 48 contiguous bytes per thread, more loads in flight, two workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o read_probe tools/micro/read_probe.hip && ./read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef const __attribute__((address_space(1))) uint32_t *gptr;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4 *gptr4;

// FORM 0: dwordx3, record i of the trip = u * NT + tid (k_s1_stream today), DEPTH trips in flight
// FORM 1: three dwordx4 per thread = 4 consecutive records, DEPTH trips in flight
template <int FORM, int NT, int UNR, int DEPTH>
__global__ __launch_bounds__(NT) void k_read(const uint32_t *buf, uint64_t n_rec, uint64_t bucket, uint32_t *out, int spin) {
  const int tid = threadIdx.x;
  uint32_t acc = 0;
  const uint64_t n_buckets = n_rec / bucket;
  constexpr int TRIP = NT * UNR;
  for (uint64_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
    const gptr g = (gptr)buf + b * bucket * 3;
    const int trips = (int)(bucket / TRIP);
    if (FORM == 0) {
      uint32_t w[DEPTH][UNR][3];
#pragma unroll
      for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const gptr p = g + ((size_t)d * TRIP + u * NT + tid) * 3;
          w[d][u][0] = p[0], w[d][u][1] = p[1], w[d][u][2] = p[2];
        }
      for (int t = 0; t < trips; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int tn = t + d + DEPTH - 1;  // the trip requested now
          const int slot = (d + DEPTH - 1) % DEPTH;
          const size_t tt = tn < trips ? tn : trips - 1;
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const gptr p = g + (tt * TRIP + u * NT + tid) * 3;
            w[slot][u][0] = p[0], w[slot][u][1] = p[1], w[slot][u][2] = p[2];
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) acc += w[d][u][0] ^ w[d][u][1] ^ w[d][u][2];
          for (int s = 0; s < spin; ++s) acc = acc * 1664525u + 1013904223u;  // stands for the inserts
        }
      }
    } else {
      u32x4 w[DEPTH][3];
      const gptr4 g4 = (gptr4)g;
#pragma unroll
      for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
        for (int j = 0; j < 3; ++j) w[d][j] = g4[((size_t)d * NT + tid) * 3 + j];
      for (int t = 0; t < trips; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int tn = t + d + DEPTH - 1;
          const int slot = (d + DEPTH - 1) % DEPTH;
          const size_t tt = tn < trips ? tn : trips - 1;
#pragma unroll
          for (int j = 0; j < 3; ++j) w[slot][j] = g4[(tt * NT + tid) * 3 + j];
#pragma unroll
          for (int j = 0; j < 3; ++j) acc += w[d][j].x ^ w[d][j].y ^ w[d][j].z ^ w[d][j].w;
          for (int s = 0; s < spin; ++s) acc = acc * 1664525u + 1013904223u;
        }
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int FORM, int NT, int UNR, int DEPTH>
void run(const char *name, const uint32_t *buf, uint64_t n_rec, int wg_per_cu, int spin) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  uint32_t *out;
  hipMalloc(&out, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const uint64_t bucket = 20480;  // records (240 KB), a multiple of every trip size here
  hipLaunchKernelGGL((k_read<FORM, NT, UNR, DEPTH>), dim3(p.multiProcessorCount * wg_per_cu), dim3(NT), 0, 0, buf, n_rec / 8, bucket, out, spin);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_read<FORM, NT, UNR, DEPTH>), dim3(p.multiProcessorCount * wg_per_cu), dim3(NT), 0, 0, buf, n_rec, bucket, out, spin);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-44s NT=%4d UNR=%d depth=%d x%d spin=%3d: %7.3f ms  %6.0f GB/s\n", name, NT, UNR, DEPTH, wg_per_cu, spin, ms, n_rec * 12.0 / ms * 1e-6);
  hipFree(out);
}

int main() {
  const uint64_t n_rec = 20480ull * 65536 / 2;  // 8 GB of 12-byte records
  uint32_t *buf;
  if (hipMalloc(&buf, n_rec * 12 + 4096) != hipSuccess) return 1;
  hipMemset(buf, 1, n_rec * 12);
  run<0, 1024, 4, 2>("dwordx3 strided (today)", buf, n_rec, 1, 0);
  run<0, 1024, 4, 2>("dwordx3 strided (today)", buf, n_rec, 1, 150);
  run<0, 1024, 4, 3>("dwordx3 strided, 2 trips ahead", buf, n_rec, 1, 0);
  run<0, 1024, 4, 3>("dwordx3 strided, 2 trips ahead", buf, n_rec, 1, 150);
  run<1, 1024, 4, 2>("3 x dwordx4 = 4 records per thread", buf, n_rec, 1, 0);
  run<1, 1024, 4, 2>("3 x dwordx4 = 4 records per thread", buf, n_rec, 1, 150);
  run<1, 1024, 4, 3>("3 x dwordx4, 2 trips ahead", buf, n_rec, 1, 0);
  run<1, 1024, 4, 3>("3 x dwordx4, 2 trips ahead", buf, n_rec, 1, 150);
  run<0, 512, 4, 2>("dwordx3 strided, two workgroups per CU", buf, n_rec, 2, 0);
  run<1, 512, 4, 2>("3 x dwordx4, two workgroups per CU", buf, n_rec, 2, 0);
  run<0, 256, 8, 2>("dwordx3 strided, 256 threads x 3 per CU", buf, n_rec, 3, 0);
  return 0;
}
