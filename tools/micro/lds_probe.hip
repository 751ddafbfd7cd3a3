// Micro-benchmark (not part of the product): what an LDS wave-op costs on this device under the access patterns of the
// stage-1 bucket streaming (k_s1_stream): random-address compare-and-swap / add / read on an 8192-word table, one
// 1024-thread workgroup per CU.  Prints ns per wave-op per CU and the implied cycles.
//   hipcc --offload-arch=gfx950 -O3 -o lds_probe tools/micro/lds_probe.hip && ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int NSLOT = 8192;
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_probe(uint32_t *out, int iters, uint32_t seed) {
  __shared__ uint32_t keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  for (int i = threadIdx.x; i < NSLOT; i += NT) {
    keys[i] = 0xFFFFFFFFu;
    cnts[i] = 0;
  }
  __syncthreads();
  uint32_t x = seed + blockIdx.x * 7919u + threadIdx.x * 104729u, acc = 0;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t h = (x >> 8) & (NSLOT - 1);
    if (MODE == 6) h = ((x >> 8) & (NSLOT - 1) & ~63u) | lane;  // conflict-free: lane i -> bank i % 32
    if (MODE == 0) {  // CAS rtn, dependent use
      const uint32_t old = atomicCAS(&keys[h], 0xFFFFFFFFu, h);
      acc += old;
    } else if (MODE == 1) {  // add, no return
      atomicAdd(&cnts[h], 1u);
    } else if (MODE == 2) {  // plain read
      acc += __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 3 || MODE == 6) {  // CAS then add when matched (the insert's pair)
      const uint32_t old = atomicCAS(&keys[h], 0xFFFFFFFFu, h);
      if (old == 0xFFFFFFFFu || old == h) atomicAdd(&cnts[h], 1u);
    } else if (MODE == 4) {  // four independent pairs back to back
      uint32_t hh[4], oo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x = x * 1664525u + 1013904223u;
        hh[u] = (x >> 8) & (NSLOT - 1);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) oo[u] = atomicCAS(&keys[hh[u]], 0xFFFFFFFFu, hh[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (oo[u] == 0xFFFFFFFFu || oo[u] == hh[u]) atomicAdd(&cnts[hh[u]], 1u);
    } else if (MODE == 5) {  // CAS with a quarter of the lanes
      if ((lane & 3) == 0) acc += atomicCAS(&keys[h], 0xFFFFFFFFu, h);
    } else if (MODE == 7) {  // 64-bit add on an interleaved {key, cnt} slot: one op per record
      atomicAdd(reinterpret_cast<unsigned long long *>(keys) + (h >> 1), 1ull);
    } else if (MODE == 8) {  // read then add (the read_first form for a key that is present)
      const uint32_t old = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (old != 12345u) atomicAdd(&cnts[h], 1u);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = cnts[5];
}

template <int MODE, int NT>
void run(const char *name, int ops_per_iter, int blocks_per_cu) {
  int dev = 0;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, dev);
  const int cus = p.multiProcessorCount, iters = 4096;
  uint32_t *out;
  hipMalloc(&out, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k_probe<MODE, NT>), dim3(cus * blocks_per_cu), dim3(NT), 0, 0, out, 64, 1u);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_probe<MODE, NT>), dim3(cus * blocks_per_cu), dim3(NT), 0, 0, out, iters, 2u);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waveops = (double)iters * ops_per_iter * (NT / 64) * blocks_per_cu;  // per CU
  const double ns = ms * 1e6 / waveops;
  printf("%-46s NT=%4d x%d  %8.3f ms  %7.2f ns per record-instruction per CU = %6.1f cycles @2.4GHz  (%.2f ns per record)\n", name, NT, blocks_per_cu, ms, ns,
         ns * 2.4, ns / 64);
  hipFree(out);
}

int main() {
  run<0, 1024>("CAS rtn, random", 1, 1);
  run<1, 1024>("add no-rtn, random", 1, 1);
  run<2, 1024>("read, random", 1, 1);
  run<3, 1024>("CAS + dependent add, random", 1, 1);
  run<4, 1024>("4 x (CAS) then 4 x add, random", 4, 1);
  run<5, 1024>("CAS rtn, 16 of 64 lanes", 1, 1);
  run<6, 1024>("CAS + add, conflict-free banks", 1, 1);
  run<7, 1024>("64-bit add on {key,cnt}", 1, 1);
  run<8, 1024>("read + dependent add", 1, 1);
  run<3, 512>("CAS + dependent add, random", 1, 2);
  run<3, 256>("CAS + dependent add, random", 1, 4);
  run<3, 512>("CAS + dependent add, random (1 WG/CU)", 1, 1);
  return 0;
}
