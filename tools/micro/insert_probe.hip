// Micro-benchmark (not part of the product): the INSERT phase of the stage-1 bucket streaming alone — keys drawn like a
// 60x bucket (89 % of the records hit one of 360 hot keys, the rest are singletons: ~2 700 distinct keys per 20 000 records),
// no global loads, table wiped between buckets — in several forms, to see what a form costs before it goes into k_s1_stream.
//   hipcc --offload-arch=gfx950 -O3 -o insert_probe tools/micro/insert_probe.hip && ./insert_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int UNR = 4;

__device__ __forceinline__ uint32_t next_key(uint32_t &x, uint32_t bucket, uint32_t serial) {
  x = x * 1664525u + 1013904223u;
  const uint32_t r = x >> 8;
  // hot: one of 360 keys of this bucket; cold: a key nobody else has
  const bool hot = (r & 0xFFu) < 228u;  // 0.89
  const uint32_t id = hot ? (r >> 8) % 360u : 1000u + serial;
  uint32_t k = (bucket * 2654435761u) ^ (id * 40503u + (id << 17));
  return (k & 0x3FFFFFC0u) | ((id * 7u) & 0x1Bu);  // 24 "mer" bits << 6 | head/tail without bits 0x24
}

// FORM 0: as k_s1_stream today: CAS loop with add, first-position store, used list through a shared counter inside the loop
// FORM 1: the same loop without the used list / shared counter (claims counted per thread)
// FORM 2: first probe straight-line for all UNR records, the unresolved ones retried together afterwards
// FORM 3: FORM 1 with the add and the first-position store moved behind the loop
template <int FORM, int NT, int LOGS>
__global__ __launch_bounds__(NT) void k_insert(uint32_t *out, int buckets, int trips, int probe_limit) {
  constexpr int NSLOT = 1 << LOGS;
  __shared__ uint32_t keys[NSLOT];
  __shared__ uint32_t cnts[NSLOT];
  __shared__ uint32_t fpos[NSLOT];
  __shared__ uint16_t used[FORM == 0 ? NSLOT : 1];
  __shared__ uint32_t s_nused, s_bad;
  const int tid = threadIdx.x;
  for (int i = tid; i < NSLOT; i += NT) {
    keys[i] = kEmpty;
    cnts[i] = 0;
  }
  if (tid == 0) s_nused = 0, s_bad = 0;
  __syncthreads();
  uint32_t x = blockIdx.x * 7919u + tid * 104729u + 12345u, claims = 0, acc = 0;
  for (int b = 0; b < buckets; ++b) {
    const uint32_t bucket = blockIdx.x * 4096u + b;
    for (int t = 0; t < trips; ++t) {
      uint32_t lk[UNR], w2[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const uint32_t serial = ((t * UNR + u) * NT + tid);
        lk[u] = next_key(x, bucket, serial);
        w2[u] = serial;
      }
      if (FORM == 0 || FORM == 1 || FORM == 3) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          uint32_t h = (lk[u] * 0x9E3779B1u) >> (32 - LOGS);
          int probes = 0;
          bool claimed = false;
          for (; probes < probe_limit; ++probes) {
            const uint32_t old = atomicCAS(&keys[h], kEmpty, lk[u]);
            if (old == kEmpty || old == lk[u]) {
              if (FORM != 3) {
                atomicAdd(&cnts[h], 1u);
                if (old == kEmpty) {
                  fpos[h] = w2[u];
                  if (FORM == 0) {
                    const uint32_t at = atomicAdd(&s_nused, 1u);
                    if (at < (uint32_t)NSLOT) used[at] = (uint16_t)h;
                    if (at >= NSLOT * 7 / 8) s_bad = 1;
                  } else {
                    ++claims;
                  }
                }
              } else {
                claimed = old == kEmpty;
              }
              break;
            }
            h = (h + 1) & (NSLOT - 1);
          }
          if (probes == probe_limit) s_bad = 1;
          if (FORM == 3) {
            atomicAdd(&cnts[h], 1u);
            if (claimed) {
              fpos[h] = w2[u];
              ++claims;
            }
          }
        }
      } else {  // FORM 2
        uint32_t h[UNR];
        uint32_t pend = 0;  // bit u: record u still looks for its slot
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          h[u] = (lk[u] * 0x9E3779B1u) >> (32 - LOGS);
          const uint32_t old = atomicCAS(&keys[h[u]], kEmpty, lk[u]);
          if (old == kEmpty || old == lk[u]) {
            atomicAdd(&cnts[h[u]], 1u);
            if (old == kEmpty) {
              fpos[h[u]] = w2[u];
              ++claims;
            }
          } else {
            pend |= 1u << u;
            h[u] = (h[u] + 1) & (NSLOT - 1);
          }
        }
        int rounds = 0;
        while (__ballot(pend != 0)) {
          if (pend) {
            // the first pending record of this lane
            uint32_t pk = lk[0], ph = h[0], pw = w2[0];
            int pu = 0;
#pragma unroll
            for (int u = UNR - 1; u >= 0; --u)
              if (pend & (1u << u)) pk = lk[u], ph = h[u], pw = w2[u], pu = u;
            const uint32_t old = atomicCAS(&keys[ph], kEmpty, pk);
            if (old == kEmpty || old == pk) {
              atomicAdd(&cnts[ph], 1u);
              if (old == kEmpty) {
                fpos[ph] = pw;
                ++claims;
              }
              pend &= ~(1u << pu);
            } else {
              ph = (ph + 1) & (NSLOT - 1);
#pragma unroll
              for (int u = 0; u < UNR; ++u)
                if (u == pu) h[u] = ph;
            }
          }
          if (++rounds > probe_limit) {
            s_bad = 1;
            break;
          }
        }
      }
    }
    __syncthreads();
    // wipe (the product's per-key phases are not part of this probe)
    for (int i = tid; i < NSLOT; i += NT) {
      acc += cnts[i];
      keys[i] = kEmpty;
      cnts[i] = 0;
    }
    if (tid == 0) s_nused = 0;
    __syncthreads();
  }
  if (acc == 0x12345u || claims == 0x7654321u) out[0] = acc + claims + fpos[tid] + s_bad;
  if (tid == 0 && blockIdx.x == 0) out[1] = s_bad;
}

template <int FORM, int NT, int LOGS>
void run(const char *name, int wg_per_cu) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int records_per_bucket = 20480, trips = records_per_bucket / (NT * UNR), buckets = 64;
  uint32_t *out;
  hipMalloc(&out, 64);
  hipMemset(out, 0, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k_insert<FORM, NT, LOGS>), dim3(cus * wg_per_cu), dim3(NT), 0, 0, out, 2, trips, 1024);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_insert<FORM, NT, LOGS>), dim3(cus * wg_per_cu), dim3(NT), 0, 0, out, buckets, trips, 1024);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  uint32_t h[2];
  hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  const double rec_per_cu = (double)buckets * trips * NT * UNR * wg_per_cu;
  const double ns_rec = ms * 1e6 / rec_per_cu;
  printf("%-58s NT=%4d slots=%5d x%d: %7.3f ms  %.3f ns/record/CU = %5.2f cycles  -> 1.33 G records on 256 CUs: %5.2f ms  (bad=%u)\n", name, NT, 1 << LOGS,
         wg_per_cu, ms, ns_rec, ns_rec * 2.4, ns_rec * 1.33e9 / 256 * 1e-6, h[1]);
  hipFree(out);
}

int main() {
  run<0, 1024, 13>("0 today: loop with add, fpos, used list", 1);
  run<1, 1024, 13>("1 loop with add, fpos; claims per thread", 1);
  run<3, 1024, 13>("3 loop of CAS only; add + fpos behind it", 1);
  run<2, 1024, 13>("2 first probes straight-line, retries together", 1);
  run<1, 512, 12>("1 two workgroups per CU, 4096 slots", 2);
  run<2, 512, 12>("2 two workgroups per CU, 4096 slots", 2);
  run<1, 512, 13>("1 one 512-thread workgroup, 8192 slots", 1);
  run<1, 256, 12>("1 four 256-thread workgroups, 4096 slots (LDS limit: 3)", 3);
  return 0;
}
