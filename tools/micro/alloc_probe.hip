// hipMalloc cost on this box: one large allocation against several smaller ones, first use (memset) included, and the same again after
// a hipFree (is the cost per byte mapped, per call, or paid once per process?).   hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const double total_gb = argc > 1 ? atof(argv[1]) : 200.0;
  hipFree(nullptr);
  for (int round = 0; round < 2; ++round)
    for (int parts : {1, 8, 40}) {
      std::vector<void *> p(parts);
      const size_t each = (size_t)(total_gb * 1e9 / parts);
      double t0 = now();
      for (int i = 0; i < parts; ++i)
        if (hipMalloc(&p[i], each) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
      double t1 = now();
      for (int i = 0; i < parts; ++i) hipMemsetAsync(p[i], 0, each, nullptr);
      hipDeviceSynchronize();
      double t2 = now();
      for (int i = 0; i < parts; ++i) hipFree(p[i]);
      double t3 = now();
      printf("{\"round\": %d, \"total_GB\": %.0f, \"allocations\": %d, \"hipMalloc_s\": %.3f, \"first_memset_s\": %.3f, \"hipFree_s\": %.3f}\n", round, total_gb, parts, t1 - t0, t2 - t1, t3 - t2);
      fflush(stdout);
    }
  return 0;
}
