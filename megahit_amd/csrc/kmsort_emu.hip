// Reference-exact tie order for read2sdbg --need_mercy.
//
// Read2SdbgS1::Lv2Postprocess re-reads the FIRST item of every (k-1)-mer group
// (reference src/sorting/read_to_sdbg_s1.cpp:399), so its mercy candidates depend on the order in which
// kmlib::kmsort (reference src/kmlib/kmsort.h:23-122) leaves records with equal keys.  That order is a
// deterministic function of each lv1 bucket's input order (global read order), but it is the result of
// in-place American-flag permutations, not of any stable rule.  To be bit-identical we replay exactly that
// permutation: records are first grouped by lv1 bucket with a STABLE pass (so every bucket holds its records
// in the reference's input order), then one GPU thread per bucket runs the same algorithm the reference
// runs with one CPU thread per bucket (65536-way parallel, serial inside a bucket).
#include "dev_prims.h"
#include "mhx_internal.h"

namespace mhx {

struct EmuSeg {
  uint32_t lo, hi;
  int byte;
};

template <int S>
struct EmuRec {
  uint32_t w[S];
};

template <int S>
__device__ __forceinline__ int emu_byte(const uint32_t *it, int key_words, int b) {  // Substr::kth_byte, kmsort_selector.cpp:28-32
  return (it[key_words - 1 - b / 4] >> ((b % 4) * 8)) & 0xFF;
}
template <int S>
__device__ __forceinline__ bool emu_less(const uint32_t *a, const uint32_t *b, int key_words) {  // Substr::operator<
  for (int i = 0; i < key_words; ++i) {
    if (a[i] < b[i]) return true;
    if (a[i] > b[i]) return false;
  }
  return false;
}
template <int S>
__device__ __forceinline__ void emu_load(const uint32_t *p, EmuRec<S> &r) {
#pragma unroll
  for (int i = 0; i < S; ++i) r.w[i] = p[i];
}
template <int S>
__device__ __forceinline__ void emu_store(uint32_t *p, const EmuRec<S> &r) {
#pragma unroll
  for (int i = 0; i < S; ++i) p[i] = r.w[i];
}

// insert_sort_core, kmsort.h:23-35 (stable)
template <int S>
__device__ void emu_insertion(uint32_t *s, uint32_t n, int key_words) {
  for (uint32_t i = 1; i < n; ++i) {
    if (emu_less<S>(s + (size_t)i * S, s + (size_t)(i - 1) * S, key_words)) {
      EmuRec<S> tmp;
      emu_load<S>(s + (size_t)i * S, tmp);
      uint32_t j = i;
      do {
        EmuRec<S> m;
        emu_load<S>(s + (size_t)(j - 1) * S, m);
        emu_store<S>(s + (size_t)j * S, m);
        --j;
      } while (j > 0 && emu_less<S>(tmp.w, s + (size_t)(j - 1) * S, key_words));
      emu_store<S>(s + (size_t)j * S, tmp);
    }
  }
}

template <int S>
__global__ __launch_bounds__(64) void k_kmsort_emulate(uint32_t *__restrict__ items, const uint64_t *__restrict__ bstart, int key_words,
                                                       EmuSeg *__restrict__ stacks, int stack_cap, uint32_t *__restrict__ overflow) {
  const uint32_t bk = blockIdx.x * blockDim.x + threadIdx.x;
  if (bk >= MHX_NUM_BUCKETS) return;
  const uint64_t s0 = bstart[bk];
  const uint64_t n64 = bstart[bk + 1] - s0;
  if (n64 <= 1) return;
  if (n64 > 0xFFFFFFF0ull) {
    atomicOr(overflow, 2u);
    return;
  }
  uint32_t *base = items + s0 * S;
  const uint32_t n = (uint32_t)n64;
  const int n_bytes = 4 * key_words - 2;  // kmsort_selector.cpp:16-17
  if (n <= 64) {                          // radix_sort_entry, kmsort.h:109-115
    emu_insertion<S>(base, n, key_words);
    return;
  }
  EmuSeg *stack = stacks + (size_t)bk * stack_cap;
  int sp = 0;
  stack[sp++] = {0u, n, n_bytes - 1};
  uint32_t count[256], cur_[257];
  uint32_t *cursor = cur_ + 1;  // cursor[-1] is valid, like the reference's last_[]
  while (sp > 0) {
    const EmuSeg sg = stack[--sp];
    const int b = sg.byte;
    // radix_sort_core, kmsort.h:45-106
    for (int i = 0; i < 256; ++i) count[i] = 0;
    for (uint32_t i = sg.lo; i < sg.hi; ++i) count[emu_byte<S>(base + (size_t)i * S, key_words, b)]++;
    cur_[0] = cur_[1] = sg.lo;
    for (int i = 1; i < 256; ++i) cursor[i] = cursor[i - 1] + count[i - 1];
    for (int i = 0; i < 256; ++i) {
      const uint32_t bin_end = cursor[i - 1] + count[i];
      if (bin_end == sg.hi) {
        cursor[i] = sg.hi;
        break;
      }
      while (cursor[i] != bin_end) {
        EmuRec<S> hold;
        emu_load<S>(base + (size_t)cursor[i] * S, hold);
        int tag = emu_byte<S>(hold.w, key_words, b);
        if (tag != i) {
          do {
            uint32_t *dst = base + (size_t)(cursor[tag]++) * S;
            EmuRec<S> t2;
            emu_load<S>(dst, t2);
            emu_store<S>(dst, hold);
            hold = t2;
          } while ((tag = emu_byte<S>(hold.w, key_words, b)) != i);
          emu_store<S>(base + (size_t)cursor[i] * S, hold);
        }
        ++cursor[i];
      }
    }
    if (b > 0) {
      for (int i = 0; i < 256; ++i) {
        const uint32_t lo = cursor[i - 1], hi = cursor[i];
        if (count[i] > 64) {
          if (sp < stack_cap) stack[sp++] = {lo, hi, b - 1};
          else atomicOr(overflow, 1u);
        } else if (count[i] > 1) {
          emu_insertion<S>(base + (size_t)lo * S, hi - lo, key_words);
        }
      }
    }
  }
}

__global__ void k_bucket_bounds(const uint32_t *__restrict__ items, uint64_t n, int stride, uint64_t *__restrict__ bstart) {
  const uint32_t bk = blockIdx.x * blockDim.x + threadIdx.x;  // first record whose bucket >= bk, bk = 0..65536
  if (bk > MHX_NUM_BUCKETS) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((items[mid * stride] >> 16) < bk) lo = mid + 1;
    else hi = mid;
  }
  bstart[bk] = lo;
}

// in: n records of stride S (key words first) in the reference's global emission order, in buf_a.
// out: pointer to the records sorted exactly as the reference's per-bucket kmsort leaves them.
uint32_t *kmsort_exact(mhx_ctx *c, uint32_t *buf_a, uint32_t *buf_b, uint64_t n, int S, int key_words) {
  if (n == 0) return buf_a;
  hipStream_t st = c->stream;
  // 1. stable grouping by lv1 bucket (top 16 bits of word 0)
  uint32_t *grouped = radix_sort(c, buf_a, buf_b, n, S, key_words, make_passes(key_words, key_words * 32 - 16, key_words * 32));
  // 2. bucket boundaries
  uint64_t *bstart = c->ws("emu_bstart", (MHX_NUM_BUCKETS + 2) * 8).as<uint64_t>();
  MHX_LAUNCH(c, "bucket_bounds", (double)MHX_NUM_BUCKETS * 8 * 30,
             hipLaunchKernelGGL(k_bucket_bounds, dim3((MHX_NUM_BUCKETS + 1 + 255) / 256), dim3(256), 0, st, grouped, n, S, bstart));
  // 3. replay kmsort, one thread per bucket
  const int stack_cap = 1536;
  EmuSeg *stacks = c->ws("emu_stacks", (size_t)MHX_NUM_BUCKETS * stack_cap * sizeof(EmuSeg)).as<EmuSeg>();
  uint32_t *ovf = c->ws("emu_overflow", 64).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(ovf, 0, 4, st));
  const double bytes = (double)n * S * 4 * 2 * (4 * key_words - 2);
#define MHX_CASE(SV)                                                                                                              \
  case SV:                                                                                                                        \
    MHX_LAUNCH(c, "kmsort_emulate", bytes,                                                                                        \
               hipLaunchKernelGGL(k_kmsort_emulate<SV>, dim3(MHX_NUM_BUCKETS / 64), dim3(64), 0, st, grouped, bstart, key_words, stacks, \
                                  stack_cap, ovf));                                                                               \
    break;
  switch (S) {
    MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
    default: throw Error("kmsort_exact: unsupported record stride");
  }
#undef MHX_CASE
  uint32_t h = 0;
  MHX_HIP(hipMemcpyAsync(&h, ovf, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h) throw Error(h & 2u ? "kmsort_exact: a bucket holds more than 2^32 records" : "kmsort_exact: segment stack overflow");
  return grouped;
}

}  // namespace mhx
