// Reference-exact tie order for read2sdbg --need_mercy.
//
// Read2SdbgS1::Lv2Postprocess re-reads the FIRST item of every (k-1)-mer group
// (reference src/sorting/read_to_sdbg_s1.cpp:399), so its mercy candidates depend on the order in which
// kmlib::kmsort (reference src/kmlib/kmsort.h:23-122) leaves records with equal keys.  That order is a
// deterministic function of each lv1 bucket's input order (global read order), but it is the result of
// in-place American-flag permutations, not of any stable rule.  To be bit-identical we replay exactly that
// permutation: records are first grouped by lv1 bucket with a STABLE pass (so every bucket holds its records
// in the reference's input order), then one GPU thread per bucket runs the same algorithm the reference
// runs with one CPU thread per bucket.
//
// The replay works on tags and indices, not on records.  kmsort's only unstable step is the in-place American-flag
// permutation of radix_sort_core (kmsort.h:45-85); which record ends where in it depends on nothing but the sequence of
// the records' current radix bytes ("tags"):
//   every position is read exactly once, always at the head of some bin's unprocessed range (its cursor); the record read
//   there moves to the cursor of its own bin, and the next read happens at that very position (or, when the chain closes,
//   at the cursor of the bin being filled)
// so one wave per lv1 bucket keeps the tags of the current segment in LDS, lane 0 follows the chain over them (two LDS
// reads per record) and emits a permutation, all 64 lanes apply it to an index array; segments of <= 64 records are
// insertion-sorted (stable, kmsort.h:23-35) by one lane each, 64 at a time.  The records themselves move once, in a final
// gather into the spare buffer.  Buckets are binned by size so that 12 / 6 / 3 / 1 waves share a CU's 160 KB of LDS; only
// a bucket too large for LDS keeps its tags in global memory.
// The original one-thread-per-bucket replay on whole records stays behind the option kmsort_emu_legacy (A/B test).
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

// ---- wave-per-bucket replay on tags + indices ----------------------------------------------------------------------

constexpr int kEmuClasses = 5;  // 4 LDS-tag capacities + "tags in global memory"
struct EmuClassify {
  uint32_t count[kEmuClasses];
  uint32_t max_n[kEmuClasses];
  uint32_t next[kEmuClasses];
  uint32_t overflow;
};
struct EmuCaps {
  uint32_t cap[kEmuClasses - 1];
};
__global__ void k_emu_classify(const uint64_t *__restrict__ bstart, EmuCaps caps, EmuClassify *__restrict__ cl, uint32_t *__restrict__ lists) {
  const uint32_t bk = blockIdx.x * blockDim.x + threadIdx.x;
  if (bk >= MHX_NUM_BUCKETS) return;
  const uint64_t n = bstart[bk + 1] - bstart[bk];
  if (n == 0) return;
  if (n > 0xFFFFFFF0ull) {
    atomicOr(&cl->overflow, 2u);
    return;
  }
  int cls = kEmuClasses - 1;
  for (int i = kEmuClasses - 2; i >= 0; --i)
    if (n <= caps.cap[i]) cls = i;
  lists[(size_t)cls * MHX_NUM_BUCKETS + atomicAdd(&cl->count[cls], 1u)] = bk;
  atomicMax(&cl->max_n[cls], (uint32_t)n);
}

struct EmuArgs {
  const uint32_t *rec;    // records grouped by lv1 bucket, each bucket in the reference's input order
  uint32_t *out;          // same layout, every bucket in kmsort's output order
  const uint64_t *bstart;
  const uint32_t *list;   // buckets of this size class
  uint32_t n_list;
  uint32_t *next;         // work counter
  uint32_t *scratch;      // per workgroup: ord[cap] tmp[cap] perm[cap] stack[3 * stack_cap]
  uint64_t slot_words;
  uint32_t cap;
  uint32_t stack_cap;
  uint8_t *gtags;         // per workgroup cap bytes (class "global" only)
  int key_words;
  uint32_t *overflow;
};

template <int S>
__device__ __forceinline__ bool emu_less_idx(const uint32_t *__restrict__ rec, uint32_t a, uint32_t b, int key_words) {
  return emu_less<S>(rec + (size_t)a * S, rec + (size_t)b * S, key_words);
}
// insert_sort_core (kmsort.h:23-35) on indices: ord[lo .. lo+n)
template <int S>
__device__ void emu_insertion_idx(const uint32_t *__restrict__ rec, uint32_t *ord, uint32_t lo, uint32_t n, int key_words) {
  for (uint32_t i = 1; i < n; ++i) {
    const uint32_t oi = ord[lo + i];
    if (emu_less_idx<S>(rec, oi, ord[lo + i - 1], key_words)) {
      uint32_t j = i;
      do {
        ord[lo + j] = ord[lo + j - 1];
        --j;
      } while (j > 0 && emu_less_idx<S>(rec, oi, ord[lo + j - 1], key_words));
      ord[lo + j] = oi;
    }
  }
}

template <int S, bool LDS_TAGS>
__global__ __launch_bounds__(64) void k_kmsort_wave(EmuArgs a) {
  extern __shared__ uint8_t emu_lds_tags[];
  __shared__ uint32_t cnt[256], start[256], cur[256];
  __shared__ uint32_t sh_sp, sh_item, sh_moved;
  const uint32_t lane = threadIdx.x;
  uint32_t *ord = a.scratch + (size_t)blockIdx.x * a.slot_words;
  uint32_t *tmp = ord + a.cap;
  uint32_t *perm = tmp + a.cap;
  uint32_t *stack = perm + a.cap;
  uint8_t *gt = LDS_TAGS ? nullptr : a.gtags + (size_t)blockIdx.x * a.cap;
  auto tag_ld = [&](uint32_t p) -> uint32_t {
    if constexpr (LDS_TAGS) return emu_lds_tags[p];
    else return gt[p];
  };
  auto tag_st = [&](uint32_t p, uint32_t v) {
    if constexpr (LDS_TAGS) emu_lds_tags[p] = (uint8_t)v;
    else gt[p] = (uint8_t)v;
  };
  const int kw = a.key_words;
  const int n_bytes = 4 * kw - 2;  // kmsort_selector.cpp:16-17
  for (;;) {
    if (lane == 0) sh_item = atomicAdd(a.next, 1u);
    __syncthreads();
    const uint32_t li = sh_item;
    __syncthreads();
    if (li >= a.n_list) break;
    const uint32_t bk = a.list[li];
    const uint64_t s0 = a.bstart[bk];
    const uint32_t n = (uint32_t)(a.bstart[bk + 1] - s0);
    const uint32_t *rec = a.rec + s0 * S;
    for (uint32_t p = lane; p < n; p += 64) ord[p] = p;
    __syncthreads();
    if (n <= 64) {  // radix_sort_entry, kmsort.h:109-115
      if (lane == 0 && n > 1) emu_insertion_idx<S>(rec, ord, 0, n, kw);
    } else {
      if (lane == 0) {
        stack[0] = 0;
        stack[1] = n;
        stack[2] = (uint32_t)(n_bytes - 1);
        sh_sp = 1;
      }
      __syncthreads();
      for (;;) {
        const uint32_t sp = sh_sp;
        if (sp == 0) break;
        const uint32_t lo = stack[3 * (sp - 1)], hi = stack[3 * (sp - 1) + 1];
        const int b = (int)stack[3 * (sp - 1) + 2];
        __syncthreads();
        if (lane == 0) {
          sh_sp = sp - 1;
          sh_moved = 0;
        }
        for (uint32_t t = lane; t < 256; t += 64) cnt[t] = 0;
        __syncthreads();
        // radix_sort_core, kmsort.h:45-106.  tags + histogram
        const int wi = kw - 1 - b / 4, sh = (b & 3) * 8;
        for (uint32_t p = lo + lane; p < hi; p += 64) {
          const uint32_t tg = (rec[(size_t)ord[p] * S + wi] >> sh) & 255u;
          tag_st(p, tg);
          atomicAdd(&cnt[tg], 1u);
          perm[p] = p;
        }
        __syncthreads();
        {  // bin starts
          const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
          const uint32_t tot = c0 + c1 + c2 + c3;
          const uint32_t ex = lo + wave_inclusive_sum(tot) - tot;
          start[4 * lane] = cur[4 * lane] = ex;
          start[4 * lane + 1] = cur[4 * lane + 1] = ex + c0;
          start[4 * lane + 2] = cur[4 * lane + 2] = ex + c0 + c1;
          start[4 * lane + 3] = cur[4 * lane + 3] = ex + c0 + c1 + c2;
        }
        __syncthreads();
        if (lane == 0) {  // the permutation (kmsort.h:63-84) as a chain over the tags
          uint32_t moved = 0;
          for (uint32_t i = 0; i < 256; ++i) {
            const uint32_t ci = cnt[i];
            if (ci == 0) continue;
            const uint32_t end = start[i] + ci;
            if (end == hi) break;
            uint32_t cpos = cur[i];
            while (cpos != end) {
              uint32_t t = tag_ld(cpos);
              if (t == i) {
                ++cpos;
                continue;
              }
              const uint32_t hole = cpos;
              uint32_t at = cpos;
              do {
                const uint32_t w = cur[t];
                cur[t] = w + 1;
                perm[at] = w;
                at = w;
                t = tag_ld(at);
              } while (t != i);
              perm[at] = hole;
              cpos = hole + 1;
              moved = 1;
            }
          }
          sh_moved = moved;
        }
        __syncthreads();
        if (sh_moved) {
          for (uint32_t p = lo + lane; p < hi; p += 64) tmp[perm[p]] = ord[p];
          __syncthreads();
          for (uint32_t p = lo + lane; p < hi; p += 64) ord[p] = tmp[p];
          __syncthreads();
        }
        if (b > 0) {  // kmsort.h:87-105
          for (uint32_t t = lane; t < 256; t += 64) {
            const uint32_t ct = cnt[t];
            if (ct > 64) {
              const uint32_t idx = atomicAdd(&sh_sp, 1u);
              if (idx < a.stack_cap) {
                stack[3 * idx] = start[t];
                stack[3 * idx + 1] = start[t] + ct;
                stack[3 * idx + 2] = (uint32_t)(b - 1);
              } else {
                atomicOr(a.overflow, 1u);
              }
            } else if (ct > 1) {
              emu_insertion_idx<S>(rec, ord, start[t], ct, kw);
            }
          }
        }
        __syncthreads();
        if (sh_sp > a.stack_cap) {  // overflow recorded: stop on this bucket
          __syncthreads();
          if (lane == 0) sh_sp = 0;
          __syncthreads();
        }
      }
    }
    __syncthreads();
    uint32_t *out = a.out + s0 * S;
    for (uint32_t p = lane; p < n; p += 64) {
      const uint2 *src = reinterpret_cast<const uint2 *>(rec + (size_t)ord[p] * S);
      uint2 *dst = reinterpret_cast<uint2 *>(out + (size_t)p * S);
#pragma unroll
      for (int w = 0; w < S / 2; ++w) dst[w] = src[w];
    }
    __syncthreads();
  }
}

template <int S>
static void emu_launch_classes(mhx_ctx *c, const uint32_t *grouped, uint32_t *other, const uint64_t *bstart, int key_words) {
  hipStream_t st = c->stream;
  // tag capacities: (cap + 3.1 KB static) x {12, 6, 3, 1} workgroups <= 160 KB
  EmuCaps caps = {{10240u, 22528u, 49152u, 155648u}};
  const int waves_per_cu[kEmuClasses] = {12, 6, 3, 1, 2};
  bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_kmsort_wave<S, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)caps.cap[3]) == hipSuccess;
  if (!big_lds) {
    (void)hipGetLastError();
    caps.cap[3] = caps.cap[2];  // no workgroup-sized LDS allocation: those buckets keep their tags in global memory
  }
  EmuClassify *cl = c->ws("emu_classify", sizeof(EmuClassify)).as<EmuClassify>();
  uint32_t *lists = c->ws("emu_lists", (size_t)kEmuClasses * MHX_NUM_BUCKETS * 4).as<uint32_t>();
  MHX_HIP(hipMemsetAsync(cl, 0, sizeof(EmuClassify), st));
  hipLaunchKernelGGL(k_emu_classify, dim3(MHX_NUM_BUCKETS / 256), dim3(256), 0, st, bstart, caps, cl, lists);
  EmuClassify h;
  MHX_HIP(hipMemcpyAsync(&h, cl, sizeof h, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (h.overflow & 2u) throw Error("kmsort_exact: a bucket holds more than 2^32 records");
  const int n_cu = 256;  // MI355X
  static const char *ws_names[kEmuClasses] = {"emu_scratch_0", "emu_scratch_1", "emu_scratch_2", "emu_scratch_3", "emu_scratch_4"};
  for (int cls = 0; cls < kEmuClasses; ++cls) {
    if (h.count[cls] == 0) continue;
    const bool lds = cls < kEmuClasses - 1;
    EmuArgs a;
    a.rec = grouped;
    a.out = other;
    a.bstart = bstart;
    a.list = lists + (size_t)cls * MHX_NUM_BUCKETS;
    a.n_list = h.count[cls];
    a.next = &cl->next[cls];
    a.cap = lds ? std::min(caps.cap[cls], h.max_n[cls]) : h.max_n[cls];
    a.cap = (a.cap + 63u) & ~63u;
    a.stack_cap = a.cap / 65 + 2;
    a.key_words = key_words;
    a.overflow = &cl->overflow;
    a.slot_words = 3ull * a.cap + 3ull * a.stack_cap + 2;
    a.slot_words = (a.slot_words + 1) & ~1ull;
    uint32_t slots = std::min<uint64_t>(h.count[cls], (uint64_t)n_cu * waves_per_cu[cls]);
    if (!lds) {  // bound the scratch of the rare very large buckets to ~4 GB
      const uint64_t per_slot = a.slot_words * 4 + a.cap;
      slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(slots, (4ull << 30) / per_slot));
    }
    a.scratch = c->ws(ws_names[cls], (size_t)slots * a.slot_words * 4).as<uint32_t>();
    a.gtags = lds ? nullptr : c->ws("emu_gtags", (size_t)slots * a.cap).as<uint8_t>();
    const size_t lds_bytes = lds ? caps.cap[cls] : 0;
    if (lds)
      MHX_LAUNCH(c, "kmsort_wave", 0.0, hipLaunchKernelGGL((k_kmsort_wave<S, true>), dim3(slots), dim3(64), lds_bytes, st, a));
    else
      MHX_LAUNCH(c, "kmsort_wave_g", 0.0, hipLaunchKernelGGL((k_kmsort_wave<S, false>), dim3(slots), dim3(64), 0, st, a));
  }
  uint32_t ovf = 0;
  MHX_HIP(hipMemcpyAsync(&ovf, &cl->overflow, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (ovf) throw Error("kmsort_exact: segment stack overflow");
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
  // 3. replay kmsort: one wave per bucket on tags + indices, gathered into the spare buffer
  if (!c->opt("kmsort_emu_legacy", 0)) {
    uint32_t *other = grouped == buf_a ? buf_b : buf_a;
#define MHX_CASE(SV)                                                   \
  case SV:                                                             \
    emu_launch_classes<SV>(c, grouped, other, bstart, key_words);      \
    break;
    switch (S) {
      MHX_CASE(4) MHX_CASE(6) MHX_CASE(8) MHX_CASE(10) MHX_CASE(12) MHX_CASE(14) MHX_CASE(16) MHX_CASE(18) MHX_CASE(20)
      default: throw Error("kmsort_exact: unsupported record stride");
    }
#undef MHX_CASE
    return other;
  }
  // legacy: one thread per bucket, in place on whole records
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
