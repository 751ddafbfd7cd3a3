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
// reads per record) and emits a permutation, all 64 lanes apply it to an index array; segments of 65..255 records are
// replayed 16 at a time, one lane each, with byte-wide counters (a segment whose records agree in every remaining byte is
// dropped: nothing in it can move at any depth); segments of <= 64 records are insertion-sorted (stable,
// kmsort.h:23-35) by one lane each, 64 at a time.  The records themselves move once, in a final
// gather into the spare buffer.  Buckets are binned by size so that 8 / 4 / 2 / 1 waves share a CU's 160 KB of LDS; only
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

constexpr int kEmuClasses = 7;  // 6 LDS-tag capacities + "tags in global memory"
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
  uint32_t *scratch;      // per workgroup: ord[cap] tmp[cap] perm[cap] stack[3 * stack_cap] midq[3 * stack_cap] small[2 * (cap / 2 + 2)]
  uint64_t slot_words;
  uint32_t cap;
  uint32_t stack_cap;
  uint8_t *gtags;         // per workgroup cap bytes (class "global" only)
  int key_words;
  uint32_t *overflow;
  uint32_t lds_bytes;          // size of the LDS tag region (0: tags in global memory)
  unsigned long long *timing;  // MHX_EMU_TIMING builds: 16 phase counters
};

template <int S>
__device__ __forceinline__ bool emu_less_idx(const uint32_t *__restrict__ rec, uint32_t a, uint32_t b, int key_words) {
  return emu_less<S>(rec + (size_t)a * S, rec + (size_t)b * S, key_words);
}
// insert_sort_core (kmsort.h:23-35) on indices: ord[lo .. lo+n).  The key of the record in front stays in registers: a run
// of equal or ascending keys (the common case: copies of one k-mer) costs one index load + one key load per record.
template <int S>
__device__ void emu_insertion_idx(const uint32_t *__restrict__ rec, uint32_t *ord, uint32_t lo, uint32_t n, int key_words) {
  uint32_t prev[S], key[S];
  {
    const uint32_t *r = rec + (size_t)ord[lo] * S;
#pragma unroll
    for (int w = 0; w < S; ++w) prev[w] = w < key_words ? r[w] : 0u;
  }
  for (uint32_t i = 1; i < n; ++i) {
    const uint32_t oi = ord[lo + i];
    const uint32_t *r = rec + (size_t)oi * S;
    int c = 0;  // key(oi) vs the key in front
#pragma unroll
    for (int w = 0; w < S; ++w) {
      key[w] = w < key_words ? r[w] : 0u;
      if (c == 0 && key[w] != prev[w]) c = key[w] < prev[w] ? -1 : 1;
    }
    if (c >= 0) {
#pragma unroll
      for (int w = 0; w < S; ++w) prev[w] = key[w];
      continue;
    }
    {  // moves: the record in front of position i stays the largest so far, prev is unchanged
      uint32_t j = i;
      do {
        ord[lo + j] = ord[lo + j - 1];
        --j;
      } while (j > 0 && emu_less_idx<S>(rec, oi, ord[lo + j - 1], key_words));
      ord[lo + j] = oi;
    }
  }
}

// MHX_EMU_TIMING: per-phase wall-clock ticks (100 MHz) summed over all workgroups into EmuArgs::timing (diagnostic build only)
#ifdef MHX_EMU_TIMING
#define EMU_T(slot)                                     \
  do {                                                  \
    const unsigned long long now_ = wall_clock64();     \
    tacc[slot] += now_ - tlast;                         \
    tlast = now_;                                       \
  } while (0)
#define EMU_C(slot, v) tacc[slot] += (v)
#else
#define EMU_T(slot) \
  do {              \
  } while (0)
#define EMU_C(slot, v) \
  do {                 \
  } while (0)
#endif
constexpr int kEmuMidLanes = 16;      // segments of 65..255 records are replayed side by side, one lane each
constexpr uint32_t kEmuMidMax = 255;  // their counters and cursors are bytes

template <int S, bool LDS_TAGS>
__global__ __launch_bounds__(64) void k_kmsort_wave(EmuArgs a) {
  constexpr int G = kEmuMidLanes;
  extern __shared__ __attribute__((aligned(16))) uint8_t emu_lds_tags[];
  __shared__ uint32_t cnt[256], start[256], cur[256];
  __shared__ uint32_t mid_cnt[G * 64], mid_cur[G * 64];  // per lane 256 byte counters / cursors, dword d of lane g at g*64 + (d ^ 4g)
  __shared__ uint32_t rs_lo[G], rs_n[G], rs_b[G], rs_off[G + 1], rs_flag[G];
  __shared__ unsigned long long rs_mask[G];  // counter dwords (4 tags each) that are not empty
  __shared__ uint32_t sh_sp, sh_mid, sh_item, sh_moved, sh_small;
  const uint32_t lane = threadIdx.x;
  uint32_t *ord = a.scratch + (size_t)blockIdx.x * a.slot_words;
  uint32_t *tmp = ord + a.cap;
  uint32_t *perm = tmp + a.cap;
  uint32_t *stack = perm + a.cap;                    // segments of > 255 records: (lo, hi, byte)
  uint32_t *midq = stack + 3 * (size_t)a.stack_cap;  // segments of 65..255 records: (lo, n, byte)
  uint32_t *small = midq + 3 * (size_t)a.stack_cap;  // bins of 2..64 records: (lo, n); sorted 64 at a time at the end
  uint8_t *gt = LDS_TAGS ? nullptr : a.gtags + (size_t)blockIdx.x * a.cap;
  auto tag_ld = [&](uint32_t p) -> uint32_t {
    if constexpr (LDS_TAGS) return emu_lds_tags[p];
    else return gt[p];
  };
  auto tag_st = [&](uint32_t p, uint32_t v) {
    if constexpr (LDS_TAGS) emu_lds_tags[p] = (uint8_t)v;
    else gt[p] = (uint8_t)v;
  };
  auto push_seg = [&](uint32_t *q, uint32_t *counter, uint32_t x, uint32_t y, uint32_t z) {
    const uint32_t idx = atomicAdd(counter, 1u);
    if (idx < a.stack_cap) {
      q[3 * idx] = x;
      q[3 * idx + 1] = y;
      q[3 * idx + 2] = z;
    } else {
      atomicOr(a.overflow, 1u);
    }
  };
  auto push_small = [&](uint32_t x, uint32_t cn) {
    const uint32_t idx = atomicAdd(&sh_small, 1u);
    small[2 * idx] = x;
    small[2 * idx + 1] = cn;
  };
  const int kw = a.key_words;
  const int n_bytes = 4 * kw - 2;  // kmsort_selector.cpp:16-17
#ifdef MHX_EMU_TIMING
  unsigned long long tacc[16] = {0}, tlast = wall_clock64();
#endif
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
    if (lane == 0) {
      sh_sp = sh_mid = sh_small = 0;
      if (n > kEmuMidMax) {
        stack[0] = 0;
        stack[1] = n;
        stack[2] = (uint32_t)(n_bytes - 1);
        sh_sp = 1;
      } else if (n > 64) {
        midq[0] = 0;
        midq[1] = n;
        midq[2] = (uint32_t)(n_bytes - 1);
        sh_mid = 1;
      }
    }
    __syncthreads();
    EMU_T(0);
    if (n <= 64) {  // radix_sort_entry, kmsort.h:109-115
      if (lane == 0 && n > 1) emu_insertion_idx<S>(rec, ord, 0, n, kw);
    }
    for (;;) {
      const uint32_t sp = min(sh_sp, a.stack_cap), nm = min(sh_mid, a.stack_cap);
      __syncthreads();
      if (sp > 0) {
        // ---- one segment of > 255 records, all lanes: radix_sort_core, kmsort.h:45-106
        const uint32_t lo = stack[3 * (sp - 1)], hi = stack[3 * (sp - 1) + 1];
        const int b = (int)stack[3 * (sp - 1) + 2];
        EMU_C(15, 1);
        __syncthreads();
        if (lane == 0) {
          sh_sp = sp - 1;
          sh_mid = nm;
          sh_moved = 0;
        }
        for (uint32_t t = lane; t < 256; t += 64) cnt[t] = 0;
        __syncthreads();
        const int wi = kw - 1 - b / 4, sh = (b & 3) * 8;
        for (uint32_t p0 = lo + lane; p0 < hi; p0 += 256) {  // tags + histogram; 4 independent load chains per lane
          uint32_t o4[4], w4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) o4[u] = p0 + 64 * u < hi ? ord[p0 + 64 * u] : 0u;
#pragma unroll
          for (int u = 0; u < 4; ++u) w4[u] = p0 + 64 * u < hi ? rec[(size_t)o4[u] * S + wi] : 0u;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t p = p0 + 64 * u;
            if (p < hi) {
              const uint32_t tg = (w4[u] >> sh) & 255u;
              tag_st(p, tg);
              atomicAdd(&cnt[tg], 1u);
              perm[p] = p;
            }
          }
        }
        __syncthreads();
        EMU_T(1);
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
        EMU_T(2);
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
        EMU_T(3);
        if (sh_moved) {
          for (uint32_t p0 = lo + lane; p0 < hi; p0 += 256) {
            uint32_t d4[4], o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const bool in = p0 + 64 * u < hi;
              d4[u] = in ? perm[p0 + 64 * u] : 0u;
              o4[u] = in ? ord[p0 + 64 * u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (p0 + 64 * u < hi) tmp[d4[u]] = o4[u];
          }
          __syncthreads();
          for (uint32_t p0 = lo + lane; p0 < hi; p0 += 256) {
            uint32_t o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o4[u] = p0 + 64 * u < hi ? tmp[p0 + 64 * u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (p0 + 64 * u < hi) ord[p0 + 64 * u] = o4[u];
          }
          __syncthreads();
        }
        EMU_T(4);
        if (b > 0) {  // kmsort.h:87-105
          for (uint32_t t = lane; t < 256; t += 64) {
            const uint32_t ct = cnt[t];
            if (ct > kEmuMidMax) push_seg(stack, &sh_sp, start[t], start[t] + ct, (uint32_t)(b - 1));
            else if (ct > 64) push_seg(midq, &sh_mid, start[t], ct, (uint32_t)(b - 1));
            else if (ct > 1) push_small(start[t], ct);  // disjoint from every later segment: its insertion sort can wait
          }
        }
        __syncthreads();
        EMU_T(5);
      } else if (nm > 0) {
        // ---- up to G segments of 65..255 records, one lane each; tags, equality test and the index moves by all lanes
        const uint32_t base = nm > (uint32_t)G ? nm - G : 0, cr = nm - base;
        EMU_C(14, 1);
        if (lane < cr) {
          rs_lo[lane] = midq[3 * (base + lane)];
          rs_n[lane] = midq[3 * (base + lane) + 1];
          rs_b[lane] = midq[3 * (base + lane) + 2];
          rs_flag[lane] = 0;
          rs_mask[lane] = 0;
        }
        for (uint32_t i = lane; i < cr * 64; i += 64) mid_cnt[i] = 0;
        __syncthreads();
        if (lane == 0) {
          sh_mid = base;
          uint32_t run = 0;
          for (uint32_t g = 0; g < cr; ++g) {
            rs_off[g] = run;
            run += rs_n[g];
          }
          rs_off[cr] = run;
        }
        __syncthreads();
        const uint32_t total = rs_off[cr];
        auto seg_of = [&](uint32_t e) -> uint32_t {
          uint32_t g = 0;
          for (uint32_t q = 1; q < cr; ++q) g += rs_off[q] <= e ? 1u : 0u;
          return g;
        };
        for (uint32_t e0 = lane; e0 < total; e0 += 256) {  // 4 independent load chains per lane
          uint32_t g4[4], p4[4], o4[4], f4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t e = e0 + 64 * u;
            g4[u] = e < total ? seg_of(e) : 0u;
            p4[u] = rs_lo[g4[u]] + (e < total ? e - rs_off[g4[u]] : 0u);
            o4[u] = ord[p4[u]];
            f4[u] = ord[rs_lo[g4[u]]];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e0 + 64 * u >= total) continue;
            const uint32_t g = g4[u], p = p4[u];
            const int b = (int)rs_b[g];
            const int wi = kw - 1 - b / 4, sh = (b & 3) * 8;
            const uint32_t *kp = rec + (size_t)o4[u] * S, *k0 = rec + (size_t)f4[u] * S;
            const uint32_t word = kp[wi];
            const uint32_t tg = (word >> sh) & 255u;
            tag_st(p, tg);
            atomicAdd(&mid_cnt[g * 64 + ((tg >> 2) ^ ((g * 4) & 63))], 1u << ((tg & 3) * 8));  // byte counters: <= 255 records
            atomicOr(&rs_mask[g], 1ull << (tg >> 2));
            perm[p] = p;
            // a segment whose records agree in every remaining byte never moves again, at any depth: drop it
            const uint32_t mask = sh == 24 ? 0xffffffffu : ((1u << (sh + 8)) - 1u);
            bool ne = ((word ^ k0[wi]) & mask) != 0;
            for (int w = wi + 1; w < kw; ++w) ne = ne || kp[w] != k0[w];
            if (ne) rs_flag[g] = 1;
          }
        }
        __syncthreads();
        EMU_T(6);
        if (lane < cr && rs_flag[lane]) {
          const uint32_t g = lane, lo = rs_lo[g], m = rs_n[g];
          const int b = (int)rs_b[g];
          const uint32_t bd = g * 64, sw = (g * 4) & 63;
          const unsigned long long nonempty = rs_mask[g];
          uint32_t run = 0;
          for (unsigned long long mk = nonempty; mk; mk &= mk - 1) {
            const uint32_t d = (uint32_t)__builtin_ctzll(mk);
            const uint32_t c4 = mid_cnt[bd + (d ^ sw)];
            const uint32_t c0 = c4 & 255u, c1 = (c4 >> 8) & 255u, c2 = (c4 >> 16) & 255u, c3 = c4 >> 24;
            mid_cur[bd + (d ^ sw)] = run | ((run + c0) << 8) | ((run + c0 + c1) << 16) | ((run + c0 + c1 + c2) << 24);
            run += c0 + c1 + c2 + c3;
          }
          uint32_t moved = 0;
          run = 0;
          bool done = false;
          for (unsigned long long mk = nonempty; mk && !done; mk &= mk - 1) {
            const uint32_t d = (uint32_t)__builtin_ctzll(mk);
            const uint32_t c4 = mid_cnt[bd + (d ^ sw)];
            for (uint32_t j = 0; j < 4; ++j) {
              const uint32_t ci = (c4 >> (8 * j)) & 255u;
              if (ci == 0) continue;
              const uint32_t i = 4 * d + j, end = run + ci;
              if (end == m) {
                done = true;
                break;
              }
              uint32_t cpos = (mid_cur[bd + (d ^ sw)] >> (8 * j)) & 255u;
              while (cpos != end) {
                uint32_t t = tag_ld(lo + cpos);
                if (t == i) {
                  ++cpos;
                  continue;
                }
                const uint32_t hole = cpos;
                uint32_t at = cpos;
                do {
                  const uint32_t di = bd + ((t >> 2) ^ sw), sb = (t & 3) * 8;
                  const uint32_t v = mid_cur[di];
                  const uint32_t w = (v >> sb) & 255u;
                  mid_cur[di] = v + (1u << sb);
                  perm[lo + at] = lo + w;
                  at = w;
                  t = tag_ld(lo + at);
                } while (t != i);
                perm[lo + at] = lo + hole;
                cpos = hole + 1;
                moved = 1;
              }
              run = end;
            }
          }
          if (b > 0) {
            run = 0;
            for (unsigned long long mk = nonempty; mk; mk &= mk - 1) {
              const uint32_t d = (uint32_t)__builtin_ctzll(mk);
              const uint32_t c4 = mid_cnt[bd + (d ^ sw)];
              for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t ci = (c4 >> (8 * j)) & 255u;
                if (ci > 64) push_seg(midq, &sh_mid, lo + run, ci, (uint32_t)(b - 1));
                else if (ci > 1) push_small(lo + run, ci);
                run += ci;
              }
            }
          }
          if (moved) rs_flag[g] = 3;
        }
        __syncthreads();
        EMU_T(7);
        const bool any_moved = __ballot(lane < cr && (rs_flag[lane < cr ? lane : 0] & 2u)) != 0;
        if (any_moved) {
          for (uint32_t e0 = lane; e0 < total; e0 += 256) {
            uint32_t p4[4], d4[4], o4[4];
            bool in4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t e = e0 + 64 * u;
              const uint32_t g = e < total ? seg_of(e) : 0u;
              in4[u] = e < total && (rs_flag[g] & 2u);
              p4[u] = in4[u] ? rs_lo[g] + e - rs_off[g] : rs_lo[0];
              d4[u] = perm[p4[u]];
              o4[u] = ord[p4[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (in4[u]) tmp[d4[u]] = o4[u];
          }
          __syncthreads();
          for (uint32_t e0 = lane; e0 < total; e0 += 256) {
            uint32_t p4[4], o4[4];
            bool in4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t e = e0 + 64 * u;
              const uint32_t g = e < total ? seg_of(e) : 0u;
              in4[u] = e < total && (rs_flag[g] & 2u);
              p4[u] = in4[u] ? rs_lo[g] + e - rs_off[g] : rs_lo[0];
              o4[u] = tmp[p4[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (in4[u]) ord[p4[u]] = o4[u];
          }
        }
        __syncthreads();
        EMU_T(8);
      } else {
        break;
      }
    }
    __syncthreads();
    // ---- bins of 2..64 records: insert_sort_core (kmsort.h:23-35) is a stable sort; any stable sort leaves the same order
    const uint32_t n_small = sh_small;
    if constexpr (LDS_TAGS) {
      // rank sort out of LDS.  The tag region is free now: it stages the first DW key words + index of as many bins as fit
      // (<= 64 at a time); every lane then ranks one record among the records of its bin (#smaller + #equal in front of it)
      // and writes its index to that place.  Interleaved copies of two k-mers, which cost an insertion sort hundreds of
      // moves per bin, cost this nothing extra, and the 64 lanes stay busy whatever the bin sizes are.
      constexpr int DW = S <= 4 ? 2 : 4;
      const int dw = kw < DW ? kw : DW;
      const uint32_t lcap = a.lds_bytes / (4 * (DW + 1));
      typedef unsigned long long u64;
      u64 *l_key = reinterpret_cast<u64 *>(emu_lds_tags);  // record e: DW/2 64-bit words (big-endian word pairs), side by side
      uint32_t *l_id = reinterpret_cast<uint32_t *>(emu_lds_tags) + (size_t)DW * lcap;
      uint32_t *b_off = cnt, *b_lo = start, *b_cn = cur;  // per bin of the batch: staging offset, first position, size
      EMU_C(11, n_small);
      for (uint32_t e0 = 0; e0 < n_small;) {
        uint32_t take = min(64u, n_small - e0), mycn, off, total;
        for (;;) {  // as many bins as fit
          mycn = lane < take ? small[2 * (e0 + lane) + 1] : 0u;
          const uint32_t inc = wave_inclusive_sum(mycn);
          total = __shfl(inc, 63, 64);
          off = inc - mycn;
          if (total <= lcap || take == 1) break;
          take >>= 1;
        }
        EMU_C(12, 1);
        EMU_C(13, total);
        b_off[lane] = off;
        b_cn[lane] = mycn;
        b_lo[lane] = lane < take ? small[2 * (e0 + lane)] : 0u;
        __syncthreads();
        auto bin_of = [&](uint32_t e) -> uint32_t {  // last lane whose offset is <= e (offsets ascend, empty lanes sit at total)
          uint32_t g = 0;
#pragma unroll
          for (uint32_t stp = 32; stp > 0; stp >>= 1)
            if (b_off[g + stp] <= e) g += stp;
          return g;
        };
        for (uint32_t ea = lane; ea < total; ea += 256) {
          uint32_t o4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t e = ea + 64 * u;
            const uint32_t g = e < total ? bin_of(e) : 0u;
            o4[u] = ord[b_lo[g] + (e < total ? e - b_off[g] : 0u)];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t e = ea + 64 * u;
            if (e < total) {
              const uint32_t *r = rec + (size_t)o4[u] * S;
#pragma unroll
              for (int w = 0; w < DW; w += 2)
                l_key[(size_t)e * (DW / 2) + w / 2] = ((u64)(w < dw ? r[w] : 0u) << 32) | (w + 1 < dw ? r[w + 1] : 0u);
              l_id[e] = o4[u];
            }
          }
        }
        __syncthreads();
        for (uint32_t e = lane; e < total; e += 64) {
          const uint32_t g = bin_of(e);
          const uint32_t base = b_off[g], cn = b_cn[g], x = e - base;
          u64 mine[DW / 2];
#pragma unroll
          for (int w = 0; w < DW / 2; ++w) mine[w] = l_key[(size_t)e * (DW / 2) + w];
          const uint32_t my_id = l_id[e];
          uint32_t rank = 0;
          if (kw <= DW) {  // the staged words are the whole key: branch-free, one LDS read per pair
#pragma unroll 4
            for (uint32_t j = 0; j < cn; ++j) {
              bool lt, eq;
              if constexpr (DW == 2) {
                const u64 kj = l_key[base + j];
                lt = kj < mine[0];
                eq = kj == mine[0];
              } else {
                const u64 k0 = l_key[(size_t)(base + j) * 2], k1 = l_key[(size_t)(base + j) * 2 + 1];
                lt = k0 < mine[0] || (k0 == mine[0] && k1 < mine[1]);
                eq = k0 == mine[0] && k1 == mine[1];
              }
              rank += (lt || (eq && j < x)) ? 1u : 0u;
            }
          } else {  // longer keys: ties of the staged words are decided in global memory
            for (uint32_t j = 0; j < cn; ++j) {
              int c = 0;  // key j vs mine
#pragma unroll
              for (int w = 0; w < DW / 2; ++w) {
                const u64 kj = l_key[(size_t)(base + j) * (DW / 2) + w];
                if (c == 0 && kj != mine[w]) c = kj < mine[w] ? -1 : 1;
              }
              if (c == 0 && j != x) {
                const uint32_t oj = l_id[base + j];
                c = emu_less_idx<S>(rec, oj, my_id, kw) ? -1 : (emu_less_idx<S>(rec, my_id, oj, kw) ? 1 : 0);
              }
              rank += (c < 0 || (c == 0 && j < x)) ? 1u : 0u;
            }
          }
          ord[b_lo[g] + rank] = my_id;
        }
        __syncthreads();
        e0 += take;
      }
    } else {
      for (uint32_t e = lane; e < n_small; e += 64) emu_insertion_idx<S>(rec, ord, small[2 * e], small[2 * e + 1], kw);
    }
    __syncthreads();
    EMU_T(9);
    uint32_t *out = a.out + s0 * S;
    for (uint32_t p0 = lane; p0 < n; p0 += 256) {
      uint32_t o4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) o4[u] = p0 + 64 * u < n ? ord[p0 + 64 * u] : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p0 + 64 * u < n) {
          const uint2 *src = reinterpret_cast<const uint2 *>(rec + (size_t)o4[u] * S);
          uint2 *dst = reinterpret_cast<uint2 *>(out + (size_t)(p0 + 64 * u) * S);
#pragma unroll
          for (int w = 0; w < S / 2; ++w) dst[w] = src[w];
        }
      }
    }
    __syncthreads();
    EMU_T(10);
  }
#ifdef MHX_EMU_TIMING
  if (lane == 0 && a.timing)
    for (int i = 0; i < 16; ++i) atomicAdd(&a.timing[i], tacc[i]);
#endif
}

template <int S>
static void emu_launch_classes(mhx_ctx *c, const uint32_t *grouped, uint32_t *other, const uint64_t *bstart, int key_words) {
  hipStream_t st = c->stream;
  // tag capacities: (cap + 11.7 KB static) x {8, 5, 4, 3, 2, 1} workgroups <= 160 KB of LDS per CU
  EmuCaps caps = {{8192u, 16384u, 24576u, 40960u, 69632u, 147456u}};
  const int waves_per_cu[kEmuClasses] = {8, 5, 4, 3, 2, 1, 2};
  // more than 64 KB of LDS for one workgroup has to be asked for; without it those buckets keep their tags in global memory
  int n_lds = kEmuClasses - 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_kmsort_wave<S, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)caps.cap[kEmuClasses - 2]) != hipSuccess) {
    (void)hipGetLastError();
    n_lds = 4;
    caps.cap[4] = caps.cap[5] = caps.cap[3];
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
  static const char *ws_names[kEmuClasses] = {"emu_scratch_0", "emu_scratch_1", "emu_scratch_2", "emu_scratch_3",
                                              "emu_scratch_4", "emu_scratch_5", "emu_scratch_6"};
#ifdef MHX_EMU_TIMING
  MHX_HIP(hipMemsetAsync(c->ws("emu_timing", 16 * 8).p, 0, 16 * 8, st));
#endif
  // the classes run side by side, each on a stream of its own (largest buckets first): their workgroups share the CUs' LDS,
  // and no class waits for the tail of another
  while (c->side_streams.size() < (size_t)kEmuClasses) {
    hipStream_t ns;
    MHX_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
    c->side_streams.push_back(ns);
  }
  while (c->side_events.size() < (size_t)kEmuClasses + 1) {
    hipEvent_t ne;
    MHX_HIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
    c->side_events.push_back(ne);
  }
  c->prof_begin("kmsort_wave", 0.0);
  MHX_HIP(hipEventRecord(c->side_events[kEmuClasses], st));
  for (int cls = kEmuClasses - 1; cls >= 0; --cls) {
    if (h.count[cls] == 0) continue;
    const bool lds = cls < n_lds;
    hipStream_t ss = c->side_streams[cls];
    MHX_HIP(hipStreamWaitEvent(ss, c->side_events[kEmuClasses], 0));
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
    a.lds_bytes = lds ? caps.cap[cls] : 0;
    a.timing = nullptr;
#ifdef MHX_EMU_TIMING
    a.timing = c->ws("emu_timing", 16 * 8).as<unsigned long long>();
#endif
    a.slot_words = 3ull * a.cap + 6ull * a.stack_cap + 2ull * (a.cap / 2 + 2) + 2;
    a.slot_words = (a.slot_words + 1) & ~1ull;
    uint32_t slots = std::min<uint64_t>(h.count[cls], (uint64_t)n_cu * waves_per_cu[cls]);
    if (!lds) {  // bound the scratch of the rare very large buckets to ~4 GB
      const uint64_t per_slot = a.slot_words * 4 + a.cap;
      slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(slots, (4ull << 30) / per_slot));
    }
    a.scratch = c->ws(ws_names[cls], (size_t)slots * a.slot_words * 4).as<uint32_t>();
    a.gtags = lds ? nullptr : c->ws("emu_gtags", (size_t)slots * a.cap).as<uint8_t>();
#ifdef MHX_EMU_TIMING
    fprintf(stderr, "emu class %d: %u buckets, largest %u records, %u workgroups\n", cls, h.count[cls], h.max_n[cls], slots);
#endif
    if (lds) hipLaunchKernelGGL((k_kmsort_wave<S, true>), dim3(slots), dim3(64), (size_t)caps.cap[cls], ss, a);
    else hipLaunchKernelGGL((k_kmsort_wave<S, false>), dim3(slots), dim3(64), 0, ss, a);
    MHX_HIP(hipGetLastError());
    MHX_HIP(hipEventRecord(c->side_events[cls], ss));
    MHX_HIP(hipStreamWaitEvent(st, c->side_events[cls], 0));
  }
  c->prof_end();
#ifdef MHX_EMU_TIMING
  {
    unsigned long long t[16];
    MHX_HIP(hipMemcpyAsync(t, c->ws("emu_timing", 16 * 8).p, sizeof t, hipMemcpyDeviceToHost, st));
    MHX_HIP(hipStreamSynchronize(st));
    static const char *names[11] = {"init", "big fill", "big scan", "big walk", "big apply", "big children", "mid fill", "mid lanes", "mid apply", "small sort", "gather"};
    for (int i = 0; i < 11; ++i) fprintf(stderr, "emu timing %-12s %10.3f ms summed over workgroups\n", names[i], t[i] / 1e5);
    fprintf(stderr, "emu counts: small bins %llu, small batches %llu, records in small bins %llu, mid rounds %llu, big passes %llu\n", t[11], t[12], t[13], t[14], t[15]);
  }
#endif
  uint32_t ovf = 0;
  MHX_HIP(hipMemcpyAsync(&ovf, &cl->overflow, 4, hipMemcpyDeviceToHost, st));
  MHX_HIP(hipStreamSynchronize(st));
  if (ovf) throw Error("kmsort_exact: segment stack overflow");
}

// bstart[b] = first record whose top `pbits` key bits are >= b, b = 0 .. 2^pbits (records sorted on those bits; 16 = lv1 buckets)
__global__ void k_bucket_bounds(const uint32_t *__restrict__ items, uint64_t n, int stride, uint64_t *__restrict__ bstart, int pbits) {
  const uint32_t bk = blockIdx.x * blockDim.x + threadIdx.x;
  if (bk > (1u << pbits)) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((items[mid * stride] >> (32 - pbits)) < bk) lo = mid + 1;
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
             hipLaunchKernelGGL(k_bucket_bounds, dim3((MHX_NUM_BUCKETS + 1 + 255) / 256), dim3(256), 0, st, grouped, n, S, bstart, 16));
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
