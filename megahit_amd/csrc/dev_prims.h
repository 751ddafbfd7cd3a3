// Device-side primitives shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mhx {

constexpr int kWave = 64;
constexpr unsigned kSentinel = 4;  // '$'

__device__ __forceinline__ int lane_id() { return __lane_id(); }

// ---- wave / block scans -------------------------------------------------
template <class T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T o = __shfl_up(v, d, kWave);
    if (lane_id() >= d) v += o;
  }
  return v;
}
template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}

// Block-wide exclusive sum for a 1-D block of NT threads (NT multiple of 64, <= 1024).
// `smem` must hold NT/64 + 1 elements of T.  Returns the exclusive prefix of v; *total = block sum.
template <class T, int NT>
__device__ __forceinline__ T block_exclusive_sum(T v, T *smem, T *total) {
  constexpr int NW = NT / kWave;
  const int w = threadIdx.x / kWave, l = lane_id();
  T inc = wave_inclusive_sum(v);
  if (l == kWave - 1) smem[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      T t = smem[i];
      smem[i] = run;
      run += t;
    }
    smem[NW] = run;
  }
  __syncthreads();
  T res = smem[w] + inc - v;
  if (total) *total = smem[NW];
  __syncthreads();
  return res;
}

// ---- 2-bit packed sequences ---------------------------------------------
__device__ __forceinline__ unsigned base_at(const uint32_t *__restrict__ seq, uint64_t i) {
  return (seq[i >> 4] >> (30 - 2 * (unsigned)(i & 15))) & 3u;
}

// reverse the 16 bases of a word and complement them
__device__ __forceinline__ uint32_t rc_word(uint32_t x) {
  x = __builtin_bitreverse32(x);
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  return ~x;
}
__device__ __forceinline__ uint32_t rev_word(uint32_t x) {  // reverse bases, no complement
  x = __builtin_bitreverse32(x);
  return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// (hi:lo) << sh, upper 32 bits, sh in [0,31]
__device__ __forceinline__ uint32_t funnel_l(uint32_t hi, uint32_t lo, unsigned sh) {
  return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}

// Chars [abs, abs+n) of the packed store, MSB-first, zero padded to KW words.
// Reads words abs/16 .. abs/16+KW (the store is padded so this never leaves the allocation).
template <int KW>
__device__ __forceinline__ void load_chars(const uint32_t *__restrict__ seq, uint64_t abs, int n, uint32_t (&out)[KW]) {
  const uint64_t w0 = abs >> 4;
  const unsigned sh = (unsigned)(abs & 15) * 2;
  uint32_t cur = seq[w0];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    uint32_t nxt = seq[w0 + i + 1];
    out[i] = funnel_l(cur, nxt, sh);
    cur = nxt;
  }
  const int full = n >> 4, rem = n & 15;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    if (i > full) out[i] = 0;
    else if (i == full) out[i] = rem ? (out[i] & (0xFFFFFFFFu << (32 - 2 * rem))) : 0u;
  }
}

// Reverse complement of the first n chars of `in` (MSB-first, KW words), zero padded.
// Requires KW*16 - n < 32 (always true for the item layouts of this path).
template <int KW>
__device__ __forceinline__ void rc_chars(const uint32_t (&in)[KW], int n, uint32_t (&out)[KW]) {
  uint32_t t[KW + 2];
#pragma unroll
  for (int i = 0; i < KW; ++i) t[i] = rc_word(in[KW - 1 - i]);
  t[KW] = 0;
  t[KW + 1] = 0;
  const int drop = KW * 16 - n;  // leading chars (complemented padding) to shift out
  const bool ws = drop >= 16;
  const unsigned bs = (unsigned)(drop & 15) * 2;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    uint32_t a = ws ? t[i + 1] : t[i];
    uint32_t b = ws ? t[i + 2] : t[i + 1];
    out[i] = funnel_l(a, b, bs);
  }
  // the bits below 2n came from t[KW..] = 0 only when drop chars were shifted in from zero words;
  // chars beyond n are already zero because t[KW], t[KW+1] are zero.
}

template <int KW>
__device__ __forceinline__ int cmp_words(const uint32_t (&a)[KW], const uint32_t (&b)[KW]) {
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}

__device__ __forceinline__ unsigned comp_or_sentinel(unsigned c) { return c == kSentinel ? kSentinel : 3u - c; }

// id of the sequence containing absolute base offset `off`: start[id] <= off < start[id+1]
__device__ __forceinline__ uint64_t seq_of_offset(const uint64_t *__restrict__ start, uint64_t n_seqs, uint32_t fixed_len,
                                                  uint64_t off) {
  if (fixed_len) return off / fixed_len;
  uint64_t lo = 0, hi = n_seqs;
  while (hi - lo > 1) {
    uint64_t mid = (lo + hi) >> 1;
    if (start[mid] <= off) lo = mid;
    else hi = mid;
  }
  return lo;
}

// dispatch a runtime key-word count to a compile-time template argument
#define MHX_DISPATCH_KW(kw, ...)                                             \
  switch (kw) {                                                              \
    case 1: { constexpr int KW = 1; __VA_ARGS__; } break;                    \
    case 2: { constexpr int KW = 2; __VA_ARGS__; } break;                    \
    case 3: { constexpr int KW = 3; __VA_ARGS__; } break;                    \
    case 4: { constexpr int KW = 4; __VA_ARGS__; } break;                    \
    case 5: { constexpr int KW = 5; __VA_ARGS__; } break;                    \
    case 6: { constexpr int KW = 6; __VA_ARGS__; } break;                    \
    case 7: { constexpr int KW = 7; __VA_ARGS__; } break;                    \
    case 8: { constexpr int KW = 8; __VA_ARGS__; } break;                    \
    case 9: { constexpr int KW = 9; __VA_ARGS__; } break;                    \
    case 10: { constexpr int KW = 10; __VA_ARGS__; } break;                  \
    case 11: { constexpr int KW = 11; __VA_ARGS__; } break;                  \
    case 12: { constexpr int KW = 12; __VA_ARGS__; } break;                  \
    case 13: { constexpr int KW = 13; __VA_ARGS__; } break;                  \
    case 14: { constexpr int KW = 14; __VA_ARGS__; } break;                  \
    case 15: { constexpr int KW = 15; __VA_ARGS__; } break;                  \
    case 16: { constexpr int KW = 16; __VA_ARGS__; } break;                  \
    case 17: { constexpr int KW = 17; __VA_ARGS__; } break;                  \
    default: throw mhx::Error("unsupported key width");                      \
  }

}  // namespace mhx
