// Digit description shared by the radix sort (sort.hip) and the extraction kernels that pre-compute its histograms.
#pragma once
#include <cstdint>

namespace mhx {

// where a pass's digit lives: up to two bit fields (word index, bit offset, mask); field 2 is the upper part
struct DigitSpec {
  int wi1;
  unsigned bit1, mask1;
  int wi2;
  unsigned bit2, mask2, sh2;  // mask2 == 0: single field
  unsigned prev_mask = 0;     // bits of word wi1 that the earlier passes of the plan have sorted (k_radix_onesweep_u RANK 2)
};
constexpr int kMaxFusedPasses = 16;  // digit histograms taken in one read of the input
struct DigitSpecs {
  DigitSpec d[kMaxFusedPasses];
  int n;
};

// digit of a record held in registers: bits [bit, bit+nbits) of word wi (and wi-1 when straddling)
template <int S>
__device__ __forceinline__ unsigned words_digit(const uint32_t (&w)[S], int wi, unsigned bit, unsigned mask) {
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    if (i == wi) lo = w[i];
    if (i == wi - 1) hi = w[i];
  }
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return (unsigned)(v >> bit) & mask;
}
template <int S>
__device__ __forceinline__ unsigned words_digit2(const uint32_t (&w)[S], const DigitSpec &ds) {
  unsigned d = words_digit<S>(w, ds.wi1, ds.bit1, ds.mask1);
  if (ds.mask2) d |= words_digit<S>(w, ds.wi2, ds.bit2, ds.mask2) << ds.sh2;
  return d;
}

}  // namespace mhx
