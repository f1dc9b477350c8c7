// Per-workgroup record format of the fused field backward and its fixed-order reduction, shared by the backward
// translation units (fused_bwd_common.h) and tail.hip.
#pragma once
#include "cnr_common.h"

namespace cnr_rec {
using namespace cnr;
constexpr int ROWS_LDS = 4;  // rows_per_class <= 4: per-object bias-row sums travel in the record / the fixed-point table
constexpr int REC_FLOATS = ((TRUNK + 126 + ROWS_LDS * 128 + 255) / 256) * 256;  // one workgroup's record
constexpr double ROWS_FIX_SCALE = 1099511627776.0;  // 2^40: bias-row sums as int64 fixed point (order-free atomics)
constexpr int ROWS_FIX_COPIES = 8;  // the table is replicated: a workgroup adds into copy (its index & 7), which
                                    // cuts the same-address atomic queue 8-fold; consumers add the copies (exact)

// Entries no launch writes (latent-layer biases, padding, unused row sums) are skipped, so the workspace needs no
// clearing.
__device__ __forceinline__ bool rec_entry_written(int i, int rows_per_class) {
  if (i < TRUNK)
    return !((i >= OFF_S1_B && i < OFF_S1_B + 32) || (i >= OFF_CAT_B && i < OFF_CAT_B + 32) ||
             (i >= OFF_S2_B && i < OFF_S2_B + 32) || (i >= OFF_T1_B && i < OFF_T1_B + 32));
  if (i < TRUNK + 126) return true;
  return rows_per_class <= ROWS_LDS && (i - (TRUNK + 126)) < rows_per_class * 128;
}
// sum of entry i over records [w0, w1) of one class (r = that class's first record + i), 8 loads in flight
__device__ __forceinline__ float record_range_sum(const float* __restrict__ r, int w0, int w1) {
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.0f;
  int w = w0;
  for (; w + 7 < w1; w += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += r[(size_t)(w + u) * REC_FLOATS];
  }
  for (; w < w1; ++w) a[0] += r[(size_t)w * REC_FLOATS];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
}  // namespace cnr_rec
