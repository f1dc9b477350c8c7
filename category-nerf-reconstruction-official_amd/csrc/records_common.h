// Per-workgroup record format of the fused field backward and its fixed-order reduction, shared by the backward
// translation units (fused_bwd_common.h) and tail.hip.
#pragma once
#include "cnr_common.h"

namespace cnr_rec {
using namespace cnr;
// rows_per_class <= 15: per-object bias-row sums travel in the record / the fixed-point table (the 8-wave kernel's
// row-sum blocks have 32 rows each: one block of 4 latent slots x <= 7 objects + 2 bias rows, or two blocks of
// 2 latent slots x <= 15 objects, the first with the 2 bias rows)
constexpr int ROWS_MAX = 15;
// rows_per_class <= 128 on the one-launch path with one object per tile (fused_bwd_pipe8.hip, WIDE = 3): the sums go to the
// fixed-point table only, the record carries none.  (The reference caps a scene at n_models = 100 categories + instances,
// configs/Replica/config_replica_room0.json:15; nothing in the kernels depends on the count any more -- the tail launch's latent
// blocks read only the rows they need, latent_common.h.)
constexpr int ROWS_TILE_MAX = 128;
constexpr int REC_ENTRIES = ((TRUNK + 126 + ROWS_MAX * 128 + 255) / 256) * 256;  // one workgroup's record
// A record entry is a bf16 (round 4; fp32 before): 256 records x 64.5 KB written by the field kernel and read once by the step's
// last launch were 15 of the 21 MB the step body moves at configs[1].  Each workgroup's partial sum is rounded once (2^-9
// relative, unbiased); the fixed-order sum over the records stays fp32, so the summed gradient carries ~2^-9 / sqrt(records) =
// 1e-4 of rounding noise -- under the f16 forward's own (tests/test_fullsize_gpu.py bars unchanged).  bf16, not f16: a workgroup's
// share of a small gradient entry sits far below f16's 6e-8.  The per-object bias-row sums of the step go through the int64
// fixed-point table as before (exact).
typedef unsigned short rec_t;
__device__ __forceinline__ rec_t rec_pack(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }   // v_cvt_pk_bf16_f32: RNE
__device__ __forceinline__ float rec_unpack(rec_t b) { return __builtin_bit_cast(float, (unsigned int)b << 16); }
constexpr double ROWS_FIX_SCALE = 1099511627776.0;  // 2^40: bias-row sums as int64 fixed point (order-free atomics)
constexpr int TAIL_EPB = 64;  // record entries per reducing block of the tail launch (256 / TAIL_EPB sub-ranges each; 32 and 16 measured slower)
constexpr int ROWS_FIX_COPIES = 8;  // the table is replicated: a workgroup adds into copy (its index & 7), which
                                    // cuts the same-address atomic queue 8-fold; consumers add the copies (exact)

// Entries no launch writes (latent-layer biases, padding, unused row sums) are skipped, so the workspace needs no
// clearing.
__device__ __forceinline__ bool rec_entry_written(int i, int rows_per_class) {
  if (i < TRUNK)
    return !((i >= OFF_S1_B && i < OFF_S1_B + 32) || (i >= OFF_CAT_B && i < OFF_CAT_B + 32) ||
             (i >= OFF_S2_B && i < OFF_S2_B + 32) || (i >= OFF_T1_B && i < OFF_T1_B + 32));
  if (i < TRUNK + 126) return true;
  return rows_per_class <= ROWS_MAX && (i - (TRUNK + 126)) < rows_per_class * 128;
}
// sum of entry i over records [w0, w1) of one class (r = that class's first record + i).  32 loads in flight per
// thread: the reducing kernels run a few waves per CU, so the loads in flight per thread are what hides the memory
// latency (8 in flight: 64 records = 8 round trips = 8 us; 32: 2 round trips).  Fixed order -> reproducible bits.
__device__ __forceinline__ float record_range_sum(const rec_t* __restrict__ r, int w0, int w1) {
  constexpr int U = 32;
  float a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = 0.0f;
  int w = w0;
  // (loads into their own registers, a scheduling barrier, then the adds: written as a[u] += r[..] the compiler is free to
  //  serialise load -> wait -> add per record, and without the SLP vectoriser it does: 19 instead of 5 us per reduction)
  for (; w + U - 1 < w1; w += U) {
    rec_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = r[(size_t)(w + u) * REC_ENTRIES];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] += rec_unpack(v[u]);
  }
  {
    rec_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (w + u < w1) ? r[(size_t)(w + u) * REC_ENTRIES] : (rec_t)0;   // the rest (< U records), issued together
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] += rec_unpack(v[u]);
  }
#pragma unroll
  for (int st = U / 2; st >= 1; st >>= 1) {
#pragma unroll
    for (int u = 0; u < st; ++u) a[u] += a[u + st];
  }
  return a[0];
}
}  // namespace cnr_rec
