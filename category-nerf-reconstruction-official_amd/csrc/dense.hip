// Dense layers of the background model: vMAP-style OccupancyMap, src/model.py:86-155 --
// seven torch.nn.Linear (+ ReLU) with K = 87 / 128 / 215 / 170, N = 128 / 1 / 3, called un-vmapped on
// 1200 rays x 14 samples per step (train.py:113-121,172-180).  Exact fp32 like the modular tier: the matrix core does
// fp32 x fp32 products with fp32 accumulation (v_mfma_f32_32x32x2_f32), so the results sit within summation order of
// torch.nn.functional.linear.
//
// One kernel covers the three products of a layer,  C[i][j] = sum_r A(i, r) * B(j, r):
//   forward   y  = act(x W^T + b)   : A = x  (M x K),   B = W  (N x K)              r = K index
//   backward  dx = dpre W           : A = dpre (M x N), B = W^T (r = N index runs over ROWS of W)
//   backward  dW = dpre^T x , db    : A = dpre^T, B = x^T  (r = M index runs over rows of both) ; column K of the
//                                     result is sum_m dpre = db (B(K, r) := 1)
// with dpre = dy * (y > 0) formed on load when a ReLU mask is given.  Workgroup = 4 waves, 64 x 64 tile of C, each
// wave a 32 x 32 quadrant; 32-deep slabs of A and B go through LDS as [i][r] / [j][r] (row stride 33: the per-lane
// operand reads hit 32 different banks) whatever their orientation in memory -- global reads always run along the
// contiguous dimension.
//
// HALF = true (f16_operands != 0 in the C-ABI): the same kernel with both operands rounded to f16 when they are staged in
// LDS and v_mfma_f32_32x32x16_f16 (fp32 accumulation): 2 instead of 16 matrix instructions and 2 x 16-byte instead of
// 32 x 4-byte LDS reads per lane and 32-deep slab.  Gradient operands are multiplied by `a_scale` (a power of two, the loss
// scale of the f16 chain) on the way in and the result by 1 / a_scale on the way out.  This is the f16 tier of the
// background model (the exact-fp32 form stays the parity tier).
#include "cnr_common.h"

namespace {
typedef float f16acc __attribute__((ext_vector_type(16)));
typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
constexpr int TM = 64, TN = 64, TK = 32, LDT = TK + 1, LDH = TK + 8;   // LDH: f16 row stride (80 B: 16-byte aligned rows)
constexpr int EPT = TM * TK / 256;                                     // slab elements per thread and operand

struct DenseArgs {
  const float* A; int64_t lda; int a_trans;   // a_trans: A(i, r) = A[r * lda + i], else A[i * lda + r]
  const float* B; int64_t ldb; int b_trans;
  const float* mask; int mask_on_a;           // same layout as A: element used only where mask > 0
  const float* bias; int relu;                // epilogue (per column j)
  float* C; int64_t ldc;
  float* extra_col; int extra_j;              // column j == extra_j (>= 0) goes to extra_col[i]; B(extra_j, r) = 1
  int I, J, Rn;
  int r_chunk; int64_t c_zstride;             // split over r: block z takes r in [z r_chunk, (z+1) r_chunk) and
  float a_scale;                              // writes its partial product to C + z c_zstride ; HALF: A x a_scale, C / a_scale
};

template <bool HALF>
__global__ __launch_bounds__(256) void dense_kernel(DenseArgs p) {
  __shared__ __attribute__((aligned(16))) float As[TM * LDT];   // HALF: reinterpreted as _Float16 [TM][LDH]
  __shared__ __attribute__((aligned(16))) float Bs[TN * LDT];
  _Float16* Ah = reinterpret_cast<_Float16*>(As);
  _Float16* Bh = reinterpret_cast<_Float16*>(Bs);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
  const int wi = (wv >> 1) * 32, wj = (wv & 1) * 32;
  f16acc acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  const int r_lo = blockIdx.z * p.r_chunk, r_hi = r_lo + p.r_chunk < p.Rn ? r_lo + p.r_chunk : p.Rn;
  float* Cz = p.C + (int64_t)blockIdx.z * p.c_zstride;
  // ---- slab loads: 64 x TK elements each, EPT per thread, consecutive threads along the contiguous dimension (TK = 64 was
  // measured: 1.16 against 1.07 ms per background step -- the slab loop is bound by its per-element loads, not by round trips).  The loads of
  // slab k + 1 are issued before the products of slab k (registers -> LDS after them): a block used to walk load -> barrier
  // -> products -> barrier per slab, one memory round trip per 32 of K with nothing to overlap it
  float an[EPT], bn[EPT];
  auto load_slab = [&](int r0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * 256 + threadIdx.x;
      int ii, rr;
      if (p.a_trans) { ii = idx & 63; rr = idx >> 6; } else { rr = idx & (TK - 1); ii = idx / TK; }
      const int gi = i0 + ii, gr = r0 + rr;
      float v = 0.0f;
      if (gi < p.I && gr < r_hi) {
        const int64_t off = p.a_trans ? (int64_t)gr * p.lda + gi : (int64_t)gi * p.lda + gr;
        v = p.A[off];
        if (p.mask && !(p.mask[off] > 0.0f)) v = 0.0f;
      }
      an[e] = v;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * 256 + threadIdx.x;
      int jj, rr;
      if (p.b_trans) { jj = idx & 63; rr = idx >> 6; } else { rr = idx & (TK - 1); jj = idx / TK; }
      const int gj = j0 + jj, gr = r0 + rr;
      float v = 0.0f;
      if (gr < r_hi) {
        if (gj == p.extra_j) v = 1.0f;
        else if (gj < p.J) v = p.B[p.b_trans ? (int64_t)gr * p.ldb + gj : (int64_t)gj * p.ldb + gr];
      }
      bn[e] = v;
    }
  };
  load_slab(r_lo);
  for (int r0 = r_lo; r0 < r_hi; r0 += TK) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * 256 + threadIdx.x;
      int ii, rr, jj, rb;
      if (p.a_trans) { ii = idx & 63; rr = idx >> 6; } else { rr = idx & (TK - 1); ii = idx / TK; }
      if (p.b_trans) { jj = idx & 63; rb = idx >> 6; } else { rb = idx & (TK - 1); jj = idx / TK; }
      if (HALF) {
        Ah[ii * LDH + rr] = (_Float16)fminf(fmaxf(an[e] * p.a_scale, -60000.0f), 60000.0f);
        Bh[jj * LDH + rb] = (_Float16)bn[e];
      } else {
        As[ii * LDT + rr] = an[e];
        Bs[jj * LDT + rb] = bn[e];
      }
    }
    __syncthreads();
    if (r0 + TK < r_hi) load_slab(r0 + TK);
    // ---- 16 MFMAs of depth 2: lane (lane & 31) = row of its operand, (lane >> 5) = which of the two r ----
    if constexpr (HALF) {  // lane (lane & 31) = row of its operand, (lane >> 5) = which 8 of the k-step's 16 r
#pragma unroll
      for (int ks = 0; ks < TK; ks += 16) {
        const hv8 a = *reinterpret_cast<const hv8*>(Ah + (wi + (lane & 31)) * LDH + ks + 8 * (lane >> 5));
        const hv8 b = *reinterpret_cast<const hv8*>(Bh + (wj + (lane & 31)) * LDH + ks + 8 * (lane >> 5));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < TK; kk += 2) {
        const float a = As[(wi + (lane & 31)) * LDT + kk + (lane >> 5)];
        const float b = Bs[(wj + (lane & 31)) * LDT + kk + (lane >> 5)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: accumulator layout col = lane & 31 (j), row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5) (i) ----
  const int j = j0 + wj + (lane & 31);
  const float bj = (p.bias && j < p.J) ? p.bias[j] : 0.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + wi + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    if (i >= p.I) continue;
    float v = (HALF ? acc[q] * (1.0f / p.a_scale) : acc[q]) + bj;
    if (p.relu) v = fmaxf(v, 0.0f);
    if (j < p.J) Cz[(int64_t)i * p.ldc + j] = v;
    else if (j == p.extra_j) {
      if (p.extra_col) p.extra_col[i] = v;
      else Cz[(int64_t)i * p.ldc + j] = v;   // split mode: the extra column is column J of the partial
    }
  }
}

// fixed-order sum of the split partials: out[i][j] = sum_z ws[z][i][j] (j < J), extra[i] = sum_z ws[z][i][J]
__global__ __launch_bounds__(256) void dense_reduce_kernel(const float* __restrict__ ws, int nz, int I, int J,
                                                           int has_extra, float* __restrict__ out,
                                                           float* __restrict__ extra) {
  const int ld = J + (has_extra ? 1 : 0);
  const int64_t n = (int64_t)I * ld;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float s = 0.0f;   // partials in order, eight loads in flight (one dependent load per partial was 13 of this launch's 15 us)
    int z = 0;
    for (; z + 8 <= nz; z += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ws[(int64_t)(z + u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; z < nz; ++z) s += ws[(int64_t)z * n + e];
    const int i = (int)(e / ld), j = (int)(e % ld);
    if (j < J) out[(int64_t)i * J + j] = s;
    else if (extra) extra[i] = s;
  }
}

// dW (N x K, + column K = db) = (dy masked by y > 0)^T x over M samples, the contraction that matters for the background step
// (M = 16 800, the result tiny): no LDS tiling at all.  The operand layout of v_mfma_f32_32x32x2_f32 IS a coalesced global
// read here -- A[i = n][k = m]: lane (n, m & 1) reads dy[m][n0 + n]; B[k = m][j]: lane (j, m & 1) reads x[m][k0 + j] -- 128
// contiguous bytes per half wave and instruction, eight sample pairs in flight per wave.  A workgroup = one 32 x 32 tile of
// the result and one chunk of samples; its four waves split the chunk and add their accumulators through LDS, so one
// partial per workgroup goes to the workspace (dense_reduce_kernel adds the partials in order).  46 -> ~15 us per layer
// against the LDS-tiled form (whose 64-chunk split ran nine load -> barrier -> scatter -> barrier rounds per workgroup).
__global__ __launch_bounds__(256) void dense_dw_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       const float* __restrict__ x, float* __restrict__ ws, int M, int K,
                                                       int N, int has_db, int chunk, int gx, int gy, int nz) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  // XCD-aware mapping of the 1-D grid: blocks b and b + 8 share an XCD (its L2), so ALL gx * gy result tiles of a sample
  // chunk -- which read the same rows of dy, y and x -- are dealt to one XCD: chunk z = 8 (s / T) + (b & 7), tile s % T of the
  // s = b >> 3-th block of that XCD.  Dealt tile-fastest instead, the eight L2s each fetched every chunk (25.7 us per launch).
  const int T = gx * gy, sx = blockIdx.x >> 3;
  const int z = 8 * (sx / T) + (blockIdx.x & 7), tt = sx % T;
  if (z >= nz) return;
  const int k0 = (tt % gx) * 32, n0 = (tt / gx) * 32;
  const int ld = K + (has_db ? 1 : 0);
  const int n = n0 + l31, kc = k0 + l31;
  const bool n_ok = n < N, k_in = kc < K, k_one = has_db && kc == K;
  const int m_lo = z * chunk, m_hi = min(M, m_lo + chunk);
  const int per = ((m_hi - m_lo + 3) / 4 + 1) / 2 * 2;              // samples per wave (a quarter of the chunk), even
  const int w_lo = m_lo + wv * per, w_hi = min(m_hi, w_lo + per);
  f16acc acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  constexpr int U = 8;
  // two register sets: the loads of the next 16 samples are in flight while the MFMAs of the current 16 run (a wave walks only
  // ~9 such blocks; without the overlap every block waited out a full memory round trip: 31 us per launch instead of ~12)
  float a0[U], b0[U], k0m[U], a1[U], b1[U], k1m[U];
  auto fetch = [&](float (&a)[U], float (&b)[U], float (&mk)[U], int m0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + 2 * u + half;
      const bool ok = m < w_hi;
      a[u] = (ok && n_ok) ? dy[(int64_t)m * N + n] : 0.0f;
      mk[u] = (y && ok && n_ok) ? y[(int64_t)m * N + n] : 1.0f;
      b[u] = ok ? (k_in ? x[(int64_t)m * K + kc] : (k_one ? 1.0f : 0.0f)) : 0.0f;
    }
  };
  auto mma = [&](const float (&a)[U], const float (&b)[U], const float (&mk)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mk[u] > 0.0f ? a[u] : 0.0f, b[u], acc, 0, 0, 0);
  };
  fetch(a0, b0, k0m, w_lo);
  for (int m0 = w_lo; m0 < w_hi; m0 += 4 * U) {
    fetch(a1, b1, k1m, m0 + 2 * U);
    mma(a0, b0, k0m);
    fetch(a0, b0, k0m, m0 + 4 * U);
    mma(a1, b1, k1m);
  }
  __shared__ float red[3][16][64];
  if (wv > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wv - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (wv == 0) {
    float* out = ws + (int64_t)z * N * ld;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float v = ((acc[q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
      const int i = n0 + (q & 3) + 8 * (q >> 2) + 4 * half;
      if (i < N && kc < ld) out[(int64_t)i * ld + kc] = v;
    }
  }
}

int launch(const DenseArgs& p, void* stream, int nz = 1, bool half = false) {
  const int jn = p.extra_j >= 0 ? p.J + 1 : p.J;
  dim3 grid((jn + TN - 1) / TN, (p.I + TM - 1) / TM, nz);
  if (half) hipLaunchKernelGGL(dense_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(dense_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
}  // namespace

extern "C" int cnr_dense_fwd(const float* x, const float* W, const float* b, float* y, int M, int K, int N, int relu,
                             int f16_operands, void* stream) {
  if (!x || !W || !y || M <= 0 || K <= 0 || N <= 0) return CNR_E_ARG;
  DenseArgs p{x, K, 0, W, K, 0, nullptr, 0, b, relu, y, N, nullptr, -1, M, N, K, K, 0, 1.0f};
  return launch(p, stream, 1, f16_operands != 0);
}

// number of sample chunks of the dW product: enough blocks to fill the chip, at least 256 samples each
static int dense_split(int M) {
  int nz = (M + 255) / 256;
  if (nz > 64) nz = 64;
  return nz < 1 ? 1 : nz;
}

extern "C" int64_t cnr_dense_bwd_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const int nz = dense_split(M);
  return nz == 1 ? 0 : (int64_t)nz * N * (K + 1) * (int64_t)sizeof(float);
}

extern "C" int cnr_dense_bwd(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW,
                             float* db, int M, int K, int N, int relu, void* workspace, int64_t workspace_bytes,
                             int f16_operands, float grad_scale, void* stream) {
  if (!x || !W || !dy || !dW || M <= 0 || K <= 0 || N <= 0 || (relu && !y)) return CNR_E_ARG;
  const bool half = f16_operands != 0;
  const float gs = half ? (grad_scale > 0.0f ? grad_scale : 1.0f) : 1.0f;
  const float* mask = relu ? y : nullptr;
  if (dx) {  // dx (M x K) = dpre (M x N) W (N x K): r = N, B(j = k, r = n) = W[n * K + k]
    DenseArgs p{dy, N, 0, W, K, 1, mask, 1, nullptr, 0, dx, K, nullptr, -1, M, K, N, N, 0, gs};
    const int rc = launch(p, stream, 1, half);
    if (rc) return rc;
  }
  // dW (N x K) = dpre^T x: r = M, A(i = n, r = m) = dy[m * N + n], B(j = k, r = m) = x[m * K + k]; column K = db.
  // The contraction runs over all M samples while the result is tiny: split it over the grid's z (each block a
  // chunk of samples, partial products to the workspace) and add the partials in a fixed order.
  const int nz = dense_split(M);
  if (nz == 1) {
    DenseArgs p{dy, N, 1, x, K, 1, mask, 1, nullptr, 0, dW, K, db, db ? K : -1, N, K, M, M, 0, gs};
    return launch(p, stream, 1, half);
  }
  const int ld = K + (db ? 1 : 0);
  if (!workspace || workspace_bytes < (int64_t)nz * N * ld * (int64_t)sizeof(float)) return CNR_E_ARG;
  const int chunk = ((M + nz - 1) / nz + TK - 1) / TK * TK;
  const int nzl = (M + chunk - 1) / chunk;
  {   // both tiers: fp32 operands straight from global memory (dense_dw_kernel; exact fp32 products also for the f16 tier)
    const int gx = (ld + 31) / 32, gy = (N + 31) / 32;
    dim3 grid((unsigned)(8 * ((nzl + 7) / 8) * gx * gy));
    hipLaunchKernelGGL(dense_dw_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, mask, x, (float*)workspace, M, K, N,
                       db ? 1 : 0, chunk, gx, gy, nzl);
    CNR_LAUNCH_CHECK();
  }
  const int64_t n = (int64_t)N * ld;
  hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)workspace, nzl, N, K, db ? 1 : 0, dW, db);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
