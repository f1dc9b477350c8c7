// The tail of the fused trainer's step in ONE launch: latent backward, AdamW on every parameter, and the step
// epilogue -- three jobs that the stock sequence (cnr_latent_bwd -> cnr_adamw_step -> cnr_step_epilogue) runs as
// dependent launches.  What makes them independent here:
//   * parameters ping-pong like the step state: every kernel of step k READS theta_in and this launch WRITES
//     theta_out (AdamW out of place; exp_avg / exp_avg_sq are touched by AdamW only).  The latent backward reads
//     trunk weights, latent weights and codes while other blocks update exactly those -- out of place, no race;
//   * the thread that produces a latent-path gradient element applies AdamW to it straight away (one producer per
//     element); the trunk entries the latent path adds to are finished by the AdamW blocks themselves
//     (latent_trunk_term: n_obj multiply-adds from dbiasrows and zl);
//   * the epilogue blocks only read the render-loss partials, the depth pool and the CURRENT state.
//   * (records != NULL) the fixed-order reduction of the field backward's records moves in as well: a block sums 64
//     record entries over all workgroups and applies AdamW to them at once; the latent path takes its input, the
//     per-object bias-row sums, from a fixed-point table the field backward filled with integer atomics (order-free,
//     hence still bitwise reproducible) instead of waiting for that reduction.
// grid: [ do_latent ? NL latent blocks per class : 0 ] [ NA AdamW blocks over the flat (C, P) buffer, or with records
//        NR = REC_ENTRIES / TAIL_EPB reduce-and-update blocks per class ] [ C epilogue ].
// With do_latent = 0 (the gradient was completed by cnr_latent_bwd, e.g. before a multi-GPU all-reduce) the AdamW
// blocks cover every parameter and nothing else changes.
#include "adamw_common.h"
#include "latent_common.h"
#include "records_common.h"
#include "render_common.h"

namespace {
using namespace cnr;

struct TailArgs {
  const float* theta_in; float* theta_out; float* grad; float* m; float* v;
  FlatLayout lay; int64_t off_B; int C;
  const float* zl; float* dbiasrows; float reg_scale;
  float lr, b1, b2, eps, wd;
  float code_lr, code_wd;   // the code tables' own AdamW group (train.py:40,54-64: code_lr / code_weight_decay); = lr, wd when not given
  const int64_t* state_cur; int64_t* state_next; int64_t add_rows;
  const float* partials; int nb; float* losses; int32_t* flags;
  const float* depth; int64_t pool_rows; const int* perm; int R; float* max_bound;
  int do_latent, NL, NA;
  int local_latent;   // latent blocks in the per-block form of latent_bwd_block_local (more than four objects, rows from the table)
  // optional: the field backward's per-workgroup records (nwg per class) are reduced HERE, and the per-object
  // bias-row sums come from the fixed-point table the field backward accumulated with integer atomics
  const cnr_rec::rec_t* records; int nwg; const long long* rows_fix; int NR;
  // grad_only: stop at the finished gradient (no AdamW, no epilogue blocks): the multi-GPU step all-reduces it first
  int grad_only;
  const int* n_obj_cls;  // optional (C,): objects each class really has (<= lay.n_obj; the rest of its rows are padding)
  int* clamp_flags;  // optional (C,): bits the field backward raised this step (cnr_field_bwd_pipe); or-ed into flags, cleared
};

// (is_code: the element belongs to a shape / texture code table of its class row)
__device__ __forceinline__ bool is_code_off(const cnr::FlatLayout& lay, int64_t q) {
  const int64_t n = (int64_t)lay.n_obj * lay.L;
  return (q >= lay.shape && q < lay.shape + n) || (q >= lay.tex && q < lay.tex + n);
}
__device__ __forceinline__ void adam_one(const TailArgs& a, int64_t e, float g, float step_size, float inv_bc2_sqrt,
                                         bool is_code) {
  float pi = a.theta_in[e] * (1.0f - (is_code ? a.code_lr * a.code_wd : a.lr * a.wd));
  if (is_code) step_size *= a.code_lr / a.lr;
  const float mi = a.m[e] + (g - a.m[e]) * (1.0f - a.b1);
  const float vi = a.v[e] * a.b2 + (1.0f - a.b2) * g * g;
  const float denom = sqrtf(vi) * inv_bc2_sqrt + a.eps;
  pi -= step_size * (mi / denom);
  a.theta_out[e] = pi; a.m[e] = mi; a.v[e] = vi;
}

struct AdamSink {  // latent-path gradient element -> gradient buffer (kept for inspection) + AdamW, same thread
  const TailArgs& a; int64_t row0; float step_size, inv_bc2_sqrt;
  mutable float p0, m0, v0;  // parameter and moments of the element in work, loaded while its gradient is computed
  __device__ __forceinline__ void trunk_add(int, float) const {}
  __device__ __forceinline__ void prefetch(int64_t idx) const {
    if (a.grad_only) return;
    p0 = a.theta_in[row0 + idx]; m0 = a.m[row0 + idx]; v0 = a.v[row0 + idx];
  }
  __device__ __forceinline__ void latent_set(int64_t idx, float g) const {
    const int64_t e = row0 + idx;
    a.grad[e] = g;
    if (a.grad_only) return;
    const bool is_code = is_code_off(a.lay, idx);
    float pi = p0 * (1.0f - (is_code ? a.code_lr * a.code_wd : a.lr * a.wd));
    const float mi = m0 + (g - m0) * (1.0f - a.b1);
    const float vi = v0 * a.b2 + (1.0f - a.b2) * g * g;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + a.eps;
    pi -= (is_code ? step_size * (a.code_lr / a.lr) : step_size) * (mi / denom);
    a.theta_out[e] = pi; a.m[e] = mi; a.v[e] = vi;
  }
};

// this class's (n_obj,4,32) bias-row gradient as floats in LDS, from the fixed-point table
__device__ __forceinline__ void load_rows_fix(const TailArgs& a, int c, float* dst) {
  const int n = a.lay.n_obj * 128;
  // four entries per thread at a time, all 32 table loads in flight together (one entry at a time was one memory round trip
  // per 256 entries: four trips for a class of seven objects)
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
    long long f[4][cnr_rec::ROWS_FIX_COPIES];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
#pragma unroll
      for (int k = 0; k < cnr_rec::ROWS_FIX_COPIES; ++k) f[u][k] = i < n ? a.rows_fix[((size_t)k * a.C + c) * n + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      long long t = 0;
#pragma unroll
      for (int k = 0; k < cnr_rec::ROWS_FIX_COPIES; ++k) t += f[u][k];
      if (i < n) dst[i] = (float)((double)t * (1.0 / cnr_rec::ROWS_FIX_SCALE));
    }
  }
  __syncthreads();
}

// ... and straight to the float rows in global memory (no staging: the row count is not bounded by the block's LDS)
__device__ __forceinline__ void publish_rows_fix(const TailArgs& a, int c) {
  const int n = a.lay.n_obj * 128;
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
    long long f[4][cnr_rec::ROWS_FIX_COPIES];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
#pragma unroll
      for (int k = 0; k < cnr_rec::ROWS_FIX_COPIES; ++k) f[u][k] = i < n ? a.rows_fix[((size_t)k * a.C + c) * n + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      long long t = 0;
#pragma unroll
      for (int k = 0; k < cnr_rec::ROWS_FIX_COPIES; ++k) t += f[u][k];
      if (i < n) a.dbiasrows[(int64_t)c * n + i] = (float)((double)t * (1.0 / cnr_rec::ROWS_FIX_SCALE));
    }
  }
}

#ifdef CNR_TAIL_STAMPS  // tools/exp only
__device__ unsigned long long g_tail_t[8];  // [type 0..2][min start, max end], [6] = min start over all
#define TAIL_T0() const unsigned long long tt0 = __builtin_amdgcn_s_memrealtime(); \
  if (threadIdx.x == 0) atomicMin(&g_tail_t[6], tt0)
#define TAIL_T1(type) do { __syncthreads(); if (threadIdx.x == 0) { atomicMin(&g_tail_t[2 * (type)], tt0); \
  atomicMax(&g_tail_t[2 * (type) + 1], __builtin_amdgcn_s_memrealtime()); } } while (0)
extern "C" int cnr_tail_stamps(unsigned long long* host, int reset) {
  if (reset) {
    unsigned long long init[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tail_t), init, sizeof(init));
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tail_t), sizeof(unsigned long long) * 8);
}
#else
#define TAIL_T0() do {} while (0)
#define TAIL_T1(type) do {} while (0)
#endif

__global__ __launch_bounds__(256) void tail_kernel(TailArgs a) {
  extern __shared__ float sm[];
  TAIL_T0();
  const int C = a.C, P = (int)a.lay.stride;
  const int nlat = a.do_latent ? a.NL * C : 0;
  int b = blockIdx.x;
  float step_size = 0.0f, inv_bc2_sqrt = 0.0f;
  if (!a.grad_only) {
    AdamArgs co{nullptr, nullptr, nullptr, nullptr, 0, a.lr, a.b1, a.b2, a.eps, a.wd, 1.0f};
    adam_coefficients(co, a.state_cur[2] + 1, step_size, inv_bc2_sqrt);
  }
  if (b < nlat) {  // ---- latent backward of class c + AdamW on what it produces
    const int c = b / a.NL, blk = b % a.NL;
    const int n_real = a.n_obj_cls ? a.n_obj_cls[c] : a.lay.n_obj;
    const float reg_c = n_real > 1 ? a.reg_scale : 0.0f;        // src/loss.py:5-15: classes with one object: no regulariser
    AdamSink sink{a, (int64_t)c * P, step_size, inv_bc2_sqrt};
    const float* dbr = a.dbiasrows + (int64_t)c * a.lay.n_obj * 128;
    float* scratch = sm;
    const int64_t nlat_out = (int64_t)4 * 32 * a.lay.L + 128 + (int64_t)2 * a.lay.n_obj * a.lay.L;
    if (a.rows_fix && a.lay.n_obj <= 4 && a.lay.L <= 256 && (int64_t)a.NL * 256 >= nlat_out) {
      // the common case: one element per thread, every load of the block in one round trip
      const int n = a.lay.n_obj * 128;
      latent_bwd_block_1trip(a.theta_in + (int64_t)c * P, a.lay, a.zl + (int64_t)c * n, a.rows_fix + (size_t)c * n,
                             (int64_t)a.C * n, cnr_rec::ROWS_FIX_COPIES, 1.0 / cnr_rec::ROWS_FIX_SCALE,
                             blk == 0 ? a.dbiasrows + (int64_t)c * n : nullptr, reg_c, sm, sink, blk, n_real);
      TAIL_T1(0);
      return;
    }
    if (a.rows_fix && a.local_latent) {
      // more than four objects: every block builds only the d pre entries its own elements read (latent_common.h); the float
      // rows are published by the class's epilogue block
      const int n = a.lay.n_obj * 128;
      latent_bwd_block_local(a.theta_in + (int64_t)c * P, a.lay, a.zl + (int64_t)c * n, a.rows_fix + (size_t)c * n,
                             (int64_t)a.C * n, cnr_rec::ROWS_FIX_COPIES, 1.0 / cnr_rec::ROWS_FIX_SCALE, reg_c, sm, sink, blk,
                             n_real);
      if (a.grad_only && blk == 0 && a.dbiasrows) publish_rows_fix(a, c);   // (no epilogue blocks in the gradient-only launch)
      TAIL_T1(0);
      return;
    }
    if (a.rows_fix) {  // rows from the fixed-point table (block 0 of the class also publishes them as floats)
      float* rows = sm;
      load_rows_fix(a, c, rows);
      if (blk == 0)
        for (int i = threadIdx.x; i < a.lay.n_obj * 128; i += 256) a.dbiasrows[(int64_t)c * a.lay.n_obj * 128 + i] = rows[i];
      dbr = rows;
      scratch = sm + a.lay.n_obj * 128;
    }
    latent_bwd_block(a.theta_in + (int64_t)c * P, a.lay, a.zl + (int64_t)c * a.lay.n_obj * 128, dbr, reg_c,
                     scratch, sink, blk, a.NL, false, n_real);
    TAIL_T1(0);
    return;
  }
  b -= nlat;
  if (a.records) {  // ---- reduce 64 record entries of class c over all workgroups, finish the gradient, AdamW
    if (b < a.NR * C) {
      const int c = b / a.NR, blk = b % a.NR;
      // EPB record entries per block, NQ = 256 / EPB sub-ranges of the workgroup range per entry: every thread sums
      // nwg / NQ records.  Measured at configs[1] (256 records): EPB = 64 (two rounds of 32 loads per thread, 256-byte
      // segments per wave load) 30.75 M rays/s; EPB = 32 29.9 M; EPB = 16 (one round, 64-byte segments, 4 x the blocks) 24.2 M
      constexpr int EPB = cnr_rec::TAIL_EPB, NQ = 256 / EPB;
      float* part = sm;                      // [NQ][EPB] (+ [NQ][EPB] for the second dB addend, + [NQ][EPB] for the latent-path term)
      const int e = threadIdx.x % EPB, q = threadIdx.x / EPB;
      const int i = blk * EPB + e;
      // The latent path's share of a latent-conditioned layer's weight / bias gradient, sum over objects of d biasrow[ob][k][o] *
      // (z[ob][k][j] | 1): the NQ threads of an entry split the objects (ob = q, q + NQ, ..) and read their table entries straight
      // into registers -- no walk over the whole table, no barrier, nothing that grows with the object count beyond n_obj / NQ
      // entries per thread; the loads go out together with the record loads below.
      float lat_part = 0.0f;
      {
        int lk, lo_, lj;
        if (i < TRUNK && latent_trunk_index(i, lk, lo_, lj)) {
          const int n_obj = a.lay.n_obj, n = n_obj * 128;
          const long long* tabc = a.rows_fix + (size_t)c * n;
          const float* zc = a.zl + (int64_t)c * n;
          for (int ob0 = q; ob0 < n_obj; ob0 += 4 * NQ) {       // four objects (32 table loads) in flight at a time
            long long f[4][cnr_rec::ROWS_FIX_COPIES];
            float zz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int ob = ob0 + u * NQ, on = ob < n_obj, idx = ((on ? ob : 0) * 4 + lk) * 32;
#pragma unroll
              for (int k = 0; k < cnr_rec::ROWS_FIX_COPIES; ++k) f[u][k] = on ? tabc[(size_t)k * a.C * n + idx + lo_] : 0;
              zz[u] = on ? (lj < 32 ? zc[idx + lj] : 1.0f) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              long long t = 0;
#pragma unroll
              for (int k = 0; k < cnr_rec::ROWS_FIX_COPIES; ++k) t += f[u][k];
              lat_part = fmaf((float)((double)t * (1.0 / cnr_rec::ROWS_FIX_SCALE)), zz[u], lat_part);
            }
          }
        }
      }
      // trunk entries and the first dB half own an output; the second dB half is the other addend of the first
      const bool owner = i < TRUNK + 63;
      float s0 = 0.0f, s1 = 0.0f;
      const int64_t idx = (int64_t)c * P + (i < TRUNK ? i : a.off_B + (i - TRUNK));
      float p0 = 0.f, m0 = 0.f, v0 = 0.f;
      if (q == 0 && owner && !a.grad_only) { p0 = a.theta_in[idx]; m0 = a.m[idx]; v0 = a.v[idx]; }   // in flight under the record sum
      if (owner) {
        const int per = (a.nwg + NQ - 1) / NQ, w0 = q * per, w1 = min(a.nwg, w0 + per);
        const cnr_rec::rec_t* r = a.records + (size_t)c * a.nwg * cnr_rec::REC_ENTRIES + i;
        if (cnr_rec::rec_entry_written(i, a.lay.n_obj)) s0 = cnr_rec::record_range_sum(r, w0, w1);
        if (i >= TRUNK) s1 = cnr_rec::record_range_sum(r + 63, w0, w1);
      }
      part[q * EPB + e] = s0; part[256 + q * EPB + e] = s1; part[512 + q * EPB + e] = lat_part;
      __syncthreads();
      if (q == 0 && owner) {
        auto tree = [&](const float* pp) {       // fixed-order pairwise sum of the NQ partials of entry e
          float t[NQ];
#pragma unroll
          for (int k = 0; k < NQ; ++k) t[k] = pp[k * EPB + e];
#pragma unroll
          for (int st = NQ / 2; st >= 1; st >>= 1) {
#pragma unroll
            for (int k = 0; k < st; ++k) t[k] += t[k + st];
          }
          return t[0];
        };
        float g = tree(part);
        if (i >= TRUNK) g += tree(part + 256);
        if (i < TRUNK) g += tree(part + 512);
        a.grad[idx] = g;
        if (!a.grad_only) {
          float pi = p0 * (1.0f - a.lr * a.wd);
          const float mi = m0 + (g - m0) * (1.0f - a.b1);
          const float vi = v0 * a.b2 + (1.0f - a.b2) * g * g;
          const float denom = sqrtf(vi) * inv_bc2_sqrt + a.eps;
          pi -= step_size * (mi / denom);
          a.theta_out[idx] = pi; a.m[idx] = mi; a.v[idx] = vi;
        }
      }
      TAIL_T1(1);
      return;
    }
    b -= a.NR * C;
  } else if (b < a.NA) {  // ---- AdamW over the flat buffer; with do_latent only the trunk and B ranges (+ the latent-path term)
    const int64_t n = (int64_t)C * P;
    for (int64_t e = (int64_t)b * 256 + threadIdx.x; e < n; e += (int64_t)a.NA * 256) {
      const int c = (int)(e / P), q = (int)(e - (int64_t)c * P);
      float g = a.grad[e];
      if (a.do_latent) {
        const bool is_trunk = q < CNR_TRUNK_PARAMS, is_B = q >= a.off_B && q < a.off_B + 63;
        if (!is_trunk && !is_B) continue;  // a latent block owns this element
        if (is_trunk) {
          const float t = latent_trunk_term(q, a.zl + (int64_t)c * a.lay.n_obj * 128,
                                            a.dbiasrows + (int64_t)c * a.lay.n_obj * 128, a.lay.n_obj);
          if (t != 0.0f) { g += t; a.grad[e] = g; }
        }
      }
      adam_one(a, e, g, step_size, inv_bc2_sqrt, is_code_off(a.lay, q));
    }
    return;
  } else {
    b -= a.NA;
  }
  // ---- epilogue of class b: loss values + flags, next slice's max depth, next step state (class 0)
  const int c = b, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* red = sm;
  if (a.local_latent && a.rows_fix && a.dbiasrows)     // the class's bias-row gradient as floats (the latent blocks no longer walk the
    publish_rows_fix(a, c);                            // whole table): table -> registers -> global, any number of objects
  const int64_t cursor = a.state_cur[0] + a.add_rows;
  if (wv == 0) {
    cnr_rl::finish_class(a.partials, a.nb, a.losses, a.flags, C, c, lane);
    if (a.clamp_flags && lane == 0) { a.flags[c] |= a.clamp_flags[c]; a.clamp_flags[c] = 0; }   // same lane wrote flags[c]
  }
  if (a.max_bound) {
    float mx = -INFINITY;
    const int64_t cbase = (int64_t)c * a.pool_rows, base = cbase + cursor;
    // this block is the launch's critical path (cursor -> permutation entry -> depth are dependent loads): eight rays
    // per thread at a time, all permutation entries together, then all depths together = two round trips per 2048 rays
    for (int r0 = 0; r0 < a.R; r0 += 8 * 256) {
      int64_t at[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u * 256 + threadIdx.x;
        at[u] = r < a.R ? (a.perm ? cbase + a.perm[base + r] : base + r) : -1;
      }
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = at[u] >= 0 ? a.depth[at[u]] : -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) mx = fmaxf(mx, d[u]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    if (threadIdx.x == 0) a.max_bound[c] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  if (c == 0 && threadIdx.x == 0) {
    a.state_next[0] = cursor; a.state_next[1] = a.state_cur[1] + 1; a.state_next[2] = a.state_cur[2] + 1;
  }
  TAIL_T1(2);
}
}  // namespace

constexpr int LOCAL_MIN_OBJ = 5;   // classes of up to four objects take the one-trip latent blocks (latent_bwd_block_1trip)
// dynamic LDS of a tail block: the reducing blocks' three partial arrays (768 floats), the general latent blocks' two whole
// (n_obj, 4, 32) tables + norms, or the per-block form's own rows (latent_local_lds_floats) -- which is what lets a class hold up to
// ROWS_TILE_MAX objects: 100 objects need 29 KB there, where the whole tables would need 103 KB and one block per CU
static size_t tail_lds_bytes(int do_latent, int local_latent, int L, int n_obj) {
  const int general = 2 * n_obj * 128 + 2 * n_obj + 520;
  const int local = latent_local_lds_floats(L, n_obj) + 8;
  const int need = !do_latent ? 776 : local_latent ? (local > 776 ? local : 776) : general;
  return (size_t)need * sizeof(float);
}

extern "C" int cnr_step_tail(const cnr_step_tail_args* args, void* stream) {
  if (!args || args->struct_size != sizeof(cnr_step_tail_args) || args->abi_version != CNR_ABI_VERSION) return CNR_E_ARG;
  const float* theta_in = args->theta_in; float* theta_out = args->theta_out; float* grad = args->grad;
  float* exp_avg = args->exp_avg; float* exp_avg_sq = args->exp_avg_sq;
  const int64_t class_stride = args->class_stride, off_B = args->off_B, off_latW = args->off_latW, off_latb = args->off_latb,
                off_shape = args->off_shape, off_tex = args->off_tex, add_rows = args->add_rows, pool_rows = args->pool_rows;
  const int L = args->L, n_obj = args->n_obj, C = args->C, do_latent = args->do_latent, R = args->R, nwg = args->nwg,
            rl_blocks = args->rl_blocks;
  const float* zl = args->zl; float* dbiasrows = args->dbiasrows;
  const float reg_scale = args->reg_scale, lr = args->lr, beta1 = args->beta1, beta2 = args->beta2, eps = args->eps,
              weight_decay = args->weight_decay, code_lr = args->code_lr, code_weight_decay = args->code_weight_decay;
  const int64_t* state_cur = args->state_cur; int64_t* state_next = args->state_next;
  const void* rl_workspace = args->rl_workspace; float* losses = args->losses; int32_t* flags = args->flags;
  const float* depth = args->depth; const int* perm = args->perm; float* next_max_bound = args->next_max_bound;
  const void* records = args->records; const long long* rows_fix = args->rows_fix; int* clamp_flags = args->clamp_flags;
  const int* n_obj_cls = args->n_obj_cls;
  if (!theta_in || !theta_out || theta_in == theta_out || !grad || !exp_avg || !exp_avg_sq || class_stride <= 0 ||
      L <= 0 || n_obj <= 0 || C <= 0 || !state_cur || !state_next || state_cur == state_next || !rl_workspace ||
      !losses || !flags || R <= 0)
    return CNR_E_ARG;
  if (do_latent && (!zl || !dbiasrows)) return CNR_E_ARG;
  // (more than ROWS_MAX object rows: the records carry no row sums, the fixed-point table has them all -- cnr_field_train's
  //  one-object-per-tile form)
  if (records && (!do_latent || !rows_fix || nwg <= 0 || n_obj > cnr_rec::ROWS_TILE_MAX)) return CNR_E_ARG;
  if (rows_fix && !do_latent) return CNR_E_ARG;
  if (n_obj > cnr_rec::ROWS_TILE_MAX) return CNR_E_SHAPE;
  if (n_obj > 32 && do_latent && !(rows_fix && latent_local_ok(L, n_obj))) return CNR_E_SHAPE;   // the general latent blocks keep whole tables in LDS
  if (next_max_bound && (!depth || pool_rows < R)) return CNR_E_ARG;
  TailArgs a{};
  a.theta_in = theta_in; a.theta_out = theta_out; a.grad = grad; a.m = exp_avg; a.v = exp_avg_sq;
  a.lay = FlatLayout{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  a.off_B = off_B; a.C = C; a.zl = zl; a.dbiasrows = dbiasrows; a.reg_scale = reg_scale; a.n_obj_cls = n_obj_cls;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.code_lr = code_lr > 0.0f ? code_lr : lr; a.code_wd = code_lr > 0.0f ? code_weight_decay : weight_decay;
  if (!(lr > 0.0f)) return CNR_E_ARG;
  a.state_cur = state_cur; a.state_next = state_next; a.add_rows = add_rows;
  const int rpb = cnr_rl::rl_rays_per_block(C, R);
  a.partials = (const float*)rl_workspace; a.nb = rl_blocks > 0 ? rl_blocks : (R + rpb - 1) / rpb;
  a.losses = losses; a.flags = flags;
  a.depth = depth; a.pool_rows = pool_rows; a.perm = perm; a.R = R; a.max_bound = next_max_bound;
  a.do_latent = do_latent ? 1 : 0;
  const int64_t nlat_out = (int64_t)4 * 32 * L + 128 + (int64_t)2 * n_obj * L;
  a.NL = (int)((nlat_out + 255) / 256);
  if (a.NL > 256) a.NL = 256;
  a.local_latent = (do_latent && rows_fix && n_obj >= LOCAL_MIN_OBJ && latent_local_ok(L, n_obj)) ? 1 : 0;
  if (a.local_latent) a.NL = latent_local_blocks(L, n_obj);
  const int64_t n = (int64_t)C * class_stride;
  int64_t na = (n + 255) / 256;
  if (na > 2048) na = 2048;
  a.NA = (int)na;
  a.records = (const cnr_rec::rec_t*)records; a.nwg = nwg; a.rows_fix = rows_fix; a.NR = cnr_rec::REC_ENTRIES / cnr_rec::TAIL_EPB;
  a.clamp_flags = clamp_flags;
  const unsigned grid = (unsigned)((a.do_latent ? a.NL * C : 0) + (records ? a.NR * C : a.NA) + C);
  const size_t lds = tail_lds_bytes(a.do_latent, a.local_latent, L, n_obj);
  hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

// The gradient half of cnr_step_tail on its own, for a host that needs the finished gradient before the optimiser
// (data parallel: all-reduce in between): fixed-order reduction of the field backward's records + latent backward +
// code regulariser, side by side in one launch, into `grad` (class_stride floats per class, the layout of theta).
// Afterwards: all-reduce, then cnr_step_tail(..., do_latent = 0, records = NULL, ...) for AdamW + epilogue.
extern "C" int cnr_step_grad(const float* theta, float* grad, int64_t class_stride, int64_t off_B, int64_t off_latW,
                             int64_t off_latb, int64_t off_shape, int64_t off_tex, int L, int n_obj, int C,
                             const float* zl, float* dbiasrows, float reg_scale, const void* records, int nwg,
                             const long long* rows_fix, const int* n_obj_cls, void* stream) {
  if (!theta || !grad || class_stride <= 0 || L <= 0 || n_obj <= 0 || C <= 0 || !zl || !dbiasrows || !records ||
      nwg <= 0 || !rows_fix)
    return CNR_E_ARG;
  if (n_obj > cnr_rec::ROWS_TILE_MAX) return CNR_E_SHAPE;
  TailArgs a{};
  a.theta_in = theta; a.grad = grad;
  a.lay = FlatLayout{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  a.off_B = off_B; a.C = C; a.zl = zl; a.dbiasrows = dbiasrows; a.reg_scale = reg_scale; a.n_obj_cls = n_obj_cls;
  a.do_latent = 1; a.grad_only = 1;
  const int64_t nlat_out = (int64_t)4 * 32 * L + 128 + (int64_t)2 * n_obj * L;
  a.NL = (int)((nlat_out + 255) / 256);
  if (a.NL > 256) a.NL = 256;
  a.local_latent = (n_obj >= LOCAL_MIN_OBJ && latent_local_ok(L, n_obj)) ? 1 : 0;
  if (a.local_latent) a.NL = latent_local_blocks(L, n_obj);
  a.records = (const cnr_rec::rec_t*)records; a.nwg = nwg; a.rows_fix = rows_fix; a.NR = cnr_rec::REC_ENTRIES / cnr_rec::TAIL_EPB;
  const unsigned grid = (unsigned)(a.NL * C + a.NR * C);
  if (n_obj > 32 && !a.local_latent) return CNR_E_SHAPE;
  const size_t lds = tail_lds_bytes(1, a.local_latent, L, n_obj);
  hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
