// C-ABI entry points of the 8-wave pipelined kernel (fused_bwd_pipe8_kernel.h) and the stand-alone backward's instantiations.
#include "fused_bwd_pipe8_kernel.h"

// the one-launch instantiations of one WIDE each (fused_bwd_pipe8_w0 .. w3.hip); internal to the library
#define CNR_HIDDEN __attribute__((visibility("hidden")))
CNR_HIDDEN int cnr_ft_wide0(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);
CNR_HIDDEN int cnr_ft_wide1(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);
CNR_HIDDEN int cnr_ft_wide2(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);
CNR_HIDDEN int cnr_ft_wide3(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);

static int cnr_field_bwd_pipe8_launch(const float* pts, const float* B, const void* packed, const float* biasrows,
                                          const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                                          float grad_scale, int C, int R, int S, int rows_per_class, int blocks,
                                          void* workspace, int64_t B_stride, long long* rows_fix, int* clamp_flags,
                                          void* stream) {
  const TrainArgs none{};
  if (rows_per_class > 7)
    return launch_p8<2, 0>(pts, B, packed, nullptr, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S,
                           rows_per_class, blocks, workspace, B_stride, rows_fix, clamp_flags, none, stream);
  if (rows_per_class > 4)
    return launch_p8<1, 0>(pts, B, packed, nullptr, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S,
                           rows_per_class, blocks, workspace, B_stride, rows_fix, clamp_flags, none, stream);
  return launch_p8<0, 0>(pts, B, packed, nullptr, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S,
                         rows_per_class, blocks, workspace, B_stride, rows_fix, clamp_flags, none, stream);
}

// ---- cnr_field_bwd_pipe: the stand-alone backward (cnr_field_bwd's contract) on the 8-wave kernel + the record reduction ----
// (chain_waves is kept in the signature: 4 = this kernel, the only pipelined form; the 4-wave forms with 2 / 3 chain waves of
//  rounds 1-2 are gone.  No ray_row, or more than 15 rows per class: the block-split kernels of fused_bwd.hip.)
extern "C" int cnr_field_bwd_pipe_blocks(int R, int S, int chain_waves, int max_blocks) {
  if (R <= 0 || S <= 0 || chain_waves != 4) return 0;
  const int64_t ntiles = ((int64_t)R * S + 31) / 32;
  int64_t blocks = (ntiles + chain_waves - 1) / chain_waves;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int)(blocks > cap ? cap : blocks);
}

extern "C" int cnr_field_bwd_pipe(const float* pts, const float* B, const void* packed, const float* biasrows,
                                  const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                                  float grad_scale, float* dtrunk, float* dB, float* dbiasrows, int C, int R, int S,
                                  int rows_per_class, int max_blocks, int chain_waves, void* workspace,
                                  int64_t workspace_bytes, int64_t B_stride, int64_t dtrunk_stride, int64_t dB_stride,
                                  long long* rows_fix, int skip_reduce, int* clamp_flags, void* stream) {
  if (!pts || !B || !packed || !biasrows || !d_sigma || !d_rgb || !dtrunk || !dB || !dbiasrows || !workspace)
    return CNR_E_ARG;
  if (C <= 0 || R <= 0 || S <= 0 || !(scale > 0.f) || !(grad_scale > 0.f)) return CNR_E_ARG;
  if (chain_waves != 4) return CNR_E_ARG;
  // the pipeline keeps per-object row sums in its row-sum accumulator blocks: class-major rows, at most ROWS_MAX = 15 per
  // class.  Everything else (one row per ray, more objects) takes the block-split kernels.
  if (ray_row == nullptr || rows_per_class < 1 || rows_per_class > cnr_rec::ROWS_MAX) {
    if (rows_fix || skip_reduce) return CNR_E_ARG;   // those two need the pipelined kernel's per-object rows
    return cnr_field_bwd(pts, B, packed, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, dtrunk, dB, dbiasrows, C,
                         R, S, rows_per_class, max_blocks, workspace, workspace_bytes, B_stride, dtrunk_stride, dB_stride,
                         stream);
  }
  if (S > 240) return CNR_E_SHAPE;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)biasrows & 15) != 0 || ((uintptr_t)workspace & 15) != 0)
    return CNR_E_ALIGN;
  const int64_t N = (int64_t)R * S;
  if (N > (int64_t)0x7fffff00) return CNR_E_SHAPE;  // tile and sample indices are 32-bit inside the kernel
  const int blocks = cnr_field_bwd_pipe_blocks(R, S, 4, max_blocks);
  if (workspace_bytes < (int64_t)C * blocks * REC_FLOATS * (int64_t)sizeof(float)) return CNR_E_ARG;
  const int rc = cnr_field_bwd_pipe8_launch(pts, B, packed, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S,
                                            rows_per_class, blocks, workspace, B_stride, rows_fix, clamp_flags, stream);
  if (rc != CNR_OK) return rc;
  if (skip_reduce) return CNR_OK;   // the caller reduces the records itself (cnr_step_tail / cnr_step_grad)
  hipLaunchKernelGGL(reduce_records_kernel, dim3(REC_FLOATS / 64, (unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, blocks, dtrunk, dB, dbiasrows, rows_per_class,
                     dtrunk_stride > 0 ? dtrunk_stride : (int64_t)TRUNK, dB_stride > 0 ? dB_stride : (int64_t)63);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

// ---- the ONE-launch step body: a8-a15 forward, losses, and the whole backward (see TrainArgs) ------------------------
// padded sample slots per ray of the one-launch form: 16 (two rays per tile), 32, 64 or 128; 0 = not supported
static int train_slots(int S) { return S <= 0 ? 0 : S <= 16 ? 16 : S <= 32 ? 32 : S <= 64 ? 64 : S <= 128 ? 128 : 0; }
extern "C" int cnr_field_train_blocks(int R, int S, int max_blocks) {
  const int sp = train_slots(S);
  if (R <= 0 || !sp) return 0;
  const int64_t ntiles = ((int64_t)R * sp + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int)(blocks > cap ? cap : blocks);
}
extern "C" int64_t cnr_field_train_workspace_bytes(int C, int R, int S, int max_blocks) {
  const int nb = cnr_field_train_blocks(R, S, max_blocks);
  return nb ? ((int64_t)C * nb * 3 + (int64_t)C * 4) * (int64_t)sizeof(float) : 0;
}
extern "C" int cnr_field_train(const cnr_field_train_args* a, void* stream) {
  if (!a || a->struct_size != sizeof(cnr_field_train_args) || a->abi_version != CNR_ABI_VERSION) return CNR_E_ARG;
  const int C = a->C, R = a->R, S = a->S, rows_per_class = a->rows_per_class;
  if (!a->pts || !a->B || !a->packed || !a->biasrows || !a->ray_row || !a->z || !a->gt_depth || !a->gt_rgb || !a->labels ||
      !a->depth_mask || !a->counts_tab || !a->records || !a->loss_workspace || C <= 0 || R <= 0 || !(a->scale > 0.f) ||
      !(a->grad_scale > 0.f))
    return CNR_E_ARG;
  const int blocks = cnr_field_train_blocks(R, S, a->max_blocks);
  const int sp = train_slots(S);
  // up to ROWS_MAX object rows per class in the row-sum blocks; more (up to ROWS_TILE_MAX) with one object per tile, i.e. a whole
  // ray per tile group (S > 16) and the fixed-point table
  if (!blocks || rows_per_class < 1 || rows_per_class > cnr_rec::ROWS_TILE_MAX) return CNR_E_SHAPE;
  if (rows_per_class > cnr_rec::ROWS_MAX && (sp == 16 || !a->rows_fix)) return CNR_E_SHAPE;
  if ((int64_t)C * R * sp >= ((int64_t)1 << 31)) return CNR_E_SHAPE;   // 32-bit slot indices inside the kernel
  if (((uintptr_t)a->packed & 15) != 0 || ((uintptr_t)a->records & 15) != 0 || ((uintptr_t)a->packed_lo & 15) != 0 ||
      ((uintptr_t)a->biasrows & 15) != 0)
    return CNR_E_ALIGN;
  if (a->records_bytes < (int64_t)C * blocks * REC_FLOATS * (int64_t)sizeof(float) ||
      a->loss_workspace_bytes < cnr_field_train_workspace_bytes(C, R, S, a->max_blocks))
    return CNR_E_ARG;
  const bool pad = S != sp;   // exact fit: the plain index arithmetic (2 % faster at configs[1] than the padded form)
  // one object per tile whenever a ray has its own tiles and the table is there: it is also the faster form for 5 .. 15 objects
  // (no one-hot table, one row-sum block: 38.4 us at 2048 x 64 for any count, against 40.7 / 41.9 us at 7 / 12 objects in the blocks)
  if (rows_per_class > cnr_rec::ROWS_MAX || (rows_per_class > 4 && sp >= 32 && a->rows_fix)) return cnr_ft_wide3(a, blocks, sp, pad, stream);
  if (rows_per_class > 7) return cnr_ft_wide2(a, blocks, sp, pad, stream);
  if (rows_per_class > 4) return cnr_ft_wide1(a, blocks, sp, pad, stream);
  return cnr_ft_wide0(a, blocks, sp, pad, stream);
}

