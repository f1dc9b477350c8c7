// C-ABI entry points of the 8-wave pipelined kernel (fused_bwd_pipe8_kernel.h) and the stand-alone backward's instantiations.
#include "fused_bwd_pipe8_kernel.h"

// the one-launch instantiations of one WIDE each (fused_bwd_pipe8_w0 .. w3.hip); internal to the library
#define CNR_HIDDEN __attribute__((visibility("hidden")))
CNR_HIDDEN int cnr_ft_wide0(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);
CNR_HIDDEN int cnr_ft_wide1(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);
CNR_HIDDEN int cnr_ft_wide2(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);
CNR_HIDDEN int cnr_ft_wide3(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream);

static int cnr_field_bwd_pipe8_launch(const float* pts, const float* B, const void* packed, const void* packed_lo,
                                          const float* biasrows, const int* ray_row, float scale, const float* d_sigma,
                                          const float* d_rgb, float grad_scale, int C, int R, int S, int rows_per_class, int blocks,
                                          void* workspace, int64_t B_stride, long long* rows_fix, int* clamp_flags,
                                          void* stream) {
  const TrainArgs none{};
#define CNR_BP(W, G)                                                                                                      \
  return launch_p8<W, 0, false, false, G>(pts, B, packed, packed_lo, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S, \
                                          rows_per_class, blocks, workspace, B_stride, rows_fix, clamp_flags, none, stream)
  if (packed_lo) {   // the forward that was rendered ran the precise geometry branch: recompute exactly that
    if (rows_per_class > 7) CNR_BP(2, true);
    if (rows_per_class > 4) CNR_BP(1, true);
    CNR_BP(0, true);
  }
  if (rows_per_class > 7) CNR_BP(2, false);
  if (rows_per_class > 4) CNR_BP(1, false);
  CNR_BP(0, false);
#undef CNR_BP
}

// ---- cnr_field_bwd_pipe: the stand-alone backward on the 8-wave kernel + the record reduction ----------------------------------
// (chain_waves is kept in the signature: 4 = this kernel, the only form.  Class-major rows, 1 .. 15 per class; per-ray rows and
//  larger classes had the block-split kernels of rounds 1-3, which are gone: a class of more than 15 objects trains on the
//  one-launch path, cnr_field_train, which takes up to 128.)
extern "C" int64_t cnr_field_bwd_workspace_bytes(int C, int max_blocks) {
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int64_t)C * cap * REC_ENTRIES * (int64_t)sizeof(rec_t);
}
extern "C" int cnr_field_bwd_pipe_blocks(int R, int S, int chain_waves, int max_blocks) {
  if (R <= 0 || S <= 0 || chain_waves != 4) return 0;
  const int64_t ntiles = ((int64_t)R * S + 31) / 32;
  int64_t blocks = (ntiles + chain_waves - 1) / chain_waves;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int)(blocks > cap ? cap : blocks);
}

extern "C" int cnr_field_bwd_pipe(const float* pts, const float* B, const void* packed, const void* packed_lo,
                                  const float* biasrows, const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                                  float grad_scale, float* dtrunk, float* dB, float* dbiasrows, int C, int R, int S,
                                  int rows_per_class, int max_blocks, int chain_waves, void* workspace,
                                  int64_t workspace_bytes, int64_t B_stride, int64_t dtrunk_stride, int64_t dB_stride,
                                  long long* rows_fix, int skip_reduce, int* clamp_flags, void* stream) {
  if (!pts || !B || !packed || !biasrows || !d_sigma || !d_rgb || !dtrunk || !dB || !dbiasrows || !workspace)
    return CNR_E_ARG;
  if (C <= 0 || R <= 0 || S <= 0 || !(scale > 0.f) || !(grad_scale > 0.f)) return CNR_E_ARG;
  if (chain_waves != 4) return CNR_E_ARG;
  // per-object row sums live in the kernel's row-sum accumulator blocks: class-major rows, at most ROWS_MAX = 15 per class
  if (ray_row == nullptr || rows_per_class < 1 || rows_per_class > cnr_rec::ROWS_MAX) return CNR_E_SHAPE;
  if (S > 240) return CNR_E_SHAPE;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)packed_lo & 15) != 0 || ((uintptr_t)biasrows & 15) != 0 ||
      ((uintptr_t)workspace & 15) != 0)
    return CNR_E_ALIGN;
  const int64_t N = (int64_t)R * S;
  if (N > (int64_t)0x7fffff00) return CNR_E_SHAPE;  // tile and sample indices are 32-bit inside the kernel
  const int blocks = cnr_field_bwd_pipe_blocks(R, S, 4, max_blocks);
  if (workspace_bytes < (int64_t)C * blocks * REC_ENTRIES * (int64_t)sizeof(rec_t)) return CNR_E_ARG;
  const int rc = cnr_field_bwd_pipe8_launch(pts, B, packed, packed_lo, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S,
                                            rows_per_class, blocks, workspace, B_stride, rows_fix, clamp_flags, stream);
  if (rc != CNR_OK) return rc;
  if (skip_reduce) return CNR_OK;   // the caller reduces the records itself (cnr_step_tail / cnr_step_grad)
  hipLaunchKernelGGL(reduce_records_kernel, dim3(REC_ENTRIES / 64, (unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     (const rec_t*)workspace, blocks, dtrunk, dB, dbiasrows, rows_per_class,
                     dtrunk_stride > 0 ? dtrunk_stride : (int64_t)TRUNK, dB_stride > 0 ? dB_stride : (int64_t)63);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

// ---- the ONE-launch step body: a8-a15 forward, losses, and the whole backward (see TrainArgs) ------------------------
// padded sample slots per ray of the one-launch form: 16 (two rays per tile), 32, 64 or 128; 0 = not supported
// (more than ROWS_MAX object rows per class: one object per tile, which needs a tile group per ray -- 16-slot rays, two to a tile,
//  are padded to 32 slots there)
static int train_slots(int S, int rows_per_class) {
  const int sp = S <= 0 ? 0 : S <= 16 ? 16 : S <= 32 ? 32 : S <= 64 ? 64 : S <= 128 ? 128 : 0;
  return (sp == 16 && rows_per_class > cnr_rec::ROWS_MAX) ? 32 : sp;
}
extern "C" int cnr_field_train_blocks(int R, int S, int max_blocks, int rows_per_class) {
  const int sp = train_slots(S, rows_per_class);
  if (R <= 0 || !sp) return 0;
  const int64_t ntiles = ((int64_t)R * sp + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int)(blocks > cap ? cap : blocks);
}
extern "C" int64_t cnr_field_train_workspace_bytes(int C, int R, int S, int max_blocks, int rows_per_class) {
  const int nb = cnr_field_train_blocks(R, S, max_blocks, rows_per_class);
  return nb ? ((int64_t)C * nb * 3 + (int64_t)C * 4) * (int64_t)sizeof(float) : 0;
}
extern "C" int cnr_field_train(const cnr_field_train_args* a, void* stream) {
  if (!a || a->struct_size != sizeof(cnr_field_train_args) || a->abi_version != CNR_ABI_VERSION) return CNR_E_ARG;
  const int C = a->C, R = a->R, S = a->S, rows_per_class = a->rows_per_class;
  if (!a->pts || !a->B || !a->packed || !a->biasrows || !a->ray_row || !a->z || !a->gt_depth || !a->gt_rgb || !a->labels ||
      !a->depth_mask || !a->counts_tab || !a->records || !a->loss_workspace || C <= 0 || R <= 0 || !(a->scale > 0.f) ||
      !(a->grad_scale > 0.f))
    return CNR_E_ARG;
  const int blocks = cnr_field_train_blocks(R, S, a->max_blocks, rows_per_class);
  const int sp = train_slots(S, rows_per_class);
  // up to ROWS_MAX object rows per class in the row-sum blocks; more (up to ROWS_TILE_MAX) with one object per tile, i.e. a whole
  // ray per tile group (S > 16) and the fixed-point table
  if (!blocks || rows_per_class < 1 || rows_per_class > cnr_rec::ROWS_TILE_MAX) return CNR_E_SHAPE;
  if (rows_per_class > cnr_rec::ROWS_MAX && (sp == 16 || !a->rows_fix)) return CNR_E_SHAPE;
  if ((int64_t)C * R * sp >= ((int64_t)1 << 31)) return CNR_E_SHAPE;   // 32-bit slot indices inside the kernel
  if (((uintptr_t)a->packed & 15) != 0 || ((uintptr_t)a->records & 15) != 0 || ((uintptr_t)a->packed_lo & 15) != 0 ||
      ((uintptr_t)a->biasrows & 15) != 0)
    return CNR_E_ALIGN;
  if (a->records_bytes < (int64_t)C * blocks * REC_ENTRIES * (int64_t)sizeof(rec_t) ||
      a->loss_workspace_bytes < cnr_field_train_workspace_bytes(C, R, S, a->max_blocks, rows_per_class))
    return CNR_E_ARG;
  const bool pad = S != sp;   // exact fit: the plain index arithmetic (2 % faster at configs[1] than the padded form)
  // one object per tile whenever a ray has its own tiles and the table is there: it is also the faster form for 5 .. 15 objects
  // (no one-hot table, one row-sum block: 38.4 us at 2048 x 64 for any count, against 40.7 / 41.9 us at 7 / 12 objects in the blocks)
  if (rows_per_class > cnr_rec::ROWS_MAX || (rows_per_class > 4 && sp >= 32 && a->rows_fix)) return cnr_ft_wide3(a, blocks, sp, pad, stream);
  if (rows_per_class > 7) return cnr_ft_wide2(a, blocks, sp, pad, stream);
  if (rows_per_class > 4) return cnr_ft_wide1(a, blocks, sp, pad, stream);
  return cnr_ft_wide0(a, blocks, sp, pad, stream);
}

