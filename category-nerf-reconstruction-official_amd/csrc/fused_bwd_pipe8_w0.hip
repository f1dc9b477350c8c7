// One-launch instantiations (cnr_field_train) of the 8-wave kernel for WIDE = 0: up to four object rows per class.
#include "fused_bwd_pipe8_kernel.h"

__attribute__((visibility("hidden"))) int cnr_ft_wide0(const cnr_field_train_args* a, int blocks, int sp, bool pad, void* stream) {
  const int C = a->C, R = a->R, S = a->S, rows_per_class = a->rows_per_class;
  const TrainArgs ta{a->z, a->gt_depth, a->gt_rgb, a->labels, a->depth_mask, a->counts_tab, a->d_state, a->color_scaling,
                     a->opacity_scaling, a->loss_scale, a->depth, a->var, a->rgb, a->opacity, (float*)a->loss_workspace};
#define CNR_FT(KR, TWO, PAD, GEO)                                                                                       \
  return launch_p8<0, KR, TWO, PAD, GEO>(a->pts, a->B, a->packed, a->packed_lo, a->biasrows, a->ray_row, a->scale,   \
                                          nullptr, nullptr, a->grad_scale, C, R, S, rows_per_class, blocks, a->records, \
                                          a->B_stride, a->rows_fix, a->clamp_flags, ta, stream)
#define CNR_FT_ALL(G)                                                                                     \
  {                                                                                                       \
    if (sp == 16) CNR_FT(1, true, true, G);                                                               \
    if (sp == 32) { if (pad) CNR_FT(1, false, true, G); CNR_FT(1, false, false, G); }                     \
    if (sp == 64) { if (pad) CNR_FT(2, false, true, G); CNR_FT(2, false, false, G); }                     \
    if (pad) CNR_FT(4, false, true, G);                                                                   \
    CNR_FT(4, false, false, G);                                                                           \
  }
  if (a->packed_lo) CNR_FT_ALL(true)
  CNR_FT_ALL(false)
#undef CNR_FT_ALL
#undef CNR_FT
}

#ifdef CNR_PIPE_STAMPS
extern "C" int cnr_pipe8_read_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pipe8_stamps), sizeof(long long) * 8 * 64);
}
#endif
