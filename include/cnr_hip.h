/*
 * cnr_hip.h -- C-ABI of libcnr_hip.so: MI355X (gfx950) kernels for the volumetric-rendering
 * train-step hot path of Taekbum/category-nerf-reconstruction-official.
 *
 * The reference is pure Python/PyTorch: there is no FFI to replace.  Each entry
 * point below therefore cites the reference *function* whose arithmetic it implements; the Python
 * binding (category-nerf-reconstruction-official_amd/_C.py, ctypes) sits under the reference's own
 * call surface (trainer.py / scene_cateogries.py / embedding.py / model.py / render_rays.py /
 * loss.py).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-allocated, contiguous, row-major memory;
 *     f32 unless the name says otherwise.  No hidden allocation, no host sync, no global state.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - return 0 on success, a CNR_E_* code (<0) for argument errors, or a positive hipError_t.
 *   - C = classes (the functorch vmap axis of train.py:154-155), R = rays per class,
 *     S = samples per ray, N = R*S, L = latent dim, E = 129 = E1 (87) + E2 (42).
 */
#ifndef CNR_HIP_H
#define CNR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNR_OK 0
#define CNR_E_ARG (-1)     /* bad size / null pointer                                  */
#define CNR_E_SHAPE (-2)   /* shape not supported by this kernel (see function doc)    */
#define CNR_E_ALIGN (-3)   /* pointer not aligned as documented                        */

#define CNR_E1 87
#define CNR_E2 42
#define CNR_E 129
#define CNR_W 32
#define CNR_NDIR 21
#define CNR_NFREQ 6
/* fp32 trunk blob: the 10 per-sample Linear layers of CodeNeRF (src/model.py:35-54) in
 * state_dict order, each as weight (out x in, row-major) followed by bias (out):
 *   encoding_xyz.0 87->32 | shape_layer_1.0 32->32 | shape_layer_2.0 32->32 | cat_layer.0 119->32
 *   encoding_shape 32->32 | sigma.0 32->1 | encoding_viewdir.0 74->32 | texture_layer_1.0 32->32
 *   rgb.0 32->16 | rgb.2 16->3                                              = 13 892 floats   */
#define CNR_TRUNK_PARAMS 13892
/* latent slots of zlat (.., 4, 32): post-ReLU outputs of the four latent layers, per ray:
 *   0 shape_latent_layer_1 | 1 cat_latent_layer | 2 shape_latent_layer_2 | 3 texture_latent_layer_1 */
#define CNR_NLAT 4

/* library / device probe: returns CNR_OK and fills n_cu, lds_bytes when a gfx950 device is current */
int cnr_version(void);
int cnr_device_info(int* n_cu, int* lds_bytes, int* gcn_arch_is_gfx950);

/* ---- a1: cameraInfo.get_rays_dirs (src/scene_cateogries.py:613-629) --------------------------------------------------
 * dirs (W,H,3): dirs[w][h] = ((w - cx) / fx, (h - cy) / fy, 1) -- indexed [w, h] (the reference transposes images at load),
 * NOT normalised (z-depth convention).  Correctly rounded fp32 subtraction and division: bit-equal to the reference. */
int cnr_camera_rays(float* dirs, int W, int H, float fx, float fy, float cx, float cy, void* stream);

/* ---- a2-a5: ray transform + depth-guided sampling (src/scene_cateogries.py:24-47, 51-96, 453-546)
 * One pool slice of R rays (per class; C slices batched).  Inputs:
 *   rgbs  (C,R,4) u8  [r,g,b,state]      depth (C,R)      dirs_c (C,R,3)     T (C,R,4,4)
 *   u     (C,R,n1+n2) uniform [0,1) draws, g (C,R,n2) N(0,(eps/3)^2) draws  (parity mode), or both
 *         NULL with `seed`/`offset` != 0 for in-kernel Philox draws (perf mode).
 *   world_frame: 0 -> origin_dirs_O (T is T_CO, inverted in-kernel, sim3 closed form is NOT assumed:
 *         a general 3x3 inverse is used), 1 -> origin_dirs_W (T is T_WC).
 *   max_bound (C,) : per-class max(depth) over the slice (reference :486); computed by
 *         cnr_sample_maxdepth below (or by the previous step's cnr_step_epilogue) so that no host sync is
 *         needed; NULL: every wave takes the maximum itself (same value, R extra loads per ray).
 * Outputs: z (C,R,S) with S = n1+n2, pts (C,R,S,3), origins (C,R,3) and dirs_o (C,R,3) (may be NULL),
 *   gt_rgb (C,R,3) f32 already /255 (train.py:144), gt_depth (C,R) copy of the slice's depth (may be
 *   NULL), depth_mask (C,R) u8, labels (C,R) u8; ray_row (C,R) i32 = pool_indices[slice] + c * n_obj, the row of
 *   each ray in class-major (C*n_obj, ...) code / bias-row tables (NULL to skip; pool_indices is (C,R) or,
 *   with pool_rows > 0, the (C,pool_rows) int64 base). */
/* cnr_sample_rays, max_bound_slices = k > 1: max_bound is a (C, k) table over the epoch's slices (cnr_slice_maxdepth), indexed on
 * the device by d_state[0] / R -- no per-step maximum launch; 0 or 1: one value per class for this step's slice. */
int cnr_sample_maxdepth(const float* depth, float* max_bound, const int64_t* d_state, int64_t pool_rows,
                        const int* perm, int C, int R, void* stream);
int cnr_sample_rays(const uint8_t* rgbs, const float* depth, const float* dirs_c, const float* T,
                    const float* u, const float* g, uint64_t seed, uint64_t offset,
                    const int64_t* d_state, int64_t pool_rows,
                    const float* max_bound, int world_frame, int C, int R, int n1, int n2,
                    float eps, float stop_eps, float min_bound,
                    float* z, float* pts, float* origins, float* dirs_o,
                    float* gt_rgb, float* gt_depth, uint8_t* depth_mask, uint8_t* labels,
                    const int64_t* pool_indices, int n_obj, int* ray_row, const int* perm, int max_bound_slices,
                    void* stream);
/* Device-resident step state int64[3] = {pool cursor (rows), rng step, optimiser step}.  With pool_rows > 0
 * the four pool pointers above are the BASES of (C, pool_rows, ...) pools and the slice starts at row
 * d_state[0] (src/scene_cateogries.py:422-431's i_batch); the Philox offset advances by d_state[1].
 * cnr_step_advance adds (add_rows, 1, 1): a captured hipGraph of the train step replays unchanged.
 * perm (C,pool_rows) i32, optional with pool_rows > 0: row r of the slice is pool row perm[c][cursor + r] -- the
 * per-epoch reshuffle (src/scene_cateogries.py:439-449) becomes a new permutation instead of moving the pool. */
int cnr_step_advance(int64_t* d_state, int64_t add_rows, void* stream);

/* ---- a8: UniDirsEmbed (src/embedding.py:82-92).  x (C,N,3), B (C,21,3) -> e (C,N,129).
 * e[0:3] = x/scale ; e[3+21k+j] = sin(pi * 2^k * (B_j . x/scale)), k = 0..5.
 * bwd: de (C,N,129) -> dB (C,21,3) (accumulated with atomics: zero it first), dx NULL or (C,N,3). */
int cnr_pe_fwd(const float* x, const float* B, float* e, int C, int64_t N, float scale, void* stream);
int cnr_pe_bwd(const float* x, const float* B, const float* de, float* dB, float* dx,
               int C, int64_t N, float scale, void* stream);

/* ---- a9: CodeNeRF trunk, exact fp32 (src/model.py:56-84 with do_cat=True, noise_std=None).
 * e (C,R,S,129), zlat (C,R,4,32) post-ReLU latent-layer outputs per ray (slots above),
 * trunk (C,13892) -> sigmas (C,R,S) (= raw*10), rgbs (C,R,S,3).
 * bwd recomputes the forward, then: dsig (C,R,S), drgb (C,R,S,3) ->
 *   de (C,R,S,129), dzlat (C,R,4,32) and dtrunk (C,13892); dzlat/dtrunk are ACCUMULATED (zero first). */
int cnr_mlp_fwd_f32(const float* e, const float* zlat, const float* trunk, float* sigmas, float* rgbs,
                    int C, int R, int S, void* stream);
int cnr_mlp_bwd_f32(const float* e, const float* zlat, const float* trunk,
                    const float* dsig, const float* drgb,
                    float* de, float* dzlat, float* dtrunk, int C, int R, int S, void* stream);

/* ---- a11-a13: occupancy_activation + occupancy_to_termination + render x4
 * (src/render_rays.py:3-7,25-33,46-50 ; src/loss.py:41-48).  NR = C*R rays.
 * alpha (NR,S), color (NR,S,3), z (NR,S) -> term (NR,S) [may be NULL], depth, var, opacity (NR,), rgb (NR,3).
 * bwd: d_depth, d_opacity (NR,), d_rgb (NR,3) [var is detached in the reference] and optional
 * d_term (NR,S) [NULL = none] -> d_alpha (NR,S), d_color (NR,S,3).
 * in_is_occ = 1: `alpha` already holds occupancies (occupancy_to_termination called on its own, as
 * src/trainer.py:147 / src/category_registration.py:152 do); d_alpha is then d/d(occupancy).
 * color/z and the outputs that need them may be NULL. */
int cnr_composite_fwd(const float* alpha, const float* color, const float* z, float* term,
                      float* depth, float* var, float* rgb, float* opacity,
                      int64_t NR, int S, int in_is_occ, void* stream);
int cnr_composite_bwd(const float* alpha, const float* color, const float* z,
                      const float* d_depth, const float* d_rgb, const float* d_opacity,
                      const float* d_term, float* d_alpha, float* d_color,
                      int64_t NR, int S, int in_is_occ, void* stream);

/* ---- a14-a15 fused: masks + L1 losses + masked means + their gradient w.r.t. the rendered values
 * (src/loss.py:18-74, src/render_rays.py:52-95).  Per class c: mask counts are reduced in-kernel
 * (no host sync, no workspace).  Reference quirk kept: if ANY class has an empty mask for a term, that term is
 * zero (no gradient) for ALL classes (render_rays.py:67-72).
 * inputs (C,R): depth, var, opacity, gt_depth ; rgb, gt_rgb (C,R,3) ; labels, depth_mask (C,R) u8.
 * outputs: losses (3,C) [depth,color,opacity]; flags (C,) i32: bit0 = "loss explode" (> 1e5,
 *   render_rays.py:87-89 -- reported, never exit()), bits1-3 = the depth / colour / opacity term was
 *   zeroed because some class has an empty mask, bit4 (fused trainer only) = the field backward clipped a scaled
 *   gradient (cnr_step_tail); d_depth/d_opacity (C,R), d_rgb (C,R,3) = dLoss/d(render),
 *   already multiplied by color_scaling / opacity_scaling and by `grad_scale`. */
int cnr_loss_fwd_bwd(const float* depth, const float* var, const float* rgb, const float* opacity,
                     const float* gt_depth, const float* gt_rgb, const uint8_t* labels,
                     const uint8_t* depth_mask, float color_scaling, float opacity_scaling,
                     float grad_scale, float* losses, int32_t* flags,
                     float* d_depth, float* d_rgb, float* d_opacity, int C, int R, void* stream);

/* ---- a18: AdamW on one flat fp32 buffer (train.py:40,183; torch.optim.AdamW semantics,
 * amsgrad=False, maximize=False).  step_count is the 1-based step number held on the host, or, when
 * d_state != NULL, the step is d_state[2] + 1 read on the device. */
int cnr_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay,
                   int64_t step_count, float grad_unscale, const int64_t* d_state, void* stream);

/* ---- a7 + latent layers over the fused trainer's flat parameter rows (csrc/latent.hip).  theta / grad: (C,
 * class_stride) floats, each row [trunk 13892 | ...]; off_* = float offsets of latent W (4,32,L), latent b (4,32),
 * shape codes (n_obj,L), texture codes (n_obj,L) inside a row.  fwd -> zl (C*n_obj,4,32) post-ReLU latent outputs,
 * biasrows (C*n_obj,4,32).  bwd: dbiasrows -> grad rows: latent W/b and code entries are WRITTEN, the trunk's
 * Wt_k[:, :32] / bt_k entries are ADDED (cnr_field_bwd's dtrunk must already be there); the code-norm regulariser
 * reg_scale * code / ||code|| (src/loss.py:5-15, train.py:165-167) is included when n_obj > 1. */
int cnr_latent_fwd(const float* theta, int64_t class_stride, int64_t off_latW, int64_t off_latb,
                   int64_t off_shape, int64_t off_tex, int L, int n_obj, int C, float* zl, float* biasrows,
                   void* stream);
int cnr_latent_bwd(const float* theta, int64_t class_stride, int64_t off_latW, int64_t off_latb,
                   int64_t off_shape, int64_t off_tex, int L, int n_obj, int C, const float* zl,
                   const float* dbiasrows, float reg_scale, float* grad, const int* n_obj_cls, void* stream);
/* n_obj_cls (optional, (C,) int32 on the device; also cnr_step_tail / cnr_step_grad): the number of objects each class really
 * has when classes differ (train.py:92-96): rows n_obj_cls[c] .. n_obj - 1 of class c are padding (no ray refers to them); the
 * regulariser applies to a class's real rows, and only when it has more than one object (src/loss.py:5-15).  NULL: n_obj. */

/* ================= fused f16-MFMA field path (a8 + a9 in one launch) ================================
 * Packed operand image per class, produced on device from the fp32 trunk blob: forward A fragments,
 * fp32 constants, transposed (backward) A fragments -- layout contract in csrc/fused_common.h.
 * The four latent layers and the code tables stay in PyTorch (per-object work);
 * they reach the kernels as effective bias rows
 *     biasrows[row][slot][32] = W_l . relu(latent_l(code[row])) + b_l ,
 *     slot 0 shape_layer_1 | 1 cat_layer (first 32 columns) | 2 shape_layer_2 | 3 texture_layer_1
 * (row = object, or ray when codes are per ray) because W (a + z) + b = W a + (W z + b).
 * ray_row (C*R,) int32 maps a ray to its row (NULL: row = ray). */
int64_t cnr_pack_bytes(void);
int cnr_pack_weights(const float* trunk, void* packed, int C, void* stream);

/* pts (C,R,S,3), B (C,21,3), packed (C,cnr_pack_bytes()), biasrows (rows,4,32) -> sigmas (C,R,S) = raw*10,
 * rgbs (C,R,S,3).  f16 MFMA operands / fp32 accumulate, fp32 PE, fp32 sigma head.  Any S. */
/* Precise geometry branch (packed_lo != NULL in cnr_field_fwd / cnr_field_fwd_render / cnr_field_train): the layers
 * between the sample position and the x10 occupancy logit (encoding_xyz, shape_layer_1, cat_layer, shape_layer_2,
 * encoding_shape; src/model.py:56-75) are computed as THREE f16 products per fragment,
 *     W x ~= Wh xh + Wl xh + Wh xl,   Wh = f16(W), Wl = f16(W - Wh), xh = f16(x), xl = f16(x - xh),   fp32 accumulate,
 * i.e. with ~22 mantissa bits on both operands; the colour branch stays plain f16.  Needed for north_star's 1e-3 on the
 * occupancy of TRAINED models: the logit reaches several hundred, and plain f16 operands then give 1.8e-3 .. 2.0e-3
 * (weights and activations contribute equally; splitting either alone leaves 1.2e-3 .. 1.6e-3), this form 2e-6.
 * packed_lo: (C, cnr_pack_lo_bytes()) = the 20 residual fragments Wl of those layers, from cnr_pack_weights_lo(trunk
 * (C,13892), ...) or cnr_step_prologue.  The backward's data-gradient chain uses f16(W) alone. */
int64_t cnr_pack_lo_bytes(void);
int cnr_pack_weights_lo(const float* trunk, void* packed_lo, int C, void* stream);

/* Class strides (floats; 0 = dense): B_stride between the (21,3) direction matrices of consecutive classes,
 * dtrunk_stride / dB_stride between the per-class gradient blocks -- so that B, dtrunk and dB may be views of a flat
 * (C, P) parameter / gradient buffer (the fused trainer's layout) with no gather or scatter copies. */
int cnr_field_fwd(const float* pts, const float* B, const void* packed, const float* biasrows,
                  const int* ray_row, float scale, float* sigmas, float* rgbs, int C, int R, int S,
                  int64_t B_stride, const void* packed_lo,
                  void* stream);

/* Backward of cnr_field_fwd: cnr_field_bwd_pipe, declared below beside the one-launch step body it shares its kernel with
 * (the block-split cnr_field_bwd of rounds 1-3 is gone).  Size of its per-workgroup records: */
int64_t cnr_field_bwd_workspace_bytes(int C, int max_blocks);

/* ---- dense layers of the background model, OccupancyMap (src/model.py:86-155) -----------------
 * y (M,N) = act(x (M,K) W^T + b), W (N,K) row-major as torch.nn.Linear stores it, relu != 0: act = ReLU.
 * Exact fp32 on the matrix core (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation).
 * Backward: dpre = dy * (y > 0) (relu) or dy; dx (M,K) = dpre W (NULL to skip), dW (N,K) = dpre^T x, db (N,) =
 * column sums of dpre (NULL to skip); all three OVERWRITTEN.  y is only read when relu != 0.
 * workspace: >= cnr_dense_bwd_workspace_bytes(M, K, N) bytes (0 for M <= 256): the dW product is split over
 * chunks of samples whose partial results are added in a fixed order (reproducible, no atomics). */
int cnr_dense_fwd(const float* x, const float* W, const float* b, float* y, int M, int K, int N, int relu,
                  int f16_operands, void* stream);
int64_t cnr_dense_bwd_workspace_bytes(int M, int K, int N);
int cnr_dense_bwd(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW, float* db,
                  int M, int K, int N, int relu, void* workspace, int64_t workspace_bytes, int f16_operands,
                  float grad_scale, void* stream);
/* f16_operands != 0: both operands of every product are rounded to f16 as they are staged in LDS and the matrix instruction
 * is v_mfma_f32_32x32x16_f16 (fp32 accumulation, fp32 results): the f16 tier of the background model.  In the backward the
 * gradient operand is multiplied by grad_scale (a power of two, the loss scale of the f16 chain; <= 0: 1) on the way in and
 * the products by 1 / grad_scale on the way out. */

/* ---- ray-pool construction from frames (src/scene_cateogries.py:164-260, 262-325) ------------
 * images (F,W,H,3) u8, depth (F,W,H) f32, obj_mask (F,W,H) i32 (instance id per pixel, -1 unknown), rays_dir (W,H,3):
 * the frames of one scene stacked in HBM.  crops (n_crops,8) i32 = {frame slot, instance id, object index, w0, w1,
 * h0, h1, frame index in the instance's own frame list}; crop_offset (n_crops + 1) i64 = exclusive prefix sum of the crop areas; T_crop (n_crops,4,4) = the pose
 * every ray of the crop carries (T_co = inv(T_wc) T_obj, or T_wc for the background / a single object).
 * perm (N,) i64 or NULL: output row r is source row perm[r] (the reference's global shuffle x = x[shuffled_idx]).
 * Outputs, N = crop_offset[n_crops] rows: rgbs (N,4) u8 [r,g,b,state], depth (N,), dirs (N,3), T (N,4,4) (NULL to
 * skip), indices (N,) i64 = object index (NULL to skip), frame (N,) i64 = that frame index (NULL to skip). */
int cnr_gather_pool(const uint8_t* images, const float* depth, const int32_t* obj_mask, const float* rays_dir,
                    const float* T_crop, const int32_t* crops, const int64_t* crop_offset, const int64_t* perm, int W,
                    int H, int n_crops, int64_t N, uint8_t* rgbs, float* depth_out, float* dirs, float* T_out,
                    int64_t* indices, int64_t* frame_out, void* stream);

/* Parameter-only work of one fused-trainer step in ONE launch, three independent jobs side by side in the grid:
 * cnr_pack_weights (trunk of class c at theta + c * class_stride + off_trunk), cnr_latent_fwd (same arguments), and
 * a zero fill of zero_buf[0 .. zero_count) (the gradient buffers; 16-B aligned; zero_count 0 to skip). */
int cnr_param_prep(const float* theta, int64_t class_stride, int64_t off_trunk, int64_t off_latW, int64_t off_latb,
                   int64_t off_shape, int64_t off_tex, int L, int n_obj, int C, void* packed, float* zl,
                   float* biasrows, float* zero_buf, int64_t zero_count, void* stream);

/* cnr_field_fwd and cnr_render_loss in ONE launch for S in {32, 64, 96, 128}: a wave owns whole rays (S / 32
 * consecutive tiles), composites them in registers and writes d_sigmas / d_colors (and the optional renders) instead of
 * sigmas / colours.  Arguments as the two calls; results agree with them to fp32 summation order (lane sums run over
 * 32-lane tiles here).  workspace >= cnr_field_fwd_render_workspace_bytes(C, R, S): per-block loss partials for
 * cnr_render_loss_finish / cnr_step_tail, which must be told rl_blocks = cnr_field_fwd_render_blocks(R, S).
 * Returns CNR_E_SHAPE for any other S (use the two calls). */
int cnr_field_fwd_render_blocks(int R, int S);
int64_t cnr_field_fwd_render_workspace_bytes(int C, int R, int S);
int cnr_field_fwd_render(const float* pts, const float* B, const void* packed, const float* biasrows,
                         const int* ray_row, float scale, const float* z, const float* gt_depth, const float* gt_rgb,
                         const uint8_t* labels, const uint8_t* depth_mask, float color_scaling, float opacity_scaling,
                         float grad_scale, float* d_sigmas, float* d_colors, float* depth, float* var, float* rgb,
                         float* opacity, int C, int R, int S, int64_t B_stride, void* workspace,
                         int64_t workspace_bytes, const void* packed_lo, const float* counts_tab,
                         const int64_t* d_state, void* stream);

/* Mask counts of every slice of an epoch in one launch (the per-step counts of src/render_rays.py:66-95 depend on the pool
 * rows only: label = pool state, depth mask = depth > min_bound): out (slices, C + 1, 4) floats,
 *   out[s][c] = {#(valid depth & label != 0), #(label != 0), #(label != 2), 0} over rows perm[c][s R .. s R + R) of class c,
 *   out[s][C] = {1 if ANY class has a zero depth / colour / opacity count in slice s, ..., ..., 0}   (render_rays.py:67-72).
 * cnr_field_fwd_render / cnr_render_loss take such a table as `counts_tab` (NULL: they count the step's labels
 * themselves) and read entry d_state[0] / R (d_state NULL: entry 0).  The host may combine tables between GPUs before
 * use -- sum the counts of ray shards of one class, or the flags of class shards -- which is how N ranks x R rays equal
 * one rank x N R rays, and how the empty-mask rule spans ranks, with no collective inside the step. */
int cnr_slice_maskcounts(const uint8_t* rgbs, const float* depth, const int* perm, int64_t pool_rows, int C, int R,
                         int slices, float min_bound, float* out, void* stream);

/* The epoch shuffle in one launch: perm (C, pool_rows) i32, perm[c] = a pseudo-random permutation of [0, pool_rows) that is a
 * function of (seed, epoch, class id) only -- class id = class_ids[c] (device, C entries) or c when NULL -- so the ranks of a
 * class-sharded run produce the same order for a class without drawing each other's (the reference reshuffles with
 * torch.randperm, src/scene_cateogries.py:439-449; the order is not a parity target).  A keyed 6-round Feistel
 * bijection, cycle-walked into range.  state_cursor != NULL: the launch also sets state_cursor[0] = cursor0 (the pool cursor of
 * the device step state starts the epoch).  pool_rows <= 2^30. */
int cnr_epoch_perm(int* perm, int64_t pool_rows, int C, uint64_t seed, uint64_t epoch, const int* class_ids,
                   int64_t* state_cursor, int64_t cursor0, void* stream);

/* Max depth of every slice of an epoch in one launch: out[c][s] = max over r < R of depth[c][perm[c][s R + r]]
 * (perm NULL: identity), s < slices, slices * R <= pool_rows.  cnr_step_prologue takes such a table as max_bound when
 * max_bound_slices = slices > 1 and indexes it with the device cursor / R -- the fused trainer fills it once per
 * reshuffle instead of computing the next slice's maximum in every step's last launch. */
int cnr_slice_maxdepth(const float* depth, const int* perm, int64_t pool_rows, int C, int R, int slices, float* out,
                       void* stream);

/* ---- versioned argument blocks -----------------------------------------------------------------------------------------
 * The three launches of the fused trainer's step (cnr_step_prologue -> cnr_field_train -> cnr_step_tail) take their
 * arguments as ONE struct by pointer instead of 35-50 positional values: `struct_size` = sizeof(the struct) and
 * `abi_version` = CNR_ABI_VERSION come first and the entry point returns CNR_E_ARG for any other value, so a caller
 * compiled (or a ctypes table written) against another revision of this header is refused instead of reading shifted
 * fields.  Field meaning is the positional entry points' of the calls they fuse (cnr_param_prep / cnr_sample_rays, ...). */
#define CNR_ABI_VERSION 3

/* cnr_param_prep and cnr_sample_rays side by side in ONE launch (max_bound must be given here, max_bound_slices = 0 or 1
 * for the per-class form): the sampler needs the ray pool and the step state only, so the first node of the fused
 * trainer's step runs it beside the parameter-only jobs instead of after them.  packed_lo (optional,
 * (C, cnr_pack_lo_bytes())): the residual image of the geometry branch (cnr_pack_weights_lo) is built in the same launch.
 * rgbs = NULL: the parameter-only jobs without the sampler blocks (a host that samples elsewhere, e.g. on another stream). */
typedef struct cnr_step_prologue_args {
  uint32_t struct_size;
  uint32_t abi_version;
  const float* theta;
  int64_t class_stride;
  int64_t off_trunk;
  int64_t off_latW;
  int64_t off_latb;
  int64_t off_shape;
  int64_t off_tex;
  int32_t L;
  int32_t n_obj;
  int32_t C;
  void* packed;
  void* packed_lo;
  float* zl;
  float* biasrows;
  float* zero_buf;
  int64_t zero_count;
  const uint8_t* rgbs;
  const float* depth;
  const float* dirs_c;
  const float* T;
  const float* u;
  const float* g;
  uint64_t seed;
  uint64_t offset;
  const int64_t* d_state;
  int64_t pool_rows;
  const float* max_bound;
  int32_t world_frame;
  int32_t R;
  int32_t n1;
  int32_t n2;
  float eps;
  float stop_eps;
  float min_bound;
  float* z;
  float* pts;
  float* origins;
  float* dirs_o;
  float* gt_rgb;
  float* gt_depth;
  uint8_t* depth_mask;
  uint8_t* labels;
  const int64_t* pool_indices;
  int32_t* ray_row;
  const int32_t* perm;
  int32_t max_bound_slices;
  int32_t rng_c0;
  int32_t rng_cstride;
  int32_t rng_R;
  int32_t rng_r0;
} cnr_step_prologue_args;
int cnr_step_prologue(const cnr_step_prologue_args* args, void* stream);
/* rng_*: the Philox counter of ray r of local class c is ((c * rng_cstride + rng_c0) * rng_R + rng_r0 + r) * 64 + lane,
 * i.e. the ray's index in the GLOBAL batch when classes (rng_c0 = rank, rng_cstride = ranks) or rays (rng_R = global rays
 * per class, rng_r0 = rank * R) are sharded over GPUs: N ranks then draw exactly the samples one rank would.  All zero:
 * the local index c * R + r. */

/* a11-a15 fused for the render + loss step of the fused trainer: cnr_composite_fwd -> cnr_loss_fwd_bwd ->
 * cnr_composite_bwd in ONE kernel (src/render_rays.py:3-7,25-33,46-95; src/loss.py:18-74).  Possible because the
 * gradient of the masked-mean losses w.r.t. one ray's renders needs that ray's values and the mask counts only.
 * sigmas (C,R,S) are the pre-sigmoid occupancy logits (CodeNeRF's raw * 10), colors (C,R,S,3), z (C,R,S).
 * Outputs: d_sigmas (C,R,S), d_colors (C,R,S,3) = dL/dsigmas, dL/dcolors of
 *   sum_c depth + color_scaling * color + opacity_scaling * opacity,  times grad_scale,
 * bit-identical to the three-call sequence; depth, var (C,R), rgb (C,R,3), opacity (C,R): the renders, each
 * optional (NULL to skip); per-block partial sums of the three loss terms into `workspace`
 * (>= cnr_render_loss_workspace_bytes(C, R) bytes).  S <= 512.
 * cnr_render_loss_finish turns those partials into losses (3,C) and flags (C), exactly as cnr_loss_fwd_bwd
 * defines them (fixed summation order, no float atomics).  It only needs the stream order after
 * cnr_render_loss, so a trainer can run it on a side stream beside the field backward. */
int64_t cnr_render_loss_workspace_bytes(int C, int R);
int cnr_render_loss(const float* sigmas, const float* colors, const float* z, const float* gt_depth,
                    const float* gt_rgb, const uint8_t* labels, const uint8_t* depth_mask, float color_scaling,
                    float opacity_scaling, float grad_scale, float* d_sigmas, float* d_colors, float* depth,
                    float* var, float* rgb, float* opacity, int C, int R, int S, void* workspace,
                    int64_t workspace_bytes, const float* counts_tab, const int64_t* d_state, void* stream);
int cnr_render_loss_finish(const void* workspace, float* losses, int32_t* flags, int C, int R, int rl_blocks,
                           void* stream);   /* rl_blocks: partials per class, 0 = cnr_render_loss's own count */
/* Last node of the fused trainer's captured step, one launch: cnr_render_loss_finish, then (next_max_bound !=
 * NULL) cnr_sample_maxdepth for the NEXT step's slice [cursor + add_rows, + R) of the (C,pool_rows) depth pool,
 * then cnr_step_advance(d_state, add_rows).  The next step's cnr_sample_rays reads next_max_bound. */
int cnr_step_epilogue(int64_t* d_state, int64_t add_rows, const void* workspace, float* losses, int32_t* flags,
                      const float* depth, int64_t pool_rows, const int* perm, float* next_max_bound, int C, int R,
                      void* stream);

/* cnr_adamw_step and cnr_step_epilogue side by side in ONE launch, for a step state that ping-pongs between two
 * int64[3] buffers: every kernel of step k reads state_cur, this launch writes state_next = state_cur + (add_rows,
 * 1, 1); the caller swaps the two for step k + 1 (two captured graphs, or a pointer swap in eager mode).  With the
 * current state read-only during the launch, the AdamW blocks (optimiser step = state_cur[2] + 1) and the epilogue
 * block need no ordering.  Arguments as in cnr_adamw_step / cnr_step_epilogue. */
int cnr_adamw_epilogue(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float grad_unscale,
                       const int64_t* state_cur, int64_t* state_next, int64_t add_rows, const void* workspace,
                       float* losses, int32_t* flags, const float* depth, int64_t pool_rows, const int* perm,
                       float* next_max_bound, int C, int R, void* stream);

/* The tail of the fused trainer's step in ONE launch: cnr_latent_bwd (do_latent != 0), AdamW on the flat parameter
 * buffer and cnr_step_epilogue, for PING-PONG parameters and step state: every kernel of step k reads theta_in /
 * state_cur, this launch writes theta_out / state_next (AdamW out of place), the caller swaps them for step k + 1.
 * With nothing written that anything in the launch reads, the three jobs run side by side; a latent-path gradient
 * element is applied by the thread that produces it, and the trunk entries the latent path adds to are finished by
 * the AdamW blocks.  do_latent = 0: grad already holds the complete gradient (cnr_latent_bwd ran, e.g. before a
 * multi-GPU all-reduce) and only AdamW + epilogue run.  grad is updated to the complete gradient either way.
 * Layout arguments as cnr_latent_bwd (trunk at offset 0 of a class row, B at off_B); rl_workspace etc. as
 * cnr_step_epilogue.
 * records != NULL (with do_latent, n_obj <= 4): the launch ALSO replaces the record reduction of
 * cnr_field_bwd_pipe(..., skip_reduce = 1): `records` is that call's workspace, nwg =
 * cnr_field_bwd_pipe_blocks(...), and rows_fix the (8, C, n_obj, 4, 32) int64 table that call accumulated (2^-40 fixed
 * point, integer atomics: any order, same sum) -- zero it before every field backward.  dbiasrows then receives the
 * float form of that table.
 * rl_blocks: loss partials per class in rl_workspace; 0 = cnr_render_loss's own block count.
 * code_lr > 0: the shape / texture code tables are an AdamW group of their own (code_lr, code_weight_decay: train.py:40,
 * 54-64, configs' code_lr / code_weight_decay); 0 = they share lr / weight_decay. */
typedef struct cnr_step_tail_args {
  uint32_t struct_size;
  uint32_t abi_version;
  const float* theta_in;
  float* theta_out;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t class_stride;
  int64_t off_B;
  int64_t off_latW;
  int64_t off_latb;
  int64_t off_shape;
  int64_t off_tex;
  int32_t L;
  int32_t n_obj;
  int32_t C;
  const float* zl;
  float* dbiasrows;
  float reg_scale;
  int32_t do_latent;
  float lr;
  float beta1;
  float beta2;
  float eps;
  float weight_decay;
  const int64_t* state_cur;
  int64_t* state_next;
  int64_t add_rows;
  const void* rl_workspace;
  float* losses;
  int32_t* flags;
  const float* depth;
  int64_t pool_rows;
  const int32_t* perm;
  float* next_max_bound;
  int32_t R;
  const void* records;
  int32_t nwg;
  const long long* rows_fix;
  int32_t rl_blocks;
  int32_t* clamp_flags;
  const int32_t* n_obj_cls;
  float code_lr;
  float code_weight_decay;
} cnr_step_tail_args;
int cnr_step_tail(const cnr_step_tail_args* args, void* stream);
/* clamp_flags (optional, (C,) int32, zero before the first step): what cnr_field_bwd_pipe raised during this step -- bit 4
 * (16) = a scaled upstream gradient |d sigma| * grad_scale exceeded 8192 and was clipped for the f16 chain; the epilogue
 * or-s the word into flags[c] and clears it. */

/* The gradient half of cnr_step_tail on its own (records reduction + latent backward + code regulariser in one launch,
 * no optimiser, no epilogue): for hosts that need the finished gradient first, e.g. to all-reduce it across GPUs.
 * Same arguments as the corresponding ones of cnr_step_tail; follow it (after the collective) with
 * cnr_step_tail(..., do_latent = 0, ..., records = NULL, nwg = 0, rows_fix = NULL, ...). */
int cnr_step_grad(const float* theta, float* grad, int64_t class_stride, int64_t off_B, int64_t off_latW,
                  int64_t off_latb, int64_t off_shape, int64_t off_tex, int L, int n_obj, int C, const float* zl,
                  float* dbiasrows, float reg_scale, const void* records, int nwg, const long long* rows_fix,
                  const int* n_obj_cls, void* stream);

/* Backward of cnr_field_fwd (recomputes the forward per tile): d_sigma (C,R,S) = dL/dsigmas, d_rgb (C,R,S,3) -> dtrunk
 * (C,13892), dB (C,21,3), dbiasrows (rows,4,32); all three ACCUMULATED (zero them first).  dtrunk receives no bias gradient for
 * the four latent-conditioned layers: those biases reach the kernel only through biasrows, whose gradient the caller
 * back-propagates (cnr_latent_bwd / cnr_step_tail).  grad_scale: power-of-two loss scale applied to d_sigma / d_rgb on load and
 * removed on store (the data-gradient chain runs on f16 MFMA operands); scaled d_sigma is clamped to +-8192.  S <= 240.
 * ONE field kernel + the record reduction: a workgroup is four chain waves that run forward recompute + data-gradient chain + PE
 * backward for one 32-sample tile each, plus four waves that own the weight-gradient accumulators and consume the chain waves'
 * per-layer images behind a workgroup barrier -- eight waves, two per SIMD (csrc/fused_bwd_pipe8_kernel.h).  chain_waves must be
 * 4 (kept in the signature).  max_blocks: workgroups per class (0 = 256).  workspace: caller-allocated, 16-B aligned, contents
 * irrelevant, >= cnr_field_bwd_workspace_bytes(C, max_blocks) bytes: every workgroup stores one record of partial sums into it
 * with plain stores -- 16 128 bf16 entries, each the workgroup's fp32 sum rounded once -- and a last small kernel sums the records
 * in fp32, in a fixed order: no float atomic anywhere, bitwise reproducible.
 * packed_lo (optional): the residual image the FORWARD ran with (cnr_field_fwd / cnr_field_fwd_render, precise geometry branch):
 * the recompute then forms the same activations and ReLU masks as that forward -- without it the backward of a precise forward
 * would be the gradient of the plain-f16 function.
 * Rows: ray_row REQUIRED, class-major, 1 .. 15 per class (their bias-row sums travel in the records and in rows_fix);
 * CNR_E_SHAPE otherwise -- larger classes (up to 128 objects) train on cnr_field_train, one object per tile. */
/* rows_fix (optional): per-object bias-row sums also accumulated as 2^-40 fixed point into this (8, C, n_obj, 4, 32)
 * int64 table (8 copies, a workgroup uses copy index & 7, to shorten the same-address atomic queues; caller zeroes it); skip_reduce != 0: stop after the field kernel, the caller reduces the nwg =
 * cnr_field_bwd_pipe_blocks(R, S, chain_waves, max_blocks) records per class in `workspace` itself (cnr_step_tail). */
int cnr_field_bwd_pipe_blocks(int R, int S, int chain_waves, int max_blocks);
int cnr_field_bwd_pipe(const float* pts, const float* B, const void* packed, const void* packed_lo, const float* biasrows,
                       const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                       float grad_scale, float* dtrunk, float* dB, float* dbiasrows, int C, int R, int S,
                       int rows_per_class, int max_blocks, int chain_waves, void* workspace,
                       int64_t workspace_bytes, int64_t B_stride,
                       int64_t dtrunk_stride, int64_t dB_stride, long long* rows_fix, int skip_reduce,
                       int* clamp_flags, void* stream);

/* ---- BASELINE.json configs[4]: the field forward on the gfx950 fp8 matrix instruction (v_mfma_f32_32x32x16_fp8_fp8) ----
 * Same contract as cnr_field_fwd with OCP e4m3 operands: weights as `terms` fp8 planes of 64 W (plane p = fp8 of what planes
 * 0..p-1 left over), activations / PE features as `terms` planes of 16 x formed on the fly, terms^2 MFMAs per fragment into
 * one fp32 accumulator, fp32 sigma head.  terms in {1, 2, 3}.  `packed` is the f16 image (its fp32 constants are used),
 * `packed8` = (C, cnr_pack_fp8_bytes(terms)) from cnr_pack_weights_fp8.  Forward only: measured against the oracle, terms = 1
 * is 6e-2 on occupancy (60x outside the 1e-3 bar), 2 -> 1.8e-3, 3 -> 1.5e-4, while the f16 kernel's ONE MFMA of the same
 * rate gives 7e-4 -- which is why the train step stays on f16. */
int64_t cnr_pack_fp8_bytes(int terms);
int cnr_pack_weights_fp8(const float* trunk, void* packed8, int C, int terms, void* stream);
int cnr_field_fwd_fp8(const float* pts, const float* B, const void* packed, const void* packed8, const float* biasrows,
                      const int* ray_row, float scale, float* sigmas, float* rgbs, int C, int R, int S, int64_t B_stride,
                      int terms, void* stream);

/* ---- the step body in ONE launch: cnr_field_fwd_render + cnr_field_bwd_pipe (chain_waves = 4) fused ---------------------
 * a8-a15 forward, the loss gradient and the whole field backward (train.py:154-182 for the object branch) with ONE field
 * forward per sample (cnr_field_bwd_pipe recomputes it), no d sigma / d colour round trip through HBM and one launch less.
 * Any S <= 128: a ray occupies 16, 32, 64 or 128 padded sample slots (the smallest that holds S; slots beyond S are dead
 * lanes).  With 16 slots two rays share a 32-sample tile, one per DPP row (the reference's real batch is 10 samples per
 * ray); with 64 / 128 the ray's 2 / 4 tiles sit in neighbouring chain waves of a workgroup iteration and exchange the tile
 * products / partial renders / suffix sums of the composite through LDS.  Arguments as the two calls it replaces;
 * counts_tab (cnr_slice_maskcounts) is REQUIRED -- the kernel never counts masks --, d_state selects its entry (NULL: 0);
 * loss_scale multiplies the loss gradient (what cnr_field_fwd_render calls grad_scale), grad_scale is the power-of-two
 * scale of the f16 data-gradient chain.  Outputs: the renders (each optional), the per-workgroup gradient records in
 * `records` (>= C * cnr_field_train_blocks() * record size = cnr_field_bwd_workspace_bytes(C, blocks); reduce with
 * cnr_step_tail / cnr_step_grad, nwg = cnr_field_train_blocks()), rows_fix as cnr_field_bwd_pipe, and per-block loss
 * partials in loss_workspace (>= cnr_field_train_workspace_bytes()) for cnr_step_tail with rl_blocks =
 * cnr_field_train_blocks().
 * rows_per_class (objects per class): up to 15 travel in the kernel's row-sum blocks (and in the records); 16 .. 32 are taken
 * when a ray has at least 32 sample slots (S > 16) and rows_fix is given: a 32-sample tile then lies inside one ray, i.e.
 * belongs to one object, and its column sums go straight to the fixed-point table at that object's row (integer atomics,
 * once per workgroup iteration); the records carry no row sums then and cnr_step_tail / cnr_step_grad take all of them from
 * the table.  Returns CNR_E_SHAPE for S > 128, for more than 32 rows per class, and for more than 15 with S <= 16 or without
 * rows_fix (use the two calls). */
int cnr_field_train_blocks(int R, int S, int max_blocks, int rows_per_class);
int64_t cnr_field_train_workspace_bytes(int C, int R, int S, int max_blocks, int rows_per_class);
/* (rows_per_class: more than 15 object rows per class need one object per tile, so rays of up to 16 samples -- otherwise two
 *  to a tile -- are padded to a 32-slot tile of their own there: the block count depends on it) */
/* packed_lo (optional, (C, cnr_pack_lo_bytes()) from cnr_pack_weights_lo / cnr_step_prologue): the forward's geometry
 * branch as three products per fragment (see cnr_pack_weights_lo) -- the trainer's default; NULL: plain f16 operands. */
typedef struct cnr_field_train_args {
  uint32_t struct_size;
  uint32_t abi_version;
  const float* pts;
  const float* B;
  const void* packed;
  const void* packed_lo;
  const float* biasrows;
  const int32_t* ray_row;
  float scale;
  const float* z;
  const float* gt_depth;
  const float* gt_rgb;
  const uint8_t* labels;
  const uint8_t* depth_mask;
  const float* counts_tab;
  const int64_t* d_state;
  float color_scaling;
  float opacity_scaling;
  float loss_scale;
  float grad_scale;
  float* depth;
  float* var;
  float* rgb;
  float* opacity;
  int32_t C;
  int32_t R;
  int32_t S;
  int32_t rows_per_class;
  int32_t max_blocks;
  void* records;
  int64_t records_bytes;
  void* loss_workspace;
  int64_t loss_workspace_bytes;
  int64_t B_stride;
  long long* rows_fix;
  int32_t* clamp_flags;
} cnr_field_train_args;
int cnr_field_train(const cnr_field_train_args* args, void* stream);

/* ---- the background field's train step on f16 MFMA, fused (csrc/bg_fused.hip) ----------------------------
 * OccupancyMap(hidden 128), src/model.py:86-155, called every iteration at train.py:113-121,172-180 on R x S world-frame
 * samples (1200 x 14).  theta: the flat fp32 parameter buffer in the modules' registration order,
 *   in_layer.0 W (128,87) b | mid1.0.0 W (128,128) b | cat_layer.0 W (128,215) b | mid2.0.0 W (128,128) b | out_alpha W (1,128) b |
 *   color_linear.0 W (128,170) b | out_color W (3,128) b | B_layer.weight (21,3)          = cnr_bg_param_count() = 94 403 floats.
 * One step = cnr_bg_pack -> cnr_bg_forward -> cnr_render_loss (+ _finish) -> cnr_bg_backward -> cnr_bg_dw -> cnr_bg_tail, M = R S
 * (or, one launch less, cnr_bg_forward -> cnr_bg_backward_render -> cnr_bg_dw -> cnr_bg_tail, below):
 *   pack:     theta -> (cnr_bg_pack_bytes()) f16 MFMA fragments: forward, geometry-branch residual, transposed.
 *   forward:  pts (M,3) -> sigma (M,) = 10 raw, rgb (M,3); act (5,M,128) f16 = post-ReLU outputs of in_layer, mid1, cat_layer,
 *             mid2, color_linear; eimg (M,144) f16 = PE features (E1 in columns 0..95, E2 in 96..143).  Geometry branch
 *             (in_layer .. mid2 -> out_alpha) as three f16 products per fragment, fp32 occupancy head; colour branch plain f16.
 *   backward: d_sigma (M,), d_rgb (M,3) (already multiplied by the loss scale) -> dpre (5,M,128) f16 pre-activation gradients,
 *             records (cnr_bg_blocks(M), cnr_bg_record_floats()): per-workgroup gradients of out_color, out_alpha, B_layer.
 *   dw:       weight / bias gradients of the five 128-wide layers over sample chunks of `chunk` (a multiple of 64) samples
 *             -> partials (cnr_bg_dw_chunks(M, chunk), cnr_bg_param_count()).  d_state != NULL: the launch also advances the
 *             step state by (add_rows, 1, 1) (for steps whose backward is cnr_bg_backward_render; then pass add_rows = -1 to the tail).
 *   tail:     fixed-order sums of partials and records, x 1 / grad_scale -> grad; AdamW in place (step = d_state[2] + 1);
 *             d_state += (add_rows, 1, 1).  With d_state given to cnr_bg_backward (it then advances the state: the sampler, which
 *             reads the cursor, ran before it, the optimiser, which reads the step count, runs after it) pass add_rows = -1
 *             here: the step count is then d_state[2] as it stands and no state launch follows.
 *             packed != NULL: every weight's f16 fragment slots are refreshed by the thread that updates it (the next step then
 *             needs no cnr_bg_pack; call cnr_bg_pack once before the first step and after any outside change of theta);
 *             rl_workspace != NULL: cnr_render_loss's workspace for R rays -> losses (3,1), flags (1) (= cnr_render_loss_finish);
 *             R = 0: rl_workspace is cnr_bg_backward_render's loss workspace (one partial per backward block, nrec of them).
 * No float atomics: two runs give the same bits. */
int64_t cnr_bg_pack_bytes(void);
int cnr_bg_param_count(void);
int cnr_bg_blocks(int M);
int cnr_bg_dw_chunks(int M, int chunk);
int cnr_bg_record_floats(void);
int cnr_bg_pack(const float* theta, void* packed, void* stream);
int cnr_bg_forward(const float* pts, const float* theta, const void* packed, float scale, int M, float* sigma, float* rgb,
                   void* act, void* eimg, void* stream);   /* act = eimg = NULL: the forward alone, nothing kept for a backward
                                                              (Trainer.eval_points' meshing queries, src/trainer.py:125-151) */
int cnr_bg_backward(const float* pts, const float* theta, const void* packed, float scale, int M, const float* d_sigma,
                    const float* d_rgb, const float* rgb, const void* act, void* dpre, float* records, int64_t* d_state,
                    int64_t add_rows, void* stream);
int cnr_bg_dw(const void* act, const void* dpre, const void* eimg, int M, int chunk, float* partials, int64_t* d_state,
              int64_t add_rows, void* stream);
/* cnr_render_loss + cnr_bg_backward in ONE launch (R rays x S <= 64 samples, ray-major: sample m = ray S + s): every backward
 * workgroup first composites the (at most 32 / S + 2) rays its 32-sample tile touches -- one wave per ray, cnr_render_loss's
 * arithmetic expression for expression, so d sigma / d colour are that call's bit for bit -- and walks straight into the
 * data-gradient chain; d sigma / d colour never reach memory.  The tile that holds a ray's first sample writes its renders
 * (depth / var / rgb_render / opacity, each optional) and accounts for its loss terms: loss_workspace
 * (cnr_bg_backward_render_workspace_bytes(M)) holds one partial per workgroup, which cnr_bg_tail (rl_workspace, R = 0) turns
 * into loss values and flags.  counts_tab (cnr_slice_maskcounts, C = 1) is required; d_state (may be NULL: entry 0) is READ for
 * the cursor and not advanced here: give d_state to cnr_bg_dw.  rgb = the forward's per-sample colours.  d_sigma (M,) /
 * d_rgb (M,3), optional: the loss gradient per sample as cnr_render_loss writes it (the chain itself does not need them). */
typedef struct cnr_bg_backward_render_args {
  uint32_t struct_size;
  uint32_t abi_version;
  const float* pts;
  const float* theta;
  const void* packed;
  float scale;
  int32_t R;
  int32_t S;
  const float* sigma;
  const float* rgb;
  const float* z;
  const float* gt_depth;
  const float* gt_rgb;
  const uint8_t* labels;
  const uint8_t* depth_mask;
  const float* counts_tab;
  const int64_t* d_state;
  float color_scaling;
  float opacity_scaling;
  float grad_scale;
  const void* act;
  void* dpre;
  float* records;
  float* depth;
  float* var;
  float* rgb_render;
  float* opacity;
  float* d_sigma;
  float* d_rgb;
  void* loss_workspace;
  int64_t loss_workspace_bytes;
} cnr_bg_backward_render_args;
int64_t cnr_bg_backward_render_workspace_bytes(int M);
int cnr_bg_backward_render(const cnr_bg_backward_render_args* args, void* stream);
int cnr_bg_tail(float* theta, float* grad, float* exp_avg, float* exp_avg_sq, const float* partials, int chunks,
                const float* records, int nrec, float grad_scale, float lr, float beta1, float beta2, float eps,
                float weight_decay, int64_t* d_state, int64_t add_rows, void* packed, const void* rl_workspace, int R,
                float* losses, int32_t* flags, void* stream);

/* cnr_bg_tail with the NEXT step's sampler (cnr_sample_rays: world or object frame, device pool + cursor + permutation, in-kernel
 * Philox, C = 1) as extra blocks of the same launch: nothing in the tail reads what the sampler writes, and the step state has
 * been advanced earlier in the step (cnr_bg_dw with d_state; this entry point has cnr_bg_tail's add_rows = -1 semantics), so
 * cursor and RNG step are the next step's -- the step after then starts with cnr_bg_forward straight away.  When no whole slice
 * is left in the epoch (cursor + R > pool_rows) the sampler blocks write nothing: the host reshuffles and calls cnr_sample_rays.
 * Tail fields as cnr_bg_tail's (rl_R = its R: 0 for cnr_bg_backward_render's workspace); sampler fields as cnr_sample_rays'. */
typedef struct cnr_bg_tail_sample_args {
  uint32_t struct_size;
  uint32_t abi_version;
  float* theta;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  const float* partials;
  int32_t chunks;
  const float* records;
  int32_t nrec;
  float grad_scale;
  float lr;
  float beta1;
  float beta2;
  float adam_eps;
  float weight_decay;
  void* packed;
  const void* rl_workspace;
  int32_t rl_R;
  float* losses;
  int32_t* flags;
  const uint8_t* rgbs;
  const float* depth;
  const float* dirs_c;
  const float* T;
  uint64_t seed;
  uint64_t offset;
  const int64_t* d_state;
  int64_t pool_rows;
  const float* max_bound;
  int32_t max_bound_slices;
  int32_t world_frame;
  int32_t R;
  int32_t n1;
  int32_t n2;
  float eps;
  float stop_eps;
  float min_bound;
  const int32_t* perm;
  float* z;
  float* pts;
  float* origins;
  float* dirs_o;
  float* gt_rgb;
  float* gt_depth;
  uint8_t* depth_mask;
  uint8_t* labels;
} cnr_bg_tail_sample_args;
int cnr_bg_tail_sample(const cnr_bg_tail_sample_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNR_HIP_H */
