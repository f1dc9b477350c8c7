"""torch.autograd.Function wrappers over the C-ABI kernels (modular, exact-fp32 tier).

Every Function works on plain tensors or on class-batched tensors (leading C dim, decided by the
parameter rank) and carries a ``vmap`` rule, so the reference's own
``functorch.combine_state_for_ensemble`` + ``vmap(fmodel)(params, buffers, *stacked_inputs)``
(src/utils.py:24-28, train.py:154-155) dispatches straight into the class-batched kernels.
"""
import os

import torch
from torch.autograd import Function

from . import _C

E1, E2, E, W, NLAT = 87, 42, 129, 32, 4
TRUNK_LAYERS = [  # (state_dict prefix, out, in) in blob order (include/cnr_hip.h: CNR_TRUNK_PARAMS)
    ("encoding_xyz.0", 32, 87), ("shape_layer_1.0", 32, 32), ("shape_layer_2.0", 32, 32),
    ("cat_layer.0", 32, 119), ("encoding_shape", 32, 32), ("sigma.0", 1, 32),
    ("encoding_viewdir.0", 32, 74), ("texture_layer_1.0", 32, 32), ("rgb.0", 16, 32), ("rgb.2", 3, 16),
]
TRUNK_PARAMS = sum(o * i + o for _, o, i in TRUNK_LAYERS)
assert TRUNK_PARAMS == 13892
LATENT_LAYERS = ["shape_latent_layer_1.0", "cat_latent_layer.0", "shape_latent_layer_2.0",
                 "texture_latent_layer_1.0"]  # zlat slot order


def _to_batch0(args, in_dims, batch_size):
    out = []
    for a, d in zip(args, in_dims):
        if not torch.is_tensor(a):
            out.append(a)
        elif d is None:
            out.append(a.unsqueeze(0).expand(batch_size, *a.shape))
        else:
            out.append(a.movedim(d, 0))
    return out


def pack_trunk(params, C):
    """20 tensors (w, b per TRUNK_LAYERS entry; optional leading C) -> (C, 13892) fp32 blob."""
    return torch.cat([p.reshape(C, -1) for p in params], dim=1).contiguous()


# ------------------------------------------------------------------------------------------------
# a8  UniDirsEmbed
# ------------------------------------------------------------------------------------------------
class UniDirsEmbedFn(Function):
    """x (..., 3), B (21, 3)  or class-batched x (C, ..., 3), B (C, 21, 3) -> (..., 129)."""

    @staticmethod
    def forward(x, B, scale):
        C = B.shape[0] if B.dim() == 3 else 1
        xs = x.contiguous().float()
        N = xs.numel() // (3 * C)
        e = torch.empty(*x.shape[:-1], E, device=x.device, dtype=torch.float32)
        _C.call("cnr_pe_fwd", xs, B.contiguous(), e, C, N, float(scale))
        return e

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, B, scale = inputs
        ctx.save_for_backward(x, B)
        ctx.scale = float(scale)

    @staticmethod
    def backward(ctx, de):
        x, B = ctx.saved_tensors
        C = B.shape[0] if B.dim() == 3 else 1
        xs = x.contiguous().float()
        N = xs.numel() // (3 * C)
        dB = torch.zeros_like(B, memory_format=torch.contiguous_format)
        dx = torch.empty_like(xs) if ctx.needs_input_grad[0] else None
        _C.call("cnr_pe_bwd", xs, B.contiguous(), de.contiguous(), dB, dx, C, N, ctx.scale)
        return dx, dB, None

    @staticmethod
    def vmap(info, in_dims, x, B, scale):
        x, B, scale = _to_batch0((x, B, scale), in_dims, info.batch_size)
        return UniDirsEmbedFn.apply(x, B, scale), 0


# ------------------------------------------------------------------------------------------------
# a9  CodeNeRF trunk (the ten per-sample Linear layers), exact fp32
# ------------------------------------------------------------------------------------------------
class CodeNeRFTrunkFn(Function):
    """e (R,S,129), zlat (R,4,32), 20 trunk tensors -> sigmas (R,S,1), rgbs (R,S,3); or all with a
    leading class dim C."""

    @staticmethod
    def forward(e, zlat, *params):
        batched = params[0].dim() == 3
        C = params[0].shape[0] if batched else 1
        S = e.shape[-2]
        R = e.numel() // (C * S * E)
        trunk = pack_trunk(params, C)
        ec, zc = e.contiguous(), zlat.contiguous()
        sig = torch.empty(*e.shape[:-1], 1, device=e.device, dtype=torch.float32)
        rgb = torch.empty(*e.shape[:-1], 3, device=e.device, dtype=torch.float32)
        _C.call("cnr_mlp_fwd_f32", ec, zc, trunk, sig, rgb, C, R, S)
        return sig, rgb

    @staticmethod
    def setup_context(ctx, inputs, output):
        e, zlat = inputs[0], inputs[1]
        params = inputs[2:]
        C = params[0].shape[0] if params[0].dim() == 3 else 1
        ctx.save_for_backward(e, zlat, *params)
        ctx.C = C

    @staticmethod
    def backward(ctx, dsig, drgb):
        e, zlat, *params = ctx.saved_tensors
        C = ctx.C
        S = e.shape[-2]
        R = e.numel() // (C * S * E)
        trunk = pack_trunk(params, C)
        ec, zc = e.contiguous(), zlat.contiguous()
        de = torch.empty_like(ec)
        dz = torch.zeros_like(zc)
        dtrunk = torch.zeros_like(trunk)
        _C.call("cnr_mlp_bwd_f32", ec, zc, trunk, dsig.contiguous(), drgb.contiguous(), de, dz, dtrunk, C, R, S)
        grads, off = [], 0
        for p in params:
            cnt = p.numel() // C
            grads.append(dtrunk[:, off:off + cnt].reshape(p.shape))
            off += cnt
        return (de, dz, *grads)

    @staticmethod
    def vmap(info, in_dims, *args):
        args = _to_batch0(args, in_dims, info.batch_size)
        return CodeNeRFTrunkFn.apply(*args), (0, 0)


# ------------------------------------------------------------------------------------------------
# a11-a13  composite
# ------------------------------------------------------------------------------------------------
class CompositeFn(Function):
    """alpha (..., S), color (..., S, 3), z (..., S) -> term (..., S), depth, var, opacity (...), rgb (..., 3).
    var is non-differentiable (src/loss.py:46 detaches it)."""

    @staticmethod
    def forward(alpha, color, z):
        S = alpha.shape[-1]
        NR = alpha.numel() // S
        a, c, zz = alpha.contiguous(), color.contiguous(), z.contiguous()
        lead = alpha.shape[:-1]
        kw = dict(device=alpha.device, dtype=torch.float32)
        term = torch.empty(*lead, S, **kw)
        depth, var, opa = torch.empty(*lead, **kw), torch.empty(*lead, **kw), torch.empty(*lead, **kw)
        rgb = torch.empty(*lead, 3, **kw)
        _C.call("cnr_composite_fwd", a, c, zz, term, depth, var, rgb, opa, NR, S, 0)
        return term, depth, var, rgb, opa

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(*inputs)
        ctx.mark_non_differentiable(output[2])

    @staticmethod
    def backward(ctx, d_term, d_depth, d_var, d_rgb, d_opa):
        alpha, color, z = ctx.saved_tensors
        S = alpha.shape[-1]
        NR = alpha.numel() // S
        a, c, zz = alpha.contiguous(), color.contiguous(), z.contiguous()
        d_alpha, d_color = torch.empty_like(a), torch.empty_like(c)
        cont = lambda t: None if t is None else t.contiguous()
        _C.call("cnr_composite_bwd", a, c, zz, cont(d_depth), cont(d_rgb), cont(d_opa), cont(d_term),
                d_alpha, d_color, NR, S, 0)
        return d_alpha, d_color, None


class TerminationFn(Function):
    """occupancy (..., S) -> termination (..., S): occ_i * prod_{j<i}(1 - occ_j + 1e-10)
    (src/render_rays.py:25-33), the stand-alone form trainer / registration code calls."""

    @staticmethod
    def forward(occ):
        S = occ.shape[-1]
        o = occ.contiguous().float()
        term = torch.empty_like(o)
        _C.call("cnr_composite_fwd", o, None, None, term, None, None, None, None, o.numel() // S, S, 1)
        return term

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(inputs[0])

    @staticmethod
    def backward(ctx, d_term):
        (occ,) = ctx.saved_tensors
        S = occ.shape[-1]
        o = occ.contiguous().float()
        d_occ = torch.empty_like(o)
        _C.call("cnr_composite_bwd", o, None, None, None, None, None, d_term.contiguous(), d_occ, None,
                o.numel() // S, S, 1)
        return d_occ


# ------------------------------------------------------------------------------------------------
# a14-a15  masked L1 losses (one launch, no host sync)
# ------------------------------------------------------------------------------------------------
class RenderLossFn(Function):
    """(C,R) renders + ground truth + u8 labels / depth mask -> losses (3, C) [depth, colour, opacity]
    and flags (C,) int32 (include/cnr_hip.h).  Gradients flow to depth, rgb, opacity."""

    @staticmethod
    def forward(depth, var, rgb, opacity, gt_depth, gt_rgb, labels, depth_mask):
        C, R = depth.shape
        kw = dict(device=depth.device, dtype=torch.float32)
        losses = torch.empty(3, C, **kw)
        flags = torch.empty(C, device=depth.device, dtype=torch.int32)
        dd, dr, do = torch.empty(C, R, **kw), torch.empty(C, R, 3, **kw), torch.empty(C, R, **kw)
        _C.call("cnr_loss_fwd_bwd", depth.contiguous(), var.contiguous(), rgb.contiguous(),
                opacity.contiguous(), gt_depth.contiguous().float(), gt_rgb.contiguous().float(),
                labels.contiguous(), depth_mask.contiguous(), 1.0, 1.0, 1.0, losses, flags, dd, dr, do, C, R)
        return losses, flags, dd, dr, do

    @staticmethod
    def setup_context(ctx, inputs, output):
        losses, flags, dd, dr, do = output
        ctx.save_for_backward(dd, dr, do)
        ctx.mark_non_differentiable(flags, dd, dr, do)

    @staticmethod
    def backward(ctx, g, _gf, _g1, _g2, _g3):
        dd, dr, do = ctx.saved_tensors
        return dd * g[0][:, None], None, dr * g[1][:, None, None], do * g[2][:, None], None, None, None, None


# ------------------------------------------------------------------------------------------------
# a2-a5  sampling (no gradient: poses are not optimised)
# ------------------------------------------------------------------------------------------------
def sample_rays(rgbs, depth, dirs_c, T, n1, n2, eps, stop_eps, min_bound=0.0, world_frame=False,
                u=None, g=None, seed=0, offset=0, want_rays=False, d_state=None, rays=None, out=None,
                max_bound=None,
                pool_indices=None, n_obj=0, perm=None, max_bound_slices=0):
    """Class-batched pool slice (C,R,...) -> dict(z, pts, gt_rgb, gt_depth, depth_mask, labels[, origins, dirs_o]).
    u/g given -> parity mode (identical draws); else in-kernel Philox(seed, offset).
    d_state (int64[3] device) + rays=R: the inputs are whole (C,pool_rows,...) pools and the slice starts at
    the device-side cursor d_state[0] (see cnr_step_advance).  out: dict of preallocated outputs to reuse."""
    C = depth.shape[0]
    pool_rows = 0 if d_state is None else depth.shape[1]
    R = depth.shape[1] if d_state is None else int(rays)
    S = n1 + n2
    dev = depth.device
    kw = dict(device=dev, dtype=torch.float32)
    o = out if out is not None else {}
    def buf(name, shape, dtype=torch.float32):
        if name not in o:
            o[name] = torch.empty(*shape, device=dev, dtype=dtype)
        return o[name]
    if max_bound is not None:
        mb = max_bound                # already there (cnr_step_epilogue of the previous step, or the caller)
    else:
        mb = buf("max_bound", (C,))
        _C.call("cnr_sample_maxdepth", depth, mb, d_state, pool_rows, perm, C, R)
    z, pts = buf("z", (C, R, S)), buf("pts", (C, R, S, 3))
    gt, gd = buf("gt_rgb", (C, R, 3)), buf("gt_depth", (C, R))
    dm, lab = buf("depth_mask", (C, R), torch.uint8), buf("labels", (C, R), torch.uint8)
    org = buf("origins", (C, R, 3)) if want_rays else None
    dro = buf("dirs_o", (C, R, 3)) if want_rays else None
    rr = buf("ray_row", (C, R), torch.int32) if pool_indices is not None else None
    cont = lambda t: None if t is None else t.contiguous()
    _C.call("cnr_sample_rays", rgbs.contiguous(), depth.contiguous(), dirs_c.contiguous(), T.contiguous(),
            cont(u), cont(g), int(seed), int(offset), d_state, pool_rows, mb, int(bool(world_frame)), C, R, n1, n2,
            float(eps), float(stop_eps), float(min_bound), z, pts, org, dro, gt, gd, dm, lab,
            pool_indices, int(n_obj), rr, perm, int(max_bound_slices))
    return o


def step_prologue(theta, lay, L, n_obj, packed, zl, brows, zero_buf, rgbs, depth, dirs_c, T, n1, n2, eps, stop_eps,
                  min_bound, seed, d_state, rays, out, max_bound, pool_indices, perm, max_bound_slices=0,
                  rng=(0, 0, 0, 0), packed_lo=None, sample=True):
    """cnr_step_prologue: the parameter-only jobs (pack | latent rows | gradient zero fill) and the sampler of one
    fused-trainer step in ONE launch.  Same outputs dict as :func:`sample_rays` (device pools, device cursor).
    max_bound: (C,) max depth of this step's slice, or with max_bound_slices = k > 1 a (C, k) table over the epoch's
    slices (cnr_slice_maxdepth), indexed on the device by cursor / rays.  sample=False: the parameter-only jobs, the ray buffers
    of ``out`` are left as they are (the caller samples elsewhere)."""
    C, R, S = depth.shape[0], int(rays), n1 + n2
    dev = depth.device
    def buf(name, shape, dtype=torch.float32):
        if name not in out:
            out[name] = torch.empty(*shape, device=dev, dtype=dtype)
        return out[name]
    z, pts = buf("z", (C, R, S)), buf("pts", (C, R, S, 3))
    gt, gd = buf("gt_rgb", (C, R, 3)), buf("gt_depth", (C, R))
    dm, lab = buf("depth_mask", (C, R), torch.uint8), buf("labels", (C, R), torch.uint8)
    rr = buf("ray_row", (C, R), torch.int32)
    _C.call_struct("cnr_step_prologue", theta=theta, class_stride=lay.total, off_trunk=lay.trunk[0], off_latW=lay.latW[0],
                   off_latb=lay.latb[0], off_shape=lay.shape[0], off_tex=lay.tex[0], L=L, n_obj=n_obj, C=C, packed=packed,
                   packed_lo=packed_lo, zl=zl, biasrows=brows, zero_buf=zero_buf, zero_count=zero_buf.numel(),
                   rgbs=rgbs if sample else None, depth=depth, dirs_c=dirs_c, T=T, u=None, g=None, seed=int(seed), offset=0,
                   d_state=d_state,
                   pool_rows=depth.shape[1], max_bound=max_bound, world_frame=0, R=R, n1=n1, n2=n2, eps=float(eps),
                   stop_eps=float(stop_eps), min_bound=float(min_bound), z=z, pts=pts, origins=None, dirs_o=None, gt_rgb=gt,
                   gt_depth=gd, depth_mask=dm, labels=lab, pool_indices=pool_indices, ray_row=rr, perm=perm,
                   max_bound_slices=int(max_bound_slices), rng_c0=int(rng[0]), rng_cstride=int(rng[1]), rng_R=int(rng[2]),
                   rng_r0=int(rng[3]))
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step, grad_unscale=1.0,
               d_state=None):
    _C.call("cnr_adamw_step", param, grad, exp_avg, exp_avg_sq, param.numel(), float(lr), float(betas[0]),
            float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_unscale), d_state)


def step_advance(d_state, add_rows):
    _C.call("cnr_step_advance", d_state, int(add_rows))


# ================================================================================================
# fused f16-MFMA field path
# ================================================================================================
# (weight offset, bias offset, leading dim) of the four latent-conditioned layers inside the trunk blob,
# in biasrows slot order (include/cnr_hip.h)
_LATENT_TARGETS = [(2816, 3840, 32), (4928, 8736, 119), (3872, 4896, 32), (12257, 13281, 32)]


def bias_rows(trunk, zlat):
    """trunk (C,13892), zlat (C,rows,4,32) post-ReLU latent outputs -> (C,rows,4,32) effective biases
    W_l z_l + b_l of shape_layer_1 / cat_layer[:, :32] / shape_layer_2 / texture_layer_1 (differentiable)."""
    C = trunk.shape[0]
    outs = []
    for k, (wo, bo, ld) in enumerate(_LATENT_TARGETS):
        Wk = trunk[:, wo:wo + 32 * ld].view(C, 32, ld)[:, :, :32]
        bk = trunk[:, bo:bo + 32]
        outs.append(torch.baddbmm(bk[:, None, :], zlat[:, :, k, :], Wk.transpose(1, 2)))
    return torch.stack(outs, dim=2)


def pack_weights(trunk):
    """fp32 trunk blob (C,13892) -> packed f16 operand image (C, cnr_pack_bytes()) uint8, on device."""
    C = trunk.shape[0]
    packed = torch.empty(C, _C.pack_bytes(), device=trunk.device, dtype=torch.uint8)
    _C.call("cnr_pack_weights", trunk.contiguous(), packed, C)
    return packed


# ------------------------------------------------------------------------------------------------
# SURVEY 8(f).1  dense layers of the background OccupancyMap (src/model.py:86-155)
# ------------------------------------------------------------------------------------------------
DENSE_GRAD_SCALE = 1024.0   # loss scale of the f16 dense tier's gradient operands (a power of two)


class DenseFn(Function):
    """y = act(x W^T + b) on the MFMA dense kernel (cnr_dense_fwd / cnr_dense_bwd); x (..., K), W (N, K).
    half = False: exact fp32 (v_mfma_f32_32x32x2_f32), the parity tier; half = True: operands rounded to f16 in LDS,
    v_mfma_f32_32x32x16_f16, fp32 accumulation -- the f16 tier of the background model."""

    @staticmethod
    def forward(ctx, x, W, b, relu, half=False, grad_out=None):
        # grad_out = (dW, db) tensors: the backward WRITES the parameter gradients there and reports none to autograd -- for a
        # host that keeps gradients in a flat buffer and uses the layer once per step (background.BackgroundStep: one add
        # launch per parameter and step less)
        ctx.grad_out = grad_out
        K, N = W.shape[1], W.shape[0]
        x2 = x.reshape(-1, K).contiguous()
        y = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
        _C.call("cnr_dense_fwd", x2, W.contiguous(), b.contiguous() if b is not None else None, y, x2.shape[0], K, N,
                int(bool(relu)), int(bool(half)))
        ctx.save_for_backward(x2, W, y)
        ctx.relu, ctx.xshape, ctx.has_b, ctx.half = bool(relu), x.shape, b is not None, bool(half)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        M, K, N = x2.shape[0], W.shape[1], W.shape[0]
        dy2 = dy.reshape(M, N).contiguous()
        dx = torch.empty(M, K, device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        direct = ctx.grad_out
        dW = direct[0] if direct else torch.empty(N, K, device=dy.device, dtype=torch.float32)
        db = (direct[1] if direct else torch.empty(N, device=dy.device, dtype=torch.float32)) if ctx.has_b else None
        wsb = int(_C.load().cnr_dense_bwd_workspace_bytes(M, K, N))
        ws = torch.empty(max(wsb, 16), device=dy.device, dtype=torch.uint8)
        _C.call("cnr_dense_bwd", x2, W.contiguous(), y, dy2, dx, dW, db, M, K, N, int(ctx.relu), ws, wsb, int(ctx.half),
                DENSE_GRAD_SCALE)
        if direct:
            return (dx.reshape(ctx.xshape) if dx is not None else None), None, None, None, None, None
        return (dx.reshape(ctx.xshape) if dx is not None else None), dW, db, None, None, None


def field_bwd(pts, B, packed, biasrows, ray_row, scale, d_sig, d_rgb, grad_scale, dtrunk, dB, dbiasrows, C, R, S,
              rows_per_class, max_blocks, workspace, B_stride=0, dtrunk_stride=0, dB_stride=0,
              rows_fix=None, skip_reduce=False, clamp_flags=None, packed_lo=None):
    """cnr_field_bwd_pipe: the stand-alone field backward (8-wave pipelined kernel + record reduction).  strides (floats, 0 =
    dense): B / dtrunk / dB may be views into a flat (C, P) buffer, see cnr_hip.h.  packed_lo: the residual image the forward
    ran with -- the recompute must form that forward's activations."""
    _C.call("cnr_field_bwd_pipe", pts, B, packed, packed_lo, biasrows, ray_row, scale, d_sig, d_rgb, grad_scale, dtrunk,
            dB, dbiasrows, C, R, S, rows_per_class, max_blocks, 4, workspace, workspace.numel(), int(B_stride),
            int(dtrunk_stride), int(dB_stride), rows_fix, int(bool(skip_reduce)), clamp_flags)


def pack_weights_lo(trunk):
    """residual fragments W - f16(W) for the split-weight forward: (C, cnr_pack_lo_bytes()) uint8"""
    C = trunk.shape[0]
    lo = torch.empty(C, int(_C.load().cnr_pack_lo_bytes()), device=trunk.device, dtype=torch.uint8)
    _C.call("cnr_pack_weights_lo", trunk.contiguous(), lo, C)
    return lo


def field_fwd(pts, B, packed, biasrows, ray_row, scale, packed_lo=None):
    """pts (C,R,S,3) -> sigmas (C,R,S), rgbs (C,R,S,3); biasrows (rows,4,32) flat over classes.
    packed_lo (pack_weights_lo): split-weight mode, f16(W) + f16(W - f16(W))."""
    C, R, S, _ = pts.shape
    sig = torch.empty(C, R, S, device=pts.device, dtype=torch.float32)
    rgb = torch.empty(C, R, S, 3, device=pts.device, dtype=torch.float32)
    _C.call("cnr_field_fwd", pts.contiguous(), B.contiguous(), packed, biasrows.contiguous(), ray_row,
            float(scale), sig, rgb, C, R, S, 0, packed_lo)
    return sig, rgb


def field_fwd_fp8(pts, B, trunk, packed, biasrows, ray_row, scale, terms=1):
    """BASELINE.json configs[4]: cnr_field_fwd on the fp8 matrix instruction (OCP e4m3 weights AND activations, `terms`
    residual planes each); forward only.  trunk (C,13892) fp32, packed = pack_weights(trunk) (for its fp32 constants)."""
    C, R, S, _ = pts.shape
    p8 = torch.empty(C, int(_C.load().cnr_pack_fp8_bytes(int(terms))), device=pts.device, dtype=torch.uint8)
    _C.call("cnr_pack_weights_fp8", trunk.contiguous(), p8, C, int(terms))
    sig = torch.empty(C, R, S, device=pts.device, dtype=torch.float32)
    rgb = torch.empty(C, R, S, 3, device=pts.device, dtype=torch.float32)
    _C.call("cnr_field_fwd_fp8", pts.contiguous(), B.contiguous(), packed, p8, biasrows.contiguous(), ray_row,
            float(scale), sig, rgb, C, R, S, 0, int(terms))
    return sig, rgb


class FusedFieldFn(Function):
    """pts (C,R,S,3), B (C,21,3), trunk (C,13892), biasrows (C*rows,4,32) -> sigmas (C,R,S), rgbs (C,R,S,3)
    on the f16-MFMA kernels; backward = cnr_field_bwd_pipe (recompute).  ray_row (C,R) int32: the bias row of every ray,
    class-major, rows_per_class (1 .. 15) rows per class; grad_scale = power-of-two loss scale for the f16 chain;
    precise: the geometry branch as three f16 products per fragment, forward AND recompute (include/cnr_hip.h)."""

    @staticmethod
    def forward(pts, B, trunk, biasrows, ray_row, scale, rows_per_class, grad_scale, max_blocks, precise=False):
        packed = pack_weights(trunk)
        lo = pack_weights_lo(trunk) if precise else None
        sig, rgb = field_fwd(pts, B, packed, biasrows, ray_row, scale, packed_lo=lo)
        return sig, rgb, packed, (lo if precise else packed.new_empty(0))

    @staticmethod
    def setup_context(ctx, inputs, output):
        pts, B, trunk, biasrows, ray_row, scale, rows_per_class, grad_scale, max_blocks = inputs[:9]
        ctx.save_for_backward(pts, B, biasrows, output[2], output[3])
        ctx.ray_row, ctx.scale, ctx.rpc, ctx.gs, ctx.mb = ray_row, float(scale), int(rows_per_class), float(grad_scale), int(max_blocks)
        ctx.trunk_shape = trunk.shape
        ctx.mark_non_differentiable(output[2], output[3])

    @staticmethod
    def backward(ctx, d_sig, d_rgb, _, _lo):
        pts, B, biasrows, packed, lo = ctx.saved_tensors
        C, R, S, _3 = pts.shape
        dtrunk = torch.zeros(ctx.trunk_shape, device=pts.device, dtype=torch.float32)
        dB = torch.zeros_like(B, memory_format=torch.contiguous_format)
        dbr = torch.zeros_like(biasrows, memory_format=torch.contiguous_format)
        wsp = torch.empty(_C.field_bwd_workspace_bytes(C, ctx.mb), device=pts.device, dtype=torch.uint8)
        field_bwd(pts.contiguous(), B.contiguous(), packed, biasrows.contiguous(), ctx.ray_row,
                  ctx.scale, d_sig.contiguous(), d_rgb.contiguous(), ctx.gs, dtrunk, dB, dbr, C, R, S, ctx.rpc, ctx.mb,
                  wsp, packed_lo=lo if lo.numel() else None)
        return None, dB, dtrunk, dbr, None, None, None, None, None, None
