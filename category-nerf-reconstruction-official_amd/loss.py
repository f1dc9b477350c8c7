"""Loss assembly with the reference's signature (src/loss.py:5-74).  step_batch_loss = composite
kernel + fused masked-L1 kernel: two launches, no host sync (the reference syncs six times here)."""
import torch

from .ops import CompositeFn, RenderLossFn

last_flags = None  # (C,) int32 device tensor of the latest step_batch_loss call (see check_flags)


def step_batch_loss_reg(cls_dict, cls_ids):
    """src/loss.py:5-15: sum of per-object code L2 norms for classes holding more than one object."""
    loss_reg_shape = torch.zeros(cls_ids.shape[0], device=cls_ids.device)
    loss_reg_texture = torch.zeros_like(loss_reg_shape)
    for idx, cls_k in enumerate(cls_dict.values()):
        if len(cls_k.obj_ids) > 1:
            loss_reg_shape[idx] = torch.norm(cls_k.trainer.shape_codes.weight, dim=-1).sum()
            loss_reg_texture[idx] = torch.norm(cls_k.trainer.texture_codes.weight, dim=-1).sum()
    return loss_reg_shape, loss_reg_texture


def step_batch_loss(alpha, color, gt_depth, gt_color, sem_labels, mask_depth, z_vals,
                    color_scaling=5.0, opacity_scaling=10.0):
    """alpha (C,R,S,1), color (C,R,S,3), gt_depth (C,R), gt_color (C,R,3), sem_labels (C,R) uint8,
    mask_depth (C,R) bool, z_vals (C,R,S) -> (loss, {'depth','color','opacity': (C,)}, loss_col)."""
    global last_flags
    alpha = alpha.squeeze(dim=-1)
    _term, depth, var, rgb, opacity = CompositeFn.apply(alpha, color, z_vals)
    losses, flags, _, _, _ = RenderLossFn.apply(depth, var, rgb, opacity, gt_depth, gt_color,
                                                sem_labels.to(torch.uint8), mask_depth.to(torch.uint8))
    last_flags = flags
    loss_depth, loss_col, loss_opacity = losses[0], losses[1], losses[2]
    l_batch = loss_depth + loss_col * color_scaling + loss_opacity * opacity_scaling
    return l_batch.sum(), {'depth': loss_depth, 'color': loss_col, 'opacity': loss_opacity}, loss_col


def check_flags(flags=None):
    """Host-side check of the device flags (one sync; call it at log cadence, not per step).
    Raises render_rays.LossExplode where the reference would exit(-1)."""
    from .render_rays import LossExplode
    f = last_flags if flags is None else flags
    if f is not None and bool((f & 1).any()):
        raise LossExplode("loss explode")
