"""Volume-rendering primitives with the reference's names and semantics (src/render_rays.py).
``occupancy_to_termination`` and the four ``render`` reductions of the train step run fused in the
composite HIP kernel (see :func:`composite`); the stand-alone primitives below exist because
trainer.eval_points / registration import them individually."""
import torch

from .ops import CompositeFn, TerminationFn


def occupancy_activation(alpha, distances=None):
    return torch.sigmoid(alpha)  # UniSurf style, `distances` ignored (src/render_rays.py:3-7)


def composite(alpha, color, z_vals):
    """alpha (...,S), color (...,S,3), z (...,S) -> termination, depth, var (detached), rgb, opacity:
    src/render_rays.py:25-33,46-50 + src/loss.py:41-48 in one kernel (cnr_composite_fwd/bwd)."""
    return CompositeFn.apply(alpha, color, z_vals)


def occupancy_to_termination(occupancy, is_batch=False):
    """occ_i * exclusive-cumprod(1 - occ + 1e-10) along the last dim (src/render_rays.py:25-44);
    `is_batch` only selected the rank of the ones-column in the reference, the kernel is rank-free."""
    return TerminationFn.apply(occupancy)


def render(termination, vals, dim=-1):
    return (termination * vals).sum(dim=dim)


def render_loss(render, gt, loss="L1", normalise=False):
    residual = render - gt
    if loss == "L2":
        loss_mat = residual ** 2
    elif loss == "L1":
        loss_mat = torch.abs(residual)
    else:
        raise ValueError("loss type {} not implemented!".format(loss))
    if normalise:
        loss_mat = loss_mat / gt
    return loss_mat


class LossExplode(RuntimeError):
    """Raised where the reference calls exit(-1) (src/render_rays.py:87-89)."""


def reduce_batch_loss(loss_mat, var=None, avg=True, mask=None, loss_type="L1"):
    """Stand-alone form (host-syncing like the reference); the train step uses the fused
    cnr_loss_fwd_bwd kernel through loss.step_batch_loss instead."""
    mask_num = torch.sum(mask, dim=-1)
    if (mask_num == 0).any():
        loss = torch.zeros_like(loss_mat)
        return torch.mean(loss, dim=-1) if avg else loss
    if var is not None:
        eps = 1e-4
        information = 1.0 / (var + eps) if loss_type == "L2" else 1.0 / (torch.sqrt(var) + eps)
        loss_mat = loss_mat * information
    if not avg:
        return loss_mat
    if mask is None:
        return torch.mean(loss_mat, dim=-1).sum()
    loss = torch.sum(loss_mat, dim=-1) / (torch.sum(mask, dim=-1) + 1e-10)
    if (loss > 100000).any():
        raise LossExplode("loss explode")
    return loss
