"""Ensemble glue with the reference's signature (src/utils.py:24-28).

``update_vmap(models, optimiser)`` stacks the per-class modules' parameters into ``(C, ...)`` leaves,
registers them as a new param group, and returns ``(fmodel, params, buffers)`` such that
``functorch.vmap(fmodel)(params, buffers, *stacked_inputs)`` -- the exact call at train.py:154-155 --
lands in the class-batched HIP kernels through the Functions' ``vmap`` rules (ops.py).
"""
import torch
from torch.func import functional_call, stack_module_state


def combine_state_for_ensemble(models):
    """(fmodel, params, buffers) like functorch's helper: params/buffers are tuples of stacked
    tensors in ``named_parameters()`` / ``named_buffers()`` order."""
    params, buffers = stack_module_state(models)
    pnames, bnames = list(params.keys()), list(buffers.keys())
    base = models[0]

    def fmodel(p, b, *args, **kwargs):
        state = {n: t for n, t in zip(pnames, p)}
        state.update({n: t for n, t in zip(bnames, b)})
        return functional_call(base, state, args, kwargs)

    return fmodel, tuple(params[n] for n in pnames), tuple(buffers[n] for n in bnames)


def update_vmap(models, optimiser):
    fmodel, params, buffers = combine_state_for_ensemble(models)
    [p.requires_grad_() for p in params]
    optimiser.add_param_group({"params": params})
    return (fmodel, params, buffers)


# ---- sim3 <-> 8-vector [scale, qw, qx, qy, qz, tx, ty, tz] ------------------------------------------------------------
# The form the reference keeps per object in ``sceneCategory.object_tensor_dict`` and writes into checkpoints
# (src/utils.py:367-447 get_tensor_from_transform_sim3 / get_transform_from_tensor_sim3; consumers read element 0 as the
# scale and elements 1: as quaternion + translation, train.py:231-234, src/scene_cateogries.py:378-379).
def get_tensor_from_transform_sim3(RT):
    """(4,4) sim3 (numpy or tensor) -> float32 tensor (8,).  scale = det(R s)^(1/3); the quaternion comes from scipy's
    Rotation (w first), the same third-party routine the reference calls, so the numbers are identical."""
    import numpy as np
    from scipy.spatial.transform import Rotation
    M = np.array(RT.detach().cpu().numpy() if torch.is_tensor(RT) else RT, dtype=np.float64)
    scale = np.linalg.det(M[:3, :3]) ** (1.0 / 3.0)
    scale32 = torch.tensor([scale], dtype=torch.float32)
    Rm = M[:3, :3] / scale32.numpy()                      # the reference divides by the float32 scale tensor
    x, y, z, w = Rotation.from_matrix(Rm).as_quat()
    vec = np.concatenate([[w, x, y, z], M[:3, 3]])
    return torch.cat([scale32, torch.from_numpy(vec).float()], 0)


def get_transform_from_tensor_sim3(vec):
    """(8,) or (n,8) [scale, qw, qx, qy, qz, t] -> (4,4) / (n,4,4) sim3 matrices (differentiable torch ops)."""
    single = vec.dim() == 1
    v = vec[None] if single else vec
    s, q, t = v[:, 0], v[:, 1:5], v[:, 5:8]
    qr, qi, qj, qk = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    R = torch.stack([1 - two_s * (qj * qj + qk * qk), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
                     two_s * (qi * qj + qk * qr), 1 - two_s * (qi * qi + qk * qk), two_s * (qj * qk - qi * qr),
                     two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi * qi + qj * qj)],
                    -1).view(-1, 3, 3)
    T = torch.eye(4, device=v.device, dtype=v.dtype).repeat(v.shape[0], 1, 1)
    T[:, :3, :3] = R * s[:, None, None]
    T[:, :3, 3] = t
    return T[0] if single else T
