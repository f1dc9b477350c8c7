"""The forward-only consumer of the hot path inside category registration: ``get_uncertainty_fields``
(src/category_registration.py:58-177) -- 100 x 100 probe rays on a sphere around each pretrained object, 96 jittered
samples per ray, UniDirsEmbed -> OccupancyMap -> sigmoid(10 sigma) -> non-batch occupancy_to_termination -> ray
entropies -> the per-object count of rays below a threshold that ranks the objects of a class.

Everything between the rays and the count runs on the device through the drop-in modules (cnr_pe_fwd, the dense
kernels, cnr_composite_fwd's termination); the reference moves the occupancies to the host and finishes in numpy.  The
rest of that file (point-cloud accumulation, TEASER++ / ICP alignment) is geometry pre-processing outside the hot path
 and is not here.
"""
import math
import os

import numpy as np
import torch

from . import embedding, model, render_rays
from .scene_cateogries import stratified_bins

N_PROBE = 100          # phi x theta grid (src/category_registration.py:96-98)
N_PROBE_BINS = 96      # samples per probe ray (:145)


def calculate_reliability(metric, eta=0.9, m1=0.1, m2=0.15, M1=0.57, M2=0.65):
    """src/utils.py:553-559, on tensors."""
    k = 2 * math.log(eta / (1 - eta))
    a_m, b_m = k / (m2 - m1), (m1 + m2) / 2
    a_M, b_M = k / (M2 - M1), (M1 + M2) / 2
    return 1 / (1 + torch.exp(a_m * (metric - b_m))) + 1 / (1 + torch.exp(-a_M * (metric - b_M)))


def probe_sphere():
    """Unit directions of the probe grid, (10000, 3), in the reference's order (:96-107: meshgrid of linspace(0, pi, 100)
    x linspace(0, 2 pi, 100), transposed, flattened)."""
    phi = torch.linspace(0, np.pi, N_PROBE)
    theta = torch.linspace(0, 2 * np.pi, N_PROBE)
    phi, theta = torch.meshgrid(phi, theta, indexing="ij")
    phi, theta = phi.t(), theta.t()
    return torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)], -1).reshape(-1, 3)


def uncertainty_probe(pe, fc_occ_map, center, r, device, z_vals=None, generator=None):
    """One object's probe (:131-156): rays from the sphere of radius ``r`` around ``center`` towards it, far = 2 r.
    -> (termination (N,96), entropies (N,), opacity (N,)) on ``device``.  ``z_vals`` (N,96) replaces the jittered bins
    (tests); ``generator`` seeds them."""
    r = float(r)
    rays_o_o = (r * probe_sphere()).to(device)
    viewdir = -rays_o_o / r
    rays_o = torch.as_tensor(center, dtype=torch.float32, device=device) + rays_o_o
    n = rays_o.shape[0]
    if z_vals is None:
        z_vals = stratified_bins(0, 2 * r, N_PROBE_BINS, n, device=device, z_fixed=True, generator=generator)
    xyz = rays_o[..., None, :] + viewdir[:, None, :] * z_vals.to(device)[..., None]
    with torch.no_grad():
        sigmas, _ = fc_occ_map(pe(xyz))
        occupancies = torch.sigmoid(10 * sigmas.squeeze(-1))
        term = render_rays.occupancy_to_termination(occupancies)
        entropies = torch.sum(-term * torch.log(term + 1e-10), dim=-1)
    return term, entropies, term.sum(-1)


def load_pretrained_fields(inst_dict, bbox3d_dict, pe_dict, fc_occ_map_dict, cfg):
    """The ``load_pretrained`` branch (:64-92): per object the newest file under ``cfg.weight_root/ckpt/<obj_id>/`` with the
    keys FC_state_dict, PE_state_dict, obj_scale, bbox -> an OccupancyMap(hidden_feature_size) + UniDirsEmbed on the data
    device."""
    emb1 = 21 * (3 + 1) + 3
    emb2 = 21 * (5 + 1) + 3 - emb1
    for cls_id, inst_dict_cls in inst_dict.items():
        if cls_id == 0:
            continue
        for d in (fc_occ_map_dict, pe_dict, bbox3d_dict):
            d.setdefault(cls_id, {})
        for obj_id in inst_dict_cls.keys():
            ckpt_dir = os.path.join(cfg.weight_root, "ckpt", str(obj_id))
            ckpt = torch.load(os.path.join(ckpt_dir, sorted(os.listdir(ckpt_dir))[-1]), map_location="cpu", weights_only=False)
            fc = model.OccupancyMap(emb1, emb2, hidden_size=cfg.hidden_feature_size)
            fc.load_state_dict(ckpt["FC_state_dict"])
            pe = embedding.UniDirsEmbed(max_deg=cfg.n_unidir_funcs, scale=ckpt["obj_scale"])
            pe.load_state_dict(ckpt["PE_state_dict"])
            fc_occ_map_dict[cls_id][obj_id] = fc.to(cfg.data_device)
            pe_dict[cls_id][obj_id] = pe.to(cfg.data_device)
            bbox3d_dict[cls_id][obj_id] = ckpt["bbox"]


def _points(pcs):
    return np.asarray(pcs.points if hasattr(pcs, "points") else pcs)


def get_uncertainty_fields(inst_dict, bbox3d_dict, count_dict, pe_dict, fc_occ_map_dict, cfg, name="replica",
                           load_pretrained=False, use_reliability=True, generator=None):
    """src/category_registration.py:58-177 with the reference's arguments: fills ``count_dict[cls_id][obj_id]`` with the
    number of probe rays whose (1 - reliability) is below 0.5 (``use_reliability``) or whose entropy is below 0.8 x the
    smallest per-object maximum.  ``inst_dict[cls_id][obj_id]['pcs']`` is a point cloud (anything with ``.points``, or an
    (n,3) array).  ``generator``: a device generator for the bin jitter (the reference draws from the global one)."""
    if load_pretrained:
        load_pretrained_fields(inst_dict, bbox3d_dict, pe_dict, fc_occ_map_dict, cfg)
    dev = torch.device(cfg.data_device)
    for cls_id in fc_occ_map_dict.keys():
        count_dict.setdefault(cls_id, {})
        # radius per object from its point cloud's half extents, at least 5 cm each (:115-123)
        bounds = []
        for obj_id in inst_dict[cls_id].keys():
            p = _points(inst_dict[cls_id][obj_id]["pcs"])
            bounds.append(torch.from_numpy((np.maximum(p.max(axis=0) - p.min(axis=0), 0.10) / 2).astype(np.float32)))
        rs = 1.2 * torch.sqrt(torch.square(torch.stack(bounds, dim=0)).sum(dim=-1))
        obj_ids = list(fc_occ_map_dict[cls_id].keys())
        ent_max, metrics = [], []
        for idx, obj_id in enumerate(obj_ids):
            p = _points(inst_dict[cls_id][obj_id]["pcs"])
            center = ((p.max(axis=0) + p.min(axis=0)) / 2 if name == "replica" else p.mean(axis=0)).astype(np.float32)
            term, entropies, opacity = uncertainty_probe(pe_dict[cls_id][obj_id], fc_occ_map_dict[cls_id][obj_id],
                                                         center, rs[idx], dev, generator=generator)
            ent_max.append(entropies.max())
            if use_reliability:
                heuristic = opacity * torch.exp(-0.5 * entropies)
                metrics.append(1 - calculate_reliability(heuristic, eta=0.9, m1=0.1, m2=0.15, M1=0.57, M2=0.65))
            else:
                metrics.append(entropies)
        threshold = 0.5 if use_reliability else 0.8 * torch.stack(ent_max).min()
        counts = torch.stack([(m < threshold).sum() for m in metrics]).cpu() if metrics else []   # ONE host sync per class
        for obj_id, n in zip(obj_ids, counts):
            count_dict[cls_id][obj_id] = int(n)
