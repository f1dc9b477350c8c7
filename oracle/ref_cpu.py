"""CPU restatement of the reference hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Plain PyTorch fp32 on CPU, no functorch.  Every function cites the reference file:line it follows
(paths relative to the reference root).  Op structure deliberately mirrors the reference
(materialised 129-d embedding, per-ray latent layers, class-batched matmul) so that timing this
file is an honest CPU baseline.

Conventions: C = classes (the functorch vmap axis), R = rays per class, S = samples per ray,
L = latent dim.  Class-stacked parameters carry a leading C dim exactly as
``combine_state_for_ensemble`` produces them (src/utils.py:24-28).
"""
import math

import torch

# ----------------------------------------------------------------------------------------------
# parameter inventory of CodeNeRF (src/model.py:23-54), in nn.Module registration order
# ----------------------------------------------------------------------------------------------
CODENERF_LAYERS = [
    # name,                      in,            out
    ("encoding_xyz.0", "E1", "W"),
    ("shape_latent_layer_1.0", "L", "W"),
    ("shape_layer_1.0", "W", "W"),
    ("shape_latent_layer_2.0", "L", "W"),
    ("shape_layer_2.0", "W", "W"),
    ("cat_layer.0", "W+E1", "W"),
    ("cat_latent_layer.0", "L", "W"),
    ("encoding_shape", "W", "W"),
    ("sigma.0", "W", "1"),
    ("encoding_viewdir.0", "W+E2", "W"),
    ("texture_layer_1.0", "W", "W"),
    ("texture_latent_layer_1.0", "L", "W"),
    ("rgb.0", "W", "W/2"),
    ("rgb.2", "W/2", "3"),
]

EMB1 = 21 * (3 + 1) + 3  # 87, src/trainer.py:20
EMB2 = 21 * (5 + 1) + 3 - EMB1  # 42, src/trainer.py:21
EMB = EMB1 + EMB2  # 129

# icosahedral directions, src/embedding.py:51-73 (numerical constants of the method)
UNIDIRS = [
    0.8506508, 0, 0.5257311, 0.809017, 0.5, 0.309017, 0.5257311, 0.8506508, 0, 1, 0, 0,
    0.809017, 0.5, -0.309017, 0.8506508, 0, -0.5257311, 0.309017, 0.809017, -0.5,
    0, 0.5257311, -0.8506508, 0.5, 0.309017, -0.809017, 0, 1, 0, -0.5257311, 0.8506508, 0,
    -0.309017, 0.809017, -0.5, 0, 0.5257311, 0.8506508, -0.309017, 0.809017, 0.5,
    0.309017, 0.809017, 0.5, 0.5, 0.309017, 0.809017, 0.5, -0.309017, 0.809017, 0, 0, 1,
    -0.5, 0.309017, 0.809017, -0.809017, 0.5, 0.309017, -0.809017, 0.5, -0.309017,
]


def layer_dims(W=32, L=256):
    sym = {"E1": EMB1, "L": L, "W": W, "W+E1": W + EMB1, "W+E2": W + EMB2, "W/2": W // 2,
           "1": 1, "3": 3}
    return [(n, sym[i], sym[o]) for n, i, o in CODENERF_LAYERS]


def init_codenerf_params(C, W=32, L=256, generator=None):
    """xavier_normal_ weights (src/model.py:4-6, src/trainer.py:40) and PyTorch's default Linear
    bias U(+-1/sqrt(fan_in)).  Returns {name.weight|bias: (C, ...)}."""
    p = {}
    for name, fin, fout in layer_dims(W, L):
        std = math.sqrt(2.0 / (fin + fout))
        p[name + ".weight"] = torch.randn(C, fout, fin, generator=generator) * std
        bound = 1.0 / math.sqrt(fin)
        p[name + ".bias"] = (torch.rand(C, fout, generator=generator) * 2 - 1) * bound
    return p


def init_codes(n_obj, L, generator=None):
    """src/trainer.py:52-58: randn(d, L) / sqrt(L/2)."""
    return torch.randn(n_obj, L, generator=generator) / math.sqrt(L / 2)


# ----------------------------------------------------------------------------------------------
# a1  pinhole ray directions  (src/scene_cateogries.py:613-629)
# ----------------------------------------------------------------------------------------------
def get_rays_dirs(width, height, fx, fy, cx, cy):
    idx_w = torch.arange(end=width)
    idx_h = torch.arange(end=height)
    dirs = torch.ones((width, height, 3))
    dirs[:, :, 0] = ((idx_w - cx) / fx)[:, None]
    dirs[:, :, 1] = ((idx_h - cy) / fy)
    return dirs


# ----------------------------------------------------------------------------------------------
# a2  camera -> object / world ray transform  (src/scene_cateogries.py:24-47)
# ----------------------------------------------------------------------------------------------
def origin_dirs_O(T_CO, dirs_C):
    T_OC = torch.linalg.inv(T_CO)
    dirs_O = (T_OC[:, :3, :3] @ dirs_C[..., None]).squeeze(-1)
    return T_OC[:, :3, -1], dirs_O


def origin_dirs_W(T_WC, dirs_C):
    dirs_W = (T_WC[:, :3, :3] @ dirs_C[..., None]).squeeze(-1)
    return T_WC[:, :3, -1], dirs_W


# ----------------------------------------------------------------------------------------------
# a3 / a4  bin sampling with the random draws passed in  (src/scene_cateogries.py:51-96)
# ----------------------------------------------------------------------------------------------
def stratified_bins(min_depth, max_depth, n_bins, n_rays, u):
    """u: (n_rays, n_bins) uniform [0,1) draws (the reference calls torch.rand at :73-75)."""
    bin_limits_scale = torch.linspace(0, 1, n_bins + 1, dtype=torch.float32)
    if not torch.is_tensor(min_depth):
        min_depth = torch.ones(n_rays, dtype=torch.float32) * min_depth
    if not torch.is_tensor(max_depth):
        max_depth = torch.ones(n_rays, dtype=torch.float32) * max_depth
    depth_range = max_depth - min_depth
    lower = depth_range[..., None] * bin_limits_scale + min_depth[..., None]
    lower = lower[:, :-1]
    bin_len = depth_range / n_bins
    return lower + u * bin_len[..., None]


def normal_bins_sampling(depth, n_bins, delta, g):
    """g: (n_rays, n_bins) draws of N(0, (delta/3)^2) (reference :91 normal_(0, delta/3))."""
    bins = g.sort().values
    bins = torch.clip(bins, -delta, delta)
    return depth[:, None] + bins


# ----------------------------------------------------------------------------------------------
# a5  depth-guided sampling  (src/scene_cateogries.py:453-546)
# ----------------------------------------------------------------------------------------------
def sample_3d_points(rgbs, depth, origins, dirs, u, g, n1, n2, eps, stop_eps, min_bound=0.0):
    """rgbs (R,4) uint8 [r,g,b,state]; depth (R,); u (R,n1+n2) uniform draws; g (R,n2) normal
    draws already scaled to std eps/3.  Row r of u/g is the draw the reference would have consumed
    for ray r (the reference draws per masked subset; tests/golden/gen_golden.py shows the map)."""
    R = rgbs.shape[0]
    z = torch.zeros(R, n1 + n2, dtype=depth.dtype)
    invalid = (depth <= min_bound).view(-1)
    max_bound = torch.max(depth)
    if invalid.count_nonzero():
        z[invalid, :] = stratified_bins(min_bound, max_bound, n1 + n2, int(invalid.sum()), u[invalid])
    valid = ~invalid
    if valid.count_nonzero():
        z[valid, :n1] = stratified_bins(min_bound, depth[valid] - eps, n1, int(valid.sum()),
                                        u[valid][:, :n1])
        obj = (rgbs[..., -1] == 1).view(-1) & valid
        if obj.count_nonzero():
            z[obj, n1:] = normal_bins_sampling(depth[obj], n2, eps, g[obj])
        other = (rgbs[..., -1] != 1).view(-1) & valid
        if other.count_nonzero():
            z[other, n1:] = stratified_bins(depth[other] - eps, depth[other] + stop_eps, n2,
                                            int(other.sum()), u[other][:, n1:])
    pts = origins[..., None, :] + dirs[:, None, :] * z[..., None]
    return rgbs[..., :3], depth, valid, rgbs[..., -1].view(-1), pts, z


# ----------------------------------------------------------------------------------------------
# a8  UniDirsEmbed  (src/embedding.py:82-92), class-batched
# ----------------------------------------------------------------------------------------------
def unidirs_embed(x, B, scale, n_freqs=6):
    """x (C,R,S,3); B (C,21,3); scale float or () tensor -> (C,R,S,3+21*n_freqs)."""
    t = x / scale
    proj = torch.matmul(t, B.transpose(-1, -2)[:, None])  # (C,R,S,21)
    bands = 2.0 ** torch.linspace(0, n_freqs - 1, n_freqs)
    pb = proj[..., None, :] * bands[None, None, None, :, None]
    xb = pb.reshape(list(proj.shape[:-1]) + [-1])
    return torch.cat([t, torch.sin(xb * math.pi)], dim=-1)


# ----------------------------------------------------------------------------------------------
# a9  CodeNeRF forward  (src/model.py:56-84), class-batched
# ----------------------------------------------------------------------------------------------
def _lin(p, name, x):
    w, b = p[name + ".weight"], p[name + ".bias"]
    # x (C,R,S,in) ; w (C,out,in)
    return torch.matmul(x, w.transpose(-1, -2)[:, None]) + b[:, None, None, :]


def codenerf_forward(p, e, cs, ct):
    """e (C,R,S,129); cs, ct (C,R,1,L) -> sigmas (C,R,S,1), rgbs (C,R,S,3)."""
    e1, e2 = e[..., :EMB1], e[..., EMB1:]
    y = torch.relu(_lin(p, "encoding_xyz.0", e1))
    # block j=0
    y = y + torch.relu(_lin(p, "shape_latent_layer_1.0", cs))
    y = torch.relu(_lin(p, "shape_layer_1.0", y))
    # block j=1 with do_cat
    y = y + torch.relu(_lin(p, "cat_latent_layer.0", cs))
    y = torch.relu(_lin(p, "cat_layer.0", torch.cat((y, e1), dim=-1)))
    y = y + torch.relu(_lin(p, "shape_latent_layer_2.0", cs))
    y = torch.relu(_lin(p, "shape_layer_2.0", y))
    y = _lin(p, "encoding_shape", y)
    sigmas = _lin(p, "sigma.0", y) * 10.0
    y = torch.relu(_lin(p, "encoding_viewdir.0", torch.cat((y, e2), dim=-1)))
    y = y + torch.relu(_lin(p, "texture_latent_layer_1.0", ct))
    y = torch.relu(_lin(p, "texture_layer_1.0", y))
    y = torch.relu(_lin(p, "rgb.0", y))
    rgbs = torch.sigmoid(_lin(p, "rgb.2", y))
    return sigmas, rgbs


# ----------------------------------------------------------------------------------------------
# OccupancyMap forward (src/model.py:124-155) -- "next" row SURVEY §8(f).1; single model (no C)
# ----------------------------------------------------------------------------------------------
def occupancy_map_forward(p, e):
    def lin(n, x):
        return torch.nn.functional.linear(x, p[n + ".weight"], p[n + ".bias"])
    e1, e2 = e[..., :EMB1], e[..., EMB1:]
    fc1 = torch.relu(lin("in_layer.0", e1))
    fc2 = torch.relu(lin("mid1.0.0", fc1))
    fc3 = torch.relu(lin("cat_layer.0", torch.cat((fc2, e1), dim=-1)))
    fc4 = torch.relu(lin("mid2.0.0", fc3))
    alpha = lin("out_alpha", fc4) * 10.0
    fc5 = torch.relu(lin("color_linear.0", torch.cat((fc4, e2), dim=-1)))
    color = torch.sigmoid(lin("out_color", fc5))
    return alpha, color


# ----------------------------------------------------------------------------------------------
# a11-a13  composite  (src/render_rays.py:3-7, 25-33, 46-50)
# ----------------------------------------------------------------------------------------------
def occupancy_activation(alpha):
    return torch.sigmoid(alpha)


def occupancy_to_termination(occ):
    """occ (..., S): exclusive cumprod of (1 - occ + 1e-10) along S, T_0 = 1."""
    first = torch.ones(list(occ.shape[:-1]) + [1])
    free = (1.0 - occ + 1e-10)[..., :-1]
    free = torch.cat([first, free], dim=-1)
    return occ * torch.cumprod(free, dim=-1)


def render(termination, vals, dim=-1):
    return (termination * vals).sum(dim=dim)


def composite(alpha, color, z):
    """alpha (C,R,S), color (C,R,S,3), z (C,R,S) -> occ, term, depth, var, rgb, opacity
    (src/loss.py:41-48; var is detached there)."""
    occ = occupancy_activation(alpha)
    term = occupancy_to_termination(occ)
    depth = render(term, z)
    var = render(term, (z - depth[..., None]) ** 2).detach()
    rgb = render(term[..., None], color, dim=-2)
    opacity = term.sum(-1)
    return occ, term, depth, var, rgb, opacity


# ----------------------------------------------------------------------------------------------
# a14  render_loss / reduce_batch_loss  (src/render_rays.py:52-95)
# ----------------------------------------------------------------------------------------------
class LossExplode(RuntimeError):
    """The reference calls exit(-1) (src/render_rays.py:87-89); the restatement raises."""


def reduce_batch_loss(loss_mat, var=None, mask=None):
    mask_num = torch.sum(mask, dim=-1)
    if (mask_num == 0).any():
        return torch.mean(torch.zeros_like(loss_mat), dim=-1)
    if var is not None:
        loss_mat = loss_mat * (1.0 / (torch.sqrt(var) + 1e-4))
    loss = torch.sum(loss_mat, dim=-1) / (torch.sum(mask, dim=-1) + 1e-10)
    if (loss > 100000).any():
        raise LossExplode("loss explode")
    return loss


# ----------------------------------------------------------------------------------------------
# a15 / a16  loss assembly  (src/loss.py:5-74)
# ----------------------------------------------------------------------------------------------
def step_batch_loss(alpha, color, gt_depth, gt_color, sem_labels, mask_depth, z_vals,
                    color_scaling=5.0, opacity_scaling=10.0):
    mask_obj = (sem_labels != 0).detach()
    mask_sem = (sem_labels != 2).detach()
    alpha = alpha.squeeze(dim=-1)
    _, _, depth, var, rgb, opacity = composite(alpha, color, z_vals)
    l_d = torch.abs(depth - gt_depth) * (mask_depth & mask_obj)
    l_d = reduce_batch_loss(l_d, var=var, mask=mask_depth & mask_obj)
    l_c = torch.abs(rgb - gt_color).sum(-1) * mask_obj
    l_c = reduce_batch_loss(l_c, mask=mask_obj)
    l_o = torch.abs(opacity - mask_obj.float()) * mask_sem
    l_o = reduce_batch_loss(l_o, mask=mask_sem)
    l_batch = l_d + l_c * color_scaling + l_o * opacity_scaling
    return l_batch.sum(), {"depth": l_d, "color": l_c, "opacity": l_o}, l_c


def step_batch_loss_reg(shape_tables, texture_tables):
    """lists (one per class) of (n_obj, L) tables -> two (C,) vectors (src/loss.py:5-15)."""
    rs = torch.zeros(len(shape_tables))
    rt = torch.zeros(len(shape_tables))
    for i, (s, t) in enumerate(zip(shape_tables, texture_tables)):
        if s.shape[0] > 1:
            rs[i] = torch.norm(s, dim=-1).sum()
            rt[i] = torch.norm(t, dim=-1).sum()
    return rs, rt


# ----------------------------------------------------------------------------------------------
# one full object-branch train step  (train.py:123-184), class-batched, for parity + CPU baseline
# ----------------------------------------------------------------------------------------------
def forward_loss(mlp, pe_B, scale, shape_tables, texture_tables, batch, reg_scaling=0.0005):
    """batch: dict with pts (C,R,S,3), z (C,R,S), gt_depth (C,R), gt_rgb (C,R,3) in [0,1],
    labels (C,R) uint8, depth_mask (C,R) bool, indices (C,R) int64.
    Returns (loss, aux) where aux holds every intermediate the fixtures pin."""
    cs = torch.stack([shape_tables[c][batch["indices"][c]][:, None, :]
                      for c in range(len(shape_tables))])
    ct = torch.stack([texture_tables[c][batch["indices"][c]][:, None, :]
                      for c in range(len(texture_tables))])
    e = unidirs_embed(batch["pts"], pe_B, scale)
    sig, col = codenerf_forward(mlp, e, cs, ct)
    loss, ld, lcol = step_batch_loss(sig, col, batch["gt_depth"], batch["gt_rgb"], batch["labels"],
                                     batch["depth_mask"], batch["z"])
    rs, rt = step_batch_loss_reg(shape_tables, texture_tables)
    loss = loss + reg_scaling * (rs + rt).sum()
    occ, term, depth, var, rgb, opacity = composite(sig.squeeze(-1), col, batch["z"])
    aux = dict(emb=e, sigmas=sig, rgbs=col, occ=occ, term=term, depth=depth, var=var, rgb=rgb,
               opacity=opacity, loss_depth=ld["depth"], loss_color=ld["color"],
               loss_opacity=ld["opacity"], reg_shape=rs, reg_texture=rt)
    return loss, aux


# ----------------------------------------------------------------------------------------------
# forward-only consumer: the uncertainty probe of category registration
# ----------------------------------------------------------------------------------------------
def calculate_reliability(metric, eta=0.9, m1=0.1, m2=0.15, M1=0.57, M2=0.65):
    """src/utils.py:553-559."""
    alpha_m = 2 * math.log(eta / (1 - eta)) / (m2 - m1)
    beta_m = (m1 + m2) / 2
    alpha_M = 2 * math.log(eta / (1 - eta)) / (M2 - M1)
    beta_M = (M1 + M2) / 2
    return 1 / (1 + torch.exp(alpha_m * (metric - beta_m))) + 1 / (1 + torch.exp(-alpha_M * (metric - beta_M)))


def uncertainty_probe(p, pe_B, scale, center, r, u):
    """src/category_registration.py:96-107,131-159 for ONE pretrained object: p = OccupancyMap state dict (one model, no class
    dim), pe_B (1,21,3), center (3,), r scalar, u (10000, 96) the jitter draws of stratified_bins.  Returns termination
    (10000,96), ray entropies, and 1 - reliability of opacity * exp(-entropy / 2)."""
    phi = torch.linspace(0, math.pi, 100)
    theta = torch.linspace(0, 2 * math.pi, 100)
    phi, theta = torch.meshgrid(phi, theta, indexing="ij")
    phi, theta = phi.t(), theta.t()
    x = r * torch.sin(phi) * torch.cos(theta)
    y = r * torch.sin(phi) * torch.sin(theta)
    zc = r * torch.cos(phi)
    rays_o_o = torch.stack([x, y, zc], dim=-1).reshape(-1, 3)
    viewdir = -rays_o_o / r
    rays_o = center + rays_o_o
    z_vals = stratified_bins(0.0, 2 * r, 96, rays_o.shape[0], u)
    xyz = rays_o[..., None, :] + viewdir[:, None, :] * z_vals[..., None]
    sigmas, _ = occupancy_map_forward(p, unidirs_embed(xyz[None], pe_B, scale)[0])
    occ = torch.sigmoid(10 * sigmas.squeeze(-1))
    term = occupancy_to_termination(occ)
    ent = torch.sum(-term * torch.log(term + 1e-10), dim=-1)
    heuristic = term.sum(-1) * torch.exp(-0.5 * ent)
    return term, ent, 1 - calculate_reliability(heuristic)
