"""TEST INFRASTRUCTURE: a LEARNABLE synthetic scene for convergence tests -- analytic spheres with a smooth colour
field, rays rendered from it -- in the ray-pool format of scene_cateogries.synthetic_pool (src/scene_cateogries.py:
211-249 row layout: rgbs (N,4) u8 [r,g,b,state], depth (N,), dirs (N,3), T_co (N,4,4), indices (N,)).

Object k of a category is a sphere of radius r_k at the origin of its own frame with colour
0.5 + 0.4 sin(2.5 p + phase_k) at surface point p.  A ray group shares a camera pose T_wc and an object sim3 T_wo
(scale, rotation, translation); T_co = inv(T_wc) T_wo exactly as the reference forms it (:237-238).  Pixels are chosen so
that about two thirds of the rays hit the sphere (state 1 = this object, z-depth of the hit, surface colour); the
others get state 0 (another object / background) with either no depth or the depth of a plane behind the object.
"""
import math

import torch


def _rot(gen, n):
    q = torch.randn(n, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(n, 3, 3)


def sphere_radius(k):
    return 0.35 + 0.08 * k


def surface_colour(p, k):
    ph = torch.tensor([0.3 + 0.9 * k, 1.1 + 0.5 * k, 2.0 - 0.4 * k])
    return 0.5 + 0.4 * torch.sin(2.5 * p + ph)


def analytic_pool(n_rays, n_obj, generator, group=64):
    g = generator
    N = n_rays
    ng = (N + group - 1) // group
    grp = torch.arange(N) // group
    obj = torch.randint(0, n_obj, (ng,), generator=g)
    # object sim3 (object -> world) and a camera looking at the object from 2 .. 3 world units away
    s = 0.6 + 0.4 * torch.rand(ng, generator=g)
    T_wo = torch.eye(4).repeat(ng, 1, 1)
    T_wo[:, :3, :3] = _rot(g, ng) * s[:, None, None]
    T_wo[:, :3, 3] = torch.rand(ng, 3, generator=g) * 2 - 1
    view = torch.randn(ng, 3, generator=g)
    view = view / view.norm(dim=-1, keepdim=True)
    dist = 2.0 + torch.rand(ng, generator=g)
    cam = T_wo[:, :3, 3] + view * dist[:, None]
    zc = -view                                              # camera z axis: towards the object
    up = torch.randn(ng, 3, generator=g)
    xc = torch.linalg.cross(up, zc)
    xc = xc / xc.norm(dim=-1, keepdim=True)
    yc = torch.linalg.cross(zc, xc)
    T_wc = torch.eye(4).repeat(ng, 1, 1)
    T_wc[:, :3, 0], T_wc[:, :3, 1], T_wc[:, :3, 2], T_wc[:, :3, 3] = xc, yc, zc, cam
    T_co = torch.linalg.inv(T_wc) @ T_wo
    # pixel directions: through a point of a disc (object frame) of 1.25 x the sphere radius around the centre
    rad = torch.tensor([sphere_radius(int(k)) for k in obj])[grp]
    a = torch.rand(N, generator=g) * 2 * math.pi
    rr = 1.25 * rad * torch.sqrt(torch.rand(N, generator=g))
    T_oc = torch.linalg.inv(T_co)[grp]
    ex, ey = T_oc[:, :3, 0], T_oc[:, :3, 1]                 # camera x / y axes in the object frame (scaled)
    ex, ey = ex / ex.norm(dim=-1, keepdim=True), ey / ey.norm(dim=-1, keepdim=True)
    target = ex * (rr * torch.cos(a))[:, None] + ey * (rr * torch.sin(a))[:, None]
    tc = (T_co[grp][:, :3, :3] @ target[..., None]).squeeze(-1) + T_co[grp][:, :3, 3]
    dirs = torch.stack([tc[:, 0] / tc[:, 2], tc[:, 1] / tc[:, 2], torch.ones(N)], -1)
    # the ray in the object frame exactly as origin_dirs_O forms it; pts = o + d z, z = z-depth
    o = T_oc[:, :3, 3]
    d = (T_oc[:, :3, :3] @ dirs[..., None]).squeeze(-1)
    A = (d * d).sum(-1)
    Bq = 2 * (o * d).sum(-1)
    Cq = (o * o).sum(-1) - rad * rad
    disc = Bq * Bq - 4 * A * Cq
    hit = disc > 0
    t = (-Bq - torch.sqrt(disc.clamp_min(0))) / (2 * A)
    hit &= t > 0.2
    p = o + d * t[:, None]
    k_ray = obj[grp]
    ph = torch.stack([0.3 + 0.9 * k_ray, 1.1 + 0.5 * k_ray, 2.0 - 0.4 * k_ray], -1).float()
    col = 0.5 + 0.4 * torch.sin(2.5 * p + ph)
    # rays that miss: half without depth, half on a plane one unit behind the object centre
    t_centre = -(o * d).sum(-1) / A
    behind = t_centre + 1.0 / torch.sqrt(A)
    no_depth = torch.rand(N, generator=g) < 0.5
    depth = torch.where(hit, t, torch.where(no_depth, torch.zeros(N), behind))
    state = torch.where(hit, 1, 0).to(torch.uint8)
    bg_col = torch.rand(N, 3, generator=g)
    rgb = torch.where(hit[:, None], col, bg_col)
    rgbs = torch.cat([(rgb * 255).round().clamp(0, 255).to(torch.uint8), state[:, None]], -1)
    return dict(rgbs=rgbs.contiguous(), depth=depth.float().contiguous(), dirs=dirs.contiguous(),
                T_co=T_co[grp].contiguous(), T_wc=T_wc[grp].contiguous(), indices=k_ray.contiguous())
