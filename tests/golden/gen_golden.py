#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE (build container only).

Imports the reference's own modules from /root/reference/src (read-only; the five viz/IO packages
it imports at module scope but never touches on this path are replaced by MagicMock), drives them
exactly as train.py:123-184 does (functorch combine_state_for_ensemble + vmap, step_batch_loss,
backward, AdamW), asserts that oracle/ref_cpu.py reproduces every tensor, and writes
tests/golden/<case>.npz.  Only arrays are written -- no reference source or bytecode.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

The GPU box has no /root/reference; tests read only the .npz files.
"""
import math
import os
import sys
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True
for _m in ["skimage", "skimage.measure", "cv2", "imgviz", "open3d", "trimesh", "plotly",
           "plotly.graph_objs", "plotly.subplots"]:
    sys.modules[_m] = MagicMock()

import loss as ref_loss  # noqa: E402
import render_rays as ref_rr  # noqa: E402
import scene_cateogries as ref_sc  # noqa: E402
import trainer as ref_trainer  # noqa: E402
from functorch import combine_state_for_ensemble, vmap  # noqa: E402

from oracle import ref_cpu as O  # noqa: E402

N1_N2 = {16: (2, 14), 32: (4, 28), 10: (1, 9)}


def rand_pose(gen, sim3):
    q = torch.randn(4, generator=gen)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    Rm = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = torch.eye(4)
    s = (0.3 + 0.7 * torch.rand(1, generator=gen)).item() if sim3 else 1.0
    T[:3, :3] = Rm * s
    T[:3, 3] = torch.rand(3, generator=gen) * 2 - 1
    return T


def make_pool(gen, R, n_obj, all_invalid=False, labels_all=None):
    """Synthetic pool slice per SURVEY §8(d)."""
    u = torch.randint(0, 1200, (R,), generator=gen).float()
    v = torch.randint(0, 680, (R,), generator=gen).float()
    dirs = torch.stack([(u - 599.5) / 600.0, (v - 339.5) / 600.0, torch.ones(R)], -1)
    T = torch.empty(R, 4, 4)
    for g0 in range(0, R, 16):
        T_wc = rand_pose(gen, False)
        T_wo = rand_pose(gen, True)
        T[g0:g0 + 16] = torch.linalg.inv(T_wc) @ T_wo
    depth = 0.5 + 3.0 * torch.rand(R, generator=gen)
    depth[torch.rand(R, generator=gen) < 0.05] = 0.0
    if all_invalid:
        depth[: R // 2] = 0.0
    pr = torch.rand(R, generator=gen)
    state = torch.where(pr < 0.70, 1, torch.where(pr < 0.95, 0, 2)).to(torch.uint8)
    if labels_all is not None:
        state[:] = labels_all
    rgb = torch.randint(0, 256, (R, 3), generator=gen, dtype=torch.uint8)
    rgbs = torch.cat([rgb, state[:, None]], -1)
    idx = torch.randint(0, n_obj, (R,), generator=gen)
    return rgbs, depth, dirs, T, idx


def run_case(name, seed, C, R, S, L, n_obj=4, single_obj=False, empty_mask_cls=None,
             all_invalid=False, keep_emb=True):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(1000 + seed)
    n1, n2 = N1_N2[S]
    eps, stop_eps = 0.1, 0.05
    scale = 2.0 if L == 256 else 3.0
    if single_obj:
        n_obj = 1
    cfg = SimpleNamespace(training_device="cpu", obj_scale=scale, n_unidir_funcs=5,
                          net_hyperparams=dict(shape_blocks=2, texture_blocks=1, W=32, latent_dim=L),
                          hidden_feature_size=32)
    trainers = [ref_trainer.Trainer(cfg, cls_id=c + 1, inst_ids=list(range(n_obj))) for c in range(C)]
    # perturb B so its gradient/updates are exercised away from the symmetric init
    for t in trainers:
        with torch.no_grad():
            t.pe.B_layer.weight.add_(0.01 * torch.randn(21, 3, generator=gen))

    opt = torch.optim.AdamW([torch.autograd.Variable(torch.tensor(0))], lr=1e-3, weight_decay=0.013)
    for t in trainers:
        opt.add_param_group({"params": t.shape_codes.parameters(), "lr": 1e-3, "weight_decay": 0.013})
        opt.add_param_group({"params": t.texture_codes.parameters(), "lr": 1e-3, "weight_decay": 0.013})
    fc_model, fc_param, fc_buffer = combine_state_for_ensemble([t.fc_occ_map for t in trainers])
    [p.requires_grad_() for p in fc_param]
    opt.add_param_group({"params": fc_param})
    pe_model, pe_param, pe_buffer = combine_state_for_ensemble([t.pe for t in trainers])
    [p.requires_grad_() for p in pe_param]
    opt.add_param_group({"params": pe_param})

    names = [n for n, _ in trainers[0].fc_occ_map.named_parameters()]
    mlp0 = {n: p.detach().clone() for n, p in zip(names, fc_param)}
    B0 = pe_param[0].detach().clone()
    shape0 = torch.stack([t.shape_codes.weight.detach().clone() for t in trainers])
    tex0 = torch.stack([t.texture_codes.weight.detach().clone() for t in trainers])

    # ---- sampling via the reference, recording the draws per ray -------------------------------
    pool = [make_pool(gen, R, n_obj, all_invalid=all_invalid and c == 0,
                      labels_all=(0 if empty_mask_cls == c else None)) for c in range(C)]
    ns = SimpleNamespace(n_bins_cam2surface=n1, n_bins=n2, surface_eps=eps, stop_eps=stop_eps,
                         data_device="cpu", min_bound=0.0, this_obj=1)
    outs, us, gs, origins_l, dirs_l = [], [], [], [], []
    for c in range(C):
        rgbs, depth, dirs, T, idx = pool[c]
        fn = ref_sc.origin_dirs_W if single_obj else ref_sc.origin_dirs_O
        origins, dirs_o = fn(T, dirs)
        o2, d2 = (O.origin_dirs_W if single_obj else O.origin_dirs_O)(T, dirs)
        assert torch.equal(origins, o2) and torch.equal(dirs_o, d2)
        state = torch.get_rng_state()
        ref_out = ref_sc.sceneCategory.sample_3d_points(ns, rgbs, depth, origins, dirs_o)
        # replay the draws in the reference's order (scene_cateogries.py:489-539) into per-ray rows
        torch.set_rng_state(state)
        u = torch.zeros(R, S)
        g = torch.zeros(R, n2)
        invalid = depth <= 0.0
        valid = ~invalid
        if invalid.any():
            u[invalid] = torch.rand(int(invalid.sum()), S)
        if valid.any():
            u[valid, :n1] = torch.rand(int(valid.sum()), n1)
            obj = (rgbs[:, 3] == 1) & valid
            if obj.any():
                g[obj] = torch.empty(int(obj.sum()), n2).normal_(mean=0.0, std=eps / 3.0)
            oth = (rgbs[:, 3] != 1) & valid
            if oth.any():
                u[oth, n1:] = torch.rand(int(oth.sum()), n2)
        mine = O.sample_3d_points(rgbs, depth, origins, dirs_o, u, g, n1, n2, eps, stop_eps)
        for a, b in zip(ref_out, mine):
            assert torch.equal(a, b), "oracle sampling != reference"
        outs.append(ref_out)
        us.append(u)
        gs.append(g)
        origins_l.append(origins)
        dirs_l.append(dirs_o)

    gt_rgb = torch.stack([o[0] for o in outs]) / 255.0
    gt_depth = torch.stack([o[1] for o in outs])
    depth_mask = torch.stack([o[2] for o in outs])
    labels = torch.stack([o[3] for o in outs])
    pts = torch.stack([o[4] for o in outs])
    z = torch.stack([o[5] for o in outs])
    indices = torch.stack([p[4] for p in pool])

    # ---- reference train step (train.py:136-184) -----------------------------------------------
    cs = torch.stack([trainers[c].shape_codes(indices[c])[:, None, :] for c in range(C)])
    ct = torch.stack([trainers[c].texture_codes(indices[c])[:, None, :] for c in range(C)])
    emb = vmap(pe_model)(pe_param, pe_buffer, pts)
    alpha, color = vmap(fc_model)(fc_param, fc_buffer, emb, cs, ct)
    loss, ld, lcol = ref_loss.step_batch_loss(alpha, color, gt_depth, gt_rgb, labels, depth_mask, z)
    cls_dict = {c: SimpleNamespace(trainer=trainers[c], training_device="cpu",
                                   obj_ids=list(range(n_obj))) for c in range(C)}
    rs, rt = ref_loss.step_batch_loss_reg(cls_dict, torch.arange(C))
    loss = loss + 0.0005 * (rs + rt).sum()
    loss.backward()

    occ = ref_rr.occupancy_activation(alpha.squeeze(-1))
    term = ref_rr.occupancy_to_termination(occ, is_batch=True)
    depth_r = ref_rr.render(term, z)
    var_r = ref_rr.render(term, (z - depth_r[..., None]) ** 2)
    rgb_r = ref_rr.render(term[..., None], color, dim=-2)
    opa_r = term.sum(-1)

    zero = lambda p: torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()
    grads = {n: zero(p) for n, p in zip(names, fc_param)}
    gB = zero(pe_param[0])
    gshape = torch.stack([zero(t.shape_codes.weight) for t in trainers])
    gtex = torch.stack([zero(t.texture_codes.weight) for t in trainers])
    opt.step()

    # ---- oracle must reproduce everything ------------------------------------------------------
    mlp_o = {n: v.clone().requires_grad_() for n, v in mlp0.items()}
    B_o = B0.clone().requires_grad_()
    sh_o = [shape0[c].clone().requires_grad_() for c in range(C)]
    tx_o = [tex0[c].clone().requires_grad_() for c in range(C)]
    batch = dict(pts=pts, z=z, gt_depth=gt_depth, gt_rgb=gt_rgb, labels=labels,
                 depth_mask=depth_mask, indices=indices)
    loss_o, aux = O.forward_loss(mlp_o, B_o, scale, sh_o, tx_o, batch)
    loss_o.backward()

    def close(a, b, what, tol=2e-6):
        a, b = a.detach().double(), b.detach().double()
        err = (a - b).norm() / max(b.norm().item(), 1e-30) if b.norm() > 0 else (a - b).abs().max()
        assert err <= tol, f"{name}: oracle {what} off by {err:.3e}"

    close(aux["emb"], emb, "emb")
    close(aux["sigmas"], alpha, "sigmas")
    close(aux["rgbs"], color, "rgbs")
    close(aux["term"], term, "term")
    close(aux["depth"], depth_r, "depth")
    close(aux["var"], var_r, "var")
    close(aux["rgb"], rgb_r, "rgb")
    close(aux["opacity"], opa_r, "opacity")
    close(loss_o, loss, "loss")
    for k in ("depth", "color", "opacity"):
        close(aux["loss_" + k], ld[k], "loss_" + k)
    for n in names:
        close(zero(mlp_o[n]), grads[n], "grad " + n, 2e-5)
    close(zero(B_o), gB, "grad B", 2e-5)
    close(torch.stack([zero(s) for s in sh_o]), gshape, "grad shape", 2e-5)
    close(torch.stack([zero(s) for s in tx_o]), gtex, "grad tex", 2e-5)

    # ---- write fixture -------------------------------------------------------------------------
    f = lambda t: t.detach().cpu().numpy()
    d = dict(meta=np.array([C, R, S, L, n_obj, n1, n2, int(single_obj)], dtype=np.int64),
             scale=np.float32(scale), eps=np.float32(eps), stop_eps=np.float32(stop_eps),
             pool_rgbs=f(torch.stack([p[0] for p in pool])), pool_depth=f(torch.stack([p[1] for p in pool])),
             pool_dirs=f(torch.stack([p[2] for p in pool])), pool_T=f(torch.stack([p[3] for p in pool])),
             indices=f(indices), u=f(torch.stack(us)), g=f(torch.stack(gs)),
             origins=f(torch.stack(origins_l)), dirs_o=f(torch.stack(dirs_l)),
             z=f(z), pts=f(pts), gt_rgb=f(gt_rgb), gt_depth=f(gt_depth), depth_mask=f(depth_mask),
             labels=f(labels), B=f(B0), shape_codes=f(shape0), texture_codes=f(tex0),
             sigmas=f(alpha), rgbs=f(color), occ=f(occ), term=f(term), depth=f(depth_r), var=f(var_r),
             rgb=f(rgb_r), opacity=f(opa_r), loss=f(loss), loss_depth=f(ld["depth"]),
             loss_color=f(ld["color"]), loss_opacity=f(ld["opacity"]), reg_shape=f(rs),
             reg_texture=f(rt), grad_B=f(gB), grad_shape_codes=f(gshape), grad_texture_codes=f(gtex),
             new_B=f(pe_param[0]),
             new_shape_codes=f(torch.stack([t.shape_codes.weight for t in trainers])),
             new_texture_codes=f(torch.stack([t.texture_codes.weight for t in trainers])))
    if keep_emb:
        d["emb"] = f(emb)
    for n, p in zip(names, fc_param):
        d["mlp." + n] = f(mlp0[n])
        d["grad." + n] = f(grads[n])
        d["new." + n] = f(p)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: loss={loss.item():.6f}  -> {os.path.getsize(path) / 1e6:.2f} MB")


def run_bg_case(name, seed, R, hidden=128, n1=5, n2=9):
    """Background branch of one train step (train.py:113-121,172-184): Trainer(cls_id=0) -> OccupancyMap(hidden)
    + UniDirsEmbed, world-frame rays (origin_dirs_W), sample_3d_points, PE -> OccupancyMap -> step_batch_loss with a
    class dim of 1, backward, AdamW.  Same fixture layout as run_case where the fields coincide."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(2000 + seed)
    S, eps, stop_eps, scale = n1 + n2, 0.1, 0.05, 10.0
    cfg = SimpleNamespace(training_device="cpu", obj_scale=scale, n_unidir_funcs=5, hidden_feature_size=hidden)
    tr = ref_trainer.Trainer(cfg, cls_id=0, inst_ids=[0])
    with torch.no_grad():
        tr.pe.B_layer.weight.add_(0.01 * torch.randn(21, 3, generator=gen))
    opt = torch.optim.AdamW([torch.autograd.Variable(torch.tensor(0))], lr=1e-3, weight_decay=0.013)
    opt.add_param_group({"params": tr.fc_occ_map.parameters(), "lr": 1e-3, "weight_decay": 0.013})
    opt.add_param_group({"params": tr.pe.parameters(), "lr": 1e-3, "weight_decay": 0.013})
    names = [n for n, _ in tr.fc_occ_map.named_parameters()]
    mlp0 = {n: p.detach().clone() for n, p in tr.fc_occ_map.named_parameters()}
    B0 = tr.pe.B_layer.weight.detach().clone()

    rgbs, depth, dirs, _, _ = make_pool(gen, R, 1)
    depth = depth * 2.0                                  # room-scale depths
    T = torch.stack([rand_pose(gen, False) for _ in range(R // 16 + 1)]).repeat_interleave(16, 0)[:R]   # T_wc
    origins, dirs_o = ref_sc.origin_dirs_W(T, dirs)
    o2, d2 = O.origin_dirs_W(T, dirs)
    assert torch.equal(origins, o2) and torch.equal(dirs_o, d2)
    ns = SimpleNamespace(n_bins_cam2surface=n1, n_bins=n2, surface_eps=eps, stop_eps=stop_eps,
                         data_device="cpu", min_bound=0.0, this_obj=1)
    state = torch.get_rng_state()
    ref_out = ref_sc.sceneCategory.sample_3d_points(ns, rgbs, depth, origins, dirs_o)
    torch.set_rng_state(state)
    u, g = torch.zeros(R, S), torch.zeros(R, n2)
    invalid = depth <= 0.0
    valid = ~invalid
    if invalid.any():
        u[invalid] = torch.rand(int(invalid.sum()), S)
    if valid.any():
        u[valid, :n1] = torch.rand(int(valid.sum()), n1)
        obj = (rgbs[:, 3] == 1) & valid
        if obj.any():
            g[obj] = torch.empty(int(obj.sum()), n2).normal_(mean=0.0, std=eps / 3.0)
        oth = (rgbs[:, 3] != 1) & valid
        if oth.any():
            u[oth, n1:] = torch.rand(int(oth.sum()), n2)
    mine = O.sample_3d_points(rgbs, depth, origins, dirs_o, u, g, n1, n2, eps, stop_eps)
    for a, b in zip(ref_out, mine):
        assert torch.equal(a, b), "oracle sampling != reference (bg)"
    gt_rgb, gt_depth, depth_mask, labels, pts, z = ref_out
    gt_rgb = gt_rgb / 255.0

    emb = tr.pe(pts)
    alpha, color = tr.fc_occ_map(emb)
    loss, ld, _ = ref_loss.step_batch_loss(alpha[None], color[None], gt_depth[None], gt_rgb[None], labels[None],
                                           depth_mask[None], z[None])
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in tr.fc_occ_map.named_parameters()}
    gB = tr.pe.B_layer.weight.grad.detach().clone()
    opt.step()

    # ---- the oracle must reproduce it ----------------------------------------------------------
    mlp_o = {n: v.clone().requires_grad_() for n, v in mlp0.items()}
    B_o = B0.clone().requires_grad_()
    emb_o = O.unidirs_embed(pts[None], B_o[None], scale)[0]
    a_o, c_o = O.occupancy_map_forward(mlp_o, emb_o)
    loss_o, aux, _ = O.step_batch_loss(a_o[None], c_o[None], gt_depth[None], gt_rgb[None], labels[None],
                                       depth_mask[None], z[None])
    loss_o.backward()

    def close(a, b, what, tol=2e-6):
        a, b = a.detach().double(), b.detach().double()
        err = (a - b).norm() / max(b.norm().item(), 1e-30)
        assert err <= tol, f"{name}: oracle {what} off by {err:.3e}"
    close(emb_o, emb, "emb"); close(a_o, alpha, "alpha"); close(c_o, color, "color"); close(loss_o, loss, "loss")
    for n in names:
        close(mlp_o[n].grad, grads[n], "grad " + n, 2e-5)
    close(B_o.grad, gB, "grad B", 2e-5)

    f = lambda t: t.detach().cpu().numpy()
    d = dict(meta=np.array([1, R, S, hidden, 1, n1, n2, 1], dtype=np.int64), scale=np.float32(scale),
             eps=np.float32(eps), stop_eps=np.float32(stop_eps), pool_rgbs=f(rgbs[None]), pool_depth=f(depth[None]),
             pool_dirs=f(dirs[None]), pool_T=f(T[None]), u=f(u[None]), g=f(g[None]), z=f(z[None]), pts=f(pts[None]),
             gt_rgb=f(gt_rgb[None]), gt_depth=f(gt_depth[None]), depth_mask=f(depth_mask[None]), labels=f(labels[None]),
             B=f(B0[None]), sigmas=f(alpha[None]), rgbs=f(color[None]), loss=f(loss), loss_depth=f(ld["depth"]),
             loss_color=f(ld["color"]), loss_opacity=f(ld["opacity"]), grad_B=f(gB[None]),
             new_B=f(tr.pe.B_layer.weight[None]))
    for n, p in tr.fc_occ_map.named_parameters():
        d["mlp." + n] = f(mlp0[n]); d["grad." + n] = f(grads[n]); d["new." + n] = f(p)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: loss={loss.item():.6f}  -> {os.path.getsize(path) / 1e6:.2f} MB")


def run_pool_case(name, seed, W=24, H=20, n_frames=6):
    """Ray-pool construction from frames (scene_cateogries.py:107-350) by the reference's own
    sceneCategory.__init__: one object category with three instances and the background, on small synthetic frames.
    Stored: the frames / metadata and the pools the reference builds (np.random.seed pins its shuffle)."""
    rng = np.random.RandomState(seed)
    gen = torch.Generator().manual_seed(3000 + seed)
    cfg = SimpleNamespace(data_device="cpu", training_device="cpu", bg_scale=5.0, obj_scale=2.0,
                          hidden_feature_size=32, hidden_feature_size_bg=32, n_bins_cam2surface=1,
                          n_bins_cam2surface_bg=5, n_bins=9, min_depth=0.0, max_depth=10.0, surface_eps=0.1,
                          stop_eps=0.05, n_unidir_funcs=5,
                          net_hyperparams=dict(shape_blocks=2, texture_blocks=1, W=32, latent_dim=32))
    inst_ids = [3, 7, 12]
    sample_dict = {}
    for f in range(n_frames):
        mask = rng.choice([-1, 0] + inst_ids, size=(W, H), p=[0.05, 0.35, 0.2, 0.2, 0.2]).astype(np.int64)
        sample_dict[10 + f] = dict(image=rng.randint(0, 256, size=(W, H, 3)).astype(np.uint8),
                                   depth=(rng.rand(W, H) * 4).astype(np.float32),
                                   T=rand_pose(gen, False).numpy().astype(np.float32), obj_mask=mask)
    rays_dir = torch.randn(W, H, 3, generator=gen)

    def bbox():
        w0, h0 = rng.randint(0, W - 6), rng.randint(0, H - 6)
        return [int(w0), int(w0 + rng.randint(2, 6)), int(h0), int(h0 + rng.randint(2, 6))]
    inst_dict = {}
    for k, iid in enumerate(inst_ids):
        frames = sorted(rng.choice(n_frames, size=3, replace=False).tolist())
        inst_dict[iid] = dict(T_obj=rand_pose(gen, True).numpy().astype(np.float64),
                              bbox3D=SimpleNamespace(extent=np.array([1.0, 2.0, 1.5])),
                              frame_info=[dict(frame=10 + f, bbox=bbox()) for f in frames])
    bg_dict = dict(bbox3D=SimpleNamespace(extent=np.array([6.0, 6.0, 3.0])),
                   frame_info=[dict(frame=10 + f, bbox=[0, W, 0, H]) for f in range(0, n_frames, 2)])

    import copy as _copy
    np.random.seed(777 + seed)
    sc = ref_sc.sceneCategory(cfg, 5, _copy.deepcopy(inst_dict), sample_dict, rays_dir)
    np.random.seed(888 + seed)
    bg = ref_sc.sceneCategory(cfg, 0, _copy.deepcopy(bg_dict), sample_dict, rays_dir)

    f = lambda t: t.detach().cpu().numpy()
    frames = sorted(sample_dict.keys())
    d = dict(meta=np.array([W, H, n_frames, len(inst_ids), seed], dtype=np.int64),
             frame_ids=np.array(frames, dtype=np.int64),
             images=np.stack([sample_dict[k]["image"] for k in frames]),
             depths=np.stack([sample_dict[k]["depth"] for k in frames]),
             masks=np.stack([sample_dict[k]["obj_mask"] for k in frames]),
             T_wc=np.stack([sample_dict[k]["T"] for k in frames]), rays_dir=f(rays_dir),
             inst_ids=np.array(inst_ids, dtype=np.int64),
             T_obj=np.stack([inst_dict[i]["T_obj"] for i in inst_ids]),
             obj_frames=np.array([[fi["frame"] for fi in inst_dict[i]["frame_info"]] for i in inst_ids], dtype=np.int64),
             obj_bboxes=np.array([[fi["bbox"] for fi in inst_dict[i]["frame_info"]] for i in inst_ids], dtype=np.int64),
             bg_frames=np.array([fi["frame"] for fi in bg_dict["frame_info"]], dtype=np.int64),
             bg_bboxes=np.array([fi["bbox"] for fi in bg_dict["frame_info"]], dtype=np.int64),
             obj_rgbs=f(sc.rgbs_batch_all), obj_depth=f(sc.depth_batch_all), obj_dirs=f(sc.ray_dirs_batch_all),
             obj_T_co=f(sc.t_co_batch_all), obj_indices=f(sc.batch_indices_all),
             bg_rgbs=f(bg.rgbs_batch_dict[0]), bg_depth=f(bg.depth_batch_dict[0]), bg_dirs=f(bg.ray_dirs_batch_dict[0]),
             bg_frame=f(bg.frame_batch_dict[0]), bg_T_wc=f(bg.t_wc_batch_dict[0]))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: obj pool {tuple(sc.rgbs_batch_all.shape)}, bg pool {tuple(bg.rgbs_batch_dict[0].shape)} "
          f"-> {os.path.getsize(path) / 1e6:.2f} MB")


def run_ckpt_case(name, seed, W=16, H=12, n_frames=4):
    """Checkpoint interchange: the reference's own sceneCategory.save_checkpoints (src/scene_cateogries.py:
    548-571) on reference-built categories -- one with three instances, one with a single instance, the background --
    read back with torch.load and stored as plain arrays: every state_dict entry, the per-object sim3 vectors the
    reference derives from T_obj (get_tensor_from_transform_sim3), extents, id -> row map, scalars; plus what
    Trainer.eval_points (src/trainer.py:125-151) returns for those weights on a fixed set of points."""
    import copy as _copy
    import tempfile
    rng = np.random.RandomState(seed)
    gen = torch.Generator().manual_seed(4000 + seed)
    cfg = SimpleNamespace(data_device="cpu", training_device="cpu", bg_scale=5.0, obj_scale=2.0,
                          hidden_feature_size=32, hidden_feature_size_bg=32, n_bins_cam2surface=1,
                          n_bins_cam2surface_bg=5, n_bins=9, min_depth=0.0, max_depth=10.0, surface_eps=0.1,
                          stop_eps=0.05, n_unidir_funcs=5,
                          net_hyperparams=dict(shape_blocks=2, texture_blocks=1, W=32, latent_dim=32))
    sample_dict = {}
    for f in range(n_frames):
        sample_dict[f] = dict(image=rng.randint(0, 256, size=(W, H, 3)).astype(np.uint8),
                              depth=(rng.rand(W, H) * 4).astype(np.float32),
                              T=rand_pose(gen, False).numpy().astype(np.float32),
                              obj_mask=rng.choice([-1, 0, 3, 7, 12, 20], size=(W, H)).astype(np.int64))
    rays_dir = torch.randn(W, H, 3, generator=gen)

    def inst(extent):
        return dict(T_obj=rand_pose(gen, True).numpy().astype(np.float64), bbox3D=SimpleNamespace(extent=np.array(extent)),
                    frame_info=[dict(frame=f, bbox=[1, 6, 2, 7]) for f in range(2)])
    multi = {3: inst([1.0, 2.0, 1.5]), 7: inst([0.5, 0.6, 0.7]), 12: inst([2.0, 1.0, 1.0])}
    single = {20: inst([1.2, 1.1, 1.0])}
    bg_dict = dict(bbox3D=np.array([6.0, 6.0, 3.0]), frame_info=[dict(frame=f, bbox=[0, W, 0, H]) for f in range(2)])
    torch.manual_seed(seed)
    cats = dict(multi=ref_sc.sceneCategory(cfg, 5, _copy.deepcopy(multi), sample_dict, rays_dir),
                single=ref_sc.sceneCategory(cfg, 6, _copy.deepcopy(single), sample_dict, rays_dir),
                bg=ref_sc.sceneCategory(cfg, 0, _copy.deepcopy(bg_dict), sample_dict, rays_dir))
    cats["multi"].trainer.extent_dict = cats["multi"].extent_dict        # what train.py's category set-up leaves there
    cats["single"].trainer.extent_dict = cats["single"].extent_dict
    pts = (torch.rand(300, 3, generator=gen) * 2 - 1)
    f = lambda t: t.detach().cpu().numpy()
    d = dict(meta=np.array([W, H, n_frames, seed], dtype=np.int64), points=f(pts), rays_dir=f(rays_dir),
             frames_image=np.stack([sample_dict[k]["image"] for k in range(n_frames)]),
             frames_depth=np.stack([sample_dict[k]["depth"] for k in range(n_frames)]),
             frames_mask=np.stack([sample_dict[k]["obj_mask"] for k in range(n_frames)]),
             frames_T=np.stack([sample_dict[k]["T"] for k in range(n_frames)]))
    with tempfile.TemporaryDirectory() as tmp:
        for tag, sc in cats.items():
            sc.save_checkpoints(tmp, 123)
            ck = torch.load(os.path.join(tmp, "cls_%d_iteration_00123.pth" % sc.cls_id), weights_only=False)
            d[tag + ".keys"] = np.array(sorted(ck.keys()))
            d[tag + ".scalars"] = np.array([ck["global_step"], ck["cls_id"], ck["obj_scale"]], dtype=np.float64)
            ids = sorted(ck["instance_id_to_index"].keys())
            d[tag + ".inst_ids"] = np.array(ids, dtype=np.int64)
            d[tag + ".inst_rows"] = np.array([ck["instance_id_to_index"][i] for i in ids], dtype=np.int64)
            for sd in ("PE_state_dict", "FC_state_dict", "shape_code_state_dict", "texture_code_state_dict"):
                if sd in ck:
                    d[tag + "." + sd + ".keys"] = np.array(list(ck[sd].keys()))
                    for k, v in ck[sd].items():
                        d[tag + "." + sd + "." + k] = f(v)
            if tag == "bg":
                d[tag + ".bound"] = np.asarray(ck["bound"], dtype=np.float64)
                occ, col = sc.trainer.eval_points(pts)
            else:
                src = multi if tag == "multi" else single
                d[tag + ".T_obj"] = np.stack([src[i]["T_obj"] for i in ids])
                d[tag + ".obj_tensor"] = np.stack([f(ck["obj_tensor_dict"][i]) for i in ids])
                d[tag + ".bound"] = np.stack([np.asarray(ck["bound"][i], dtype=np.float64) for i in ids])
                if "extent_dict" in ck:
                    d[tag + ".extent"] = np.stack([np.asarray(ck["extent_dict"][i], dtype=np.float64) for i in ids])
                occ, col = sc.trainer.eval_points(pts, inst_id=ids[-1])
            d[tag + ".eval_occ"], d[tag + ".eval_color"] = f(occ), f(col)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {len(d)} arrays -> {os.path.getsize(path) / 1e6:.2f} MB")


def run_camera_case(name):
    """a1: cameraInfo.get_rays_dirs (src/scene_cateogries.py:600-629) for two small cameras; the oracle must equal it."""
    d = {}
    for i, cam in enumerate((dict(W=9, H=6, fx=4.0, fy=3.5, cx=4.25, cy=2.5), dict(W=31, H=17, fx=577.87, fy=577.87, cx=15.5, cy=8.0))):
        ref = ref_sc.cameraInfo(SimpleNamespace(**cam)).rays_dir_cache
        mine = O.get_rays_dirs(cam["W"], cam["H"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        assert torch.equal(ref, mine), "oracle get_rays_dirs != reference"
        d["cam%d" % i] = np.array([cam[k] for k in ("W", "H", "fx", "fy", "cx", "cy")], dtype=np.float64)
        d["dirs%d" % i] = ref.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: ok")


def main():
    run_camera_case("cam_rays")
    run_ckpt_case("ckpt_ref", 9)
    run_case("s0_c1_r64_s16_l256", 0, 1, 64, 16, 256)
    run_case("s1_c1_r64_s16_l256", 1, 1, 64, 16, 256, keep_emb=False)
    run_case("s2_c1_r64_s16_l256", 2, 1, 64, 16, 256, keep_emb=False)
    run_case("s0_c2_r64_s16_l32", 0, 2, 64, 16, 32)
    run_case("s0_c1_r512_s32_l256", 0, 1, 512, 32, 256, keep_emb=False)
    run_case("s0_c1_r120_s10_l256", 0, 1, 120, 10, 256, keep_emb=False)  # the real config shape
    run_case("edge_empty_mask", 3, 2, 64, 16, 32, empty_mask_cls=1, keep_emb=False)
    run_case("edge_invalid_depth", 4, 1, 64, 16, 256, all_invalid=True, keep_emb=False)
    run_case("edge_single_obj_W", 5, 1, 64, 16, 256, single_obj=True, keep_emb=False)
    run_bg_case("bg_r240_s14_h128", 6, 240)            # background model: OccupancyMap(128), 5 + 9 samples
    run_bg_case("bg_r100_s14_h32", 7, 100, hidden=32)  # the pretrained per-object vMAP shape (hidden 32)
    run_pool_case("pool_w24_h20_f6", 8)                # ray-pool construction from frames


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pool":
        run_pool_case("pool_w24_h20_f6", 8)
    elif len(sys.argv) > 1 and sys.argv[1] == "ckpt":    # only the checkpoint + camera fixtures
        run_camera_case("cam_rays")
        run_ckpt_case("ckpt_ref", 9)
    elif len(sys.argv) > 1 and sys.argv[1] == "bg":      # only the background fixtures
        run_bg_case("bg_r240_s14_h128", 6, 240)
        run_bg_case("bg_r100_s14_h32", 7, 100, hidden=32)
    else:
        main()
