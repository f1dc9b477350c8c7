"""GPU: HIP kernels (through the C-ABI) against the reference's golden vectors and the oracle.

Tolerances: the modular tier computes in fp32 like the reference, so everything is held to
<= 2e-5 relative L2 (forward) / 2e-4 (gradients: fp32 atomics sum in a different order);
integer / mask outputs are bit-exact.  Where fp32 summation order matters (gradients under the
1/(sqrt(var)+1e-4) depth weighting reach 1e4 dynamic range) the bar is tied to the reference's OWN fp32 error
against an fp64 evaluation of the oracle: ours must be within 5x of it."""
import pytest
import torch

from conftest import Golden, bg_golden_names, golden_names, pool_golden_names, rel_l2
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
FWD_TOL = 2e-5
# End-to-end gradients: ReLU' is discontinuous, so ONE pre-activation that sits within fp32 rounding of
# zero (expected about once per ~1e5 units) flips a whole sample's contribution: 5.6e-4 on one layer of one
# fixture, 1.06e-3 on dB; tools/debug_grads.py shows the row pattern of a single flipped unit).  The bar
# therefore is 2e-3 (or 5x the reference's own fp32-vs-fp64 error when that is larger).
GRAD_TOL = 2e-3


@pytest.fixture(scope="module")
def cnr(dev):
    import cnr_amd
    info = cnr_amd._C.device_info()
    assert info["gfx950"], "these kernels are built for gfx950 only"
    return cnr_amd


@pytest.mark.parametrize("name", golden_names())
def test_sample_rays(cnr, dev, name):
    g = Golden(name, dev)
    out = cnr.ops.sample_rays(g.t("pool_rgbs"), g.t("pool_depth"), g.t("pool_dirs"), g.t("pool_T"), g.n1, g.n2,
                              g.eps, g.stop_eps, world_frame=g.single_obj, u=g.t("u"), g=g.t("g"), want_rays=True)
    assert torch.equal(out["labels"], g.t("labels"))
    assert torch.equal(out["depth_mask"].bool(), g.t("depth_mask"))
    assert rel_l2(out["gt_rgb"], g.t("gt_rgb")) < 1e-7
    assert rel_l2(out["origins"], g.t("origins")) < 1e-5
    assert rel_l2(out["dirs_o"], g.t("dirs_o")) < 1e-5
    assert rel_l2(out["z"], g.t("z")) < 1e-6
    assert rel_l2(out["pts"], g.t("pts")) < 1e-5


@pytest.mark.parametrize("name", [n for n in golden_names() if "emb" in Golden(n)])
def test_pe_forward(cnr, dev, name):
    g = Golden(name, dev)
    e = cnr.ops.UniDirsEmbedFn.apply(g.t("pts"), g.t("B"), g.scale)
    # sin(32*pi*p) amplifies a 1-ulp difference in p ~100x: 1e-5, not 1e-6, is the honest fp32 bar
    assert rel_l2(e, g.t("emb")) < 1e-5


@pytest.mark.parametrize("name", golden_names())
def test_mlp_forward_and_composite(cnr, dev, name):
    g = Golden(name, dev)
    e = cnr.ops.UniDirsEmbedFn.apply(g.t("pts"), g.t("B"), g.scale)
    mlp = g.mlp()
    cs = torch.stack([g.t("shape_codes")[c][g.t("indices")[c]][:, None] for c in range(g.C)])
    ct = torch.stack([g.t("texture_codes")[c][g.t("indices")[c]][:, None] for c in range(g.C)])
    zl = []
    for i, n in enumerate(cnr.ops.LATENT_LAYERS):
        code = ct if i == 3 else cs
        zl.append(torch.relu(torch.matmul(code, mlp[n + ".weight"].transpose(-1, -2)[:, None])
                             + mlp[n + ".bias"][:, None, None, :]))
    zlat = torch.cat(zl, dim=-2)
    params = []
    for n, _, _ in cnr.ops.TRUNK_LAYERS:
        params += [mlp[n + ".weight"], mlp[n + ".bias"]]
    sig, rgb = cnr.ops.CodeNeRFTrunkFn.apply(e, zlat, *params)
    assert rel_l2(sig, g.t("sigmas")) < FWD_TOL
    assert rel_l2(rgb, g.t("rgbs")) < FWD_TOL
    term, depth, var, rgbr, opa = cnr.ops.CompositeFn.apply(g.t("sigmas").squeeze(-1), g.t("rgbs"), g.t("z"))
    for got, k in ((term, "term"), (depth, "depth"), (var, "var"), (rgbr, "rgb"), (opa, "opacity")):
        assert rel_l2(got, g.t(k)) < FWD_TOL, k
    t2 = cnr.render_rays.occupancy_to_termination(g.t("occ"), is_batch=True)
    assert rel_l2(t2, g.t("term")) < FWD_TOL


def _fp64_truth_grads(g):
    """fp64 oracle gradients (CPU) for a fixture: the yardstick for fp32 summation-order noise."""
    gc = Golden(g.name)
    dd = lambda t: t.double() if t.is_floating_point() else t
    mlp = {k: dd(v).clone().requires_grad_() for k, v in gc.mlp().items()}
    B = dd(gc.t("B")).clone().requires_grad_()
    sh = [dd(gc.t("shape_codes")[c]).clone().requires_grad_() for c in range(gc.C)]
    tx = [dd(gc.t("texture_codes")[c]).clone().requires_grad_() for c in range(gc.C)]
    batch = dict(pts=dd(gc.t("pts")), z=dd(gc.t("z")), gt_depth=dd(gc.t("gt_depth")), gt_rgb=dd(gc.t("gt_rgb")),
                 labels=gc.t("labels"), depth_mask=gc.t("depth_mask"), indices=gc.t("indices"))
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        loss, _ = O.forward_loss(mlp, B, gc.scale, sh, tx, batch)
        loss.backward()
    finally:
        torch.set_default_dtype(old)
    z = lambda p: torch.zeros_like(p) if p.grad is None else p.grad
    out = {"grad." + k: z(v) for k, v in mlp.items()}
    out["grad_B"] = z(B)
    out["grad_shape_codes"] = torch.stack([z(s_) for s_ in sh])
    out["grad_texture_codes"] = torch.stack([z(s_) for s_ in tx])
    return out


def _grad_ok(got, key, g, truth):
    ref = g.t(key)
    tol = max(GRAD_TOL, 5.0 * rel_l2(ref.cpu(), truth[key]))
    err = rel_l2(got.cpu(), truth[key])
    assert err < tol, f"{key}: ours vs fp64 {err:.2e}, reference vs fp64 {rel_l2(ref.cpu(), truth[key]):.2e}"


def _update_ok(p_new, g, k_old, k_new, k_grad, lr=1e-3):
    """Compare the AdamW step where the gradient is not rounding noise: |g| >= 1e-3 max|g|."""
    ref_new, grad = g.t(k_new), g.t(k_grad)
    sel = grad.abs() >= 1e-3 * grad.abs().max()
    assert sel.any()
    assert float((p_new.detach() - ref_new)[sel].abs().max()) < 0.05 * lr, k_new
    assert rel_l2(p_new, ref_new) < 2e-3, k_new


def _build_reference_style_step(cnr, g, dev):
    """The reference's train.py:49-64,88-89,136-184 flow on our modules (functorch ensemble + vmap)."""
    from torch.func import vmap
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=g.L, obj_scale=g.scale,
                                   n_bins_cam2surface=g.n1, n_bins=g.n2)
    trainers = [cnr.trainer.Trainer(cfg, c + 1, list(range(g.n_obj))) for c in range(g.C)]
    mlp = g.mlp()
    for c, t in enumerate(trainers):
        t.fc_occ_map.load_state_dict({k: v[c] for k, v in mlp.items()})
        t.pe.load_state_dict({"B_layer.weight": g.t("B")[c], "scale": torch.tensor(g.scale)})
        with torch.no_grad():
            t.shape_codes.weight.copy_(g.t("shape_codes")[c])
            t.texture_codes.weight.copy_(g.t("texture_codes")[c])
    opt = torch.optim.AdamW([torch.zeros(1, device=dev, requires_grad=True)], lr=1e-3, weight_decay=0.013)
    for t in trainers:
        opt.add_param_group({"params": t.shape_codes.parameters(), "lr": 1e-3, "weight_decay": 0.013})
        opt.add_param_group({"params": t.texture_codes.parameters(), "lr": 1e-3, "weight_decay": 0.013})
    fc_model, fc_param, fc_buffer = cnr.utils.update_vmap([t.fc_occ_map for t in trainers], opt)
    pe_model, pe_param, pe_buffer = cnr.utils.update_vmap([t.pe for t in trainers], opt)
    idx = g.t("indices")
    cs = torch.stack([trainers[c].shape_codes(idx[c])[:, None, :] for c in range(g.C)])
    ct = torch.stack([trainers[c].texture_codes(idx[c])[:, None, :] for c in range(g.C)])
    emb = vmap(pe_model)(pe_param, pe_buffer, g.t("pts"))
    alpha, color = vmap(fc_model)(fc_param, fc_buffer, emb, cs, ct)
    loss, ld, lcol = cnr.loss.step_batch_loss(alpha, color, g.t("gt_depth"), g.t("gt_rgb"), g.t("labels"),
                                              g.t("depth_mask"), g.t("z"))
    from types import SimpleNamespace
    cls_dict = {c: SimpleNamespace(trainer=trainers[c], obj_ids=list(range(g.n_obj))) for c in range(g.C)}
    rs, rt = cnr.loss.step_batch_loss_reg(cls_dict, torch.arange(g.C, device=dev))
    loss = loss + 0.0005 * (rs + rt).sum()
    names = [n for n, _ in trainers[0].fc_occ_map.named_parameters()]
    return trainers, opt, names, fc_param, pe_param, alpha, color, loss, ld


@pytest.mark.parametrize("name", golden_names())
def test_reference_flow_loss_grads_and_update(cnr, dev, name):
    g = Golden(name, dev)
    trainers, opt, names, fc_param, pe_param, alpha, color, loss, ld = _build_reference_style_step(cnr, g, dev)
    assert rel_l2(alpha, g.t("sigmas")) < FWD_TOL and rel_l2(color, g.t("rgbs")) < FWD_TOL
    for k in ("depth", "color", "opacity"):
        assert rel_l2(ld[k], g.t("loss_" + k)) < FWD_TOL * 5, k
    assert rel_l2(loss, g.t("loss")) < FWD_TOL * 5
    loss.backward()
    truth = _fp64_truth_grads(g)
    for n, p in zip(names, fc_param):
        got = torch.zeros_like(g.t("grad." + n)) if p.grad is None else p.grad
        _grad_ok(got, "grad." + n, g, truth)
    _grad_ok(pe_param[0].grad, "grad_B", g, truth)
    _grad_ok(torch.stack([t.shape_codes.weight.grad for t in trainers]), "grad_shape_codes", g, truth)
    _grad_ok(torch.stack([t.texture_codes.weight.grad for t in trainers]), "grad_texture_codes", g, truth)
    opt.step()
    # AdamW's first step is lr * g / (|g| + eps): elements whose gradient is ~0 take a +-lr step whose
    # sign is rounding noise (in the reference too), so the post-step bar is 1e-4 of the parameter norm;
    # the optimiser arithmetic itself is pinned to 1e-6 in test_hip_adamw_matches_torch.
    for n, p in zip(names, fc_param):
        _update_ok(p, g, "mlp." + n, "new." + n, "grad." + n)
    _update_ok(pe_param[0], g, "B", "new_B", "grad_B")
    cnr.loss.check_flags()


def test_empty_mask_flags(cnr, dev):
    g = Golden("edge_empty_mask", dev)
    _ = _build_reference_style_step(cnr, g, dev)
    fl = cnr.loss.last_flags.cpu()
    assert ((fl & 2) != 0).all() and ((fl & 4) != 0).all() and ((fl & 8) == 0).all()


def test_hip_adamw_matches_torch(cnr, dev):
    torch.manual_seed(0)
    p = torch.randn(50000, device=dev)
    gr = torch.randn(50000, device=dev) * 0.1
    p_ref = p.clone().requires_grad_()
    opt = torch.optim.AdamW([p_ref], lr=1e-3, weight_decay=0.013)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        p_ref.grad = gr.clone()
        opt.step()
        cnr.ops.adamw_step(p, gr, m, v, 1e-3, (0.9, 0.999), 1e-8, 0.013, step)
        assert rel_l2(p, p_ref) < 1e-6


# ---- BASELINE.json sizes: direct comparison with the oracle + size-independent properties ------------
@pytest.mark.parametrize("C,R,S,L", [(1, 2048, 64, 256), (2, 4096, 128, 32)])
def test_full_size_against_oracle_and_properties(cnr, dev, C, R, S, L):
    gen = torch.Generator().manual_seed(1234)
    mlp = O.init_codenerf_params(C, 32, L, gen)
    B = torch.tensor(O.UNIDIRS).view(21, 3).repeat(C, 1, 1) + 0.01 * torch.randn(C, 21, 3, generator=gen)
    pts = torch.rand(C, R, S, 3, generator=gen) * 2 - 1
    z = torch.sort(torch.rand(C, R, S, generator=gen) * 3 + 0.5, dim=-1).values
    cs = torch.randn(C, R, 1, L, generator=gen) / (L / 2) ** 0.5
    ct = torch.randn(C, R, 1, L, generator=gen) / (L / 2) ** 0.5
    scale = 2.0
    e_ref = O.unidirs_embed(pts, B, scale)
    sig_ref, rgb_ref = O.codenerf_forward(mlp, e_ref, cs, ct)
    occ, term, depth, var, rgb, opa = O.composite(sig_ref.squeeze(-1), rgb_ref, z)

    d = lambda t: t.to(dev)
    e = cnr.ops.UniDirsEmbedFn.apply(d(pts), d(B), scale)
    assert rel_l2(e, e_ref) < 1e-5
    zl = []
    for i, n in enumerate(cnr.ops.LATENT_LAYERS):
        code = d(ct if i == 3 else cs)
        zl.append(torch.relu(torch.matmul(code, d(mlp[n + ".weight"]).transpose(-1, -2)[:, None])
                             + d(mlp[n + ".bias"])[:, None, None, :]))
    params = []
    for n, _, _ in cnr.ops.TRUNK_LAYERS:
        params += [d(mlp[n + ".weight"]), d(mlp[n + ".bias"])]
    sig, col = cnr.ops.CodeNeRFTrunkFn.apply(e, torch.cat(zl, dim=-2), *params)
    assert rel_l2(sig, sig_ref) < FWD_TOL and rel_l2(col, rgb_ref) < FWD_TOL
    t_, dep_, var_, rgb_, opa_ = cnr.ops.CompositeFn.apply(sig.squeeze(-1), col, d(z))
    for got, ref, k in ((t_, term, "term"), (dep_, depth, "depth"), (var_, var, "var"), (rgb_, rgb, "rgb"),
                        (opa_, opa, "opacity")):
        assert rel_l2(got, ref) < 1e-4, k   # north-star bar is 1e-3; fp32 path sits far inside it
    # properties: termination is a sub-probability distribution; render is linear in colour
    assert float(t_.min()) >= 0.0 and float(opa_.max()) <= 1.0 + 1e-5
    assert rel_l2(t_.sum(-1), opa_) < 1e-6
    _, _, _, rgb2, _ = cnr.ops.CompositeFn.apply(sig.squeeze(-1), col * 0.5, d(z))
    assert rel_l2(rgb2 * 2, rgb_) < 1e-6


@pytest.mark.parametrize("C,R,S,empty", [(1, 2048, 64, False), (2, 300, 16, False), (3, 130, 200, False),
                                        (2, 257, 64, True)])
def test_render_loss_single_launch_equals_three_calls(cnr, dev, C, R, S, empty):
    """cnr_render_loss == cnr_composite_fwd -> cnr_loss_fwd_bwd -> cnr_composite_bwd: gradients and renders
    bit-identical (same expressions), loss values to fp32 summation order, flags equal; a second launch on
    the same workspace reproduces the first bitwise."""
    _C = cnr._C
    gen = torch.Generator().manual_seed(C * 1000 + R + S)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    sig, col = rnd(C, R, S) * 3, torch.rand(C, R, S, 3, generator=gen).to(dev)
    z = (torch.rand(C, R, S, generator=gen).sort(dim=-1).values * 4 + 0.1).to(dev)
    gt_d, gt_c = (torch.rand(C, R, generator=gen) * 4).to(dev), torch.rand(C, R, 3, generator=gen).to(dev)
    labels = torch.randint(0, 3, (C, R), generator=gen).to(torch.uint8).to(dev)
    dmask = (torch.rand(C, R, generator=gen) > 0.2).to(torch.uint8).to(dev)
    if empty:
        labels[1] = 0                      # class 1 has no object ray: colour / depth terms vanish for ALL classes
    f = lambda *s: torch.empty(*s, device=dev)
    depth, var, rgb, opa = f(C, R), f(C, R), f(C, R, 3), f(C, R)
    _C.call("cnr_composite_fwd", sig, col, z, None, depth, var, rgb, opa, C * R, S, 0)
    losses, flags = f(3, C), torch.empty(C, device=dev, dtype=torch.int32)
    dd, dr, do = f(C, R), f(C, R, 3), f(C, R)
    _C.call("cnr_loss_fwd_bwd", depth, var, rgb, opa, gt_d, gt_c, labels, dmask, 5.0, 10.0, 0.5, losses, flags,
            dd, dr, do, C, R)
    dsig, dcol = f(C, R, S), f(C, R, S, 3)
    _C.call("cnr_composite_bwd", sig, col, z, dd, dr, do, None, dsig, dcol, C * R, S, 0)

    ws = torch.zeros(_C.render_loss_workspace_bytes(C, R), device=dev, dtype=torch.uint8)
    outs = []
    for _ in range(2):
        l2, f2 = f(3, C), torch.empty(C, device=dev, dtype=torch.int32)
        ds2, dc2 = f(C, R, S), f(C, R, S, 3)
        d2, v2, r2, o2 = f(C, R), f(C, R), f(C, R, 3), f(C, R)
        _C.call("cnr_render_loss", sig, col, z, gt_d, gt_c, labels, dmask, 5.0, 10.0, 0.5, ds2, dc2, d2, v2, r2, o2,
                C, R, S, ws, ws.numel(), None, None)
        _C.call("cnr_render_loss_finish", ws, l2, f2, C, R, 0)
        outs.append((l2, f2, ds2, dc2, d2, v2, r2, o2))
    l2, f2, ds2, dc2, d2, v2, r2, o2 = outs[0]
    assert torch.equal(ds2, dsig) and torch.equal(dc2, dcol)
    assert torch.equal(d2, depth) and torch.equal(v2, var) and torch.equal(r2, rgb) and torch.equal(o2, opa)
    assert torch.equal(f2, flags)
    assert rel_l2(l2, losses) < 1e-6
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    if empty:
        assert int(flags[0]) & 4 and float(losses[1].abs().sum()) == 0.0


# ---- background model (OccupancyMap) ---------------------------------------------------
@pytest.mark.parametrize("M,K,N,relu", [(16800, 87, 128, True), (16800, 215, 128, True), (16800, 128, 1, False),
                                        (1000, 170, 128, True), (77, 128, 3, False), (64, 32, 64, True),
                                        (1, 87, 32, True)])
def test_dense_kernels_against_torch(cnr, dev, M, K, N, relu):
    """cnr_dense_fwd / cnr_dense_bwd (exact-fp32 MFMA) == torch.nn.functional.linear (+ ReLU) and its autograd,
    to fp32 summation order (fp64 evaluation as the yardstick)."""
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=gen).to(dev).requires_grad_()
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev).requires_grad_()
    b = (torch.randn(N, generator=gen) * 0.1).to(dev).requires_grad_()
    y = cnr.ops.DenseFn.apply(x, W, b, relu)
    dy = torch.randn(M, N, generator=gen).to(dev)
    y.backward(dy)
    xd, Wd, bd = (t.detach().double().requires_grad_() for t in (x, W, b))
    yd = torch.nn.functional.linear(xd, Wd, bd)
    yd = torch.relu(yd) if relu else yd
    yd.backward(dy.double())
    assert rel_l2(y, yd) < 2e-6
    # ReLU'(0) ties: a pre-activation within fp32 rounding of zero can be masked differently in fp64
    tol = 1e-5 if not relu else 2e-4
    assert rel_l2(x.grad, xd.grad) < tol and rel_l2(W.grad, Wd.grad) < tol and rel_l2(b.grad, bd.grad) < tol


@pytest.mark.parametrize("name", bg_golden_names())
def test_background_step_against_reference(cnr, dev, name):
    """The background branch through the drop-in modules (sceneCategory.from_pool(cls_id=0) -> Trainer ->
    UniDirsEmbed -> OccupancyMap -> loss.step_batch_loss, train.py:113-121,172-184) against the reference's vectors:
    sampling bit-exact, alpha / colour / losses <= 2e-5, gradients within GRAD_TOL, AdamW update."""
    g = Golden(name, dev)
    hidden = g.L                                                   # meta[3] holds the hidden size here
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, obj_scale=g.scale, n_bins_cam2surface=g.n1,
                                   n_bins=g.n2)
    cfg.bg_scale, cfg.hidden_feature_size_bg, cfg.n_bins_cam2surface_bg = g.scale, hidden, g.n1
    cfg.surface_eps, cfg.stop_eps = g.eps, g.stop_eps
    pool = dict(rgbs=g.t("pool_rgbs")[0], depth=g.t("pool_depth")[0], dirs=g.t("pool_dirs")[0], T_wc=g.t("pool_T")[0])
    # a pool longer than one slice so that no reshuffle happens inside the call
    pool = {k: torch.cat([v, v, v]) for k, v in pool.items()}
    sc = cnr.scene_cateogries.sceneCategory.from_pool(cfg, 0, [0], pool)
    assert type(sc.trainer.fc_occ_map).__name__ == "OccupancyMap"
    sc.trainer.fc_occ_map.load_state_dict(g.mlp())
    with torch.no_grad():
        sc.trainer.pe.B_layer.weight.copy_(g.t("B")[0])
    sc.parity_draws = (g.t("u"), g.t("g"))
    gt_rgb, gt_depth, dmask, labels, pts, z, idx = sc.get_training_samples(g.R)
    assert torch.equal(labels, g.t("labels")[0]) and torch.equal(dmask, g.t("depth_mask")[0])
    assert rel_l2(z, g.t("z")[0]) < 1e-6 and rel_l2(pts, g.t("pts")[0]) < 1e-5 and int(idx.abs().sum()) == 0
    params = list(sc.trainer.fc_occ_map.parameters()) + list(sc.trainer.pe.parameters())
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.013)
    emb = sc.trainer.pe(pts)
    alpha, color = sc.trainer.fc_occ_map(emb)
    assert rel_l2(alpha[None], g.t("sigmas")) < FWD_TOL and rel_l2(color[None], g.t("rgbs")) < FWD_TOL
    loss, ld, _ = cnr.loss.step_batch_loss(alpha[None], color[None], gt_depth[None], gt_rgb[None] / 255.0, labels[None],
                                           dmask[None], z[None])
    assert rel_l2(loss, g.t("loss")) < FWD_TOL
    loss.backward()
    for n, p in sc.trainer.fc_occ_map.named_parameters():
        assert rel_l2(p.grad, g.t("grad." + n)) < GRAD_TOL, n
    assert rel_l2(sc.trainer.pe.B_layer.weight.grad, g.t("grad_B")[0]) < GRAD_TOL
    opt.step()
    for n, p in sc.trainer.fc_occ_map.named_parameters():
        assert rel_l2(p, g.t("new." + n)) < 1e-4, n


# ---- ray-pool construction from frames -------------------------------------------------------
@pytest.mark.parametrize("name", pool_golden_names())
def test_pool_construction_against_reference(cnr, dev, name):
    """sceneCategory(cfg, cls_id, inst_dict, sample_dict, rays) -- one cnr_gather_pool launch -- against the pools
    the reference's own constructor built from the same frames (src/scene_cateogries.py:107-350): pixel rows,
    states, object indices and the shuffle order bit-exact, poses to fp32 rounding of the 4x4 inverse."""
    import os
    from types import SimpleNamespace
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    seed = int(z["meta"][4])
    frames = [int(f) for f in z["frame_ids"]]
    sample_dict = {f: dict(image=z["images"][i], depth=z["depths"][i], T=z["T_wc"][i], obj_mask=z["masks"][i])
                   for i, f in enumerate(frames)}
    inst_dict = {}
    for k, iid in enumerate(int(i) for i in z["inst_ids"]):
        inst_dict[iid] = dict(T_obj=z["T_obj"][k], bbox3D=SimpleNamespace(extent=np.array([1.0, 2.0, 1.5])),
                              frame_info=[dict(frame=int(f), bbox=[int(v) for v in b])
                                          for f, b in zip(z["obj_frames"][k], z["obj_bboxes"][k])])
    bg_dict = dict(bbox3D=SimpleNamespace(extent=np.array([6.0, 6.0, 3.0])),
                   frame_info=[dict(frame=int(f), bbox=[int(v) for v in b]) for f, b in zip(z["bg_frames"], z["bg_bboxes"])])
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32)
    cfg.hidden_feature_size_bg = 32
    rays = torch.from_numpy(z["rays_dir"])
    t = lambda k: torch.from_numpy(z[k]).to(dev)

    np.random.seed(777 + seed)
    sc = cnr.scene_cateogries.sceneCategory(cfg, 5, inst_dict, sample_dict, rays)
    assert torch.equal(sc.rgbs_batch_all, t("obj_rgbs")) and torch.equal(sc.depth_batch_all, t("obj_depth"))
    assert torch.equal(sc.ray_dirs_batch_all, t("obj_dirs")) and torch.equal(sc.batch_indices_all, t("obj_indices"))
    assert rel_l2(sc.t_co_batch_all, t("obj_T_co")) < 1e-6
    assert type(sc.trainer.fc_occ_map).__name__ == "CodeNeRF" and sc.trainer.n_obj == 3

    np.random.seed(888 + seed)
    bg = cnr.scene_cateogries.sceneCategory(cfg, 0, bg_dict, sample_dict, rays)
    assert torch.equal(bg.rgbs_batch_dict[0], t("bg_rgbs")) and torch.equal(bg.depth_batch_dict[0], t("bg_depth"))
    assert torch.equal(bg.ray_dirs_batch_dict[0], t("bg_dirs")) and torch.equal(bg.frame_batch_dict[0], t("bg_frame"))
    assert torch.equal(bg.t_wc_batch_dict[0], t("bg_T_wc"))
    assert torch.equal(bg.t_wc_batch_all, t("bg_T_wc")[t("bg_frame")])
    # and the constructed pools feed the sampler: one slice of each
    out = sc.get_training_samples(16)
    assert out[4].shape == (16, cfg.n_bins_cam2surface + cfg.n_bins, 3) and torch.isfinite(out[4]).all()
    out = bg.get_training_samples(64)
    assert out[4].shape == (64, cfg.n_bins_cam2surface_bg + cfg.n_bins, 3) and int(out[6].abs().sum()) == 0


@pytest.mark.parametrize("name", [n for n in bg_golden_names() if n.endswith("h32")])
def test_uncertainty_field_probe_forward(cnr, dev, name):
    """the forward-only probe of get_uncertainty_fields (src/category_registration.py:127-152) --
    stratified bins (96 per ray; z_fixed is a no-op in the reference), UniDirsEmbed, a hidden-32 OccupancyMap, sigmoid(10 sigma), non-batch
    occupancy_to_termination, ray entropies -- on the drop-in modules against the oracle on the same weights."""
    g = Golden(name, dev)
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, obj_scale=g.scale)
    cfg.hidden_feature_size = g.L
    tr = cnr.trainer.Trainer(cfg, 0, [0])
    tr.fc_occ_map.load_state_dict(g.mlp())
    with torch.no_grad():
        tr.pe.B_layer.weight.copy_(g.t("B")[0])
    gen = torch.Generator().manual_seed(5)
    n_rays, r = 500, 1.3
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1)
    rays_o, viewdir = (r * d).to(dev), (-d).to(dev)
    zj = cnr.scene_cateogries.stratified_bins(0, 2 * r, 96, n_rays, device=dev, z_fixed=True)   # jittered all the same
    lo = O.stratified_bins(0.0, 2 * r, 96, n_rays, torch.zeros(n_rays, 96)).to(dev)
    assert bool(((zj >= lo) & (zj <= lo + 2 * r / 96 + 1e-6)).all()) and float((zj - lo).std()) > 1e-3
    z = O.stratified_bins(0.0, 2 * r, 96, n_rays, torch.rand(n_rays, 96, generator=gen)).to(dev)
    xyz = rays_o[..., None, :] + viewdir[:, None, :] * z[..., None]
    with torch.no_grad():
        sig, _ = tr.fc_occ_map(tr.pe(xyz))
        occ = torch.sigmoid(10 * sig.squeeze(-1))
        term = cnr.render_rays.occupancy_to_termination(occ)
        ent = (-term * torch.log(term + 1e-10)).sum(-1)
    mlp = {k: v.cpu() for k, v in g.mlp().items()}
    emb = O.unidirs_embed(xyz.cpu()[None], g.t("B").cpu(), g.scale)[0]
    sig_o, _ = O.occupancy_map_forward(mlp, emb)
    occ_o = torch.sigmoid(10 * sig_o.squeeze(-1))
    term_o = O.occupancy_to_termination(occ_o)
    ent_o = (-term_o * torch.log(term_o + 1e-10)).sum(-1)
    assert rel_l2(sig, sig_o) < FWD_TOL and rel_l2(term, term_o) < 1e-4 and rel_l2(ent, ent_o) < 1e-4
