"""GPU: the fused f16 background step (csrc/bg_fused.hip: cnr_bg_pack / _forward / _backward / _dw / _tail around
cnr_render_loss) against the vectors the REFERENCE produced for the background branch (tests/golden/bg_*_h128.npz: one
train.py:113-121,172-184 step of OccupancyMap(128) + UniDirsEmbed -- samples, outputs, losses, every gradient, the AdamW
update), and BackgroundStep(precision="fused") against the exact-fp32 tier over several steps.

Bars: the occupancy logit comes out of the precise geometry branch (three f16 products per fragment + fp32 head): 2e-5 like
the fp32 tier; colours carry f16 operands: 1e-3 (north_star's bar); gradients are an f16 chain with fp32 accumulation: each
tensor within a few per cent of the reference's and cosine > 0.999 overall, like the category kernel's."""
import pytest
import torch

from conftest import Golden, bg_golden_names, rel_l2

pytestmark = pytest.mark.gpu
NAMES = ["in_layer.0.weight", "in_layer.0.bias", "mid1.0.0.weight", "mid1.0.0.bias", "cat_layer.0.weight", "cat_layer.0.bias",
         "mid2.0.0.weight", "mid2.0.0.bias", "out_alpha.weight", "out_alpha.bias", "color_linear.0.weight",
         "color_linear.0.bias", "out_color.weight", "out_color.bias"]


def _flat(g, prefix):
    return torch.cat([g.t(prefix + n).reshape(-1) for n in NAMES] + [g.t({"mlp.": "B", "grad.": "grad_B", "new.": "new_B"}[prefix]).reshape(-1)])


@pytest.mark.parametrize("name", [n for n in bg_golden_names() if n.endswith("h128")])
def test_fused_background_step_against_the_reference_vectors(dev, name):
    import cnr_amd as cnr
    _C = cnr._C
    lib = _C.load()
    g = Golden(name, dev)
    R, S = g.R, g.S
    M = R * S
    theta = _flat(g, "mlp.").contiguous()
    n = theta.numel()
    assert n == int(lib.cnr_bg_param_count())
    f = lambda *sh, dt=torch.float32: torch.zeros(*sh, device=dev, dtype=dt)
    packed = f(int(lib.cnr_bg_pack_bytes()), dt=torch.uint8)
    sigma, rgbs = f(1, R, S), f(1, R, S, 3)
    act, eimg, dpre = f(5, M, 128, dt=torch.float16), f(M, 144, dt=torch.float16), f(5, M, 128, dt=torch.float16)
    nblk, chunk = int(lib.cnr_bg_blocks(M)), 256
    nch = int(lib.cnr_bg_dw_chunks(M, chunk))
    records, partials = f(nblk, int(lib.cnr_bg_record_floats())), f(nch, n)
    pts = g.t("pts").contiguous()
    _C.call("cnr_bg_pack", theta, packed)
    _C.call("cnr_bg_forward", pts, theta, packed, g.scale, M, sigma, rgbs, act, eimg)
    torch.cuda.synchronize()
    e_sig, e_rgb = rel_l2(sigma, g.t("sigmas").squeeze(-1)), rel_l2(rgbs, g.t("rgbs"))
    print(f"fused background forward {name}: sigma {e_sig:.2e} rgb {e_rgb:.2e}")
    assert e_sig < 2e-5 and e_rgb < 1e-3
    # activations and PE image are what the backward consumes: a1 against torch
    e1 = torch.cat([pts.view(M, 3) / g.scale, torch.sin(torch.pi * (2.0 ** torch.arange(6, device=dev))[:, None] *
                                                          ((pts.view(M, 3) / g.scale) @ g.t("B")[0].T)[:, None, :]).reshape(M, -1)], -1)
    assert rel_l2(eimg[:, :87].float(), e1[:, :87]) < 1e-3 and rel_l2(eimg[:, 96:138].float(), e1[:, 87:]) < 1e-3
    assert float(eimg[:, 87:96].abs().max()) == 0 and float(eimg[:, 138:].abs().max()) == 0
    a1 = torch.relu(e1[:, :87] @ g.t("mlp.in_layer.0.weight").T + g.t("mlp.in_layer.0.bias"))
    assert rel_l2(act[0].float(), a1) < 1e-3
    # ---- composite + losses + their gradient, then the backward ---------------------------------------------------------
    gscale = 256.0
    ws = torch.zeros(_C.render_loss_workspace_bytes(1, R), device=dev, dtype=torch.uint8)
    dsig, drgb = f(1, R, S), f(1, R, S, 3)
    depth, var, rgb, opa = f(1, R), f(1, R), f(1, R, 3), f(1, R)
    losses, flags = f(3, 1), f(1, dt=torch.int32)
    _C.call("cnr_render_loss", sigma, rgbs, g.t("z").contiguous(), g.t("gt_depth").contiguous(), g.t("gt_rgb").contiguous(),
            g.t("labels").contiguous(), g.t("depth_mask").to(torch.uint8).contiguous(), 5.0, 10.0, gscale, dsig, drgb, depth, var,
            rgb, opa, 1, R, S, ws, ws.numel(), None, None)
    _C.call("cnr_render_loss_finish", ws, losses, flags, 1, R, 0)
    for k, key in enumerate(("loss_depth", "loss_color", "loss_opacity")):
        assert rel_l2(losses[k], g.t(key)) < 1e-3, key
    _C.call("cnr_bg_backward", pts, theta, packed, g.scale, M, dsig, drgb, rgbs, act, dpre, records, None, 0)
    _C.call("cnr_bg_dw", act, dpre, eimg, M, chunk, partials, None, 0)
    # ---- the same two calls as ONE launch (cnr_bg_backward_render: every backward workgroup composites the rays its tile
    # touches): d sigma / d colour never reach memory, everything downstream must be the two-call form's bit for bit; the mask
    # counts come from the per-epoch table (here: one slice = the fixture's rays in order)
    pool_rgbs = torch.zeros(1, R, 4, device=dev, dtype=torch.uint8)
    pool_rgbs[0, :, 3] = g.t("labels").reshape(-1).to(torch.uint8)
    pool_depth = torch.where(g.t("depth_mask").reshape(1, R).bool(), torch.ones(1, R, device=dev), torch.zeros(1, R, device=dev))
    tab = f(1, 2, 4)
    _C.call("cnr_slice_maskcounts", pool_rgbs, pool_depth, None, R, 1, R, 1, 0.0, tab)
    ws2 = torch.zeros(int(lib.cnr_bg_backward_render_workspace_bytes(M)), device=dev, dtype=torch.uint8)
    dpre2, records2 = torch.zeros_like(dpre), torch.zeros_like(records)
    depth2, var2, rgb2, opa2 = f(1, R), f(1, R), f(1, R, 3), f(1, R)
    dsig2, drgb2 = f(1, R, S), f(1, R, S, 3)
    _C.call_struct("cnr_bg_backward_render", pts=pts, theta=theta, packed=packed, scale=g.scale, R=R, S=S, sigma=sigma, rgb=rgbs,
                   z=g.t("z").contiguous(), gt_depth=g.t("gt_depth").contiguous(), gt_rgb=g.t("gt_rgb").contiguous(),
                   labels=g.t("labels").contiguous(), depth_mask=g.t("depth_mask").to(torch.uint8).contiguous(), counts_tab=tab,
                   d_state=None, color_scaling=5.0, opacity_scaling=10.0, grad_scale=gscale, act=act, dpre=dpre2, records=records2,
                   depth=depth2, var=var2, rgb_render=rgb2, opacity=opa2, d_sigma=dsig2, d_rgb=drgb2, loss_workspace=ws2, loss_workspace_bytes=ws2.numel())
    torch.cuda.synchronize()
    assert torch.equal(dsig2, dsig) and torch.equal(drgb2, drgb)
    assert torch.equal(dpre2, dpre) and torch.equal(records2, records)
    assert torch.equal(depth2, depth) and torch.equal(var2, var) and torch.equal(rgb2, rgb) and torch.equal(opa2, opa)
    losses2, flags2 = f(3, 1), f(1, dt=torch.int32)
    _C.call("cnr_render_loss_finish", ws2, losses2, flags2, 1, R, nblk)       # same partial format, one per backward block
    assert rel_l2(losses2, losses) < 1e-6 and torch.equal(flags2, flags)
    # the weight-gradient kernel on its own: dW = dPre^T X, db = column sums, from the very f16 arrays it reads
    torch.cuda.synchronize()
    psum = partials.sum(0)
    xs = dict(in_layer=eimg[:, :87], mid1=act[0], cat_layer=torch.cat([act[1], eimg[:, :87]], 1), mid2=act[2],
              color_linear=torch.cat([act[3], eimg[:, 96:138]], 1))
    off = 0
    for li, (lname, x) in enumerate(xs.items()):
        k = x.shape[1]
        dwr = dpre[li].float().T @ x.float()
        dbr = dpre[li].float().sum(0)
        if lname == "color_linear":
            off = 72065
        assert rel_l2(psum[off:off + 128 * k].view(128, k), dwr) < 1e-5, lname
        assert rel_l2(psum[off + 128 * k:off + 128 * k + 128], dbr) < 1e-5, lname
        off += 128 * k + 128
    th2, grad, m, v = theta.clone(), f(n), f(n), f(n)
    state = torch.zeros(3, device=dev, dtype=torch.int64)
    packed2 = packed.clone()
    losses_t, flags_t = f(3, 1), f(1, dt=torch.int32)
    _C.call("cnr_bg_tail", th2, grad, m, v, partials, nch, records, nblk, gscale, 1e-3, 0.9, 0.999, 1e-8, 0.013, state, R, packed2,
            ws, R, losses_t, flags_t)
    # the tail refreshed the fragments of every weight it updated: identical to a fresh pack of the new parameters; and its loss
    # block is cnr_render_loss_finish
    fresh = torch.zeros_like(packed)
    _C.call("cnr_bg_pack", th2, fresh)
    assert torch.equal(packed2, fresh)
    assert torch.equal(losses_t, losses) and torch.equal(flags_t, flags)
    torch.cuda.synchronize()
    assert state.tolist() == [R, 1, 1]
    ref = _flat(g, "grad.")
    off, rep, worst = 0, [], 0.0
    for nme in NAMES + ["B"]:
        k = g.t("mlp." + nme).numel() if nme != "B" else 63
        e = rel_l2(grad[off:off + k], ref[off:off + k])
        rep.append(f"{nme}={e:.3f}")
        worst = max(worst, e)
        off += k
    cos = float((grad.double() @ ref.double()) / (grad.double().norm() * ref.double().norm()))
    print(f"fused background gradient {name}: cos {cos:.6f} worst {worst:.3f}  " + " ".join(rep))
    assert cos > 0.999 and worst < 0.06
    # AdamW (first step: -lr sign(g) where g is not noise)
    new_ref = _flat(g, "new.")
    clear = ref.abs() > 0.02 * ref.abs().max()
    assert rel_l2((th2 - theta)[clear], (new_ref - theta)[clear]) < 0.02
    # the step is a fixed-order computation: a second run gives the same bits
    th3, grad3 = theta.clone(), f(n)
    st3 = torch.zeros(3, device=dev, dtype=torch.int64)
    _C.call("cnr_bg_backward", pts, theta, packed, g.scale, M, dsig, drgb, rgbs, act, dpre, records, st3, R)   # the state moves here ...
    _C.call("cnr_bg_dw", act, dpre, eimg, M, chunk, partials, None, 0)
    _C.call("cnr_bg_tail", th3, grad3, f(n), f(n), partials, nch, records, nblk, gscale, 1e-3, 0.9, 0.999, 1e-8, 0.013,
            st3, -1, None, None, 0, None, None)                                                                                          # ... and the tail reads it as it stands
    assert torch.equal(grad3, grad) and torch.equal(th3, th2) and st3.tolist() == [R, 1, 1]


def test_launch_fusions_of_the_background_step_change_no_bit(dev, monkeypatch):
    """The fused step in its four launch layouts -- composite / losses inside the backward launch or as cnr_render_loss, the next
    step's sampler inside the last launch or as every step's first -- over 14 steps (three reshuffles, eager steps and graph
    replays): the same parameters, optimiser moments, loss values and step state, bit for bit."""
    import cnr_amd as cnr

    def run(fuse_render, sample_in_tail):
        monkeypatch.setenv("CNR_BG_FUSE_RENDER", str(fuse_render))
        monkeypatch.setenv("CNR_BG_SAMPLE_IN_TAIL", str(sample_in_tail))
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=9)
        cfg.hidden_feature_size_bg, cfg.n_bins_cam2surface_bg = 128, 5
        torch.manual_seed(3)
        pool = cnr.scene_cateogries.synthetic_pool(6 * 300, 1, torch.Generator().manual_seed(12), "cpu")
        bg = cnr.background.BackgroundStep(cfg, pool, 300, dev, seed=5, precision="fused")
        assert bg.fuse_render == bool(fuse_render) and bg.sample_in_tail == bool(sample_in_tail)
        losses = []
        for _ in range(14):
            bg.step()
            losses.append(bg.losses.clone())
        torch.cuda.synchronize()
        return bg.flat.clone(), bg.exp_avg.clone(), bg.exp_avg_sq.clone(), torch.stack(losses), bg.d_state.clone()
    ref = run(0, 0)
    assert float(ref[3].abs().sum()) > 0
    for fr, st in ((1, 0), (0, 1), (1, 1)):
        got = run(fr, st)
        for k, (x, y) in enumerate(zip(ref, got)):
            if k == 3 and fr:      # the loss VALUES are summed over other blocks (per backward block instead of per render block)
                assert rel_l2(y, x) < 1e-6
            else:
                assert torch.equal(x, y), (fr, st, k)


def test_fused_background_trainer_tracks_the_fp32_tier(dev, monkeypatch):
    """BackgroundStep(precision="fused") against precision="fp32" from the same initial parameters on the same sampled batches
    (device cursor + in-kernel Philox: identical), 12 steps across a reshuffle, graph replay included.  (With the sampler as every
    step's first launch, so that after a step both tiers' buffers hold THAT step's batch; the default layout -- the next step's
    batch drawn by the step's last launch -- is bit-equal to it: test_launch_fusions_of_the_background_step_change_no_bit.)"""
    import cnr_amd as cnr
    monkeypatch.setenv("CNR_BG_SAMPLE_IN_TAIL", "0")

    def make(prec):
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=9)
        cfg.hidden_feature_size_bg, cfg.n_bins_cam2surface_bg = 128, 5
        torch.manual_seed(3)
        pool = cnr.scene_cateogries.synthetic_pool(6 * 300, 1, torch.Generator().manual_seed(12), "cpu")
        return cnr.background.BackgroundStep(cfg, pool, 300, dev, seed=5, precision=prec)
    a, b = make("fused"), make("fp32")
    assert torch.equal(a.flat, b.flat)
    for it in range(12):
        a.step()
        b.step()
        torch.cuda.synchronize()
        assert torch.equal(a.bufs["z"], b.bufs["z"]), it
        assert rel_l2(a.losses, b.losses) < (2e-3 if it == 0 else 5e-2), (it, a.losses, b.losses)
    assert a.graph is not None and int(a.d_state[2]) == 12 and int(a.d_state[0]) == a.cursor
    assert torch.isfinite(a.flat).all()
    # both moved away from the start by the same amount and in the same direction
    start = make("fp32").flat
    da, db = a.flat - start, b.flat - start
    cos = float((da.double() @ db.double()) / (da.double().norm() * db.double().norm()))
    print("fused vs fp32 background trainer after 12 steps: cosine of the parameter displacement", round(cos, 4),
          "rel", round(rel_l2(a.flat, b.flat), 5))
    assert cos > 0.9 and rel_l2(a.flat, b.flat) < 5e-3


def test_whole_iteration_run_in_multi_iteration_graphs_equals_single_steps(dev):
    """FullStepTrainer.run(n): groups of up to eight iterations (background + categories, two forked branches each) as ONE
    hipGraph, across epoch ends of BOTH pools (which differ in length) -- against step() called n times: bit for bit."""
    import cnr_amd as cnr

    def make():
        torch.manual_seed(31)          # epoch permutations of the category trainer come from the default generator
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=28)
        gen = torch.Generator().manual_seed(21)
        pools = [cnr.scene_cateogries.synthetic_pool(11 * 256, 4, gen, "cpu") for _ in range(2)]
        tr = cnr.fused.FusedCategoryTrainer(cfg, 2, 4, pools, 256, dev, seed=1, generator=gen)
        cfg_bg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=9)
        cfg_bg.hidden_feature_size_bg, cfg_bg.n_bins_cam2surface_bg = 128, 5
        bg = cnr.background.BackgroundStep(cfg_bg, cnr.scene_cateogries.synthetic_pool(7 * 300, 1, torch.Generator().manual_seed(12), "cpu"),
                                           300, dev, seed=5, precision="fused")
        return cnr.background.FullStepTrainer(tr, bg)
    a, b = make(), make()
    n = 37
    a.run(n)
    for _ in range(n):
        b.step()
    torch.cuda.synchronize()
    assert any(isinstance(k, tuple) and k[1] >= 4 for k in a.graphs), list(a.graphs)      # multi-iteration graphs were used
    assert a.obj.steps_done == b.obj.steps_done == n and a.bg.steps_done == b.bg.steps_done == n
    assert a.obj.cursor == b.obj.cursor and a.bg.cursor == b.bg.cursor
    assert torch.equal(a.obj.theta, b.obj.theta) and torch.equal(a.obj.losses, b.obj.losses)
    assert torch.equal(a.bg.flat, b.bg.flat) and torch.equal(a.bg.losses, b.bg.losses)
    assert torch.equal(a.bg.d_state, b.bg.d_state)


def test_fused_background_forward_on_trained_weights(dev, monkeypatch):
    """The fused background forward against the fp32 oracle ON TRAINED WEIGHTS (the category kernel's occupancy left the 1e-3
    bar after training with plain f16 operands, tests/test_trained_parity_gpu.py; the background kernel has the three-product
    geometry branch from the start): 1500 steps of the fused trainer on a learnable pool, then the kernel's per-sample
    occupancy / colour and the four renders against oracle.ref_cpu on the same weights and the same samples."""
    import cnr_amd as cnr
    from oracle import ref_cpu as O
    monkeypatch.setenv("CNR_BG_SAMPLE_IN_TAIL", "0")     # after a step the sample buffers then hold THAT step's batch
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=9)
    cfg.hidden_feature_size_bg, cfg.n_bins_cam2surface_bg = 128, 5
    pool = cnr.scene_cateogries.synthetic_pool(16 * 1200, 1, torch.Generator().manual_seed(12), "cpu")
    bg = cnr.background.BackgroundStep(cfg, pool, 1200, dev, seed=5, precision="fused")
    first = None
    for it in range(1500):
        bg.step()
        if it == 3:
            first = bg.losses.clone()
    torch.cuda.synchronize()
    last = bg.losses.clone()
    assert torch.isfinite(bg.flat).all() and float(last[1]) < float(first[1])          # the colour term fell
    # one more forward of the kernels on the CURRENT weights, through the product's own step (its outputs stay in the buffers)
    w_before = bg.flat.clone()
    bg.step(use_graph=False)
    torch.cuda.synchronize()
    fb, bufs = bg.fb, bg.bufs
    pts, z = bufs["pts"].cpu(), bufs["z"].cpu()                    # (1, R, S, 3), (1, R, S)
    flat = w_before.cpu()
    sd, off = {}, 0
    for k, p_ in bg.trainer.fc_occ_map.named_parameters():
        sd[k] = flat[off:off + p_.numel()].view(p_.shape); off += p_.numel()
    Bm = flat[off:off + 63].view(1, 21, 3)
    emb = O.unidirs_embed(pts, Bm, float(bg.trainer.pe._scale))
    alpha, color = O.occupancy_map_forward(sd, emb)              # alpha: the x10 logit (src/model.py:151), colour after the sigmoid
    occ_ref, _, depth_ref, _, rgb_ref, opa_ref = O.composite(alpha.squeeze(-1), color, z)
    sig, col = fb["sigma"].cpu(), fb["rgbs"].cpu()
    errs = dict(occupancy=rel_l2(torch.sigmoid(sig), occ_ref), colour=rel_l2(col, color), depth=rel_l2(fb["depth"].cpu(), depth_ref),
                rgb=rel_l2(fb["rgb"].cpu(), rgb_ref), opacity=rel_l2(fb["opa"].cpu(), opa_ref))
    print("fused background forward on trained weights:", {k: "%.2e" % v for k, v in errs.items()},
          "max |x10 logit| %.1f" % float(sig.abs().max()), "losses first", first.tolist(), "last", last.tolist())
    for k, v in errs.items():
        assert v < 1e-3, (k, v)
