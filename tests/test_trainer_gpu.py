"""GPU: the fused single-launch-per-op train step (cnr_amd.fused) against the oracle's train step."""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr(dev):
    import cnr_amd
    return cnr_amd


def _oracle_params(cnr, tr, theta):
    v = tr.lay.views(theta.cpu())
    mlp, off = {}, 0
    for n, o, i in cnr.ops.TRUNK_LAYERS:
        mlp[n + ".weight"] = v["trunk"][:, off:off + o * i].reshape(tr.C, o, i).clone(); off += o * i
        mlp[n + ".bias"] = v["trunk"][:, off:off + o].clone(); off += o
    for k, n in enumerate(cnr.ops.LATENT_LAYERS):
        mlp[n + ".weight"] = v["latW"][:, k].clone()
        mlp[n + ".bias"] = v["latb"][:, k].clone()
    return mlp, v["B"].clone(), v["shape"].clone(), v["tex"].clone()


@pytest.mark.parametrize("C,R,n1,n2,L,n_obj", [(1, 256, 4, 28, 256, 4), (2, 128, 8, 56, 32, 4), (2, 256, 8, 56, 32, 20), (1, 256, 4, 28, 32, 31),
                                               (1, 512, 8, 56, 32, 64), (1, 600, 1, 9, 64, 100), (2, 384, 4, 28, 256, 100)])
def test_fused_train_step_against_oracle(cnr, dev, C, R, n1, n2, L, n_obj):
    """(20 / 31 / 64 / 100 objects per class: the one-launch kernel's one-object-per-tile row sums, S = 64 with the PE backward on
    the dW partner, S = 32 without, and S = 10 padded to a 32-slot tile per ray -- the latent-layer and code-table gradients below
    are what those sums feed; the tail launch's latent blocks then walk their objects in chunks of 32.)"""
    torch.manual_seed(1234)  # the trainer draws its epoch permutation from the default generator
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(7)
    pools = [cnr.scene_cateogries.synthetic_pool(8 * R, n_obj, gen, "cpu") for _ in range(C)]
    tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=3, generator=gen, use_graph=False)
    assert tr.use_records and tr.fused_tail and tr._ft_blocks
    theta0 = tr.theta.clone()
    rows = tr.perm[:, :R].long().cpu()          # pool rows of the first slice (epoch permutation)
    tr.step()
    b = {k: v.cpu() for k, v in tr.bufs.items() if torch.is_tensor(v)}
    losses = tr.losses.cpu()
    # ---- oracle on exactly the batch the kernels sampled -------------------------------------------------
    mlp, B, shape, tex = _oracle_params(cnr, tr, theta0)
    mlp = {k: v.requires_grad_() for k, v in mlp.items()}
    B.requires_grad_()
    sh = [shape[c].clone().requires_grad_() for c in range(C)]
    tx = [tex[c].clone().requires_grad_() for c in range(C)]
    idx = torch.stack([pools[c]["indices"][rows[c]] for c in range(C)])
    batch = dict(pts=b["pts"], z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"], labels=b["labels"],
                 depth_mask=b["depth_mask"].bool(), indices=idx)
    loss, aux = O.forward_loss(mlp, B, cfg.obj_scale, sh, tx, batch)
    loss.backward()
    for k, name in enumerate(("loss_depth", "loss_color", "loss_opacity")):
        assert rel_l2(losses[k], aux[name]) < 2e-3, name
    # the sampled z must be a valid depth-guided sample set of the pool slice (a5 invariants)
    d = torch.stack([pools[c]["depth"][rows[c]] for c in range(C)])
    valid = d > 0
    assert torch.equal(batch["depth_mask"], valid)
    zc = b["z"]
    assert bool((zc[..., :n1][valid] <= (d[valid] - cfg.surface_eps)[:, None] + 1e-5).all())
    # ---- gradient direction + AdamW step -------------------------------------------------------------------
    gk = tr.lay.views(tr.grad.cpu())
    off, dot, n_a, n_b = 0, 0.0, 0.0, 0.0
    for n, o, i in cnr.ops.TRUNK_LAYERS:
        for kind, cnt in (("weight", o * i), ("bias", o)):
            got = gk["trunk"][:, off:off + cnt].reshape(-1).double()
            ref = mlp[n + "." + kind].grad.reshape(-1).double()
            dot += float(got @ ref); n_a += float(got @ got); n_b += float(ref @ ref)
            off += cnt
    assert dot / (n_a * n_b) ** 0.5 > 0.99
    for k, n in enumerate(cnr.ops.LATENT_LAYERS):
        assert rel_l2(gk["latW"][:, k], mlp[n + ".weight"].grad) < 0.15, n
    assert rel_l2(gk["B"], B.grad) < 0.15
    assert rel_l2(gk["shape"], torch.stack([s.grad for s in sh])) < 0.15
    assert rel_l2(gk["tex"], torch.stack([s.grad for s in tx])) < 0.15
    # one AdamW step (first step = lr * sign(g) where g is not noise): compare the signs of the updates
    params = list(mlp.values()) + [B] + sh + tx
    opt = torch.optim.AdamW(params, lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    opt.step()
    new = tr.lay.views(tr.theta.cpu())
    old = tr.lay.views(theta0.cpu())
    upd_ref = (B.detach() - old["B"]).reshape(-1)
    upd_got = (new["B"] - old["B"]).reshape(-1)
    # ... on the entries whose reference gradient is not noise (an f16-pipeline gradient within rounding of zero takes
    # either sign; which entries those are depends on the sampled batch)
    g_ref = B.grad.reshape(-1).abs()
    clear = g_ref > 0.02 * g_ref.max()
    assert float(clear.float().mean()) > 0.5
    assert float((torch.sign(upd_ref) == torch.sign(upd_got))[clear].float().mean()) > 0.97


def test_graph_replay_trains(cnr, dev):
    """hipGraph-captured step: losses stay finite and the colour loss goes down over 60 replays."""
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=4, n_bins=28)
    gen = torch.Generator().manual_seed(11)
    pools = [cnr.scene_cateogries.synthetic_pool(64 * 512, 4, gen, "cpu")]
    tr = cnr.fused.FusedCategoryTrainer(cfg, 1, 4, pools, 512, dev, seed=5, generator=gen, use_graph=True)
    hist = []
    for it in range(60):
        tr.step()
        hist.append(tr.losses.clone())
    torch.cuda.synchronize()
    assert len(tr.graphs) == 2              # one captured graph per state parity
    h = torch.stack(hist).cpu()
    assert torch.isfinite(h).all()
    assert int(tr.d_state[2]) == 60 and int(tr.d_state[0]) == tr.cursor
    assert h[-10:, 1].mean() < h[:10, 1].mean()   # colour L1 decreases
    assert not bool((tr.flags.cpu() & 1).any())


def test_eager_one_graph_and_two_graphs_train_identically(cnr, dev):
    """The same 40 steps (same seeds, in-kernel Philox) run eagerly, as ONE captured hipGraph, and as the TWO graphs
    of the distributed step (front | all-reduce | back, here with no process group): parameters bitwise equal.  The
    kernels have no atomics on the gradient path, so this is exact, not a tolerance."""
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=4, n_bins=28)
    thetas = []
    for kw in (dict(use_graph=False), dict(use_graph=True), dict(use_graph=True, split_graph=True)):
        gen = torch.Generator().manual_seed(23)
        pools = [cnr.scene_cateogries.synthetic_pool(16 * 256, 4, gen, "cpu")]
        rng = torch.cuda.get_rng_state(dev)
        torch.cuda.manual_seed(1234)               # (CNR_EPOCH_PERM=torch: the epoch reshuffle then draws torch.randperm on the device)
        tr = cnr.fused.FusedCategoryTrainer(cfg, 1, 4, pools, 256, dev, seed=9, generator=gen, **kw)
        for _ in range(40):                        # 16 x 256 rows / 256 per step: crosses two epoch reshuffles
            tr.step()
        torch.cuda.synchronize()
        torch.cuda.set_rng_state(rng, dev)
        if kw.get("split_graph"):
            assert all(isinstance(g, tuple) for g in tr.graphs.values()) and len(tr.graphs) == 2
        thetas.append((tr.theta.clone(), tr.losses.clone(), int(tr.d_state[0])))
    for t, l, cur in thetas[1:]:
        assert torch.equal(t, thetas[0][0]) and torch.equal(l, thetas[0][1]) and cur == thetas[0][2]


@pytest.mark.parametrize("C,n_obj,L", [(1, 4, 256), (3, 2, 32), (2, 7, 64), (2, 15, 32)])
def test_param_prep_equals_separate_calls(cnr, dev, C, n_obj, L):
    """cnr_param_prep (pack | latent rows | zero fill side by side in one grid) == cnr_pack_weights +
    cnr_latent_fwd + a memset: operand image bit-identical, rows identical (same code path), buffer zero."""
    _C = cnr._C
    gen = torch.Generator().manual_seed(7 * C + n_obj)
    theta, lay = cnr.fused.init_params(C, L, n_obj, gen, dev)
    v = lay.views(theta)
    args = (lay.total, lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, C)
    zl0, br0 = torch.empty(C * n_obj, 4, 32, device=dev), torch.empty(C * n_obj, 4, 32, device=dev)
    _C.call("cnr_latent_fwd", theta, *args, zl0, br0)
    pk0 = cnr.ops.pack_weights(v["trunk"].contiguous())
    zl1, br1 = torch.empty_like(zl0), torch.empty_like(br0)
    pk1 = torch.empty_like(pk0)
    zbuf = torch.full((lay.total * C + 131,), 3.0, device=dev)       # odd length: the scalar tail is covered too
    _C.call("cnr_param_prep", theta, lay.total, lay.trunk[0], lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0],
            L, n_obj, C, pk1, zl1, br1, zbuf, zbuf.numel())
    assert torch.equal(pk0, pk1)
    assert torch.equal(zl0, zl1) and torch.equal(br0, br1)
    assert float(zbuf.abs().max()) == 0.0
    # against torch: z = relu(Wl code + bl), rows = Wt[:, :32] z + bt
    zref = []
    for k in range(4):
        code = v["tex"] if k == 3 else v["shape"]
        zref.append(torch.relu(torch.einsum("col,cnl->cno", v["latW"][:, k], code) + v["latb"][:, k][:, None, :]))
    zref = torch.stack(zref, dim=2).reshape(C * n_obj, 4, 32)
    assert rel_l2(zl1, zref) < 1e-6


def test_prologue_and_epilogue_launches_equal_their_parts(cnr, dev):
    """cnr_step_prologue == cnr_param_prep + cnr_sample_rays, and cnr_adamw_epilogue == cnr_adamw_step +
    cnr_step_epilogue (on a second state copy): every output bitwise equal -- the one-launch forms are the same
    device functions side by side in one grid."""
    _C, ops = cnr._C, cnr.ops
    C, n_obj, L, R, n1, n2 = 2, 4, 64, 200, 3, 13
    gen = torch.Generator().manual_seed(4)
    theta, lay = cnr.fused.init_params(C, L, n_obj, gen, dev)
    pools = [cnr.scene_cateogries.synthetic_pool(8 * R, n_obj, gen, "cpu") for _ in range(C)]
    st = lambda k: torch.stack([p[k] for p in pools]).to(dev).contiguous()
    rgbs, depth, dirs, T, idx = st("rgbs"), st("depth"), st("dirs"), st("T_co"), st("indices")
    pool_rows = depth.shape[1]
    perm = torch.stack([torch.randperm(pool_rows, generator=gen) for _ in range(C)]).to(torch.int32).to(dev)
    state = torch.tensor([3 * R, 7, 11], dtype=torch.int64, device=dev)
    mb = torch.empty(C, device=dev)
    _C.call("cnr_sample_maxdepth", depth, mb, state, pool_rows, perm, C, R)
    # ---- prologue
    pk = [torch.empty(C, _C.pack_bytes(), device=dev, dtype=torch.uint8) for _ in range(2)]
    zl, br = [torch.empty(C * n_obj, 4, 32, device=dev) for _ in range(2)], [torch.empty(C * n_obj, 4, 32, device=dev) for _ in range(2)]
    zb = [torch.full((1000,), 5.0, device=dev) for _ in range(2)]
    _C.call("cnr_param_prep", theta, lay.total, lay.trunk[0], lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj,
            C, pk[0], zl[0], br[0], zb[0], zb[0].numel())
    a = ops.sample_rays(rgbs, depth, dirs, T, n1, n2, 0.1, 0.05, seed=9, d_state=state, rays=R, out={}, max_bound=mb,
                        pool_indices=idx, n_obj=n_obj, perm=perm)
    b = ops.step_prologue(theta, lay, L, n_obj, pk[1], zl[1], br[1], zb[1], rgbs, depth, dirs, T, n1, n2, 0.1, 0.05, 0.0, 9,
                          state, R, {}, mb, idx, perm)
    assert torch.equal(pk[0], pk[1]) and torch.equal(zl[0], zl[1]) and torch.equal(br[0], br[1]) and torch.equal(zb[0], zb[1])
    for k in ("z", "pts", "gt_rgb", "gt_depth", "depth_mask", "labels", "ray_row"):
        assert torch.equal(a[k], b[k]), k
    # ---- the epoch's slice maxima as a table (cnr_slice_maxdepth) instead of this slice's value: same samples
    nsl = pool_rows // R
    table = torch.empty(C, nsl, device=dev)
    _C.call("cnr_slice_maxdepth", depth, perm, pool_rows, C, R, nsl, table)
    ref = torch.stack([depth[c][perm[c].long()][: nsl * R].reshape(nsl, R).max(dim=1).values for c in range(C)])
    assert torch.equal(table, ref) and torch.equal(table[:, 3], mb)
    zb2, pk2 = torch.full((1000,), 5.0, device=dev), torch.empty_like(pk[0])
    b2 = ops.step_prologue(theta, lay, L, n_obj, pk2, torch.empty_like(zl[0]), torch.empty_like(br[0]), zb2, rgbs, depth,
                           dirs, T, n1, n2, 0.1, 0.05, 0.0, 9, state, R, {}, table, idx, perm, max_bound_slices=nsl)
    for k in ("z", "pts", "gt_rgb", "gt_depth", "depth_mask", "labels", "ray_row"):
        assert torch.equal(a[k], b2[k]), k
    # ---- epilogue
    n = theta.numel()
    grad = torch.randn(n, generator=gen).to(dev) * 1e-3
    ws = torch.rand(_C.render_loss_workspace_bytes(C, R) // 4, generator=gen).to(dev)
    outs = []
    for fused in (False, True):
        p, m, v = theta.clone().reshape(-1), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        s_cur, s_next = state.clone(), torch.zeros(3, dtype=torch.int64, device=dev)
        losses, flags, mbn = torch.empty(3, C, device=dev), torch.empty(C, device=dev, dtype=torch.int32), torch.empty(C, device=dev)
        if fused:
            _C.call("cnr_adamw_epilogue", p, grad, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 0.013, 1.0, s_cur, s_next, R, ws, losses,
                    flags, depth, pool_rows, perm, mbn, C, R)
            new_state = s_next
        else:
            ops.adamw_step(p, grad, m, v, 1e-3, (0.9, 0.999), 1e-8, 0.013, 0, d_state=s_cur)
            _C.call("cnr_step_epilogue", s_cur, R, ws, losses, flags, depth, pool_rows, perm, mbn, C, R)
            new_state = s_cur
        outs.append((p, m, v, losses, flags, mbn, new_state.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert outs[1][6].tolist() == [4 * R, 8, 12]


def test_plain_f16_trainer_tracks_the_precise_default(cnr, dev):
    """precise_geometry=False (plain f16 operands in the geometry branch, the round-2 forward) against the default (three
    products per fragment): same trajectory to the f16 level (the losses of the first steps agree to 2e-3), graph capture
    included."""
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=56)
    out = []
    for sw in (True, False):
        gen = torch.Generator().manual_seed(31)
        pools = [cnr.scene_cateogries.synthetic_pool(16 * 256, 4, gen, "cpu")]
        torch.cuda.manual_seed(77)
        tr = cnr.fused.FusedCategoryTrainer(cfg, 1, 4, pools, 256, dev, seed=3, generator=gen, precise_geometry=sw)
        hist = []
        for _ in range(6):
            tr.step()
            hist.append(tr.losses.clone())
        torch.cuda.synchronize()
        out.append(torch.stack(hist).cpu())
    assert torch.isfinite(out[1]).all()
    # (step 0: same parameters, the forward differs by the weight-rounding share of the f16 error; later steps also carry
    #  the depth term's 1 / (sqrt(var) + 1e-4) amplification of the parameter differences)
    assert rel_l2(out[1][0], out[0][0]) < 2e-3 and rel_l2(out[1], out[0]) < 1e-2 and not torch.equal(out[1], out[0])


def test_process_group_path_matches_single_gpu_path(cnr, dev):
    """The N > 1 step (front graph: prologue, forward + render, field backward + record reduction, latent backward;
    all-reduce of the flat gradient; back graph: AdamW + epilogue) on ONE rank: a world-size-1 gloo group makes the
    trainer take that path, and the result must match the single-GPU step (one launch for reduction + latent
    backward + AdamW) to fp32 rounding -- the per-object bias-row sums reach the latent backward as floats on one
    path and as 2^-40 fixed point on the other."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        res = {}
        for name, pg in (("single", None), ("group", dist.group.WORLD)):
            torch.manual_seed(99)
            cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=56)
            gen = torch.Generator().manual_seed(21)
            pools = [cnr.scene_cateogries.synthetic_pool(16 * 256, 4, gen, "cpu") for _ in range(2)]
            tr = cnr.fused.FusedCategoryTrainer(cfg, 2, 4, pools, 256, dev, seed=4, generator=gen, process_group=pg,
                                                use_graph=True)
            hist, early = [], None
            for it in range(6):           # two eager steps, then captured replays of both parities
                tr.step()
                hist.append(tr.losses.clone())
                if it == 1:
                    torch.cuda.synchronize()
                    early = (tr.theta.clone(), tr.grad.clone())
            torch.cuda.synchronize()
            res[name] = (tr.theta.clone(), torch.stack(hist), early)
        g, s1 = res["group"], res["single"]
        # same forward on the same parameters: the first step's losses are the same numbers; after one update the paths
        # differ by the rounding of the bias-row sums only
        assert rel_l2(g[1][0], s1[1][0]) < 1e-6
        assert rel_l2(g[2][1], s1[2][1]) < 1e-4 and rel_l2(g[2][0], s1[2][0]) < 1e-5
        # four more steps through the captured graphs: still the same training run (the depth loss is weighted by
        # 1 / (sqrt(var) + 1e-4): it amplifies parameter rounding, hence the looser bar)
        assert torch.isfinite(g[1]).all() and rel_l2(g[1], s1[1]) < 5e-3
        assert rel_l2(g[0], s1[0]) < 1e-3
        assert cnr.parallel.params_in_sync(g[0], dist.group.WORLD)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("C,R,n1,n2", [(1, 2048, 8, 56), (2, 4096, 16, 112)])
def test_full_size_step_sampling_invariants(cnr, dev, C, R, n1, n2):
    """BASELINE sizes: the samples the fused step draws obey the reference's construction (scene_cateogries.py:453-546)
    whatever the size -- stratified bins between min depth and surface - eps, sorted draws inside surface +- eps for
    rays on the object, stratified bins up to surface + stop_eps for other rays, min .. slice max for rays without
    depth -- and the step's losses are finite."""
    torch.manual_seed(5)
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(31)
    pools = [cnr.scene_cateogries.synthetic_pool(4 * R, 4, gen, "cpu") for _ in range(C)]
    tr = cnr.fused.FusedCategoryTrainer(cfg, C, 4, pools, R, dev, seed=2, generator=gen, use_graph=False)
    rows = tr.perm[:, :R].long().cpu()
    tr.step()
    torch.cuda.synchronize()
    z = tr.bufs["z"].cpu()
    assert torch.isfinite(tr.losses).all() and torch.isfinite(z).all()
    eps, stop, lo = cfg.surface_eps, cfg.stop_eps, cfg.min_depth
    for c in range(C):
        d = pools[c]["depth"][rows[c]]
        state = pools[c]["rgbs"][rows[c]][:, 3]
        valid, on_obj = d > lo, (pools[c]["rgbs"][rows[c]][:, 3] == 1) & (d > lo)
        zc = z[c]
        # camera -> surface bins
        front = zc[valid][:, :n1]
        assert bool((front >= lo - 1e-6).all()) and bool((front <= (d[valid] - eps)[:, None] + 1e-5).all())
        assert bool((front[:, 1:] >= front[:, :-1]).all())
        # around the surface, rays on the object: ascending, clipped to +- eps
        near = zc[on_obj][:, n1:]
        assert bool((near[:, 1:] >= near[:, :-1]).all())
        assert bool(((near - d[on_obj][:, None]).abs() <= eps + 1e-5).all())
        # other rays with depth: up to surface + stop_eps
        other = valid & ~on_obj
        if bool(other.any()):
            back = zc[other][:, n1:]
            assert bool((back >= (d[other] - eps)[:, None] - 1e-5).all()) and bool((back <= (d[other] + stop)[:, None] + 1e-5).all())
        # rays without depth: min depth .. max depth of the slice
        if bool((~valid).any()):
            mb = float(d.max())
            free = zc[~valid]
            assert bool((free >= lo - 1e-6).all()) and bool((free <= mb + 1e-4).all())
        assert state.max() <= 2


@pytest.mark.parametrize("n_obj", [4, 6, 10])
def test_many_epochs_graph_equals_eager(cnr, dev, n_obj):
    """(Four objects per class, six and ten: up to fifteen stay on the record path of the 8-wave backward.)  Sixty steps over a pool of eight slices (a reshuffle every seven steps: new permutation, new per-slice max-depth
    table, cursor back to zero) -- the captured graphs must keep following the device-side state across epochs: same
    losses and parameters, bitwise, as the eager trainer."""
    res = {}
    for name, graph in (("eager", False), ("graph", True)):
        torch.manual_seed(77)          # epoch permutations come from the default generator
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=64, n_bins_cam2surface=4, n_bins=28)
        gen = torch.Generator().manual_seed(3)
        pools = [cnr.scene_cateogries.synthetic_pool(8 * 128, n_obj, gen, "cpu") for _ in range(2)]
        tr = cnr.fused.FusedCategoryTrainer(cfg, 2, n_obj, pools, 128, dev, seed=1, generator=gen, use_graph=graph)
        assert tr.fused_tail
        hist = []
        for _ in range(60):
            tr.step()
            hist.append(tr.losses.clone())
        torch.cuda.synchronize()
        res[name] = (torch.stack(hist), tr.theta.clone())
    assert torch.isfinite(res["graph"][0]).all()
    assert torch.equal(res["graph"][0], res["eager"][0]) and torch.equal(res["graph"][1], res["eager"][1])


@pytest.mark.parametrize("n_obj,n1,n2", [(20, 4, 28), (20, 4, 8), (64, 1, 9), (100, 8, 56)])
def test_many_objects_per_class_train_bitwise_repeatably(cnr, dev, n_obj, n1, n2):
    """More than fifteen objects per class (the reference allows n_models = 100): the one-launch kernel's one-object-per-tile
    form -- a tile lies inside one ray = one object, the per-object sums go straight to the fixed-point table -- for any count up
    to 128.  Rays of up to 16 samples, otherwise two to a tile, get a 32-slot tile of their own there (12 and 10 samples here: the
    reference's real shape).  Graph replay and eager stepping end bit for bit in the same place, and training makes progress."""
    def make(graph):
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=n1, n_bins=n2)
        gen = torch.Generator().manual_seed(5)
        pools = [cnr.scene_cateogries.synthetic_pool(8 * 256, n_obj, gen, "cpu") for _ in range(2)]
        return cnr.fused.FusedCategoryTrainer(cfg, 2, n_obj, pools, 256, dev, seed=2, generator=gen, use_graph=graph)
    res = {}
    for name, graph in (("eager", False), ("graph", True)):
        tr = make(graph)
        assert tr.fused_tail and tr._ft_blocks                 # the one-launch path, whatever the slot fill
        hist = []
        for _ in range(24):
            tr.step()
            hist.append(tr.losses.clone())
        torch.cuda.synchronize()
        res[name] = (torch.stack(hist), tr.theta.clone())
    assert torch.isfinite(res["graph"][0]).all()
    assert torch.equal(res["graph"][0], res["eager"][0]) and torch.equal(res["graph"][1], res["eager"][1])
    h = res["graph"][0]
    assert float(h[-4:].sum()) < float(h[:4].sum())


def test_a_class_too_large_for_any_path_is_refused(cnr, dev):
    """more than 128 objects, or more than 15 with more than 128 samples per ray: a message at construction, not a wrong step"""
    gen = torch.Generator().manual_seed(5)
    for n_obj, n1, n2 in ((129, 4, 28), (16, 16, 120)):
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=n1, n_bins=n2)
        pools = [cnr.scene_cateogries.synthetic_pool(1024, n_obj, gen, "cpu")]
        with pytest.raises(ValueError, match="one-launch step body"):
            cnr.fused.FusedCategoryTrainer(cfg, 1, n_obj, pools, 128, dev, generator=gen)


def test_code_tables_are_their_own_adamw_group(cnr, dev):
    """The reference gives the code tables their own AdamW group (code_lr / code_weight_decay, train.py:40,54-64).  With
    different values for the two groups the networks' parameters after a step are bitwise what a run with equal values gives,
    and the codes follow AdamW's first step with THEIR lr / weight decay: p (1 - lr wd) - lr g / (|g| + eps)."""
    res = {}
    for name, (clr, cwd) in (("same", (1e-3, 0.013)), ("own", (4e-3, 0.05))):
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=28)
        cfg.code_learning_rate, cfg.code_weight_decay = clr, cwd
        gen = torch.Generator().manual_seed(11)
        pools = [cnr.scene_cateogries.synthetic_pool(4 * 200, 3, gen, "cpu") for _ in range(2)]
        tr = cnr.fused.FusedCategoryTrainer(cfg, 2, 3, pools, 200, dev, seed=3, generator=gen, use_graph=False)
        th0 = tr.theta.clone()
        tr.step()
        torch.cuda.synchronize()
        res[name] = (th0, tr.theta.clone(), tr.grad.clone(), tr.lay)
    th0, a, g, lay = res["same"]
    b = res["own"][1]
    assert torch.equal(th0, res["own"][0]) and torch.equal(g, res["own"][2])
    net = slice(0, lay.shape[0])
    assert torch.equal(a[:, net], b[:, net])
    codes = slice(lay.shape[0], lay.total)
    gd = g[:, codes].double()
    for th, lr, wd in ((a, 1e-3, 0.013), (b, 4e-3, 0.05)):
        want = th0[:, codes].double() * (1 - lr * wd) - lr * gd / (gd.abs() + 1e-8)
        assert (th[:, codes].double() - want).abs().max() < 2e-6 * max(1.0, float(want.abs().max()))


def test_run_with_multi_step_graphs_equals_single_steps(cnr, dev):
    """run(n) captures groups of `unroll` steps in one hipGraph (no idle GPU between the steps of a group); cursor, RNG step,
    optimiser step and the parameter ping-pong are device-side state, so 75 steps over a pool of 13 slices (epoch ends
    inside: groups must not cross them) give bitwise the parameters, optimiser moments and per-step losses of 75 step()
    calls; check_flags sees every step of a group."""
    res = {}
    for name in ("single", "multi"):
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=28)
        gen = torch.Generator().manual_seed(3)
        pools = [cnr.scene_cateogries.synthetic_pool(13 * 96, 4, gen, "cpu") for _ in range(2)]
        tr = cnr.fused.FusedCategoryTrainer(cfg, 2, 4, pools, 96, dev, seed=1, generator=gen, unroll=6, check_every=10)
        hist = []
        if name == "single":
            for _ in range(75):
                tr.step()
                hist.append(tr.losses.clone())
        else:
            for n in (3, 20, 1, 12, 39):
                if n == 20:
                    tr.prepare_graphs()          # captures (does not run) the one-step and six-step graphs of both parities
                    assert {0, 1, (0, 6), (1, 6)} <= set(tr.graphs)
                tr.run(n)
                h = tr.loss_history()
                hist.extend(h[k].clone() for k in range(h.shape[0]))
            assert any(k[1] == 6 for k in tr.graphs if isinstance(k, tuple))      # groups of six were used
        torch.cuda.synchronize()
        res[name] = (tr.theta.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.steps_done, tr.cursor, tr.d_state.clone())
        res[name + "_hist"] = hist
    for a, b in zip(res["single"], res["multi"]):
        assert (a == b) if not torch.is_tensor(a) else torch.equal(a, b)
    # the multi run reports the steps of each launch: the last `k` of every run() call, all of them bitwise the single-step values
    single = torch.stack(res["single_hist"])
    got = res["multi_hist"]
    assert len(got) >= 5 and all(any(torch.equal(g, s) for s in single) for g in got)
    assert torch.equal(got[-1], single[-1])


@pytest.mark.parametrize("C,R,n1,n2,n_obj,L", [(1, 2048, 8, 56, 4, 256), (2, 250, 4, 28, 4, 32), (2, 249, 8, 56, 6, 32),
                                               (1, 1023, 16, 112, 4, 32), (3, 64, 4, 28, 7, 32), (1, 1, 8, 56, 1, 32),
                                               # padded rays: the reference's real shape (120 rays per object x (1 + 9)
                                               # samples: two rays per tile), an odd ray count, 16 of 16, 24 of 32, 40 of
                                               # 64, 100 of 128 slots
                                               (1, 480, 1, 9, 4, 256), (2, 249, 1, 9, 5, 32), (1, 77, 2, 14, 4, 32),
                                               (1, 130, 4, 20, 4, 32), (2, 100, 8, 32, 4, 32), (1, 50, 12, 88, 4, 32),
                                               # 8-15 objects per class: two row-sum blocks
                                               (2, 250, 4, 28, 9, 32), (1, 480, 1, 9, 15, 32), (1, 256, 8, 56, 12, 32),
                                               (1, 96, 16, 112, 8, 32)])
@pytest.mark.parametrize("precise", [False, True], ids=["plain_f16", "precise_geometry"])
def test_one_launch_step_equals_forward_render_plus_backward(cnr, dev, C, R, n1, n2, n_obj, L, precise):
    """cnr_field_train (a8-a15 forward, losses, loss gradient and the field backward in ONE launch, the ray's tiles
    exchanging composite partials between waves) against cnr_field_fwd_render + cnr_field_bwd_pipe: same f16 pipeline,
    same samples -> renders and loss values to fp32 summation order, the complete gradient (trunk, latent layers, B,
    codes) to 1e-4, parameters after AdamW; ragged tile counts (dead tiles in the last workgroup iteration), 5-7
    objects per class (the run-time row-sum stride), 8-15 (two row-sum blocks), S = 32 / 64 / 128 (1, 2, 4 tiles per ray) and rays padded to
    16 / 32 / 64 / 128 sample slots (dead lanes; two 16-slot rays per tile for S <= 16).
    Both modes run ONE arithmetic end to end on either path -- the stand-alone backward takes the residual image and recomputes the
    precise forward's activations and ReLU masks (round 4; before, it recomputed plain f16 and the bar for the default mode had to
    be 2e-2) -- so one gradient bar serves plain_f16 and precise_geometry alike.  It is 4e-3, not the 3e-4 the arithmetic alone
    would hold (measured 1e-5 .. 1.2e-4 with fp32 records): the two paths leave their partial sums in per-workgroup records whose
    entries are bf16 since round 4 (2^-9 each, independent in the two launches; these small shapes have 1 .. 32 records, where
    configs[1] has 256) -- measured 4e-4 .. 2.2e-3.  A mis-indexed row or a dropped layer is a 1e-1 effect."""
    res = {}
    for name, one in (("two", False), ("one", True)):
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2)
        gen = torch.Generator().manual_seed(5)
        pools = [cnr.scene_cateogries.synthetic_pool(max(4 * R, 8), n_obj, gen, "cpu") for _ in range(C)]
        tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=2, generator=gen, use_graph=False,
                                            one_launch=one, precise_geometry=precise)
        assert bool(tr._ft_blocks) == one
        hist = []
        for _ in range(3):
            tr.step()
            torch.cuda.synchronize()
            hist.append(dict(grad=tr.grad.clone(), losses=tr.losses.clone(), flags=tr.flags.clone(), theta=tr.theta.clone(),
                             **{k: tr.bufs[k].clone() for k in ("depth", "var", "rgb", "opa", "z")}))
        res[name] = hist
    a, b = res["one"][0], res["two"][0]
    assert torch.equal(a["z"], b["z"])
    for k in ("depth", "rgb", "opa"):
        assert rel_l2(a[k], b[k]) < 1e-6, (k, rel_l2(a[k], b[k]))
    assert rel_l2(a["var"], b["var"]) < 1e-5
    assert rel_l2(a["losses"], b["losses"]) < 5e-5 and torch.equal(a["flags"], b["flags"])   # (v_rcp / v_sqrt in the depth weight)
    # (the one-launch kernel forms its quotients with v_rcp_f32, 1 ulp: d sigma differs in the last bit, which the f16 pack of
    #  the scaled gradients turns into an f16 ulp here and there; rays that span tiles (S > 32): the one-launch kernel hands d e1 / d e2
    #  to the PE backward as f16, like every other pre-activation gradient of the chain, the two-launch form keeps them fp32: 1.2e-4
    #  on the whole gradient, all of it in dB)
    # (a batch whose only live loss term is saturated -- the one-ray case can draw an "unknown" ray with opacity exactly 1: both
    #  gradients are then f16 underflow noise of norm 1e-5, six orders under a real one -- has nothing to compare)
    noise = float(b["grad"].double().norm()) < 1e-3
    if noise:
        assert float(a["grad"].double().norm()) < 1e-3       # ... i.e. both negligible
    else:
        assert rel_l2(a["grad"], b["grad"]) < 4e-3, rel_l2(a["grad"], b["grad"])
        assert rel_l2(a["theta"], b["theta"]) < 1e-3     # first AdamW step: +-lr per entry, the sign of a ~0 gradient entry is noise
    for s in range(3):   # still the same training run two steps later (AdamW's sign-like first steps amplify rounding)
        assert torch.isfinite(res["one"][s]["grad"]).all()
        if not noise:     # (AdamW turns a noise gradient into +-lr per entry: two runs that start from one then differ for real)
            assert rel_l2(res["one"][s]["losses"], res["two"][s]["losses"]) < 4e-2


def test_classes_with_different_object_counts_and_a_single_object_class(cnr, dev):
    """The reference's categories differ (train.py:92-96): here one class with three objects (object frame, code
    regulariser) and one with a single object (WORLD frame: origin_dirs_W on T_wc, no regulariser, src/scene_cateogries.py:
    427-432, src/loss.py:5-15) in one fused trainer -- padded code rows, per-class regulariser switch, per-class frame --
    against the oracle on the sampled batch, and a state_dicts / load_state_dicts round trip."""
    C, R, n1, n2, L = 2, 512, 4, 28, 32
    n_objs = [3, 1]
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(8)
    pools = [cnr.scene_cateogries.synthetic_pool(4 * R, k, gen, "cpu") for k in n_objs]
    tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_objs, pools, R, dev, seed=4, generator=gen, use_graph=False)
    assert tr.n_obj == 3 and tr.world_frame == [False, True] and tr.n_obj_cls.tolist() == n_objs
    theta0 = tr.theta.clone()
    rows = tr.perm[:, :R].long().cpu()
    tr.step()
    torch.cuda.synchronize()
    b = {k: v.cpu() for k, v in tr.bufs.items() if torch.is_tensor(v)}
    # class 1's rays are in the world frame: pts = t_wc + R_wc d z (origin_dirs_W), to the rounding of inverting inv(T_wc)
    T = pools[1]["T_wc"][rows[1]]
    o_w, d_w = O.origin_dirs_W(T, pools[1]["dirs"][rows[1]])
    assert rel_l2(b["pts"][1], o_w[:, None, :] + d_w[:, None, :] * b["z"][1][..., None]) < 1e-5
    mlp, B, shape, tex = _oracle_params(cnr, tr, theta0)
    mlp = {k: v.requires_grad_() for k, v in mlp.items()}
    B.requires_grad_()
    sh = [shape[c, :n_objs[c]].clone().requires_grad_() for c in range(C)]
    tx = [tex[c, :n_objs[c]].clone().requires_grad_() for c in range(C)]
    idx = torch.stack([pools[c]["indices"][rows[c]] for c in range(C)])
    batch = dict(pts=b["pts"], z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"], labels=b["labels"],
                 depth_mask=b["depth_mask"].bool(), indices=idx)
    loss, aux = O.forward_loss(mlp, B, cfg.obj_scale, sh, tx, batch)
    loss.backward()
    for k, name in enumerate(("loss_depth", "loss_color", "loss_opacity")):
        assert rel_l2(tr.losses[k].cpu(), aux[name]) < 2e-3, name
    gk = tr.lay.views(tr.grad.cpu())
    for c in range(C):
        k = n_objs[c]
        assert rel_l2(gk["shape"][c, :k], sh[c].grad) < 0.1 and rel_l2(gk["tex"][c, :k], tx[c].grad) < 0.1, c
        assert float(gk["shape"][c, k:].abs().sum()) == 0.0 and float(gk["tex"][c, k:].abs().sum()) == 0.0   # padding rows
    # the single-object class carries no regulariser: its code gradient is the data term alone -- with the three-object
    # class's reg_scale code / |code| on top it would be off by ~50 % here
    assert rel_l2(gk["latW"], torch.stack([mlp[n + ".weight"].grad for n in cnr.ops.LATENT_LAYERS], 1)) < 0.1
    assert rel_l2(gk["B"], B.grad) < 0.1
    # export / import of one class in the reference's checkpoint keys (codes: the class's real rows only)
    sd = tr.state_dicts(1)
    assert sd["shape_code_state_dict"]["weight"].shape == (1, L)
    tr2 = cnr.fused.FusedCategoryTrainer(cfg, C, n_objs, pools, R, dev, seed=9, generator=torch.Generator().manual_seed(99),
                                         use_graph=False)
    assert not torch.equal(tr2.theta[1], tr.theta[1])
    for _ in range(5):          # an optimiser with history: moments and a step count of 5
        tr2.step()
    # default: the loaded class alone starts afresh -- class 0 keeps its AdamW history and the step counter keeps running
    m0, s0 = tr2.exp_avg[0].clone(), tr2.d_state2[:, 2].tolist()
    tr2.load_state_dicts(sd, 1)
    assert float(tr2.exp_avg[1].abs().max()) == 0.0 and float(tr2.exp_avg_sq[1].abs().max()) == 0.0
    assert torch.equal(tr2.exp_avg[0], m0) and float(m0.abs().max()) > 0.0 and tr2.d_state2[:, 2].tolist() == s0
    # reset="all": the reference's resume (a new AdamW for every class, train.py:40,66-68)
    tr2.load_state_dicts(sd, 1, reset="all")
    v1, v2 = tr.lay.views(tr.theta), tr2.lay.views(tr2.theta)
    for k in ("trunk", "latW", "latb", "B"):
        assert torch.equal(v1[k][1], v2[k][1]), k
    assert torch.equal(v1["shape"][1, :1], v2["shape"][1, :1]) and torch.equal(v1["tex"][1, :1], v2["tex"][1, :1])
    # a resume starts a FRESH AdamW (the reference builds a new optimiser and saves no moments): moments zero, step counter 0,
    # so that the first update after the load is a first AdamW step: |delta| = lr (1 + wd |theta|) wherever the gradient is
    # not noise -- with the old step count bias correction would have shrunk it by 1 - beta1^6 = 0.47
    assert float(tr2.exp_avg.abs().max()) == 0.0 and float(tr2.exp_avg_sq.abs().max()) == 0.0
    assert tr2.d_state2[:, 2].tolist() == [0, 0]
    before = tr2.theta.clone()
    tr2.step()
    torch.cuda.synchronize()
    g2 = tr2.grad[1]
    big = g2.abs() > 0.05 * g2.abs().max()
    delta = (tr2.theta[1] - before[1]).abs()[big]
    assert float((delta / cfg.learning_rate).min()) > 0.9 and float((delta / cfg.learning_rate).max()) < 1.1


@pytest.mark.parametrize("C,R,slices,perm_on", [(1, 2048, 7, True), (3, 100, 5, False), (9, 33, 4, True)])
def test_slice_mask_count_table_against_torch(cnr, dev, C, R, slices, perm_on):
    """cnr_slice_maskcounts: per slice and class #(valid depth & label != 0), #(label != 0), #(label != 2) and the
    any-class-empty flags (src/render_rays.py:66-72, src/loss.py:24-26) -- bit-exact against torch and against
    parallel.mask_count_table, including a slice in which one class has no ray on its object."""
    _C = cnr._C
    gen = torch.Generator().manual_seed(C * R)
    pool_rows = slices * R + 17
    pools = [cnr.scene_cateogries.synthetic_pool(pool_rows, 4, gen, "cpu") for _ in range(C)]
    pools[C - 1]["rgbs"][:, 3] = torch.where(torch.arange(pool_rows) < 2 * R, torch.zeros(pool_rows, dtype=torch.uint8),
                                             pools[C - 1]["rgbs"][:, 3])           # without perm: slices 0, 1 of the last class are empty
    st = lambda k: torch.stack([p[k] for p in pools]).to(dev).contiguous()
    rgbs, depth = st("rgbs"), st("depth")
    perm = torch.stack([torch.randperm(pool_rows, generator=gen) for _ in range(C)]).to(torch.int32).to(dev) if perm_on else None
    tab = torch.empty(slices, C + 1, 4, device=dev)
    _C.call("cnr_slice_maskcounts", rgbs, depth, perm, pool_rows, C, R, slices, 0.0, tab)
    for s_ in range(slices):
        rows = [(perm[c, s_ * R:(s_ + 1) * R].long() if perm_on else torch.arange(s_ * R, (s_ + 1) * R, device=dev)) for c in range(C)]
        labels = torch.stack([rgbs[c][rows[c], 3] for c in range(C)]).cpu()
        dmask = torch.stack([depth[c][rows[c]] > 0.0 for c in range(C)]).cpu()
        want = cnr.parallel.mask_count_table(labels, dmask)
        assert torch.equal(tab[s_].cpu(), want), s_
    if not perm_on:
        assert tab[0, C, 0] == 1 and tab[0, C, 1] == 1 and tab[slices - 1, C, 1] == 0


def test_epoch_perm_is_a_keyed_bijection(dev):
    """cnr_epoch_perm (the epoch shuffle in one launch): every row is a permutation of [0, n) for powers of two, their neighbours
    and small n; a function of (seed, epoch, class id) only -- the same class id gives the same order wherever it sits in
    class_ids, as class shards need --; different epochs / classes / seeds give different orders; no fixed structure (an epoch's
    order is not the previous one shifted); the cursor of the step state is set by the same launch."""
    import cnr_amd as cnr
    _C = cnr._C
    for n in (1, 2, 7, 64, 1000, 4096, 4097, 131072, 100003):
        ids = torch.tensor([5, 0, 9], device=dev, dtype=torch.int32)
        perm = torch.full((3, n), -1, device=dev, dtype=torch.int32)
        state = torch.tensor([123, 4, 5], device=dev, dtype=torch.int64)
        _C.call("cnr_epoch_perm", perm, n, 3, 77, 2, ids, state, 640)
        torch.cuda.synchronize()
        assert state.tolist() == [640, 4, 5]
        ar = torch.arange(n, device=dev, dtype=torch.int32)
        for c in range(3):
            assert torch.equal(perm[c].sort().values, ar), (n, c)
        if n >= 64:
            assert not torch.equal(perm[0], perm[1]) and not torch.equal(perm[1], perm[2])
            assert float((perm[0] == ar).float().mean()) < 0.2                     # not the identity
        # class 0 alone, no id table (c = class id), other position: the same order as row 1 above
        solo = torch.empty(1, n, device=dev, dtype=torch.int32)
        _C.call("cnr_epoch_perm", solo, n, 1, 77, 2, None, None, 0)
        assert torch.equal(solo[0], perm[1]), n
        if n >= 1000:
            other = torch.empty(1, n, device=dev, dtype=torch.int32)
            _C.call("cnr_epoch_perm", other, n, 1, 77, 3, None, None, 0)           # next epoch
            assert not torch.equal(other[0], solo[0])
            assert float((other[0] == solo[0]).float().mean()) < 0.01
            d = (other[0].long() - solo[0].long()) % n
            assert int(d.unique().numel()) > n // 4                                  # not a rotation of the previous epoch
            _C.call("cnr_epoch_perm", other, n, 1, 78, 2, None, None, 0)           # another seed
            assert float((other[0] == solo[0]).float().mean()) < 0.01
            # neighbours in index space land far apart: mean |perm[i + 1] - perm[i]| of a random order is n / 3
            step = (solo[0][1:].long() - solo[0][:-1].long()).abs().float().mean()
            assert 0.25 * n < float(step) < 0.42 * n, (n, float(step))
