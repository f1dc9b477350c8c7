"""CPU: host logic (autograd Functions, functorch vmap rules, modules, loss assembly, trainer bookkeeping) with the
kernels replaced by the oracle-backed test double -- the reference's train.py:49-64,88-89,136-184 flow on our
modules must reproduce the reference's golden losses and gradients."""
import pytest
import torch

from conftest import Golden, rel_l2


@pytest.fixture()
def cnr():
    import cnr_amd
    from cpu_double import Double
    cnr_amd._C.install_test_double(Double())
    yield cnr_amd
    cnr_amd._C.install_test_double(None)


@pytest.mark.parametrize("name", ["s0_c2_r64_s16_l32", "edge_empty_mask", "s0_c1_r120_s10_l256"])
def test_reference_flow_under_vmap(cnr, name):
    from test_parity_gpu import _build_reference_style_step
    g = Golden(name, "cpu")
    trainers, opt, names, fc_param, pe_param, alpha, color, loss, ld = _build_reference_style_step(cnr, g, torch.device("cpu"))
    assert alpha.shape == (g.C, g.R, g.S, 1) and color.shape == (g.C, g.R, g.S, 3)
    assert rel_l2(alpha, g.t("sigmas")) < 1e-5 and rel_l2(color, g.t("rgbs")) < 1e-5
    assert rel_l2(loss, g.t("loss")) < 1e-5
    for k in ("depth", "color", "opacity"):
        assert rel_l2(ld[k], g.t("loss_" + k)) < 1e-5
    loss.backward()
    for n, p in zip(names, fc_param):
        ref = g.t("grad." + n)
        got = torch.zeros_like(ref) if p.grad is None else p.grad
        assert rel_l2(got, ref) < 1e-4, n
    assert rel_l2(pe_param[0].grad, g.t("grad_B")) < 1e-4
    assert rel_l2(torch.stack([t.shape_codes.weight.grad for t in trainers]), g.t("grad_shape_codes")) < 1e-4


def test_state_dict_names_match_reference_checkpoints(cnr):
    """FC_state_dict / PE_state_dict keys (src/scene_cateogries.py:548-571; names listed in SURVEY section 8(a))."""
    cfg = cnr.cfg.synthetic_config(device="cpu", latent_dim=32)
    t = cnr.trainer.Trainer(cfg, 1, [3, 7])
    keys = list(t.fc_occ_map.state_dict().keys())
    assert keys[:2] == ["encoding_xyz.0.weight", "encoding_xyz.0.bias"]
    assert "shape_latent_layer_2.0.weight" in keys and "rgb.2.bias" in keys and len(keys) == 28
    assert sum(p.numel() for p in t.fc_occ_map.parameters()) == 13892 + 4 * (32 * 32 + 32)
    assert list(t.pe.state_dict().keys()) == ["scale", "B_layer.weight"]
    assert t.inst_id_to_index == {3: 0, 7: 1} and t.n_obj == 2 and (t.emb_size1, t.emb_size2) == (87, 42)
    assert t.shape_codes.weight.shape == (2, 32)


def test_flat_layout_and_export(cnr):
    lay = cnr.fused.ParamLayout(256, 4)
    assert lay.total == 13892 + 4 * 32 * 256 + 128 + 63 + 2 * 4 * 256
    flat, _ = cnr.fused.init_params(2, 256, 4, torch.Generator().manual_seed(0))
    v = lay.views(flat)
    assert v["trunk"].shape == (2, 13892) and v["latW"].shape == (2, 4, 32, 256) and v["B"].shape == (2, 21, 3)
    v["B"][1, 0, 0] = 123.0
    assert flat[1, lay.B[0]] == 123.0          # views alias the flat buffer


def test_bias_rows_fold(cnr):
    """W (a + z) + b == W a + (W z + b): the identity the fused kernels rely on, checked on the fp32 oracle."""
    from oracle import ref_cpu as O
    gen = torch.Generator().manual_seed(0)
    p = O.init_codenerf_params(1, 32, 32, gen)
    parts = []
    for n, _, _ in cnr.ops.TRUNK_LAYERS:
        parts += [p[n + ".weight"].reshape(1, -1), p[n + ".bias"].reshape(1, -1)]
    trunk = torch.cat(parts, 1)
    zlat = torch.rand(1, 5, 4, 32, generator=gen)
    br = cnr.ops.bias_rows(trunk, zlat)
    W, b = p["shape_layer_2.0.weight"][0], p["shape_layer_2.0.bias"][0]
    assert torch.allclose(br[0, :, 2], zlat[0, :, 2] @ W.T + b, atol=1e-6)
    Wc, bc = p["cat_layer.0.weight"][0], p["cat_layer.0.bias"][0]
    assert torch.allclose(br[0, :, 1], zlat[0, :, 1] @ Wc[:, :32].T + bc, atol=1e-6)


def test_synthetic_pool_and_shards(cnr):
    gen = torch.Generator().manual_seed(3)
    pool = cnr.scene_cateogries.synthetic_pool(1000, 4, gen, "cpu")
    assert pool["rgbs"].dtype == torch.uint8 and pool["T_co"].shape == (1000, 4, 4)
    frac_invalid = float((pool["depth"] == 0).float().mean())
    assert 0.01 < frac_invalid < 0.12
    lo, hi = zip(*[cnr.parallel.shard_rows(1000, r, 3) for r in range(3)])
    assert lo[0] == 0 and hi[-1] == 1000 and all(h == l2 for h, l2 in zip(hi[:-1], lo[1:]))
    sh = cnr.parallel.shard_pool(pool, 1, 3)
    assert torch.equal(sh["depth"], pool["depth"][lo[1]:hi[1]])


def test_checkpoint_round_trip_in_the_reference_format(cnr, tmp_path):
    """save_checkpoints / load_checkpoints (src/scene_cateogries.py:548-597): file name, dict keys and state_dict key
    names of the reference, for an object category and for the background model; weights survive the round trip and
    the FC_state_dict keys are exactly the reference's (taken from the golden fixtures it generated)."""
    cfg = cnr.cfg.synthetic_config(device="cpu", latent_dim=32)
    cfg.hidden_feature_size_bg = 32
    gen = torch.Generator().manual_seed(3)
    pool = cnr.scene_cateogries.synthetic_pool(256, 3, gen, "cpu")
    for cls_id, obj_ids, fixture in ((4, [11, 12, 13], "s0_c2_r64_s16_l32"), (0, [0], "bg_r100_s14_h32")):
        a = cnr.scene_cateogries.sceneCategory.from_pool(cfg, cls_id, obj_ids, pool)
        b = cnr.scene_cateogries.sceneCategory.from_pool(cfg, cls_id, obj_ids, pool, seed=1)
        a.object_tensor_dict = {i: torch.eye(4) for i in obj_ids}
        a.trainer.extent_dict = {i: [1.0, 1.0, 1.0] for i in obj_ids}
        a.trainer.bound = "bbox"
        f = a.save_checkpoints(str(tmp_path), 42)
        assert f.endswith(f"cls_{cls_id}_iteration_00042.pth")
        ck = torch.load(f, weights_only=False)
        want = {"global_step", "PE_state_dict", "FC_state_dict", "cls_id", "instance_id_to_index", "obj_scale", "bound"}
        if cls_id != 0:
            want |= {"obj_tensor_dict", "extent_dict", "shape_code_state_dict", "texture_code_state_dict"}
        assert set(ck.keys()) == want
        assert sorted(ck["FC_state_dict"].keys()) == sorted(Golden(fixture).mlp().keys())
        assert list(ck["PE_state_dict"].keys()) == ["scale", "B_layer.weight"]
        assert not torch.equal(next(iter(a.trainer.fc_occ_map.parameters())), next(iter(b.trainer.fc_occ_map.parameters())))
        b.load_checkpoints(f)
        for (n, p), (_, q) in zip(a.trainer.fc_occ_map.state_dict().items(), b.trainer.fc_occ_map.state_dict().items()):
            assert torch.equal(p, q), n
        assert b.start == 42 and b.trainer.inst_id_to_index == a.trainer.inst_id_to_index
        if cls_id != 0:
            assert torch.equal(a.trainer.shape_codes.weight, b.trainer.shape_codes.weight)


def test_camera_rays_dirs_match_the_oracle(cnr):
    """a1, cameraInfo.get_rays_dirs (src/scene_cateogries.py:613-629): [(u - cx) / fx, (v - cy) / fy, 1] indexed [w, h],
    not normalised -- against the oracle's restatement, for the Replica intrinsics and a cropped ScanNet-like camera."""
    from oracle import ref_cpu as O
    from types import SimpleNamespace
    for cam in (dict(W=1200, H=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5),
                dict(W=620, H=460, fx=577.87, fy=577.87, cx=309.5, cy=229.5), dict(W=7, H=5, fx=3.0, fy=2.0, cx=1.25, cy=4.0)):
        info = cnr.scene_cateogries.cameraInfo(SimpleNamespace(**cam))
        want = O.get_rays_dirs(cam["W"], cam["H"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        assert info.rays_dir_cache.shape == (cam["W"], cam["H"], 3)
        assert torch.equal(info.rays_dir_cache, want)
        assert torch.equal(info.rays_dir_cache[..., 2], torch.ones(cam["W"], cam["H"]))
        with pytest.raises(Exception):
            info.get_rays_dirs(depth_type="euclidean")


def test_camera_rays_dirs_match_the_reference_fixture(cnr):
    """... and against what the reference's own cameraInfo produced (tests/golden/cam_rays.npz)."""
    import os
    import numpy as np
    from types import SimpleNamespace
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cam_rays.npz"))
    for i in range(2):
        W, H, fx, fy, cx, cy = z["cam%d" % i]
        info = cnr.scene_cateogries.cameraInfo(SimpleNamespace(W=int(W), H=int(H), fx=fx, fy=fy, cx=cx, cy=cy))
        assert np.array_equal(info.rays_dir_cache.numpy(), z["dirs%d" % i])


def test_multi_step_graph_sizes_cover_every_even_length(cnr):
    """run() sends the largest captured group that fits the request and what is left of the epoch; with every even size there an
    epoch end inside a run costs two extra launches (n - 1 steps as one graph + one step), not a halving ladder of them."""
    gs = cnr.fused.FusedCategoryTrainer._group_sizes
    assert gs(20) == list(range(20, 1, -2))
    assert gs(5) == [4, 2] and gs(2) == [2] and gs(1) == []
    for left in range(2, 33):                      # what goes out in front of an epoch end `left` steps away
        first = next(u for u in gs(32) if u <= left)
        assert left - first in (0, 1)
