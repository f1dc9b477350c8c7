"""a10 (init_weights: xavier-normal weights, nn.Linear default biases; src/model.py:4-6, src/trainer.py:40,57-58) and a1
(cameraInfo.get_rays_dirs; src/scene_cateogries.py:613-629).

CPU: ``fused.init_params`` restates the initialisation by hand -- its per-layer distribution is checked against the arrays
the REFERENCE initialised (the ``mlp.*`` / code tables in tests/golden/*.npz) and against torch's own initialisers.
GPU: the ray-direction cache built on the device equals the reference's bit for bit and feeds cnr_gather_pool."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, Golden, golden_names


def _flat_layers(cnr, theta, lay, C):
    v = lay.views(theta)
    out, off = {}, 0
    for n, o, i in cnr.ops.TRUNK_LAYERS:
        out[n + ".weight"] = v["trunk"][:, off:off + o * i].reshape(C, o, i); off += o * i
        out[n + ".bias"] = v["trunk"][:, off:off + o]; off += o
    for k, n in enumerate(cnr.ops.LATENT_LAYERS):
        out[n + ".weight"], out[n + ".bias"] = v["latW"][:, k], v["latb"][:, k]
    return out, v


def test_init_params_matches_the_reference_initialised_arrays():
    import cnr_amd as cnr
    C = 64                                     # many classes -> tight statistics per layer
    for L in (256, 32):
        theta, lay = cnr.fused.init_params(C, L, 4, torch.Generator().manual_seed(5))
        mine, v = _flat_layers(cnr, theta, lay, C)
        ref = {}
        for name in golden_names():             # pool every fixture of this latent size: reference-initialised tensors
            g = Golden(name)
            if g.L != L:
                continue
            for k, t in g.mlp().items():
                ref.setdefault(k, []).append(t.reshape(-1, *t.shape[1:]))
        assert ref, L
        assert set(mine) == set(ref)
        for k, parts in ref.items():
            r = torch.cat(parts)
            m = mine[k]
            assert m.shape[1:] == r.shape[1:], k
            if k.endswith(".weight"):
                o, i = m.shape[1:]
                std = math.sqrt(2.0 / (i + o))                       # xavier_normal_, gain 1
                assert abs(float(m.std()) / std - 1) < 0.03, (k, float(m.std()), std)
                assert abs(float(m.mean())) < 4 * std / math.sqrt(m.numel()), k
                # the reference's own draw of the same layer is a sample of that distribution: its std must agree with
                # ours within the sampling error of ITS size (std error of a normal sample ~ std / sqrt(2 n))
                tol = 5 / math.sqrt(2 * r.numel()) + 0.03
                assert abs(float(r.std()) / float(m.std()) - 1) < tol, (k, float(r.std()), float(m.std()))
                # normal, not uniform: kurtosis 3 (uniform would give 1.8)
                kurt = float(((m - m.mean()) ** 4).mean() / m.var() ** 2)
                assert 2.8 < kurt < 3.2, (k, kurt)
            else:
                i = mine[k[:-5] + ".weight"].shape[2]
                bound = 1 / math.sqrt(i)                              # nn.Linear default: U(-1/sqrt(fan_in), ..)
                assert float(m.abs().max()) <= bound and float(r.abs().max()) <= bound * (1 + 1e-6), k
                assert float(m.abs().max()) > 0.97 * bound, k
                assert abs(float(m.std()) / (bound / math.sqrt(3)) - 1) < 0.05, k       # uniform: std = bound / sqrt(3)
                kurt = float(((m - m.mean()) ** 4).mean() / m.var() ** 2)
                assert 1.6 < kurt < 2.0, (k, kurt)
        # codes: randn / sqrt(L / 2) (src/trainer.py:57-58); B: the 21 icosahedral directions
        for key in ("shape", "tex"):
            assert abs(float(v[key].std()) * math.sqrt(L / 2) - 1) < 0.03
        for name in golden_names():
            g = Golden(name)
            if g.L == L:
                sc = g.t("shape_codes")
                assert abs(float(sc.std()) * math.sqrt(L / 2) - 1) < 5 / math.sqrt(2 * sc.numel()) + 0.03
        assert torch.equal(v["B"][0], torch.tensor(cnr.embedding.UNIDIRS, dtype=torch.float32).view(21, 3))


def test_init_params_against_torch_initialisers():
    """The same statistics from torch.nn.init.xavier_normal_ / nn.Linear(...).bias on a module of the reference's shape."""
    import cnr_amd as cnr
    torch.manual_seed(0)
    fc = cnr.model.CodeNeRF(87, 42, latent_dim=256, W=32, shape_blocks=2, texture_blocks=1)
    fc.apply(cnr.model.init_weights)
    theta, lay = cnr.fused.init_params(32, 256, 4, torch.Generator().manual_seed(1))
    mine, _ = _flat_layers(cnr, theta, lay, 32)
    sd = fc.state_dict()
    assert set(sd) == set(mine)
    for k, t in sd.items():
        assert tuple(t.shape) == tuple(mine[k].shape[1:]), k
        if k.endswith("weight") and t.numel() >= 1024:
            assert abs(float(t.std()) / float(mine[k].std()) - 1) < 0.08, k


@pytest.mark.gpu
def test_camera_ray_cache_on_device_matches_the_reference(dev):
    """a1 on the device: cameraInfo(cfg, device) builds the (W,H,3) cache with device ops; bit-equal to the reference's."""
    import cnr_amd as cnr
    from types import SimpleNamespace
    z = np.load(os.path.join(GOLDEN, "cam_rays.npz"))
    for i in range(2):
        W, H, fx, fy, cx, cy = z["cam%d" % i]
        info = cnr.scene_cateogries.cameraInfo(SimpleNamespace(W=int(W), H=int(H), fx=fx, fy=fy, cx=cx, cy=cy), device=dev)
        assert info.rays_dir_cache.device.type == "cuda"
        assert np.array_equal(info.rays_dir_cache.cpu().numpy(), z["dirs%d" % i])
    # the bench's synthetic camera (config_replica_room0.json:48-57) against the oracle
    from oracle import ref_cpu as O
    info = cnr.scene_cateogries.cameraInfo(SimpleNamespace(W=1200, H=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5), device=dev)
    assert torch.equal(info.rays_dir_cache.cpu(), O.get_rays_dirs(1200, 680, 600.0, 600.0, 599.5, 339.5))


@pytest.mark.gpu
def test_device_ray_cache_feeds_pool_construction(dev):
    """... and the device cache goes straight into cnr_gather_pool: the pool's directions are rows of it."""
    import cnr_amd as cnr
    from types import SimpleNamespace
    from conftest import pool_golden_names
    name = pool_golden_names()[0]
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    W, H = z["images"].shape[1:3]
    info = cnr.scene_cateogries.cameraInfo(SimpleNamespace(W=W, H=H, fx=20.0, fy=21.0, cx=W / 2 - 0.5, cy=H / 2 - 0.5), device=dev)
    frames = [int(f) for f in z["frame_ids"]]
    sample_dict = {f: dict(image=z["images"][i], depth=z["depths"][i], T=z["T_wc"][i], obj_mask=z["masks"][i])
                   for i, f in enumerate(frames)}
    inst_dict = {}
    for k, iid in enumerate(int(i) for i in z["inst_ids"]):
        inst_dict[iid] = dict(T_obj=z["T_obj"][k], bbox3D=SimpleNamespace(extent=np.array([1.0, 2.0, 1.5])),
                              frame_info=[dict(frame=int(f), bbox=[int(v) for v in b])
                                          for f, b in zip(z["obj_frames"][k], z["obj_bboxes"][k])])
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32)
    np.random.seed(777 + int(z["meta"][4]))
    sc = cnr.scene_cateogries.sceneCategory(cfg, 5, inst_dict, sample_dict, info.rays_dir_cache)
    # same frames, same crops, same shuffle as the fixture run -> the reference pool's pixels; directions now come from OUR cache:
    # every pool row's direction must be the cache entry of a pixel whose colour / depth match that row
    ref_dirs = torch.from_numpy(z["obj_dirs"])
    rays_ref = torch.from_numpy(z["rays_dir"]).reshape(-1, 3)
    # pixel index of each reference row = the row of the (random) reference cache it equals
    pix = torch.tensor([int((rays_ref == d).all(-1).nonzero()[0]) for d in ref_dirs])
    assert torch.equal(sc.ray_dirs_batch_all.cpu(), info.rays_dir_cache.cpu().reshape(-1, 3)[pix])
    assert torch.equal(sc.rgbs_batch_all.cpu(), torch.from_numpy(z["obj_rgbs"]))
