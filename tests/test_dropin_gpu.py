"""The benchmarked fused iteration behind the reference's own objects (VERDICT r03 row h; train.py:33-96 builds them,
:98-201 is the loop ``FullStepTrainer.from_scene`` replaces, :196-201 the copy-back ``sync_to_modules`` stands for).

The categories come out of the ``pool_w24_h20_f6`` fixture exactly as train.py:46-64 would build them: ``sceneCategory(cfg,
cls_id, inst_dict, sample_dict, rays)`` for a three-instance category, a single-instance category (world frame, no code
regulariser) and the background -- three pools of three different lengths."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr():
    import cnr_amd
    return cnr_amd


def _scene(cnr, dev, hidden_bg=128, seed=5):
    """(cfg, cls_dict, scene_bg): what train.py:33-64 holds after its construction loop"""
    z = np.load(os.path.join(GOLDEN, "pool_w24_h20_f6.npz"))
    frames = [int(f) for f in z["frame_ids"]]
    sample_dict = {f: dict(image=z["images"][i], depth=z["depths"][i], T=z["T_wc"][i], obj_mask=z["masks"][i])
                   for i, f in enumerate(frames)}
    inst = {}
    for k, iid in enumerate(int(i) for i in z["inst_ids"]):
        inst[iid] = dict(T_obj=z["T_obj"][k], bbox3D=SimpleNamespace(extent=np.array([1.0, 2.0, 1.5])),
                         frame_info=[dict(frame=int(f), bbox=[int(v) for v in b])
                                     for f, b in zip(z["obj_frames"][k], z["obj_bboxes"][k])])
    bg_dict = dict(bbox3D=SimpleNamespace(extent=np.array([6.0, 6.0, 3.0])),
                   frame_info=[dict(frame=int(f), bbox=[int(v) for v in b]) for f, b in zip(z["bg_frames"], z["bg_bboxes"])])
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=28)
    cfg.hidden_feature_size_bg, cfg.n_per_optim, cfg.n_per_optim_bg = hidden_bg, 4, 64
    cfg.n_bins_cam2surface_bg = 4
    rays = torch.from_numpy(z["rays_dir"])
    ids = list(inst.keys())
    np.random.seed(seed)
    torch.manual_seed(seed)
    cls_dict = {5: cnr.scene_cateogries.sceneCategory(cfg, 5, inst, sample_dict, rays),
                9: cnr.scene_cateogries.sceneCategory(cfg, 9, {ids[0]: inst[ids[0]]}, sample_dict, rays)}
    scene_bg = cnr.scene_cateogries.sceneCategory(cfg, 0, bg_dict, sample_dict, rays)
    return cfg, cls_dict, scene_bg


def _module_tensors(sc):
    t = sc.trainer
    out = {"fc." + k: v.detach().clone() for k, v in t.fc_occ_map.state_dict().items()}
    out["B"] = t.pe.B_layer.weight.detach().clone()
    if sc.cls_id != 0:
        out["shape"], out["tex"] = t.shape_codes.weight.detach().clone(), t.texture_codes.weight.detach().clone()
    return out


def test_from_scene_is_the_fused_trainer_on_the_categories_own_pools(cnr, dev):
    """Same pools, same initial parameters -> the first steps of ``from_scene`` ARE ``FusedCategoryTrainer``'s, bit for bit; the
    reference's batch size (train.py:92-96); pools of different lengths; the single-instance category in the world frame."""
    cfg, cls_dict, scene_bg = _scene(cnr, dev)
    full = cnr.background.FullStepTrainer.from_scene(cls_dict, scene_bg, cfg, seed=3)
    tr = full.obj
    cats = list(cls_dict.values())
    assert tr.R == (3 + 1) * cfg.n_per_optim // 2 and tr.n_obj_list == [3, 1] and tr.world_frame == [False, True]
    assert tr.pool_rows_cls == [c.rgbs_batch_all.shape[0] for c in cats] and tr.ragged
    assert full.bg.trainer is scene_bg.trainer and full.bg.R == cfg.n_per_optim_bg and full.bg.precision == "fused"
    # the modules' values are the trainer's initial parameters
    for k, c in enumerate(cats):
        sd = tr.state_dicts(k)
        for n, v in c.trainer.fc_occ_map.state_dict().items():
            assert torch.equal(sd["FC_state_dict"][n], v), n
        assert torch.equal(sd["shape_code_state_dict"]["weight"], c.trainer.shape_codes.weight)
        assert torch.equal(sd["texture_code_state_dict"]["weight"], c.trainer.texture_codes.weight)
    # a FusedCategoryTrainer built by hand on the same pools
    pools = [cnr.background.FullStepTrainer.pool_of(c) for c in cats]
    ref = cnr.fused.FusedCategoryTrainer(cfg, 2, [3, 1], pools, tr.R, dev, seed=3, world_frame=[False, True])
    for k, c in enumerate(cats):
        ref.load_state_dicts(cnr.background.FullStepTrainer.module_state(c.trainer), k, reset=None)
    assert torch.equal(ref.perm, tr.perm)
    for s in range(6):
        full.step()
        ref.step()
        torch.cuda.synchronize()
        assert torch.equal(tr.losses, ref.losses), s
        assert torch.equal(tr.theta, ref.theta), s
    assert torch.isfinite(tr.losses).all() and float(tr.losses.abs().sum()) > 0


def test_ragged_pools_walk_each_class_through_its_own_epochs(cnr, dev):
    """src/scene_cateogries.py:436-449 per category: slices of n rows, a reshuffle of THAT category's pool after
    ceil(N_c / n) - 1 slices.  The permutation rows the kernels read must therefore be, class by class, a sequence of
    epochs, each a run of distinct rows of the class's own pool -- across the host's (longest-class) reshuffles too."""
    cfg, cls_dict, scene_bg = _scene(cnr, dev)
    full = cnr.background.FullStepTrainer.from_scene(cls_dict, None, cfg, seed=11, use_graph=False)
    tr = full.obj
    Rg = tr.Rg
    K = -(-tr.pool_rows // Rg) - 1
    seq = [[] for _ in range(tr.C)]
    for epoch in range(5):                               # five host epochs of K steps each
        tr._pre_step()                                   # (the reshuffle is lazy: it happens in front of the epoch's first step)
        p = tr.perm.cpu()
        for c in range(tr.C):
            seq[c].append(p[c, :K * Rg].clone())
        for _ in range(K):
            full.step()
        assert tr.cursor >= tr.pool_rows - Rg           # the next step reshuffles
    torch.cuda.synchronize()
    assert torch.isfinite(tr.losses).all()
    for c, N in enumerate(tr.pool_rows_cls):
        rows = torch.cat(seq[c])
        assert int(rows.min()) >= 0 and int(rows.max()) < N, c       # never a padding row
        k_c = -(-N // Rg) - 1
        per = k_c * Rg
        n_ep = rows.numel() // per
        assert n_ep >= 5
        orders = []
        for e in range(n_ep):
            piece = rows[e * per:(e + 1) * per]
            assert piece.unique().numel() == per, (c, e)               # one epoch never reads a row twice
            orders.append(piece)
        assert not torch.equal(orders[0], orders[1])                   # and every epoch is a new order
    # the longest class: exactly one epoch per host epoch (the uniform case)
    c_long = max(range(tr.C), key=lambda c: tr.pool_rows_cls[c])
    assert -(-tr.pool_rows_cls[c_long] // Rg) - 1 == K


@pytest.mark.parametrize("hidden_bg", [128, 32])
def test_train_sync_checkpoint_round_trip(cnr, dev, hidden_bg, tmp_path):
    """N fused iterations, ``sync_to_modules()`` (train.py:196-201), then the reference's own consumers of the modules:
    ``save_checkpoints`` -> ``load_checkpoints`` into freshly built categories reproduces every tensor, and ``eval_points``
    runs on the trained weights.  hidden 128: the fused background step; 32: the exact-fp32 tier takes over."""
    cfg, cls_dict, scene_bg = _scene(cnr, dev, hidden_bg)
    before = {cid: _module_tensors(sc) for cid, sc in list(cls_dict.items()) + [(0, scene_bg)]}
    full = cnr.background.FullStepTrainer.from_scene(cls_dict, scene_bg, cfg, seed=1)
    assert full.bg.precision == ("fused" if hidden_bg == 128 else "fp32")
    full.run(40)
    ld = full.loss_dict()
    assert ld["depth"].shape == (2,) and ld["background"]["color"].shape == (1,)
    assert all(torch.isfinite(v).all() for v in (ld["depth"], ld["color"], ld["opacity"], *ld["background"].values()))
    # before the copy-back the categories' modules still hold the initial values; the background trained in place
    for cid, sc in cls_dict.items():
        assert all(torch.equal(v, before[cid][k]) for k, v in _module_tensors(sc).items())
    assert not torch.equal(scene_bg.trainer.fc_occ_map.in_layer[0].weight, before[0]["fc.in_layer.0.weight"])
    full.sync_to_modules()
    for k, (cid, sc) in enumerate(cls_dict.items()):
        now = _module_tensors(sc)
        moved = [n for n, v in now.items() if not torch.equal(v, before[cid][n])]
        assert len(moved) == len(now), set(now) - set(moved)            # every tensor trained
        sd = full.obj.state_dicts(k)
        assert torch.equal(now["B"], sd["PE_state_dict"]["B_layer.weight"])
        assert torch.equal(now["shape"], sd["shape_code_state_dict"]["weight"])
    # the reference's checkpoint round trip on the synced modules (tests/test_checkpoint.py's loader path)
    cfg2, cls2, bg2 = _scene(cnr, dev, hidden_bg, seed=6)
    for cid, sc in list(cls_dict.items()) + [(0, scene_bg)]:
        f = sc.save_checkpoints(str(tmp_path), 40)
        fresh = cls2[cid] if cid else bg2
        fresh.load_checkpoints(f, allow_pickle=cid == 0)      # (the background's `bound` is a box object, here a namespace)
        a, b = _module_tensors(sc), _module_tensors(fresh)
        assert all(torch.equal(a[n], b[n]) for n in a), cid
        assert fresh.start == 40
    # eval_points (src/trainer.py:125-151) reads the trained modules
    pts = (torch.rand(4096, 3, device=dev) - 0.5)
    sc5 = cls_dict[5]
    trained = sc5.trainer.eval_points(pts, sc5.obj_ids[1])
    fresh5 = _scene(cnr, dev, hidden_bg, seed=5)[1][5].trainer.eval_points(pts, sc5.obj_ids[1])
    assert trained is not None and fresh5 is not None and rel_l2(trained[0], fresh5[0]) > 1e-3
    # ... and a resume: the loaded modules become the fused trainers' parameters, with a fresh optimiser
    full2 = cnr.background.FullStepTrainer.from_scene(cls2, bg2, cfg2, seed=1)
    assert torch.equal(full2.obj.theta, full.obj.theta)
    if hidden_bg == 128:
        assert torch.equal(full2.bg.flat, full.bg.flat)


def test_full_run_with_an_odd_unroll_is_the_same_training(cnr, dev):
    """``FullStepTrainer.run(n, unroll=5)`` used to capture five iterations per graph and leave the parameter
    ping-pong on the wrong copy.  Group sizes are even now: any ``unroll`` trains bitwise what ``n`` calls of ``step()`` train."""
    res = []
    for mode in ("step", 5, 7, 8):
        cfg, cls_dict, scene_bg = _scene(cnr, dev)
        full = cnr.background.FullStepTrainer.from_scene(cls_dict, scene_bg, cfg, seed=2)
        n = 37
        if mode == "step":
            for _ in range(n):
                full.step()
        else:
            full.run(n, unroll=mode)
        torch.cuda.synchronize()
        res.append((full.obj.theta.clone(), full.bg.flat.clone(), full.obj.losses.clone(), full.obj.parity, full.obj.cursor))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
        assert r[3:] == res[0][3:]
