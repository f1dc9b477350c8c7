"""Checkpoint interchange with the reference (src/scene_cateogries.py:548-597).

tests/golden/ckpt_ref.npz holds what the REFERENCE's own ``sceneCategory.save_checkpoints`` wrote for three categories
it built itself (three instances / one instance / background), as plain arrays (tests/golden/gen_golden.py
run_ckpt_case), plus ``Trainer.eval_points`` of those weights.  Here a checkpoint file in the reference's schema is
rebuilt from the arrays, loaded by our ``load_checkpoints``, saved again by our ``save_checkpoints`` and compared key by
key; on the GPU the loaded weights must reproduce the reference's eval_points output."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

TAGS = ("multi", "single", "bg")


def _z():
    return np.load(os.path.join(GOLDEN, "ckpt_ref.npz"))


def _reference_checkpoint(z, tag):
    """the dict the reference's save_checkpoints wrote, from the fixture's arrays"""
    step, cls_id, obj_scale = z[tag + ".scalars"]
    ids = [int(i) for i in z[tag + ".inst_ids"]]
    sd = lambda name: {str(k): torch.from_numpy(z[f"{tag}.{name}.{k}"]) for k in z[f"{tag}.{name}.keys"]}
    ck = {"global_step": int(step), "PE_state_dict": sd("PE_state_dict"), "FC_state_dict": sd("FC_state_dict"),
          "cls_id": int(cls_id), "instance_id_to_index": {i: int(r) for i, r in zip(ids, z[tag + ".inst_rows"])},
          "obj_scale": float(obj_scale)}
    if tag == "bg":
        ck["bound"] = z[tag + ".bound"]
    else:
        ck["obj_tensor_dict"] = {i: torch.from_numpy(z[tag + ".obj_tensor"][k]) for k, i in enumerate(ids)}
        if tag + ".extent" in z.files:
            ck["extent_dict"] = {i: z[tag + ".extent"][k] for k, i in enumerate(ids)}
        ck["shape_code_state_dict"] = sd("shape_code_state_dict")
        ck["texture_code_state_dict"] = sd("texture_code_state_dict")
        ck["bound"] = {i: z[tag + ".bound"][k] for k, i in enumerate(ids)}
    assert sorted(ck.keys()) == [str(k) for k in z[tag + ".keys"]]
    return ck, ids


def _category(cnr, z, tag, device):
    cfg = cnr.cfg.synthetic_config(device=device, latent_dim=32)
    cfg.hidden_feature_size_bg, cfg.bg_scale = 32, 5.0
    ck, ids = _reference_checkpoint(z, tag)
    gen = torch.Generator().manual_seed(1)
    pool = cnr.scene_cateogries.synthetic_pool(64, max(len(ids), 1), gen, "cpu")
    sc = cnr.scene_cateogries.sceneCategory.from_pool(cfg, ck["cls_id"], ids, pool, seed=3)
    return sc, ck, ids


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), path
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif torch.is_tensor(a) or isinstance(a, np.ndarray):
        assert np.array_equal(np.asarray(a.cpu() if torch.is_tensor(a) else a), np.asarray(b.cpu() if torch.is_tensor(b) else b)), path
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("tag", TAGS)
def test_reference_checkpoint_loads_and_saves_back_identically(tag, tmp_path):
    import cnr_amd as cnr
    z = _z()
    sc, ck, ids = _category(cnr, z, tag, "cpu")
    f = os.path.join(str(tmp_path), "cls_%d_iteration_%05d.pth" % (ck["cls_id"], ck["global_step"]))
    torch.save(ck, f)
    sc.load_checkpoints(f)                                            # weights_only: nothing in the file is executed
    for name, mod in (("FC_state_dict", sc.trainer.fc_occ_map), ("PE_state_dict", sc.trainer.pe)):
        _same(ck[name], dict(mod.state_dict()), name)
    assert sc.start == ck["global_step"] and sc.trainer.inst_id_to_index == ck["instance_id_to_index"]
    assert sc.trainer.obj_scale == ck["obj_scale"]
    if tag != "bg":
        _same(ck["shape_code_state_dict"], dict(sc.trainer.shape_codes.state_dict()))
        _same(ck["obj_tensor_dict"], sc.object_tensor_dict)
        # consumers read [0] as the scale and [1:] as quaternion + translation (train.py:231-234)
        v = sc.object_tensor_dict[ids[-1]]
        T = cnr.utils.get_transform_from_tensor_sim3(v)
        assert rel_l2(T, torch.from_numpy(z[tag + ".T_obj"][-1]).float()) < 1e-6
    # save -> same file name, same keys, same values as the reference wrote
    os.makedirs(str(tmp_path / "again"))
    out = sc.save_checkpoints(str(tmp_path / "again"), ck["global_step"])
    assert os.path.basename(out) == os.path.basename(f)
    back = torch.load(out, weights_only=False)
    assert sorted(back.keys()) == sorted(ck.keys())
    _same({k: ck[k] for k in sorted(ck)}, {k: back[k] for k in sorted(back)})
    # anything that is not a category checkpoint is refused
    torch.save({"cls_id": 1, "FC_state_dict": {}}, f)
    with pytest.raises(KeyError):
        sc.load_checkpoints(f)


def test_sim3_vector_equals_the_reference(tmp_path):
    """object_tensor_dict entries: [scale, qw, qx, qy, qz, t] exactly as the reference derives them from T_obj."""
    import cnr_amd as cnr
    z = _z()
    for tag in ("multi", "single"):
        for T, want in zip(z[tag + ".T_obj"], z[tag + ".obj_tensor"]):
            got = cnr.utils.get_tensor_from_transform_sim3(T.copy())
            assert got.dtype == torch.float32 and got.shape == (8,)
            assert np.array_equal(got.numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_loaded_reference_weights_reproduce_eval_points(dev, tag, tmp_path):
    """GPU: a category built by OUR constructor from the same frames, restored from the reference's checkpoint, returns
    the reference's Trainer.eval_points values: the fused f16 forward for CodeNeRF (<= 1e-3), the exact-fp32 dense
    kernels for the background OccupancyMap (<= 2e-5)."""
    import copy
    import cnr_amd as cnr
    z = _z()
    ck, ids = _reference_checkpoint(z, tag)
    W, H, n_frames, _ = [int(v) for v in z["meta"]]
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32)
    cfg.hidden_feature_size_bg, cfg.bg_scale = 32, 5.0
    sample_dict = {f: dict(image=z["frames_image"][f], depth=z["frames_depth"][f], T=z["frames_T"][f],
                           obj_mask=z["frames_mask"][f]) for f in range(n_frames)}
    frame_info = [dict(frame=f, bbox=[1, 6, 2, 7]) for f in range(2)]
    if tag == "bg":
        inst = dict(bbox3D=z["bg.bound"], frame_info=[dict(frame=f, bbox=[0, W, 0, H]) for f in range(2)])
    else:
        inst = {i: dict(T_obj=z[tag + ".T_obj"][k], bbox3D=SimpleNamespace(extent=z[tag + ".bound"][k]),
                        frame_info=copy.deepcopy(frame_info)) for k, i in enumerate(ids)}
    sc = cnr.scene_cateogries.sceneCategory(cfg, ck["cls_id"], inst, sample_dict, torch.from_numpy(z["rays_dir"]))
    if tag != "bg":   # the constructor derives the same sim3 vectors from T_obj
        for k, i in enumerate(ids):
            assert np.array_equal(sc.object_tensor_dict[i].cpu().numpy(), z[tag + ".obj_tensor"][k])
    f = os.path.join(str(tmp_path), "ref.pth")
    torch.save(ck, f)
    sc.load_checkpoints(f)
    pts = torch.from_numpy(z["points"]).to(dev)
    occ, col = sc.trainer.eval_points(pts) if tag == "bg" else sc.trainer.eval_points(pts, inst_id=ids[-1])
    tol = 2e-5 if tag == "bg" else 1e-3
    assert rel_l2(occ, torch.from_numpy(z[tag + ".eval_occ"])) < tol
    assert rel_l2(col, torch.from_numpy(z[tag + ".eval_color"])) < tol
    # and a checkpoint written from the device model reads back the same weights
    out = sc.save_checkpoints(str(tmp_path), 7)
    back = torch.load(out, weights_only=False, map_location="cpu")
    _same({k: v for k, v in ck["FC_state_dict"].items()}, {k: v for k, v in back["FC_state_dict"].items()})
