"""Per-category ray pool + sample generator with the reference's surface
(src/scene_cateogries.py: ``sceneCategory.get_training_samples``, ``sample_3d_points``, the free
functions ``origin_dirs_O/W``, ``stratified_bins``, ``normal_bins_sampling``, ``cameraInfo``).

Differences by design: the ray pool is DEVICE-resident (the reference slices CPU
tensors and pays an H2D copy per step), and the whole of a2-a5 for a pool slice is one HIP launch
(cnr_sample_rays) with no count_nonzero host syncs.  ``sceneCategory(cfg, cls_id, inst_dict, sample_dict, rays)``
builds the pool from dataset frames (:107-350) with one device gather (cnr_gather_pool);
:func:`synthetic_pool` generates the SURVEY §8(d) random-pose pool and ``sceneCategory.from_pool`` accepts any
pool with the same fields.
"""
import copy

import numpy as np
import torch

from . import ops, trainer
from .utils import get_tensor_from_transform_sim3


# ---- free functions (reference signatures) -------------------------------------------------------
def origin_dirs_O(T_CO, dirs_C):
    """src/scene_cateogries.py:24-36."""
    assert T_CO.shape[0] == dirs_C.shape[0]
    assert T_CO.shape[1:] == (4, 4)
    assert dirs_C.shape[1] == 3
    T_OC = torch.linalg.inv(T_CO)
    return T_OC[:, :3, -1], (T_OC[:, :3, :3] @ dirs_C[..., None]).squeeze(-1)


def origin_dirs_W(T_WC, dirs_C):
    """src/scene_cateogries.py:38-47."""
    assert T_WC.shape[0] == dirs_C.shape[0]
    assert T_WC.shape[1:] == (4, 4)
    assert dirs_C.shape[1] == 3
    return T_WC[:, :3, -1], (T_WC[:, :3, :3] @ dirs_C[..., None]).squeeze(-1)


def stratified_bins(min_depth, max_depth, n_bins, n_rays, type=torch.float32, device="cuda:0", z_fixed=False, generator=None):
    """src/scene_cateogries.py:51-81 (stand-alone form; the train step samples in cnr_sample_rays)."""
    lim = torch.linspace(0, 1, n_bins + 1, dtype=type, device=device)
    if not torch.is_tensor(min_depth):
        min_depth = torch.ones(n_rays, dtype=type, device=device) * min_depth
    if not torch.is_tensor(max_depth):
        max_depth = torch.ones(n_rays, dtype=type, device=device) * max_depth
    rng = max_depth - min_depth
    lower = (rng[..., None] * lim + min_depth[..., None])[:, :-1]
    assert lower.shape == (n_rays, n_bins)
    # z_fixed is accepted and IGNORED, as in the reference (its fixed-z branch is commented out, :70-72): the probe
    # rays of category_registration.py:145 are jittered too
    return lower + torch.rand(n_rays, n_bins, device=device, dtype=torch.float32, generator=generator) * (rng / n_bins)[..., None]


def normal_bins_sampling(depth, n_bins, n_rays, delta, device="cuda:0"):
    """src/scene_cateogries.py:84-96."""
    bins = torch.empty(n_rays, n_bins, dtype=torch.float32, device=device).normal_(mean=0., std=delta / 3.)
    bins = torch.clip(bins.sort().values, -delta, delta)
    z_vals = depth[:, None] + bins
    assert z_vals.shape == (n_rays, n_bins)
    return z_vals


class cameraInfo:
    """src/scene_cateogries.py:600-629: pinhole directions, z-depth convention (not normalised)."""

    def __init__(self, cfg, device="cpu") -> None:
        """``device``: where the (W,H,3) cache is built and kept (the reference keeps it on the CPU, :617; pool construction
        here is a device gather, so a device cache saves the upload)."""
        self.width, self.height = cfg.W, cfg.H
        self.fx, self.fy, self.cx, self.cy = cfg.fx, cfg.fy, cfg.cx, cfg.cy
        self.device = torch.device(device)
        self.rays_dir_cache = self.get_rays_dirs()

    def get_rays_dirs(self, depth_type="z"):
        if depth_type == "euclidean":   # the reference raises here as well (:623-626)
            raise Exception("Get camera rays directions with euclidean depth not yet implemented")
        if self.device.type == "cuda":  # cnr_camera_rays: correctly rounded division, bit-equal to the reference's CPU tensor
            from . import _C
            dirs = torch.empty((self.width, self.height, 3), device=self.device)
            with torch.cuda.device(self.device):
                _C.call("cnr_camera_rays", dirs, int(self.width), int(self.height), float(self.fx), float(self.fy),
                        float(self.cx), float(self.cy))
            return dirs
        dirs = torch.ones((self.width, self.height, 3))
        dirs[:, :, 0] = ((torch.arange(end=self.width) - self.cx) / self.fx)[:, None]
        dirs[:, :, 1] = ((torch.arange(end=self.height) - self.cy) / self.fy)
        return dirs


# ---- synthetic pool --------------------------------------------------------------
def _rand_rot(gen, n, device):
    q = torch.randn(n, 4, generator=gen, device=device)
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(n, 3, 3)


def synthetic_pool(n_rays, n_obj, generator, device, width=1200, height=680, fx=600.0, fy=600.0,
                   cx=599.5, cy=339.5, pose_group=1024):
    """Random-pose ray pool: fields rgbs (N,4) u8 [r,g,b,state], depth (N,), dirs (N,3), T_co (N,4,4),
    T_wc (N,4,4), indices (N,) int64.  Same generator recipe on CPU (oracle timing) and GPU."""
    g = generator
    N = n_rays
    u = torch.randint(0, width, (N,), generator=g, device=device).float()
    v = torch.randint(0, height, (N,), generator=g, device=device).float()
    dirs = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones(N, device=device)], -1)
    ng = (N + pose_group - 1) // pose_group
    eye = torch.eye(4, device=device)
    T_wc = eye.repeat(ng, 1, 1)
    T_wc[:, :3, :3] = _rand_rot(g, ng, device)
    T_wc[:, :3, 3] = torch.rand(ng, 3, generator=g, device=device) * 2 - 1
    T_wo = eye.repeat(ng, 1, 1)
    s = 0.3 + 0.7 * torch.rand(ng, 1, 1, generator=g, device=device)
    T_wo[:, :3, :3] = _rand_rot(g, ng, device) * s
    T_wo[:, :3, 3] = torch.rand(ng, 3, generator=g, device=device) * 2 - 1
    T_co = torch.linalg.inv(T_wc) @ T_wo
    grp = torch.arange(N, device=device) // pose_group
    depth = 0.5 + 3.0 * torch.rand(N, generator=g, device=device)
    depth = torch.where(torch.rand(N, generator=g, device=device) < 0.05, torch.zeros_like(depth), depth)
    pr = torch.rand(N, generator=g, device=device)
    state = torch.where(pr < 0.70, 1, torch.where(pr < 0.95, 0, 2)).to(torch.uint8)
    rgb = torch.randint(0, 256, (N, 3), generator=g, device=device, dtype=torch.uint8)
    return dict(rgbs=torch.cat([rgb, state[:, None]], -1).contiguous(), depth=depth.contiguous(),
                dirs=dirs.contiguous(), T_co=T_co[grp].contiguous(), T_wc=T_wc[grp].contiguous(),
                indices=torch.randint(0, n_obj, (N,), generator=g, device=device))


# ---- per-category scene object ------------------------------------------------------------------------
class sceneCategory():
    """shared MLP + instance-specific codes for one category; single batch holds all its instances.

    Construct from the reference's dataset dictionaries (``__init__``, src/scene_cateogries.py:107-350) or from a
    ready pool (:meth:`from_pool`)."""

    other_obj, this_obj, unknown_obj = 0, 1, 2  # pixel states (:141-143)

    def __init__(self, cfg, cls_id, inst_dict, sample_dict, cached_rays_dir):
        """The reference's constructor (src/scene_cateogries.py:107-350): ``inst_dict`` {inst_id: {T_obj, bbox3D,
        frame_info: [{frame, bbox}]}} (background: {bbox3D, frame_info}), ``sample_dict`` {frame: {image (W,H,3)
        u8, depth (W,H), T (4,4) = T_wc, obj_mask (W,H)}}, ``cached_rays_dir`` (W,H,3).  The frames go to the
        device once; every crop of every instance becomes pool rows in one gather launch, global shuffle included
        (same ``np.random.shuffle`` call as the reference, so a seeded run gives the same pool order)."""
        obj_ids = list(inst_dict.keys()) if cls_id != 0 else [0]
        self._init_common(cfg, cls_id, obj_ids)
        dev = self.data_device
        init_idx = list(sample_dict.keys())[0]
        self.frames_width, self.frames_height = sample_dict[init_idx]["image"].shape[:2]
        W, H = self.frames_width, self.frames_height
        infos = {0: inst_dict} if cls_id == 0 else inst_dict
        frames = sorted({fi["frame"] for info in infos.values() for fi in info["frame_info"]})
        slot = {f: i for i, f in enumerate(frames)}
        stack = lambda key, dt: torch.from_numpy(np.stack([np.asarray(sample_dict[f][key]) for f in frames])).to(dt)
        images = stack("image", torch.uint8).to(dev).contiguous()
        depths = stack("depth", torch.float32).to(dev).contiguous()
        masks = stack("obj_mask", torch.int32).to(dev).contiguous()
        T_wc = stack("T", torch.float32).to(dev)
        rays = torch.as_tensor(cached_rays_dir, dtype=torch.float32).to(dev).contiguous()
        crops, T_crop, areas = [], [], []
        if cls_id != 0:
            self.extent_dict, self.object_tensor_dict = {}, {}
            self._T_obj = {}                 # the 4x4 itself, for the pool poses below (not part of the checkpoint)
        for inst_id in obj_ids:
            info = infos[inst_id]
            if cls_id != 0:
                self.extent_dict[inst_id] = info["bbox3D"].extent if "bbox3D" in info else np.array([2.0, 2.0, 2.0])
                T_obj = torch.from_numpy(np.asarray(info["T_obj"]).astype(np.float32)).to(dev)
                self._T_obj[inst_id] = T_obj
                # [scale, qw, qx, qy, qz, t]: what the reference keeps and checkpoints (:179-181), read back as
                # obj_tensor[0] = scale, obj_tensor[1:] = rigid transform (train.py:231-234)
                self.object_tensor_dict[inst_id] = get_tensor_from_transform_sim3(np.copy(info["T_obj"])).to(dev)
            index = 0 if len(obj_ids) == 1 else obj_ids.index(inst_id)
            sl = [slot[fi["frame"]] for fi in info["frame_info"]]
            t_wc = T_wc[sl]                                                        # (n_frames_of_inst, 4, 4)
            pose = t_wc if self.world_frame else torch.linalg.inv(t_wc) @ T_obj[None]  # :237-238
            for idx, fi in enumerate(info["frame_info"]):
                w0, w1, h0, h1 = [int(v) for v in fi["bbox"]]
                crops.append([sl[idx], int(inst_id), index, w0, w1, h0, h1, idx])
                areas.append((w1 - w0) * (h1 - h0))
            T_crop.append(pose)
            if cls_id == 0:
                self.t_wc_batch_dict = {0: t_wc}
        N = int(sum(areas))
        shuffled_idx = np.arange(N)
        np.random.shuffle(shuffled_idx)                                            # :251-252 / :309-310
        crops_t = torch.tensor(crops, dtype=torch.int32, device=dev)
        off_t = torch.tensor(np.concatenate([[0], np.cumsum(areas)]), dtype=torch.int64, device=dev)
        perm_t = torch.from_numpy(shuffled_idx).to(dev)
        T_crop = torch.cat(T_crop).contiguous()
        self.rgbs_batch_all = torch.empty(N, 4, dtype=torch.uint8, device=dev)
        self.depth_batch_all = torch.empty(N, device=dev)
        self.ray_dirs_batch_all = torch.empty(N, 3, device=dev)
        T_rows = torch.empty(N, 4, 4, device=dev)
        self.batch_indices_all = torch.empty(N, dtype=torch.int64, device=dev)
        frame_rows = torch.empty(N, dtype=torch.int64, device=dev)
        from . import _C
        _C.call("cnr_gather_pool", images, depths, masks, rays, T_crop, crops_t, off_t, perm_t, W, H, len(crops), N,
                self.rgbs_batch_all, self.depth_batch_all, self.ray_dirs_batch_all, T_rows, self.batch_indices_all,
                frame_rows)
        self.t_wc_batch_all = T_rows if self.world_frame else None
        self.t_co_batch_all = None if self.world_frame else T_rows
        if cls_id == 0:   # the reference's per-object dict view of the background pool (:318-324)
            self.rgbs_batch_dict, self.depth_batch_dict = {0: self.rgbs_batch_all}, {0: self.depth_batch_all}
            self.frame_batch_dict, self.ray_dirs_batch_dict = {0: frame_rows}, {0: self.ray_dirs_batch_all}
            self.i_batch_dict = {0: 0}
        self._make_trainer(cfg)
        if cls_id == 0:
            self.trainer.bound = inst_dict.get("bbox3D")
        elif len(obj_ids) == 1:
            self.trainer.bound_dict = {i: inst_dict[i].get("bbox3D") for i in obj_ids}
        else:
            self.trainer.extent_dict = self.extent_dict

    def _init_common(self, cfg, cls_id, obj_ids):
        self.cls_id = cls_id
        self.obj_ids = list(obj_ids) if cls_id != 0 else [0]          # src/scene_cateogries.py:109-112
        self.data_device = cfg.data_device
        self.training_device = cfg.training_device
        if cls_id == 0:   # background (:116-119): vMAP OccupancyMap, world-frame rays, its own bin counts
            self.obj_scale = cfg.bg_scale
            self.hidden_feature_size = cfg.hidden_feature_size_bg
            self.n_bins_cam2surface = cfg.n_bins_cam2surface_bg
        else:
            self.obj_scale = cfg.obj_scale
            self.hidden_feature_size = cfg.hidden_feature_size
            self.n_bins_cam2surface = cfg.n_bins_cam2surface
        self.min_bound, self.max_bound = cfg.min_depth, cfg.max_depth
        self.n_bins = cfg.n_bins
        self.surface_eps, self.stop_eps = cfg.surface_eps, cfg.stop_eps
        self.world_frame = cls_id == 0 or len(self.obj_ids) == 1       # origin_dirs_W vs origin_dirs_O (:374-386)
        self.i_batch = 0
        self.parity_draws = None  # (u, g) tensors for the next get_training_samples call (tests)

    def _make_trainer(self, cfg, seed=0):
        self._seed, self._calls = int(seed) * 7919 + int(self.cls_id) + 1, 0
        trainer_cfg = copy.copy(cfg)
        trainer_cfg.hidden_feature_size = self.hidden_feature_size     # :328-330
        trainer_cfg.obj_scale = self.obj_scale
        self.trainer = trainer.Trainer(trainer_cfg, self.cls_id, self.obj_ids)

    @classmethod
    def from_pool(cls, cfg, cls_id, obj_ids, pool, seed=0):
        """A ready pool dict {rgbs (N,4) u8, depth (N,), dirs (N,3), T_co / T_wc (N,4,4), indices (N,)} on any device."""
        self = object.__new__(cls)
        self._init_common(cfg, cls_id, obj_ids)
        dev = self.data_device
        self.rgbs_batch_all = pool["rgbs"].to(dev)
        self.depth_batch_all = pool["depth"].to(dev)
        self.ray_dirs_batch_all = pool["dirs"].to(dev)
        self.t_co_batch_all = pool["T_co"].to(dev) if not self.world_frame else None
        self.t_wc_batch_all = pool["T_wc"].to(dev) if self.world_frame else None
        self.batch_indices_all = pool["indices"].to(dev) if cls_id != 0 else \
            torch.zeros(pool["depth"].shape[0], dtype=torch.int64, device=dev)
        self._make_trainer(cfg, seed)
        return self

    def sample_3d_points(self, sampled_rgbs, sampled_depth, T, dirs_c, world_frame, u=None, g=None):
        """a2-a5 for one slice (leading class dim of 1 added for the kernel)."""
        self._calls += 1
        out = ops.sample_rays(sampled_rgbs[None], sampled_depth[None], dirs_c[None], T[None],
                              self.n_bins_cam2surface, self.n_bins, self.surface_eps, self.stop_eps,
                              min_bound=self.min_bound, world_frame=world_frame, u=u, g=g,
                              seed=self._seed, offset=self._calls * 4)
        return out

    def get_training_samples(self, n_samples):
        """-> (gt_rgb uint8 (R,3), gt_depth (R,), depth_mask bool (R,), obj_mask uint8 (R,),
        input_pcs (R,S,3), sampled_z (R,S), indices int64 (R,))  -- src/scene_cateogries.py:353-451.  The background
        (cls_id 0) is the one-pool case of the per-object loop at :353-420: world-frame rays, indices all zero."""
        sl = slice(self.i_batch, self.i_batch + n_samples)
        batch_indices = self.batch_indices_all[sl]
        rgbs, depth, dirs = self.rgbs_batch_all[sl], self.depth_batch_all[sl], self.ray_dirs_batch_all[sl]
        single = self.world_frame
        T = self.t_wc_batch_all[sl] if single else self.t_co_batch_all[sl]
        u, g = self.parity_draws if self.parity_draws is not None else (None, None)
        self.parity_draws = None
        out = self.sample_3d_points(rgbs, depth, T, dirs, single, u=u, g=g)
        self.i_batch += n_samples
        if self.i_batch >= self.rgbs_batch_all.shape[0] - n_samples:  # reshuffle per epoch (:439-449)
            rand_idx = torch.randperm(self.rgbs_batch_all.shape[0], device=self.data_device)
            self.batch_indices_all = self.batch_indices_all[rand_idx]
            self.rgbs_batch_all = self.rgbs_batch_all[rand_idx]
            self.depth_batch_all = self.depth_batch_all[rand_idx]
            self.ray_dirs_batch_all = self.ray_dirs_batch_all[rand_idx]
            if single:
                self.t_wc_batch_all = self.t_wc_batch_all[rand_idx]
            else:
                self.t_co_batch_all = self.t_co_batch_all[rand_idx]
            self.i_batch = 0
        return (rgbs[:, :3], depth, out["depth_mask"][0].bool(), out["labels"][0], out["pts"][0],
                out["z"][0], batch_indices)

    # ---- checkpoint interchange (src/scene_cateogries.py:548-597) ------------------------------------------------------
    # File name and key schema of the reference:
    #   global_step int | cls_id int | obj_scale float | instance_id_to_index {inst_id: row}
    #   PE_state_dict {scale, B_layer.weight} | FC_state_dict {<layer>.weight / .bias}
    #   background (cls_id 0):  bound = trainer.bound (the scene's bbox3D object)
    #   object category:        obj_tensor_dict {inst_id: (8,) sim3 vector}, extent_dict {inst_id: (3,)} (only with > 1
    #                           object), shape_code_state_dict / texture_code_state_dict {weight}, bound = trainer.extent_dict
    CKPT_KEYS_BG = ("global_step", "PE_state_dict", "FC_state_dict", "cls_id", "instance_id_to_index", "obj_scale", "bound")
    CKPT_KEYS_OBJ = CKPT_KEYS_BG + ("obj_tensor_dict", "shape_code_state_dict", "texture_code_state_dict")

    def checkpoint_dict(self, iter):
        t = self.trainer
        d = {"global_step": iter, "PE_state_dict": t.pe.state_dict(), "FC_state_dict": t.fc_occ_map.state_dict(),
             "cls_id": self.cls_id, "instance_id_to_index": t.inst_id_to_index, "obj_scale": t.obj_scale}
        if self.cls_id == 0:
            d["bound"] = getattr(t, "bound", None)
            return d
        d["obj_tensor_dict"] = getattr(self, "object_tensor_dict", {})
        if len(self.obj_ids) > 1:
            d["extent_dict"] = getattr(self, "extent_dict", None)
        d["shape_code_state_dict"] = t.shape_codes.state_dict()
        d["texture_code_state_dict"] = t.texture_codes.state_dict()
        d["bound"] = t.extent_dict
        return d

    def save_checkpoints(self, path, iter):
        import os
        ckpt_file = os.path.join(path, "cls_" + str(self.cls_id) + "_iteration_{:05d}.pth".format(iter))
        torch.save(self.checkpoint_dict(iter), ckpt_file)
        return ckpt_file

    def load_checkpoints(self, ckpt_file, allow_pickle=False):
        """Restore from a checkpoint in the reference's format (written by either code base).  The file is read with
        ``weights_only=True`` (tensors, containers, numbers, numpy arrays): nothing in it is executed.  A reference
        checkpoint of the BACKGROUND carries an open3d bounding box under ``bound``; pass ``allow_pickle=True`` to
        unpickle such a file -- only for files you trust.  A missing file prints a note and returns, as the reference
        does.  The reference assigns ``bound`` / ``extent_dict`` to the opposite trainer attribute of the one its saver
        read them from (:585-593); here each goes back where it came from, so save -> load -> save is the identity."""
        import os
        if not os.path.exists(ckpt_file):
            print("ckpt not exist ", ckpt_file)
            return
        if allow_pickle:
            ck = torch.load(ckpt_file, map_location=self.training_device, weights_only=False)
        else:
            try:
                import numpy._core.multiarray as _ma
            except ImportError:                                       # numpy < 2
                import numpy.core.multiarray as _ma
            with torch.serialization.safe_globals([_ma._reconstruct, np.ndarray, np.dtype, type(np.dtype(np.float64)),
                                                   type(np.dtype(np.float32)), type(np.dtype(np.int64))]):
                ck = torch.load(ckpt_file, map_location=self.training_device, weights_only=True)
        want = self.CKPT_KEYS_BG if ck["cls_id"] == 0 else self.CKPT_KEYS_OBJ
        missing = [k for k in want if k not in ck]
        if missing:
            raise KeyError(f"{ckpt_file}: not a category checkpoint, missing {missing}")
        t, dev = self.trainer, self.training_device
        self.cls_id = ck["cls_id"]
        t.fc_occ_map.load_state_dict(ck["FC_state_dict"])
        t.pe.load_state_dict(ck["PE_state_dict"])
        t.fc_occ_map.to(dev)
        t.pe.to(dev)
        # ("bound" lands on both attributes: trainer.bound is where the reference's loader puts an object category's
        #  value, trainer.extent_dict where its saver read it from -- and the other way round for the background)
        t.bound = ck["bound"]
        if self.cls_id == 0:
            t.extent_dict = ck["bound"]
        else:
            self.object_tensor_dict = ck["obj_tensor_dict"]
            if "extent_dict" in ck:
                self.extent_dict = ck["extent_dict"]
            t.shape_codes.load_state_dict(ck["shape_code_state_dict"])
            t.texture_codes.load_state_dict(ck["texture_code_state_dict"])
            t.shape_codes.to(dev)
            t.texture_codes.to(dev)
            t.extent_dict = ck["bound"]
        t.inst_id_to_index = ck["instance_id_to_index"]
        t.obj_scale = ck["obj_scale"]
        self.start = ck["global_step"]
