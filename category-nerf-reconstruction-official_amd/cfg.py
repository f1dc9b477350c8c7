"""Config reader for the keys the hot path consumes, same attribute names as the reference's
``cfg.Config`` (src/cfg.py:6-97); dataset / registration keys are read when present and otherwise
ignored (those subsystems are out of scope)."""
import json


class Config:
    def __init__(self, config_file):
        with open(config_file) as f:
            config = json.load(f)
        tr, rd, md, op = config["trainer"], config["render"], config["model"], config["optimizer"]["args"]
        self.training_device, self.data_device = tr["train_device"], tr["data_device"]
        self.max_n_models, self.max_iter = tr["n_models"], tr["max_iter"]
        self.save_iter, self.log_iter = tr["save_iter"], tr["log_iter"]
        self.depth_scale = 1 / tr["scale"]
        self.min_depth, self.max_depth = rd["depth_range"][0], rd["depth_range"][1]
        cam = config["camera"]
        self.mh, self.mw, self.height, self.width = cam["mh"], cam["mw"], cam["h"], cam["w"]
        self.H, self.W = self.height - 2 * self.mh, self.width - 2 * self.mw
        if "fx" in cam:
            self.fx, self.fy = cam["fx"], cam["fy"]
            self.cx, self.cy = cam["cx"] - self.mw, cam["cy"] - self.mh
        self.n_per_optim, self.n_per_optim_bg = rd["n_per_optim"], rd["n_per_optim_bg"]
        self.obj_scale, self.bg_scale = md["obj_scale"], md["bg_scale"]
        self.hidden_feature_size, self.hidden_feature_size_bg = md["hidden_feature_size"], md["hidden_feature_size_bg"]
        self.n_bins_cam2surface, self.n_bins_cam2surface_bg = rd["n_bins_cam2surface"], rd["n_bins_cam2surface_bg"]
        self.n_bins = rd["n_bins"]
        self.n_unidir_funcs = md["n_unidir_funcs"]
        self.surface_eps, self.stop_eps = md["surface_eps"], md["other_eps"]
        self.net_hyperparams = md["net_hyperparams"]
        self.learning_rate, self.code_learning_rate = op["lr"], op["code_lr"]
        self.weight_decay, self.code_weight_decay = op["weight_decay"], op["code_weight_decay"]


def synthetic_config(device="cuda:0", latent_dim=256, obj_scale=2.0, n_bins_cam2surface=8, n_bins=56):
    """The Replica room_0 model/optimiser settings (configs/Replica/config_replica_room0.json) with the
    sample split of a synthetic scale-up; no file needed."""
    c = object.__new__(Config)
    c.training_device = c.data_device = device
    c.max_n_models, c.max_iter, c.save_iter, c.log_iter = 100, 10001, 2000, 100
    c.min_depth, c.max_depth = 0.0, 8.0
    c.W, c.H, c.fx, c.fy, c.cx, c.cy = 1200, 680, 600.0, 600.0, 599.5, 339.5
    c.n_per_optim, c.n_per_optim_bg = 120, 1200
    c.obj_scale, c.bg_scale = obj_scale, 5.0
    c.hidden_feature_size, c.hidden_feature_size_bg = 32, 128
    c.n_bins_cam2surface, c.n_bins_cam2surface_bg, c.n_bins = n_bins_cam2surface, 5, n_bins
    c.n_unidir_funcs = 5
    c.surface_eps, c.stop_eps = 0.1, 0.05
    c.net_hyperparams = dict(shape_blocks=2, texture_blocks=1, W=32, latent_dim=latent_dim)
    c.learning_rate = c.code_learning_rate = 0.001
    c.weight_decay = c.code_weight_decay = 0.013
    return c
