"""The background branch of the train step (train.py:113-121,172-184) as a device-resident, graph-capturable step, and the
WHOLE iteration of train.py:113-184 -- background + every object category -- replayed as ONE hipGraph.

The background is one vMAP-style ``OccupancyMap(hidden_feature_size_bg)`` (src/model.py:86-155) trained on world-frame rays
(src/scene_cateogries.py:353-420) with the same composite / loss as the categories.  Its arithmetic stays on the exact-fp32
kernels (cnr_pe_*, cnr_dense_*: fp32 MFMA, cnr_composite_*, cnr_loss_fwd_bwd) under torch autograd -- the tier pinned to the
reference's ``bg_*.npz`` vectors at 2e-5 -- but nothing in a step touches the host: the ray pool, its permutation and the
cursor live on the device (the sampler reads the cursor, cnr_step_advance moves it), the optimiser is captured with the
step.  Gradients of background and categories are disjoint (separate parameters, train.py:54-64), so running the two
backward passes one after the other inside the graph is the same step as the reference's single ``batch_loss.backward()``.
"""
import torch

from . import _C, loss as loss_mod, ops, trainer as trainer_mod


class BackgroundStep:
    def __init__(self, cfg, pool, rays_per_step=None, device=None, seed=0, precision="fp32"):
        import copy
        self.cfg = cfg
        self.device = torch.device(device or cfg.training_device)
        self.R = int(rays_per_step or cfg.n_per_optim_bg)
        self.n1, self.n2 = cfg.n_bins_cam2surface_bg, cfg.n_bins
        tcfg = copy.copy(cfg)
        tcfg.hidden_feature_size, tcfg.obj_scale, tcfg.training_device = cfg.hidden_feature_size_bg, cfg.bg_scale, str(self.device)
        self.trainer = trainer_mod.Trainer(tcfg, 0, [0])                     # .pe, .fc_occ_map (src/trainer.py:23-25)
        assert precision in ("fp32", "f16")
        # "f16": the hidden layers' products on f16 MFMA operands with fp32 accumulation (ops.DenseFn half=True; the x10
        # occupancy head stays fp32, like the sigma head of the category kernel); "fp32": the exact tier (default)
        self.trainer.fc_occ_map.half = precision == "f16"
        dev = self.device
        self.pool = dict(rgbs=pool["rgbs"].to(dev)[None].contiguous(), depth=pool["depth"].to(dev)[None].contiguous(),
                         dirs=pool["dirs"].to(dev)[None].contiguous(), T=pool["T_wc"].to(dev)[None].contiguous())
        self.pool_rows = self.pool["depth"].shape[1]
        assert self.pool_rows >= 2 * self.R
        self.perm = torch.empty(1, self.pool_rows, device=dev, dtype=torch.int32)
        self._perm_gen = torch.Generator(device=dev)
        self._perm_gen.manual_seed(0xB6 + 7919 * int(seed))
        self.d_state = torch.zeros(3, device=dev, dtype=torch.int64)
        self._zero = torch.zeros(1, device=dev, dtype=torch.int64)
        self.seed = int(seed) + 101
        self.params = list(self.trainer.fc_occ_map.parameters()) + list(self.trainer.pe.parameters())
        # one flat buffer for the parameters and one for their gradients (the modules' tensors become views): the optimiser is
        # ONE launch (cnr_adamw_step, torch.optim.AdamW semantics, step count read from the device state) and the gradient
        # reset one fill, instead of the ~50 small kernels of a capturable torch AdamW + per-tensor zero fills per step
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=dev)
        self.gflat = torch.zeros(n, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.gflat[off:off + k].view_as(p)
            off += k
        self.exp_avg, self.exp_avg_sq = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        # every Linear of the field is used once per step: its weight / bias gradients are written straight into their views
        # of the flat gradient buffer by the dense backward (model.OccupancyMap.grad_out), no per-parameter add launch
        import torch.nn as nn
        self.trainer.fc_occ_map.grad_out = {id(m): (m.weight.grad, m.bias.grad if m.bias is not None else None)
                                            for m in self.trainer.fc_occ_map.modules() if isinstance(m, nn.Linear)}
        self.bufs = {}
        self.loss = torch.zeros((), device=dev)
        self.losses = torch.zeros(3, device=dev)
        self.cursor = 0
        self.steps_done = 0
        self.graph = None
        self._reshuffle()

    def _reshuffle(self):
        """src/scene_cateogries.py:439-449 as a new permutation; cursor back to 0."""
        self.perm[0].copy_(torch.randperm(self.pool_rows, device=self.device, generator=self._perm_gen))
        self.d_state[0:1].copy_(self._zero)
        self.cursor = 0

    def _body(self):
        """sample -> PE -> OccupancyMap -> composite + losses -> backward -> AdamW -> advance (all stream-ordered)."""
        cfg, t = self.cfg, self.trainer
        self.gflat.zero_()
        b = ops.sample_rays(self.pool["rgbs"], self.pool["depth"], self.pool["dirs"], self.pool["T"], self.n1, self.n2,
                            cfg.surface_eps, cfg.stop_eps, min_bound=cfg.min_depth, world_frame=True, seed=self.seed,
                            d_state=self.d_state, rays=self.R, out=self.bufs, perm=self.perm)
        alpha, color = t.fc_occ_map(t.pe(b["pts"][0]))
        loss, ld, _ = loss_mod.step_batch_loss(alpha[None], color[None], b["gt_depth"], b["gt_rgb"], b["labels"],
                                               b["depth_mask"], b["z"])
        loss.backward()
        ops.adamw_step(self.flat, self.gflat, self.exp_avg, self.exp_avg_sq, cfg.learning_rate, (0.9, 0.999), 1e-8,
                       cfg.weight_decay, 0, 1.0, d_state=self.d_state)          # step = d_state[2] + 1, advanced below
        self.loss.copy_(loss.detach())
        self.losses.copy_(torch.stack([ld["depth"][0], ld["color"][0], ld["opacity"][0]]).detach())
        _C.call("cnr_step_advance", self.d_state, self.R)

    def pre_step(self):
        if self.cursor >= self.pool_rows - self.R:
            self._reshuffle()

    def post_step(self):
        self.cursor += self.R
        self.steps_done += 1

    def step(self, use_graph=True):
        """One background step; after three eager steps it is captured and replayed."""
        self.pre_step()
        if not use_graph or self.steps_done < 3:
            self._body()
        else:
            if self.graph is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body()
                self.graph = g
            self.graph.replay()
        self.post_step()


class FullStepTrainer:
    """train.py:113-184 in one replay: the background step and the fused category step captured into ONE hipGraph per
    state parity of the category trainer (its parameters / step state ping-pong between two copies)."""

    def __init__(self, categories, background):
        self.obj, self.bg = categories, background
        assert not self.obj.grad_exchange, "ray-sharded category steps run a collective between their two graphs"
        self.graphs = {}
        self.steps_done = 0

    def step(self):
        o, b = self.obj, self.bg
        o._pre_step()
        b.pre_step()
        par = o.parity
        if self.steps_done < 3:
            b._body()
            o._step_body()
        else:
            if par not in self.graphs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    b._body()
                    o._step_body()
                self.graphs[par] = g
            self.graphs[par].replay()
        o._post_step()
        b.post_step()
        self.steps_done += 1
