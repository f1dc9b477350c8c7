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
    def __init__(self, cfg, pool, rays_per_step=None, device=None, seed=0, precision="fp32", trainer=None):
        """``trainer``: an existing ``Trainer(cls_id=0)`` to train in place (``scene_bg.trainer`` of train.py:49-53) -- its
        modules' parameters become views of this step's flat buffer, so whatever reads them afterwards (``save_checkpoints``,
        ``eval_points``, meshing) sees the trained values with no copy-back.  Default: a fresh one."""
        import copy
        self.cfg = cfg
        self.device = torch.device(device or cfg.training_device)
        self.R = int(rays_per_step or cfg.n_per_optim_bg)
        self.n1, self.n2 = cfg.n_bins_cam2surface_bg, cfg.n_bins
        if trainer is None:
            tcfg = copy.copy(cfg)
            tcfg.hidden_feature_size, tcfg.obj_scale, tcfg.training_device = cfg.hidden_feature_size_bg, cfg.bg_scale, str(self.device)
            trainer = trainer_mod.Trainer(tcfg, 0, [0])                      # .pe, .fc_occ_map (src/trainer.py:23-25)
        else:
            assert trainer.cls_id == 0 and next(trainer.fc_occ_map.parameters()).device == self.device
        self.trainer = trainer
        assert precision in ("fp32", "f16", "fused")
        # "fp32": the exact tier -- one launch per layer (cnr_dense_*: fp32 MFMA) under torch autograd, pinned to the reference's
        #         bg_*.npz vectors at 2e-5 (the default of this class; the parity tier);
        # "f16":  the same launches with f16 operands in the hidden layers (ops.DenseFn half=True);
        # "fused": the throughput tier (csrc/bg_fused.hip): the whole step in seven launches, every activation of a 64-sample
        #         tile in LDS, f16 MFMA operands with the geometry branch as three products per fragment (the x10 occupancy
        #         logit, like the category kernel), no autograd, no float atomics.  Hidden size 128 only.
        self.precision = precision
        self.trainer.fc_occ_map.half = precision == "f16"
        dev = self.device
        self.pool = dict(rgbs=pool["rgbs"].to(dev)[None].contiguous(), depth=pool["depth"].to(dev)[None].contiguous(),
                         dirs=pool["dirs"].to(dev)[None].contiguous(), T=pool["T_wc"].to(dev)[None].contiguous())
        self.pool_rows = self.pool["depth"].shape[1]
        assert self.pool_rows >= 2 * self.R
        self.perm = torch.empty(1, self.pool_rows, device=dev, dtype=torch.int32)
        self._perm_gen = torch.Generator(device=dev)
        self._perm_gen.manual_seed(0xB6 + 7919 * int(seed))
        # epoch shuffle: cnr_epoch_perm (one launch) unless CNR_EPOCH_PERM=torch asks for torch.randperm (a dozen launches)
        self._torch_perm = __import__("os").environ.get("CNR_EPOCH_PERM", "kernel") == "torch"
        self._perm_seed, self._epoch = 0xB6 + 7919 * int(seed), 0
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
        if precision == "fused":
            self._init_fused(n)
        self.bufs = {}
        if precision != "fused":
            self.loss = torch.zeros((), device=dev)
            self.losses = torch.zeros(3, device=dev)
        self.cursor = 0
        self.steps_done = 0
        self.graph = None
        self._reshuffle()

    def _reshuffle(self):
        """src/scene_cateogries.py:439-449 as a new permutation; cursor back to 0.  Fused tier: the max depth of every slice of
        the epoch in one launch (cnr_slice_maxdepth; src/scene_cateogries.py:486 needs it per batch), so that no step computes it."""
        if self._torch_perm:
            self.perm[0].copy_(torch.randperm(self.pool_rows, device=self.device, generator=self._perm_gen))
            self.d_state[0:1].copy_(self._zero)
        else:     # one launch: the keyed bijection of cnr_epoch_perm, and the cursor back to row 0
            _C.call("cnr_epoch_perm", self.perm, self.pool_rows, 1, self._perm_seed, self._epoch, None, self.d_state, 0)
            self._epoch += 1
        self.cursor = 0
        if self.precision == "fused":
            _C.call("cnr_slice_maxdepth", self.pool["depth"], self.perm, self.pool_rows, 1, self.R, self.n_slices, self.slice_max)
            # ... and its mask counts (src/render_rays.py:66-95 are pool-row properties): the composite / loss launch reads its
            # slice's entry instead of every block counting the batch's 1200 labels first
            _C.call("cnr_slice_maskcounts", self.pool["rgbs"], self.pool["depth"], self.perm, self.pool_rows, 1, self.R,
                    self.n_slices, float(self.cfg.min_depth), self.counts_tab)
            if self.sample_in_tail:
                self._sample()      # the epoch's first batch; every later one is drawn by the previous step's last launch

    def _body(self):
        """sample -> PE -> OccupancyMap -> composite + losses -> backward -> AdamW -> advance (all stream-ordered)."""
        if self.precision == "fused":
            return self._body_fused()
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

    # ---- the fused f16 tier ----------------------------------------------------------------------------------------------------
    def _init_fused(self, n):
        import math
        lib, dev = _C.load(), self.device
        fc = self.trainer.fc_occ_map
        assert self.cfg.hidden_feature_size_bg == 128 and n == int(lib.cnr_bg_param_count()), \
            "the fused background step is built for OccupancyMap(hidden 128) + UniDirsEmbed"
        # the kernels address the flat buffer by fixed offsets: the modules' registration order must be the header's
        names = [k for k, _ in fc.named_parameters()]
        assert names == ["in_layer.0.weight", "in_layer.0.bias", "mid1.0.0.weight", "mid1.0.0.bias", "cat_layer.0.weight",
                         "cat_layer.0.bias", "mid2.0.0.weight", "mid2.0.0.bias", "out_alpha.weight", "out_alpha.bias",
                         "color_linear.0.weight", "color_linear.0.bias", "out_color.weight", "out_color.bias"], names
        assert [tuple(p.shape) for p in self.trainer.pe.parameters()] == [(21, 3)]
        self.S = self.n1 + self.n2
        M = self.M = self.R * self.S
        # composite / losses inside the backward launch (cnr_bg_backward_render: a ray on one wave, S <= 64); CNR_BG_FUSE_RENDER=0:
        # the separate cnr_render_loss launch
        self.fuse_render = self.S <= 64 and __import__('os').environ.get('CNR_BG_FUSE_RENDER', '1') != '0'
        # the NEXT step's rays sampled by extra blocks of this step's last launch (cnr_bg_tail_sample): the step then starts with the
        # forward; CNR_BG_SAMPLE_IN_TAIL=0: cnr_sample_rays as every step's first launch
        self.sample_in_tail = __import__('os').environ.get('CNR_BG_SAMPLE_IN_TAIL', '1') != '0'
        # samples per weight-gradient workgroup: 384 -> 5 layers x 44 chunks = 220 workgroups, one round on 256 CUs (measured 17.5 us;
        # 256: 20.2, 512: 21.2)
        self.dw_chunk = int(__import__('os').environ.get('CNR_BG_DW_CHUNK', '384'))
        self.nblk, self.nchunk = int(lib.cnr_bg_blocks(M)), int(lib.cnr_bg_dw_chunks(M, self.dw_chunk))
        self.gscale = float(2 ** round(math.log2(max(self.R, 2))))      # power-of-two loss scale of the f16 gradient chain
        self.n_slices = self.pool_rows // self.R
        self.slice_max = torch.zeros(1, self.n_slices, device=dev)
        self.counts_tab = torch.zeros(self.n_slices, 2, 4, device=dev)
        self._packed_for = None          # version of self.flat the fragment images were packed from
        f = lambda *sh, dt=torch.float32: torch.empty(*sh, device=dev, dtype=dt)
        self.fb = dict(packed=f(int(lib.cnr_bg_pack_bytes()), dt=torch.uint8), sigma=f(1, self.R, self.S),
                       rgbs=f(1, self.R, self.S, 3), act=f(5, M, 128, dt=torch.float16), eimg=f(M, 144, dt=torch.float16),
                       dpre=f(5, M, 128, dt=torch.float16), records=f(self.nblk, int(lib.cnr_bg_record_floats())),
                       partials=f(self.nchunk, n), dsig=f(1, self.R, self.S), drgb=f(1, self.R, self.S, 3),
                       depth=f(1, self.R), var=f(1, self.R), rgb=f(1, self.R, 3), opa=f(1, self.R),
                       rl_ws=torch.zeros(max(_C.render_loss_workspace_bytes(1, self.R),
                                             int(lib.cnr_bg_backward_render_workspace_bytes(M))), device=dev, dtype=torch.uint8),
                       losses=torch.zeros(3, 1, device=dev), flags=torch.zeros(1, device=dev, dtype=torch.int32))

    @property
    def losses(self):
        """(3,) depth / colour / opacity terms of the last step (src/loss.py:18-74)"""
        return self.fb["losses"][:, 0] if self.precision == "fused" else self._losses

    @losses.setter
    def losses(self, v):
        self._losses = v

    @property
    def loss(self):
        """the step's scalar loss: depth + 5 colour + 10 opacity (formed on demand in the fused tier: nobody reads it per step)"""
        if self.precision == "fused":
            l = self.fb["losses"][:, 0]
            return l[0] + 5.0 * l[1] + 10.0 * l[2]
        return self._loss

    @loss.setter
    def loss(self, v):
        self._loss = v

    def _body_fused(self):
        """forward -> composite + losses + their gradient + backward -> weight gradients -> reduce + AdamW + fragment refresh + loss
        values + the NEXT step's sampler: four launches, nothing under autograd (train.py:113-121,172-184 for the background; six
        launches with CNR_BG_FUSE_RENDER=0 CNR_BG_SAMPLE_IN_TAIL=0, bit-equal)."""
        cfg, o = self.cfg, self.fb
        b = self.bufs if self.sample_in_tail else self._sample()      # (in the tail: drawn by the previous step, or by _reshuffle)
        scale, M = float(self.trainer.pe._scale), self.M
        _C.call("cnr_bg_forward", b["pts"], self.flat, o["packed"], scale, M, o["sigma"], o["rgbs"], o["act"], o["eimg"])
        if self.fuse_render:
            # composite + losses + their gradient in front of the backward chain, in the backward's launch (one launch and the
            # d sigma / d colour round trip less); the step state then moves in the weight-gradient launch
            _C.call_struct("cnr_bg_backward_render", pts=b["pts"], theta=self.flat, packed=o["packed"], scale=scale, R=self.R,
                           S=self.S, sigma=o["sigma"], rgb=o["rgbs"], z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"],
                           labels=b["labels"], depth_mask=b["depth_mask"], counts_tab=self.counts_tab, d_state=self.d_state,
                           color_scaling=5.0, opacity_scaling=10.0, grad_scale=self.gscale, act=o["act"], dpre=o["dpre"],
                           records=o["records"], depth=o["depth"], var=o["var"], rgb_render=o["rgb"], opacity=o["opa"], d_sigma=None, d_rgb=None,
                           loss_workspace=o["rl_ws"], loss_workspace_bytes=o["rl_ws"].numel())
            _C.call("cnr_bg_dw", o["act"], o["dpre"], o["eimg"], M, self.dw_chunk, o["partials"], self.d_state, self.R)
        else:
            _C.call("cnr_render_loss", o["sigma"], o["rgbs"], b["z"], b["gt_depth"], b["gt_rgb"], b["labels"], b["depth_mask"],
                    5.0, 10.0, self.gscale, o["dsig"], o["drgb"], o["depth"], o["var"], o["rgb"], o["opa"], 1, self.R, self.S,
                    o["rl_ws"], o["rl_ws"].numel(), self.counts_tab, self.d_state)
            _C.call("cnr_bg_backward", b["pts"], self.flat, o["packed"], scale, M, o["dsig"], o["drgb"], o["rgbs"], o["act"],
                    o["dpre"], o["records"], self.d_state, self.R)
            _C.call("cnr_bg_dw", o["act"], o["dpre"], o["eimg"], M, self.dw_chunk, o["partials"], None, 0)
        if self.sample_in_tail:
            _C.call_struct("cnr_bg_tail_sample", theta=self.flat, grad=self.gflat, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                           partials=o["partials"], chunks=self.nchunk, records=o["records"], nrec=self.nblk, grad_scale=self.gscale,
                           lr=cfg.learning_rate, beta1=0.9, beta2=0.999, adam_eps=1e-8, weight_decay=cfg.weight_decay,
                           packed=o["packed"], rl_workspace=o["rl_ws"], rl_R=0 if self.fuse_render else self.R, losses=o["losses"],
                           flags=o["flags"], rgbs=self.pool["rgbs"], depth=self.pool["depth"], dirs_c=self.pool["dirs"],
                           T=self.pool["T"], seed=int(self.seed), offset=0, d_state=self.d_state, pool_rows=self.pool_rows,
                           max_bound=self.slice_max, max_bound_slices=self.n_slices, world_frame=1, R=self.R, n1=self.n1,
                           n2=self.n2, eps=float(cfg.surface_eps), stop_eps=float(cfg.stop_eps), min_bound=float(cfg.min_depth),
                           perm=self.perm, z=b["z"], pts=b["pts"], origins=None, dirs_o=None, gt_rgb=b["gt_rgb"],
                           gt_depth=b["gt_depth"], depth_mask=b["depth_mask"], labels=b["labels"])
        else:
            _C.call("cnr_bg_tail", self.flat, self.gflat, self.exp_avg, self.exp_avg_sq, o["partials"], self.nchunk, o["records"],
                    self.nblk, self.gscale, cfg.learning_rate, 0.9, 0.999, 1e-8, cfg.weight_decay, self.d_state, -1, o["packed"],
                    o["rl_ws"], 0 if self.fuse_render else self.R, o["losses"], o["flags"])
        # (self.losses / self.loss are views of / derived from o["losses"]: see the properties -- no torch kernel in the step)

    def _sample(self):
        """a2-a6 for the step at the device cursor (cnr_sample_rays, world frame, per-epoch max-depth table) into self.bufs"""
        cfg = self.cfg
        return ops.sample_rays(self.pool["rgbs"], self.pool["depth"], self.pool["dirs"], self.pool["T"], self.n1, self.n2,
                               cfg.surface_eps, cfg.stop_eps, min_bound=cfg.min_depth, world_frame=True, seed=self.seed,
                               d_state=self.d_state, rays=self.R, out=self.bufs, perm=self.perm, max_bound=self.slice_max,
                               max_bound_slices=self.n_slices)

    def repack(self):
        """Fused tier: rebuild the f16 fragment images from ``self.flat`` (cnr_bg_pack).  Needed once before the first step and
        after any outside change of the parameters (load_state_dict, a copy into ``flat``): inside the training loop every
        weight's fragments are refreshed by the optimiser launch that updates it."""
        _C.call("cnr_bg_pack", self.flat, self.fb["packed"])
        self._packed_for = self.flat._version

    def reset_optimizer(self):
        """a fresh AdamW (both moments zero, step counter 0) -- what the reference's resume amounts to (train.py:40,66-68)"""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.d_state[2:3].zero_()

    def pre_step(self):
        if self.precision == "fused" and (self._packed_for is None or self._packed_for != self.flat._version):
            self.repack()
        if self.cursor >= self.pool_rows - self.R:
            self._reshuffle()

    def post_step(self):
        self.cursor += self.R
        self.steps_done += 1
        if self.precision == "fused":
            self._packed_for = self.flat._version     # (the step's own update: kernels do not bump torch's version counter)

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

    def __init__(self, categories, background, concurrent=None):
        """concurrent: capture the background step on a second stream (forked from / joined to the capturing stream), so that
        the graph holds two independent chains -- the two branches share no parameter and no buffer.  Default: on.
        ``background`` may be None (a scene without a background model, train.py:113 ``if scene_bg is not None``): the iteration is
        then the category step alone and ``run`` / ``step`` are the category trainer's."""
        import os
        self.obj, self.bg = categories, background
        assert not self.obj.grad_exchange, "ray-sharded category steps run a collective between their two graphs"
        self.graphs = {}
        self.steps_done = 0
        self.concurrent = bool(int(os.environ.get("CNR_FULLSTEP_CONCURRENT", "1"))) if concurrent is None else bool(concurrent)
        self.concurrent = self.concurrent and background is not None
        self._side = torch.cuda.Stream(device=self.obj.device) if self.concurrent else None
        # run(): fork once per graph (True) or once per iteration (False: every iteration ends with a join of the two chains)
        self.free_chains = os.environ.get("CNR_FULLSTEP_FREE", "1") != "0"
        self.categories, self.scene_bg = None, None        # the reference-side objects (from_scene)

    # ---- behind the reference's objects (train.py:33-96 builds them, :98-201 is what this class replaces) -----------------------
    @staticmethod
    def pool_of(scene_category):
        """the ``*_batch_all`` pool of a ``sceneCategory`` (src/scene_cateogries.py:211-249) as the fused trainers' pool dict"""
        sc = scene_category
        pool = dict(rgbs=sc.rgbs_batch_all, depth=sc.depth_batch_all, dirs=sc.ray_dirs_batch_all, indices=sc.batch_indices_all)
        if sc.world_frame:           # background and single-object categories: world-frame rays (:374-386, :427-432)
            pool["T_wc"] = sc.t_wc_batch_all
        else:
            pool["T_co"] = sc.t_co_batch_all
        return pool

    @staticmethod
    def module_state(trainer):
        """a ``Trainer``'s parameters in the checkpoint schema ``FusedCategoryTrainer.load_state_dicts`` reads"""
        return dict(FC_state_dict=trainer.fc_occ_map.state_dict(), PE_state_dict=trainer.pe.state_dict(),
                    shape_code_state_dict=trainer.shape_codes.state_dict(),
                    texture_code_state_dict=trainer.texture_codes.state_dict())

    @classmethod
    def from_scene(cls, cls_dict, scene_bg, cfg, rays_per_step=None, seed=0, bg_precision="fused", concurrent=None, **trainer_kw):
        """The fused iteration behind the objects the reference's ``train.py`` builds: ``cls_dict`` {cls_id: sceneCategory}
        (object categories, train.py:55-64), ``scene_bg`` the background ``sceneCategory`` or None (train.py:50-53), ``cfg`` the
        ``Config``.  Takes each category's ray pool as it stands (``*_batch_all``: any lengths, any object counts, world-frame
        single-object categories), its ``trainer.fc_occ_map`` / ``pe`` / ``shape_codes`` / ``texture_codes`` as the initial
        parameters, and the reference's batch size ``n_objs * cfg.n_per_optim // n_cls`` rays per category (train.py:92-96)
        unless ``rays_per_step`` says otherwise.  The background model trains IN PLACE (its modules' parameters become views of
        the step's flat buffer); the categories' modules are refreshed by :meth:`sync_to_modules` -- the copy-back of
        train.py:196-201, needed only before something reads the modules (checkpoint, ``eval_points``, meshing), not per step.

            full = FullStepTrainer.from_scene(cls_dict, scene_bg, cfg)
            for it in range(start, cfg.max_iter, cfg.log_iter):
                full.run(cfg.log_iter)                         # train.py:98-201, cfg.log_iter times
                losses = full.loss_dict()                      # what train.py:186-192 logs
                if it % cfg.save_iter == 0:
                    full.sync_to_modules()
                    for k in vis_dict.values(): k.save_checkpoints(ckpt_dir, it)
        """
        from . import fused
        cats = list(cls_dict.values())
        assert cats, "no object category"
        dev = torch.device(cfg.training_device)
        n_objs = [len(c.obj_ids) for c in cats]
        R = int(rays_per_step) if rays_per_step else sum(n_objs) * cfg.n_per_optim // len(cats)
        tr = fused.FusedCategoryTrainer(cfg, len(cats), n_objs, [cls.pool_of(c) for c in cats], R, dev, seed=seed,
                                        world_frame=[bool(c.world_frame) for c in cats], **trainer_kw)
        for k, c in enumerate(cats):
            tr.load_state_dicts(cls.module_state(c.trainer), k, reset=None)
        bg = None
        if scene_bg is not None:
            hidden = scene_bg.trainer.fc_occ_map.in_layer[0].out_features
            prec = bg_precision if (bg_precision != "fused" or hidden == 128) else "fp32"   # the fused step is built for 128
            bg = BackgroundStep(cfg, cls.pool_of(scene_bg), cfg.n_per_optim_bg, dev, seed=seed, precision=prec,
                                trainer=scene_bg.trainer)
        self = cls(tr, bg, concurrent=concurrent)
        self.categories, self.scene_bg, self.cls_ids = cats, scene_bg, list(cls_dict.keys())
        return self

    def sync_to_modules(self):
        """train.py:196-201: the trained values back into every category's ``trainer.fc_occ_map`` / ``trainer.pe`` -- and, since
        the fused step owns the code tables too, ``shape_codes`` / ``texture_codes`` -- so that ``save_checkpoints``,
        ``Trainer.eval_points`` and meshing read them.  The background's modules are views of the live buffer already."""
        assert self.categories is not None, "built without the reference's objects: use state_dicts()"
        with torch.no_grad():
            for k, c in enumerate(self.categories):
                sd, t = self.obj.state_dicts(k), c.trainer
                t.fc_occ_map.load_state_dict(sd["FC_state_dict"])
                t.pe.B_layer.weight.copy_(sd["PE_state_dict"]["B_layer.weight"])
                t.shape_codes.weight.copy_(sd["shape_code_state_dict"]["weight"])
                t.texture_codes.weight.copy_(sd["texture_code_state_dict"]["weight"])

    def load_from_modules(self):
        """the other direction: after ``sceneCategory.load_checkpoints`` (a resume, train.py:66-86) the modules' values become the
        fused trainers' parameters; the optimiser starts afresh, as the reference's does"""
        assert self.categories is not None
        for k, c in enumerate(self.categories):
            self.obj.load_state_dicts(self.module_state(c.trainer), k, reset=None)
        self.obj.reset_optimizer()
        if self.bg is not None:
            self.bg.reset_optimizer()
            if self.bg.precision == "fused":       # the modules are views of bg.flat: their new values need new fragment images
                self.bg.repack()

    def loss_dict(self):
        """the last iteration's loss terms in the layout train.py:156-180 logs: {'depth','color','opacity': (C,)} for the
        categories, 'background': the same three for the background model (device tensors, no sync)"""
        l = self.obj.loss_values()
        d = {"depth": l[0], "color": l[1], "opacity": l[2]}
        if self.bg is not None:
            b = self.bg.losses
            d["background"] = {"depth": b[0:1], "color": b[1:2], "opacity": b[2:3]}
        return d

    def _both(self):
        """background + categories, on one stream or forked onto two"""
        if self.bg is None:
            return self.obj._step_body()
        if not self.concurrent:
            self.bg._body()
            self.obj._step_body()
            return
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.bg._body()
        self.obj._step_body()
        cur.wait_stream(self._side)

    def run(self, n, unroll=8):
        """``n`` iterations, the same arithmetic as ``n`` calls of ``step()`` (bitwise: tests/test_bg_fused_gpu.py), with groups of up
        to ``unroll`` iterations captured as ONE hipGraph -- both branches keep cursor, RNG step and optimiser step on the device, so
        a group replays unchanged; it never crosses an epoch end of either pool (the reshuffles are host-launched).  Between two
        graph launches the GPU idles ~8 us: 5 % of a 0.15 ms iteration."""
        o, b = self.obj, self.bg
        if b is None:
            self.steps_done += n
            return o.run(n, unroll)
        # even group sizes only, like FusedCategoryTrainer.run: a group leaves the parameter / state ping-pong where it found it
        unroll = max(2, int(unroll) // 2 * 2)
        while n > 0:
            o._pre_step()
            b.pre_step()
            left = min(-(-(o.pool_rows - o.Rg - o.cursor) // o.Rg), -(-(b.pool_rows - b.R - b.cursor) // b.R))
            U = 0
            if self.steps_done >= 3 and o.use_graph:
                for u in o._group_sizes(min(unroll, o.unroll)):
                    if u <= n and u <= left:
                        U = u
                        break
            if not U:
                self.step()
                n -= 1
                continue
            key = (o.parity, U)
            if key not in self.graphs:
                par0 = o.parity
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if self.concurrent and self.free_chains:
                        # the two chains of all U iterations as two branches that meet only at the end of the graph: iteration
                        # i + 1 of one chain waits for nothing of the other (they share no parameter and no buffer)
                        cur = torch.cuda.current_stream()
                        self._side.wait_stream(cur)
                        with torch.cuda.stream(self._side):
                            for i in range(U):
                                self.bg._body()
                        for i in range(U):
                            o._out_slot = i if i < U - 1 else None
                            o._step_body()
                            o.parity ^= 1
                        cur.wait_stream(self._side)
                    else:
                        for i in range(U):
                            o._out_slot = i if i < U - 1 else None
                            self._both()
                            o.parity ^= 1
                o._out_slot, o.parity = None, par0
                self.graphs[key] = g
            assert U % 2 == 0
            self.graphs[key].replay()
            before = o.steps_done
            o._last_multi = U - 1
            o.cursor += U * o.Rg
            o.steps_done += U                     # (U is even: the state parity is where it was)
            b.cursor += U * b.R
            b.steps_done += U
            if b.precision == "fused":
                b._packed_for = b.flat._version
            self.steps_done += U
            n -= U
            if o.check_every and o.steps_done // o.check_every != before // o.check_every:
                o.check_flags()

    def step(self):
        o, b = self.obj, self.bg
        if b is None:
            self.steps_done += 1
            return o.step()
        o._pre_step()
        b.pre_step()
        par = o.parity
        if self.steps_done < 3:
            self._both()
        else:
            if par not in self.graphs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._both()
                self.graphs[par] = g
            self.graphs[par].replay()
        o._post_step()
        b.post_step()
        self.steps_done += 1
