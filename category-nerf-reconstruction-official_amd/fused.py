"""Single-process fast path of the train step (train.py:98-201, object branch): every hot op is one of
the HIP kernels, all parameters of all classes live in ONE flat fp32 buffer (one AdamW launch, one
gradient all-reduce), the ray pool is device-resident, and the whole step -- sample -> latent rows ->
fused field forward -> composite -> loss -> composite backward -> fused field backward -> latent backward ->
AdamW -> advance -- is captured once into a hipGraph and replayed (its cursor, RNG and optimiser step
counters live on the device, cnr_step_advance).

Per class the flat parameter row is
    [ trunk 13892 | latent W (4,32,L) | latent b (4,32) | B (21,3) | shape codes (n_obj,L) | texture codes (n_obj,L) ]
with the latent slots ordered shape_latent_layer_1, cat_latent_layer, shape_latent_layer_2,
texture_latent_layer_1 (include/cnr_hip.h).  ``state_dicts()`` exports reference-named tensors
(src/scene_cateogries.py:548-571 checkpoint keys).
"""
import math
import os

import torch

from . import _C, ops, parallel
from .embedding import UNIDIRS
from .ops import LATENT_LAYERS, TRUNK_LAYERS, TRUNK_PARAMS

_LAT_FANIN_TARGET = ops._LATENT_TARGETS  # (w_off, b_off, ld) of the trunk layer each latent slot feeds


class ParamLayout:
    def __init__(self, L, n_obj):
        self.L, self.n_obj = L, n_obj
        o = 0
        self.trunk = (o, o + TRUNK_PARAMS); o += TRUNK_PARAMS
        self.latW = (o, o + 4 * 32 * L); o += 4 * 32 * L
        self.latb = (o, o + 4 * 32); o += 4 * 32
        self.B = (o, o + 63); o += 63
        self.shape = (o, o + n_obj * L); o += n_obj * L
        self.tex = (o, o + n_obj * L); o += n_obj * L
        self.total = o

    def views(self, flat):
        """flat (C, total) -> dict of views (no copies)."""
        C = flat.shape[0]
        s = lambda r: flat[:, r[0]:r[1]]
        return dict(trunk=s(self.trunk), latW=s(self.latW).view(C, 4, 32, self.L), latb=s(self.latb).view(C, 4, 32),
                    B=s(self.B).view(C, 21, 3), shape=s(self.shape).view(C, self.n_obj, self.L),
                    tex=s(self.tex).view(C, self.n_obj, self.L))


def init_params(C, L, n_obj, generator=None, device="cpu"):
    """Reference initialisation: xavier-normal weights + nn.Linear default biases (src/model.py:4-6,
    src/trainer.py:40), codes randn/sqrt(L/2) (src/trainer.py:57-58), B = icosahedral directions."""
    lay = ParamLayout(L, n_obj)
    flat = torch.zeros(C, lay.total)
    v = lay.views(flat)
    g = generator

    def linear(out_f, in_f):
        w = torch.randn(C, out_f, in_f, generator=g) * math.sqrt(2.0 / (in_f + out_f))
        b = (torch.rand(C, out_f, generator=g) * 2 - 1) / math.sqrt(in_f)
        return w, b
    off = 0
    for _, o, i in TRUNK_LAYERS:
        w, b = linear(o, i)
        v["trunk"][:, off:off + o * i] = w.reshape(C, -1); off += o * i
        v["trunk"][:, off:off + o] = b; off += o
    for k in range(4):
        w, b = linear(32, L)
        v["latW"][:, k] = w
        v["latb"][:, k] = b
    v["B"][:] = torch.tensor(UNIDIRS, dtype=torch.float32)
    v["shape"][:] = torch.randn(C, n_obj, L, generator=g) / math.sqrt(L / 2)
    v["tex"][:] = torch.randn(C, n_obj, L, generator=g) / math.sqrt(L / 2)
    return flat.to(device), lay


class FusedCategoryTrainer:
    """C classes x n_obj objects, R rays per class per step, S = n1 + n2 samples per ray.

    Multi-GPU (one process per GPU, ``process_group`` = a torch.distributed group; backend "nccl" is RCCL), two ways --
    SURVEY.md section 8(e):

    * ``shard="class"``: this rank owns the classes ``class_ids`` (default ``rank::world`` of ``n_cls_global``); ``n_cls``
      and ``pools`` are the LOCAL classes.  Classes share nothing (train.py:58-64, src/loss.py:70), so there is NO
      gradient collective and the step stays one captured graph.  The one coupling, the any-class-empty rule of
      src/render_rays.py:67-72, crosses ranks once per EPOCH: three flags per slice, all-reduced when the pool is
      reshuffled (see ``_reshuffle``).
    * ``shard="ray"``: every rank holds the same pools and parameters and takes rows ``[rank R, rank R + R)`` of each
      global slice of ``world * R`` rows.  Masked means divide by the class's GLOBAL mask counts and the invalid-depth
      rays use the global slice's max depth -- both come from per-epoch tables every rank computes from its own copy of
      the pool, no collective -- the Philox draws are indexed by the global ray, and the flat gradient is summed by ONE
      all-reduce per step: N ranks x R rays are one rank x N R rays to fp32 summation order (tests/test_multigpu_gpu.py).
    """

    def __init__(self, cfg, n_cls, n_obj, pools, rays_per_step, device, seed=0, generator=None,
                 grad_scale=None, bwd_blocks=0, process_group=None, use_graph=True, split_graph=False,
                 fuse_render=True, split_weights=None, shard=None, n_cls_global=None, class_ids=None,
                 check_every=0, world_frame=None, dp_rank=None, dp_world=None, one_launch=None, unroll=16,
                 precise_geometry=None):
        # n_obj: one count for every class, or one per (local) class -- the reference's categories differ (train.py:92-96).  The
        # flat layout uses the largest; a smaller class keeps unused rows (no ray refers to them, no regulariser on them)
        n_obj_list = [int(n_obj)] * n_cls if isinstance(n_obj, int) else [int(v) for v in n_obj]
        assert len(n_obj_list) == n_cls and min(n_obj_list) >= 1
        n_obj = max(n_obj_list)
        self.n_obj_list = n_obj_list
        self.cfg, self.C, self.n_obj, self.R = cfg, n_cls, n_obj, rays_per_step
        self.device = torch.device(device)
        self.n_obj_cls = None if min(n_obj_list) == n_obj else torch.tensor(n_obj_list, device=self.device, dtype=torch.int32)
        self.n1, self.n2 = cfg.n_bins_cam2surface, cfg.n_bins
        self.S = self.n1 + self.n2
        self.L = cfg.net_hyperparams["latent_dim"]
        self.scale = float(cfg.obj_scale)
        self.lr, self.wd = cfg.learning_rate, cfg.weight_decay
        # the code tables are an AdamW group of their own in the reference (train.py:40,54-64); the shipped configs give both
        # groups the same values
        self.code_lr, self.code_wd = float(cfg.code_learning_rate), float(cfg.code_weight_decay)
        # ---- sharding ---------------------------------------------------------------------------------------------
        self.pg = process_group
        pg_world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        pg_rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        # (dp_rank / dp_world without a process group: one rank of a ray-sharded world, the gradient exchange left to the
        #  caller -- tests emulate N ranks in one process this way)
        self.world = int(dp_world) if dp_world is not None else pg_world
        self.rank = int(dp_rank) if dp_rank is not None else pg_rank
        self.shard = shard if shard is not None else ("ray" if self.world > 1 else None)
        assert self.shard in (None, "ray", "class")
        self.ray_world = self.world if self.shard == "ray" else 1       # ranks that share one class's rays
        self.ray_rank = self.rank if self.shard == "ray" else 0
        self.Rg = self.R * self.ray_world                                # rows of a global slice
        if self.shard == "class":
            self.n_cls_global = int(n_cls_global) if n_cls_global is not None else n_cls * self.world
            self.class_ids = list(class_ids) if class_ids is not None else list(range(self.rank, self.n_cls_global, self.world))
            assert len(self.class_ids) == n_cls
        else:
            self.n_cls_global, self.class_ids = n_cls, list(range(n_cls))
        theta_all, self.lay = init_params(self.n_cls_global, self.L, n_obj, generator, "cpu")
        theta0 = theta_all[self.class_ids].contiguous()
        for c, k in enumerate(n_obj_list):          # code rows beyond a class's own objects: unused, kept at zero
            v0 = self.lay.views(theta0)
            v0["shape"][c, k:] = 0
            v0["tex"][c, k:] = 0
        theta0 = theta0.to(self.device)
        # parameters in two copies, like the step state: step k reads copy k & 1, its last launch (AdamW, out of place)
        # writes copy (k + 1) & 1 -- gradient kernels and the optimiser never touch the same copy in one launch
        self.theta2 = torch.stack([theta0, theta0.clone()])
        self.parity = 0
        # gradient of the flat parameters and of the per-object bias rows in ONE allocation: one fill per step
        # ... plus (<= 15 objects per class with the 8-wave backward, <= 4 otherwise) the int64 fixed-point table of the per-object bias-row sums
        # that the field backward fills with integer atomics for cnr_step_tail
        n_th, n_db = self.theta.numel(), n_cls * n_obj * 128
        # the field backward leaves per-workgroup records + the fixed-point table; ONE later launch reduces them next to
        # the latent backward: with AdamW and the epilogue when no gradient exchange follows (cnr_step_tail: single GPU,
        # class sharding), gradient only when rays are sharded (cnr_step_grad: the all-reduce comes between gradient and
        # optimiser)
        # Forward + render / loss + the whole backward in ONE launch (cnr_field_train, the 8-wave kernel's record path): a ray takes
        # 16 / 32 / 64 / 128 padded sample slots; taken when at least 60 % of them are real samples (S = 10, 16, 20-32, 39-64,
        # 77-128), else two calls, whose extra forward costs less than the dead lanes would.  CNR_ONE_LAUNCH=0 keeps the two calls.
        one = bool(int(os.environ.get("CNR_ONE_LAUNCH", "1"))) if one_launch is None else bool(one_launch)
        slots = 16 if self.S <= 16 else 32 if self.S <= 32 else 64 if self.S <= 64 else 128
        # per-object bias-row sums: up to 15 objects per class in the kernel's row-sum blocks (any path); 16 .. 128 on the one-launch
        # path only, with a whole ray per tile group (>= 32 slots per ray: one object per tile, sums straight to the fixed-point
        # table) -- such a class takes that path whatever its slot fill (the reference allows 100 instances, n_models)
        if n_obj > 15:
            if not (fuse_render and self.S <= 128 and n_obj <= 128):
                raise ValueError(f"a class of {n_obj} objects trains on the one-launch step body: at most 128 objects, at most 128 "
                                 f"samples per ray (got S = {self.S}, fuse_render = {fuse_render})")
            one = True              # (rays of up to 16 samples get a 32-slot tile of their own there: cnr_field_train_blocks)
        else:
            one = one and fuse_render and self.S <= 128 and self.S >= 0.6 * slots
        self.use_records = True
        self.grad_exchange = self.shard == "ray" and self.world > 1
        self.fused_tail = not self.grad_exchange and self.use_records
        fix_off = (n_th + n_db + 3) // 4 * 4                      # 16-byte aligned
        self._gbuf = torch.zeros(fix_off + 2 * 8 * n_db, device=self.device)  # 8 copies
        self.rows_fix = self._gbuf[fix_off:].view(torch.int64)
        self.grad = self._gbuf[:self.theta.numel()].view_as(self.theta)
        self.exp_avg = torch.zeros_like(self.theta)
        self.exp_avg_sq = torch.zeros_like(self.theta)
        # device-resident pools stacked over classes: (C, Npool, ...).  A class that trains in the WORLD frame (the
        # reference does that for single-object categories: origin_dirs_W on T_wc, src/scene_cateogries.py:427-432) keeps
        # inv(T_wc) in the pose slot: the sampler's origin_dirs_O inverts it back (rigid, so exact to fp32 rounding)
        if world_frame is None:
            world_frame = [k == 1 and "T_wc" in p for k, p in zip(n_obj_list, pools)]
        self.world_frame = [bool(world_frame)] * n_cls if isinstance(world_frame, (bool, int)) else [bool(w) for w in world_frame]
        # Pools of different lengths (every real scene: a category's pool is the sum of its instances' crop areas,
        # src/scene_cateogries.py:164-260) are padded to the longest; a padding row is never referenced, because the step
        # reaches the pool only through the permutation `perm`, which _reshuffle fills per class from the class's OWN rows
        self.pool_rows_cls = [int(p["depth"].shape[0]) for p in pools]
        self.pool_rows = max(self.pool_rows_cls)
        self.ragged = min(self.pool_rows_cls) != self.pool_rows

        def padded(t):
            t = t.to(self.device)
            if t.shape[0] == self.pool_rows:
                return t
            return torch.cat([t, t[:1].expand(self.pool_rows - t.shape[0], *t.shape[1:])])
        st = lambda k: torch.stack([padded(p[k]) for p in pools]).contiguous()
        T = torch.stack([padded(torch.linalg.inv(p["T_wc"]) if wf else p["T_co"]) for p, wf in zip(pools, self.world_frame)])
        self.pool = dict(rgbs=st("rgbs"), depth=st("depth"), dirs=st("dirs"), T=T.contiguous(), indices=st("indices"))
        assert min(self.pool_rows_cls) >= 2 * self.Rg, "every class's pool must hold two slices of rays"
        if self.shard == "class" and self.pg is not None and self.world > 1:
            # the per-epoch OR of the empty flags is a collective issued at every reshuffle: all ranks must reshuffle in the
            # same steps, i.e. hold pools of the same length and take the same rays per step -- checked here, where a
            # mismatch is a message instead of a hang at the first epoch end
            mine = torch.tensor([self.pool_rows, self.Rg], dtype=torch.int64,
                                device=self.device if torch.distributed.get_backend(self.pg) == "nccl" else "cpu")
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            torch.distributed.all_gather(every, mine, group=self.pg)
            every = [tuple(int(v) for v in t.tolist()) for t in every]
            if len(set(every)) != 1:
                raise ValueError(f"class shards need the same pool length and rays per step on every rank, got (pool_rows, rays) = {every}")
        # device-side step state {pool cursor, rng step, optimiser step}, two copies: step k reads copy k & 1 and its
        # last kernel writes copy (k + 1) & 1 -- no kernel ever writes a state another kernel of the same step reads
        self.d_state2 = torch.zeros(2, 3, device=self.device, dtype=torch.int64)
        # epoch shuffle as an index permutation (C, pool_rows): the pool itself never moves.  Drawn from a generator of
        # its own so that every rank of a sharded run draws the same permutations, whatever else uses the default one
        self.perm = torch.empty(n_cls, self.pool_rows, device=self.device, dtype=torch.int32)
        self._perm_gen = torch.Generator(device=self.device)
        self._perm_gen.manual_seed(0x5EED + 7919 * int(seed))
        self._cursor0 = torch.tensor([self.ray_rank * self.R], device=self.device, dtype=torch.int64)
        # epoch shuffle: cnr_epoch_perm (one launch; the order is a function of seed, epoch and global class id) unless
        # CNR_EPOCH_PERM=torch asks for torch.randperm from one generator, as rounds 1-3 did (a dozen launches per epoch end)
        self._torch_perm = os.environ.get("CNR_EPOCH_PERM", "kernel") == "torch"
        assert not (self.ragged and self._torch_perm), "pools of different lengths need the kernel permutation (cnr_epoch_perm)"
        # ragged pools: each class walks its OWN epochs (a reshuffle whenever i_batch >= N_c - n, src/scene_cateogries.py:439-449);
        # per class the epoch it is in and the slice of that epoch the next filled permutation row starts with
        self._cls_epoch = [0] * n_cls
        self._cls_slice = [0] * n_cls
        self._perm_scratch = torch.empty(self.pool_rows, device=self.device, dtype=torch.int32) if self.ragged else None
        self._perm_seed, self._epoch, self._cursor0_host = 0x5EED + 7919 * int(seed), 0, self.ray_rank * self.R
        self._class_ids_dev = torch.tensor(self.class_ids, device=self.device, dtype=torch.int32)
        # per-epoch tables, one entry per LOCAL slice (index = device cursor / R): the max depth of the (global) slice
        # (scene_cateogries.py:486) and the mask counts + any-class-empty flags of the (global) slice
        # (render_rays.py:66-95) -- no step computes a maximum or counts a mask
        self.n_gslices = self.pool_rows // self.Rg
        self.n_slices = self.n_gslices * self.ray_world
        self.slice_max = torch.zeros(n_cls, self.n_slices, device=self.device)
        self.counts_tab = torch.zeros(self.n_slices, n_cls + 1, 4, device=self.device)
        self.cursor = 0
        self.seed = int(seed) + 1
        self.grad_scale = float(grad_scale) if grad_scale else float(2 ** round(math.log2(max(self.Rg, 2))))
        # workgroups per class of the field kernels (0 = this default): every workgroup copies the class's 61 KB operand image in
        # and leaves a 58 KB gradient record, so with several classes per GPU fewer, longer-running workgroups per class win --
        # 256 in total, one per CU, in one round.  Measured (2048 x 64 per class, M rays/s; 256 per class -> 256 / C per class):
        # 2 classes 39.5 -> 43.1, 3: 41.2 (= at 170), 4: 43.7 -> 47.7 (48.7 at 128), 8: 50.9 (at 64) -> 53.7, 16: 46.4 -> 57.0
        # (241 MB of records per step -> 15 MB); 1.5 rounds (2 x 192) is the worst choice: 33.5
        self.bwd_blocks = int(bwd_blocks) if int(bwd_blocks) > 0 else max(4, 256 // max(n_cls, 1))
        self.bufs = {}
        self.losses = torch.zeros(3, n_cls, device=self.device)
        self.flags = torch.zeros(n_cls, device=self.device, dtype=torch.int32)
        self.clamp = torch.zeros(n_cls, device=self.device, dtype=torch.int32)
        self.check_every = int(check_every)
        # run(n): `unroll` (even) steps per hipGraph launch.  Consecutive graph launches leave ~8 us of idle GPU between them
        # (measured, rocprofv3 kernel trace: the kernels INSIDE a graph follow each other without a gap), which is 12 % of a
        # 65 us step; the steps before the last of such a launch keep their loss values / flags in these slots
        self.unroll = max(2, int(unroll) // 2 * 2)
        self._loss_hist = torch.zeros(self.unroll - 1, 3, n_cls, device=self.device)
        self._flags_hist = torch.zeros(self.unroll - 1, n_cls, device=self.device, dtype=torch.int32)
        self._out_slot = None       # capture of a multi-step graph: which history slot the step being recorded writes
        self._last_multi = 0        # steps in the last launch that left their values in the history slots
        self.dbias = self._gbuf[n_th:n_th + n_db].view(n_cls * n_obj, 4, 32)
        self._nwg = int(_C.load().cnr_field_bwd_pipe_blocks(self.R, self.S, 4, self.bwd_blocks))
        # Precise geometry branch (default ON): the layers between the sample and the x10 occupancy logit as three f16
        # products per fragment, Wh xh + Wl xh + Wh xl (include/cnr_hip.h, cnr_pack_weights_lo).  Plain f16 operands hold
        # north_star's 1e-3 on the occupancy only at initialisation (6.7e-4); after 400 / 5000 training steps they give
        # 1.8e-3 / 2.0e-3 (the logit reaches hundreds), the three-product form 2e-6 (tests/test_trained_parity_gpu.py,
        # DESIGN.md section 3.3).  precise_geometry=False / CNR_PRECISE_GEOMETRY=0: the plain-f16 forward, for the cost
        # comparison only.  (split_weights: the former name of the option.)
        if precise_geometry is None:
            precise_geometry = split_weights
        self.precise = bool(int(os.environ.get("CNR_PRECISE_GEOMETRY", "1"))) if precise_geometry is None \
            else bool(precise_geometry)
        self.split_weights = self.precise
        # forward + render/loss in one launch where the shape allows (S a multiple of 32 up to 128), else two launches
        self._rl_blocks = int(_C.load().cnr_field_fwd_render_blocks(self.R, self.S)) if fuse_render else 0
        # ... and forward + render / loss + the whole backward in ONE launch (``one``, decided above): no second forward, no
        # d sigma / d colour round trip
        self._ft_blocks = int(_C.load().cnr_field_train_blocks(self.R, self.S, self.bwd_blocks, n_obj)) if (one and fuse_render) else 0
        if self._ft_blocks:
            self._nwg = self._ft_blocks
        self.use_graph = use_graph
        self.split_graph = bool(split_graph)     # the two-graph form of the distributed step, for single-GPU tests
        self.graphs = {}                         # parity -> captured graph (or (front, back) pair)
        self.steps_done = 0
        self._reshuffle()

    # ---- one step, eager (also the body that gets captured) ------------------------------------------
    @property
    def theta(self):
        """the current parameters (C, P): what the NEXT step will read"""
        return self.theta2[self.parity]

    @property
    def d_state(self):
        """the state the NEXT step will read (int64[3] view)"""
        return self.d_state2[self.parity]

    def _step_body(self):
        self._step_front()
        if self.grad_exchange and self.pg is not None:
            parallel.allreduce_sum_(self.grad, self.pg)                    # one flat buffer, one collective
        self._step_back()

    def _step_front(self):
        """param prep ... latent backward: everything up to the (optional) gradient all-reduce."""
        C, R, S, n_obj, L = self.C, self.R, self.S, self.n_obj, self.L
        cfg, v, lay = self.cfg, self.lay.views(self.theta), self.lay
        gv = self.lay.views(self.grad)
        o = self.bufs
        if "zl" not in o:
            kw = dict(device=self.device, dtype=torch.float32)
            o["zl"] = torch.empty(C * n_obj, 4, 32, **kw)
            o["brows"] = torch.empty(C * n_obj, 4, 32, **kw)
            o["packed"] = torch.empty(C, _C.pack_bytes(), device=self.device, dtype=torch.uint8)
            # residual weight image of the geometry branch, rebuilt from theta by the step's first launch like `packed`
            o["packed_lo"] = torch.empty(C, int(_C.load().cnr_pack_lo_bytes()), device=self.device, dtype=torch.uint8) \
                if self.precise else None
            o["sig"] = torch.empty(C, R, S, **kw)
            o["rgbs"] = torch.empty(C, R, S, 3, **kw)
            for name, shape in (("depth", (C, R)), ("var", (C, R)), ("rgb", (C, R, 3)), ("opa", (C, R)),
                                ("dsig", (C, R, S)), ("drgb", (C, R, S, 3))):
                o[name] = torch.empty(*shape, **kw)
            o["rl_ws"] = torch.zeros(max(_C.render_loss_workspace_bytes(C, R),
                                         int(_C.load().cnr_field_fwd_render_workspace_bytes(C, R, S)),
                                         int(_C.load().cnr_field_train_workspace_bytes(C, R, S, self.bwd_blocks, n_obj))),
                                     device=self.device, dtype=torch.uint8)
            o["bwd_ws"] = torch.empty(_C.field_bwd_workspace_bytes(C, self.bwd_blocks), device=self.device,
                                      dtype=torch.uint8)
        zl, brows, packed = o["zl"], o["brows"], o["packed"]
        lat_args = (lay.total, lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, C)
        # B, dtrunk, dB: class 0's slice of the flat (C, P) buffers + the class stride P (no gather / scatter copies)
        P = lay.total
        Bc = self.theta[0, lay.B[0]:lay.B[1]]
        g_trunk, g_B = self.grad[0, lay.trunk[0]:lay.trunk[1]], self.grad[0, lay.B[0]:lay.B[1]]
        # One stream, one chain of kernels.  (Parallel hipGraph branches -- parameter prep beside sampling, the loss
        # values beside the field backward -- were measured: every cross-queue edge costs more than the few
        # microseconds of overlap it buys at this step size, 0.154 -> 0.170 ms per step.)
        # First node, one launch, four independent jobs side by side in the grid: zero the gradient buffers | a7 +
        # latent layers (per-object bias rows) | f16 operand image of the trunk | a2-a6: slice the device pool at
        # the device cursor, transform, sample (the slice's max depth comes out of self.slice_max, which _reshuffle
        # filled for the whole epoch)
        b = ops.step_prologue(self.theta, lay, L, n_obj, packed, zl, brows, self._gbuf, self.pool["rgbs"],
                              self.pool["depth"], self.pool["dirs"], self.pool["T"], self.n1, self.n2, cfg.surface_eps,
                              cfg.stop_eps, cfg.min_depth, self.seed, self.d_state2[self.parity], R, self.bufs,
                              self.slice_max, self.pool["indices"], self.perm, max_bound_slices=self.n_slices,
                              rng=self._rng_map(), packed_lo=o["packed_lo"])
        ray_row = b["ray_row"]
        lo = o["packed_lo"]
        # the loss gradient of a ray carries 1 / (GLOBAL mask count of its class) out of counts_tab: ray shards add up to
        # the whole batch's gradient, nothing to rescale; the code regulariser is formed on every ray shard -> 1 / world
        inv_w = 1.0
        st = self.d_state2[self.parity]
        if self._ft_blocks:
            # a8-a15 forward, losses, their gradient and the field backward in ONE launch
            _C.call_struct("cnr_field_train", **self._field_train_args(b, o, Bc, st, inv_w, self.rows_fix, self.clamp))
        elif self._rl_blocks:
            # a8-a15 in one launch (S = 32 k): field forward, composite, losses, their gradient, composite backward;
            # sigma / colour per sample never leave registers
            _C.call("cnr_field_fwd_render", b["pts"], Bc, packed, brows, ray_row, self.scale, b["z"], b["gt_depth"],
                    b["gt_rgb"], b["labels"], b["depth_mask"], 5.0, 10.0, inv_w, o["dsig"], o["drgb"], o["depth"],
                    o["var"], o["rgb"], o["opa"], C, R, S, P, o["rl_ws"], o["rl_ws"].numel(), lo, self.counts_tab, st)
        else:
            # a8 + a9 fused forward, then a11-a15 in one launch
            sig, rgb = o["sig"], o["rgbs"]
            _C.call("cnr_field_fwd", b["pts"], Bc, packed, brows, ray_row, self.scale, sig, rgb, C, R, S, P, lo)
            _C.call("cnr_render_loss", sig, rgb, b["z"], b["gt_depth"], b["gt_rgb"], b["labels"], b["depth_mask"],
                    5.0, 10.0, inv_w, o["dsig"], o["drgb"], o["depth"], o["var"], o["rgb"], o["opa"], C, R, S,
                    o["rl_ws"], o["rl_ws"].numel(), self.counts_tab, st)
        # fused backward: dtrunk, dB, dbiasrows straight into the flat gradient buffer views
        if not self._ft_blocks:
            ops.field_bwd(b["pts"], Bc, packed, brows, ray_row, self.scale, o["dsig"], o["drgb"], self.grad_scale,
                          g_trunk, g_B, self.dbias, C, R, S, n_obj, self.bwd_blocks, o["bwd_ws"],
                          B_stride=P, dtrunk_stride=P, dB_stride=P, rows_fix=self.rows_fix, skip_reduce=True,
                          clamp_flags=self.clamp, packed_lo=lo)
        self._reg = 0.0005 / self.ray_world      # code regulariser scale: loss.py:5-15, train.py:165-167
        if self.grad_exchange:                   # ray shards: the all-reduce needs the complete gradient first
            # record reduction + latent backward in one launch, gradient only
            _C.call("cnr_step_grad", self.theta, self.grad, lay.total, lay.B[0], lay.latW[0], lay.latb[0],
                    lay.shape[0], lay.tex[0], L, n_obj, C, zl, self.dbias, self._reg, o["bwd_ws"], self._nwg,
                    self.rows_fix, self.n_obj_cls)

    def _field_train_args(self, b, o, Bc, st, loss_scale, rows_fix, clamp):
        """the argument block of cnr_field_train on the live buffers of a step (include/cnr_hip.h, cnr_field_train_args)"""
        return dict(pts=b["pts"], B=Bc, packed=o["packed"], packed_lo=o["packed_lo"], biasrows=o["brows"], ray_row=b["ray_row"],
                    scale=self.scale, z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"], labels=b["labels"],
                    depth_mask=b["depth_mask"], counts_tab=self.counts_tab, d_state=st, color_scaling=5.0, opacity_scaling=10.0,
                    loss_scale=float(loss_scale), grad_scale=self.grad_scale, depth=o["depth"], var=o["var"], rgb=o["rgb"],
                    opacity=o["opa"], C=self.C, R=self.R, S=self.S, rows_per_class=self.n_obj, max_blocks=self.bwd_blocks,
                    records=o["bwd_ws"], records_bytes=o["bwd_ws"].numel(), loss_workspace=o["rl_ws"],
                    loss_workspace_bytes=o["rl_ws"].numel(), B_stride=self.lay.total, rows_fix=rows_fix, clamp_flags=clamp)

    def _step_back(self):
        """Last launch (cnr_step_tail): the fixed-order reduction of the field backward's records (single GPU, <= 15
        objects per class; otherwise cnr_field_bwd_pipe did it), latent backward + code regulariser (single GPU; with a
        process group it ran before the all-reduce), AdamW out of place into the other parameter copy, and the
        epilogue (loss values +
        flags from the render kernel's partials, the next slice's max depth, next step state into the other state
        copy) -- side by side in one grid."""
        C, R, o, par, lay = self.C, self.R, self.bufs, self.parity, self.lay
        _C.call_struct(
            "cnr_step_tail", theta_in=self.theta2[par], theta_out=self.theta2[1 - par], grad=self.grad, exp_avg=self.exp_avg,
            exp_avg_sq=self.exp_avg_sq, class_stride=lay.total, off_B=lay.B[0], off_latW=lay.latW[0], off_latb=lay.latb[0],
            off_shape=lay.shape[0], off_tex=lay.tex[0], L=self.L, n_obj=self.n_obj, C=C, zl=o["zl"], dbiasrows=self.dbias,
            reg_scale=self._reg, do_latent=0 if self.grad_exchange else 1, lr=self.lr, beta1=0.9, beta2=0.999, eps=1e-8,
            weight_decay=self.wd, state_cur=self.d_state2[par], state_next=self.d_state2[1 - par], add_rows=self.Rg,
            rl_workspace=o["rl_ws"], losses=self.losses if self._out_slot is None else self._loss_hist[self._out_slot],
            flags=self.flags if self._out_slot is None else self._flags_hist[self._out_slot], depth=None,
            pool_rows=self.pool_rows, perm=None, next_max_bound=None, R=R,
            records=o["bwd_ws"] if self.fused_tail else None, nwg=self._nwg if self.fused_tail else 0,
            rows_fix=self.rows_fix if self.fused_tail else None, rl_blocks=self._ft_blocks or self._rl_blocks,
            clamp_flags=self.clamp, n_obj_cls=self.n_obj_cls, code_lr=self.code_lr, code_weight_decay=self.code_wd)

    def step(self):
        """One train step.  Returns nothing; ``self.losses`` (3,C) / ``self.flags`` (C,) hold the device-side
        loss terms (depth, colour, opacity) and flags of the step just run.

        After two eager steps the step is captured, once per state parity: one hipGraph on a single GPU; with a
        process group TWO graphs around the all-reduce (front graph, eager RCCL call, back graph) -- three host
        calls per step, and no collective inside a capture."""
        self._pre_step()
        split = self.grad_exchange or self.split_graph
        par = self.parity
        if not self.use_graph or self.steps_done < 2:
            self._step_body()
        elif par not in self.graphs:
            if split:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    self._step_front()
                ga.replay()                  # capture only records: run the part it stands for
                if self.grad_exchange and self.pg is not None:
                    parallel.allreduce_sum_(self.grad, self.pg)
                with torch.cuda.graph(gb, pool=ga.pool()):
                    self._step_back()
                gb.replay()
                self.graphs[par] = (ga, gb)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_body()
                g.replay()
                self.graphs[par] = g
        elif split:
            self.graphs[par][0].replay()
            if self.grad_exchange and self.pg is not None:
                parallel.allreduce_sum_(self.grad, self.pg)
            self.graphs[par][1].replay()
        else:
            self.graphs[par].replay()
        self._last_multi = 0
        self._post_step()

    def run(self, n, unroll=None):
        """``n`` train steps, the same arithmetic as ``n`` calls of ``step()`` (bitwise: tests/test_trainer_gpu.py), with
        groups of ``unroll`` steps captured in ONE hipGraph (2 x unroll x 3 kernel nodes, device-side cursor / RNG /
        optimiser state and the parameter ping-pong carry from step to step inside it).  Between two graph launches the
        GPU idles for ~8 us, inside a graph the kernels follow each other directly -- at a 65 us step that is the
        difference between 30.7 and 35 M rays/s.  A group never crosses an epoch end (the reshuffle is host-launched)
        and is not used around a gradient all-reduce; those steps go out one by one.  ``self.losses`` / ``self.flags``
        hold the LAST step's values, ``loss_history()`` the values of every step of the last launch; ``check_flags``
        looks at all of them."""
        U0 = self.unroll if unroll is None else max(2, int(unroll) // 2 * 2)
        U0 = min(U0, self.unroll)
        multi_ok = self.use_graph and not (self.grad_exchange or self.split_graph)
        while n > 0:
            if self.cursor >= self.pool_rows - self.Rg:
                self._reshuffle()
            left = -(-(self.pool_rows - self.Rg - self.cursor) // self.Rg)     # steps before the next reshuffle
            # the largest even group that fits what is left of the request and of the epoch (each launch boundary idles the GPU ~8 us)
            U = 0
            if multi_ok and self.steps_done >= 2:
                for u in self._group_sizes(U0):
                    if u <= n and u <= left:
                        U = u
                        break
            if not U:
                self.step()
                n -= 1
                continue
            key = (self.parity, U)
            if key not in self.graphs:
                self._capture_multi(self.parity, U)
            self.graphs[key].replay()
            self._last_multi = U - 1
            before = self.steps_done
            self.cursor += U * self.Rg
            self.steps_done += U
            n -= U
            if self.check_every and self.steps_done // self.check_every != before // self.check_every:
                self.check_flags()

    @staticmethod
    def _group_sizes(U0):
        """group sizes of the multi-step graphs, largest first: EVERY even size U0, U0 - 2, .., 2.  (A halving ladder -- U0,
        U0 / 2, .. -- sent the 7 steps in front of an epoch end and the 13 behind it out as 4 + 2 + 1 and 10 + 2 + 1: six graph
        launches and the reshuffle's three kernels in a row, ~400 us of host work against ~400 us of GPU work queued: a 20-step
        region with an epoch end inside ran 82 us per step.  With every even size it is 6 + 1 and 12 + 1.)"""
        return list(range(int(U0) // 2 * 2, 1, -2))

    def _capture_multi(self, par, U):
        """Record (not run) U steps starting at state parity ``par`` as one graph."""
        par0 = self.parity
        self.parity = par
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(U):
                self._out_slot = i if i < U - 1 else None
                self._step_body()
                self.parity ^= 1
        self._out_slot, self.parity = None, par0
        self.graphs[(par, U)] = g

    def prepare_graphs(self, unroll=None):
        """Capture every graph ``step()`` / ``run()`` can need -- one step and ``unroll`` steps, from either state parity --
        without running them, so that no capture (a millisecond of host work) lands inside a timed or latency-sensitive
        region later.  Needs two steps done (the buffers exist); a no-op without graphs or around a gradient all-reduce."""
        if not self.use_graph or self.grad_exchange or self.split_graph or self.steps_done < 2:
            return
        U = min(self.unroll if unroll is None else max(2, int(unroll) // 2 * 2), self.unroll)
        par0 = self.parity
        for par in (0, 1):
            if par not in self.graphs:
                self.parity = par
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_body()
                self.graphs[par] = g
            self.parity = par0
            for u in self._group_sizes(U):
                if (par, u) not in self.graphs:
                    self._capture_multi(par, u)
        self.parity = par0

    def loss_history(self):
        """(k, 3, C): the loss terms of the k steps of the last launch (k = 1 after ``step()``), oldest first."""
        return torch.cat([self._loss_hist[:self._last_multi], self.losses[None]])

    def _pre_step(self):
        if self.cursor >= self.pool_rows - self.Rg:    # epoch end: reshuffle (i_batch >= N - n, scene_cateogries.py:439-449)
            self._reshuffle()

    def _post_step(self):
        self.parity ^= 1
        self.cursor += self.Rg
        self.steps_done += 1
        if self.check_every and self.steps_done % self.check_every == 0:
            self.check_flags()

    def _rng_map(self):
        """(first global class, class stride, global rays per class, first ray of this rank): the Philox counter of a ray
        is its index in the GLOBAL batch, so shards draw what a single GPU would (include/cnr_hip.h, cnr_step_prologue)."""
        if self.shard == "class":
            ids = self.class_ids
            stride = ids[1] - ids[0] if len(ids) > 1 else 1
            if all(ids[k] == ids[0] + k * stride for k in range(len(ids))):
                return (ids[0], stride, self.R, 0)
            return (0, 0, 0, 0)       # irregular ownership: local indices (still a valid stream, just not the single-GPU one)
        if self.shard == "ray":
            return (0, 1, self.Rg, self.ray_rank * self.R)
        return (0, 0, 0, 0)

    def check_flags(self):
        """The reference exits on a loss above 1e5 (src/render_rays.py:87-89); the step itself never syncs, so the
        explode bit is looked at here (a host sync: call at the logging cadence, or pass check_every=k).  Returns the
        flag words; bit 4 = the field backward clipped a scaled gradient this step (reported, not fatal)."""
        from .render_rays import LossExplode
        fl = self.flags.cpu()
        if self._last_multi:           # every step of the last multi-step launch
            for row in self._flags_hist[:self._last_multi].cpu():
                fl = fl | row
        if bool((fl & 1).any()):
            raise LossExplode(f"loss explode: step {self.steps_done}, losses {self.losses.cpu().tolist()}")
        return fl

    def loss_values(self):
        """(3, C) depth / colour / opacity terms of the last step over the WHOLE batch: with ray shards every rank holds its
        rays' share of the class sums (already divided by the global counts), so the values add up over ranks."""
        out = self.losses.clone()
        if self.grad_exchange and self.pg is not None:
            torch.distributed.all_reduce(out, group=self.pg)
        return out

    def time_field_train(self, iters=50):
        """Average duration (ms) of the one-launch step body (cnr_field_train) on the live buffers of the last step,
        launched back to back between two HIP events on the launch stream (records / renders are overwritten with the same
        values; the integer row table is scratch here)."""
        assert self._ft_blocks
        C, R, S, o, b, lay = self.C, self.R, self.S, self.bufs, self.bufs, self.lay
        Bc = self.theta[0, lay.B[0]:lay.B[1]]
        fix = torch.zeros_like(self.rows_fix)
        clamp = torch.zeros_like(self.clamp)
        st = self.d_state2[self.parity]
        args = self._field_train_args(b, o, Bc, st, 1.0, fix, clamp)
        # the argument block is filled ONCE (~40 us of host work per call otherwise: as long as the kernel itself, so that a slow
        # host moment showed up as kernel time -- 135 instead of 38 us in one run of eight); median of five batches
        blk = _C._prepare_struct("cnr_field_train", args)
        run = lambda: _C._invoke_struct("cnr_field_train", blk)
        for _ in range(3):
            run()
        per = max(10, iters // 5)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / per)
        return sorted(ts)[2]

    def time_field_bwd(self, iters=50):
        """Average duration (ms) of the dominant call -- the fused field backward on the live buffers of the last step
        -- launched back to back between two HIP events on the launch stream (outputs go to scratch)."""
        C, R, S, n_obj, o, b = self.C, self.R, self.S, self.n_obj, self.bufs, self.bufs
        v = self.lay.views(self.theta)
        Bc = v["B"].contiguous()
        kw = dict(device=self.device, dtype=torch.float32)
        dtrunk, dB, dbias = torch.zeros(C, TRUNK_PARAMS, **kw), torch.zeros(C, 21, 3, **kw), torch.zeros_like(self.dbias)
        run = lambda: ops.field_bwd(b["pts"], Bc, o["packed"], o["brows"], b["ray_row"], self.scale, o["dsig"], o["drgb"],
                                    self.grad_scale, dtrunk, dB, dbias, C, R, S, n_obj, self.bwd_blocks, o["bwd_ws"],
                                    packed_lo=o["packed_lo"])
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def _reshuffle(self):
        """New permutation, cursor back to this rank's first row (scene_cateogries.py:439-449), and the epoch's tables:
        per GLOBAL slice of world * R rows the max depth (cnr_slice_maxdepth) and the mask counts + any-class-empty flags
        (cnr_slice_maskcounts), each entry repeated for the ray ranks that share the slice (the kernels index by their own
        cursor / R).  All on the device; the only communication is, with class shards, one all-reduce (MAX) of the
        (slices, 3) empty flags per epoch."""
        C, ng, w = self.C, self.n_gslices, self.ray_world
        if self._torch_perm:
            for cg in range(self.n_cls_global):    # every rank draws every class's permutation: same epoch everywhere
                p = torch.randperm(self.pool_rows, device=self.device, generator=self._perm_gen)
                if cg in self.class_ids:
                    self.perm[self.class_ids.index(cg)].copy_(p)
            self.d_state2[self.parity, 0:1].copy_(self._cursor0)
        elif self.ragged:
            self._ragged_perm()
        else:
            # one launch: a keyed bijection per (seed, epoch, GLOBAL class id) -- a rank computes its own classes' orders, every
            # rank that holds a class gets the same one -- and the cursor of the current state copy back to this rank's first row
            _C.call("cnr_epoch_perm", self.perm, self.pool_rows, C, self._perm_seed, self._epoch, self._class_ids_dev,
                    self.d_state2[self.parity], int(self._cursor0_host))
            self._epoch += 1
        self.cursor = 0
        direct = w == 1        # no ray shards: the tables go straight into the buffers the kernels read
        smax = self.slice_max if direct else torch.empty(C, ng, device=self.device)
        _C.call("cnr_slice_maxdepth", self.pool["depth"], self.perm, self.pool_rows, C, self.Rg, ng, smax)
        tab = self.counts_tab if direct else torch.empty(ng, C + 1, 4, device=self.device)
        _C.call("cnr_slice_maskcounts", self.pool["rgbs"], self.pool["depth"], self.perm, self.pool_rows, C, self.Rg, ng,
                float(self.cfg.min_depth), tab)
        if self.shard == "class" and self.pg is not None and self.world > 1:
            parallel.allreduce_any_(tab[:, C, :3], self.pg)
        if not direct:
            self.slice_max.copy_(smax.repeat_interleave(w, dim=1))
            self.counts_tab.copy_(tab.repeat_interleave(w, dim=0))

    def _ragged_perm(self):
        """Pools of different lengths.  The reference gives every category its own cursor: class c takes slices of n rows and
        reshuffles ITS pool when i_batch >= N_c - n (src/scene_cateogries.py:436-449), i.e. after k_c = ceil(N_c / n) - 1 slices.
        The step's kernels share one device cursor, so the permutation row of class c is laid out as what that class will read
        over the next K = max_c k_c steps: the rest of its current epoch's order, then as many further epochs of ITS OWN pool
        (each a keyed bijection of (seed, class epoch, global class id) on [0, N_c): cnr_epoch_perm) as fit.  Where a class is
        in its epoch carries over to the next call.  Host work per (longest-class) epoch, nothing per step."""
        Rg = self.Rg
        K = -(-self.pool_rows // Rg) - 1               # steps until the host reshuffles again (= the longest class's epoch)
        for c, N in enumerate(self.pool_rows_cls):
            k_c = -(-N // Rg) - 1                      # slices per epoch of this class
            row, filled = self.perm[c], 0
            while filled < K:
                _C.call("cnr_epoch_perm", self._perm_scratch, N, 1, self._perm_seed, self._cls_epoch[c],
                        self._class_ids_dev[c:c + 1], None, 0)
                take = min(k_c - self._cls_slice[c], K - filled)
                a = self._cls_slice[c] * Rg
                row[filled * Rg:(filled + take) * Rg].copy_(self._perm_scratch[a:a + take * Rg])
                filled += take
                self._cls_slice[c] += take
                if self._cls_slice[c] == k_c:
                    self._cls_slice[c], self._cls_epoch[c] = 0, self._cls_epoch[c] + 1
            row[K * Rg:].zero_()                       # never read (the host reshuffles after K slices); kept in range
        self.d_state2[self.parity, 0:1].copy_(self._cursor0)

    # ---- reference-named export ------------------------------------------------------------------------
    def state_dicts(self, c=0):
        v = self.lay.views(self.theta)
        fc, off = {}, 0
        for n, o, i in TRUNK_LAYERS:
            fc[n + ".weight"] = v["trunk"][c, off:off + o * i].reshape(o, i).clone(); off += o * i
            fc[n + ".bias"] = v["trunk"][c, off:off + o].clone(); off += o
        for k, n in enumerate(LATENT_LAYERS):
            fc[n + ".weight"] = v["latW"][c, k].clone()
            fc[n + ".bias"] = v["latb"][c, k].clone()
        return dict(FC_state_dict=fc, PE_state_dict={"B_layer.weight": v["B"][c].clone(), "scale": torch.tensor(self.scale)},
                    shape_code_state_dict={"weight": v["shape"][c, :self.n_obj_list[c]].clone()},
                    texture_code_state_dict={"weight": v["tex"][c, :self.n_obj_list[c]].clone()}, obj_scale=self.scale)

    def load_state_dicts(self, d, c=0, reset="class"):
        """Inverse of :meth:`state_dicts` for local class ``c``: a checkpoint dict in the reference's key schema
        (src/scene_cateogries.py:548-571: FC_state_dict, PE_state_dict, shape_code_state_dict, texture_code_state_dict)
        goes into BOTH parameter copies.  ``reset``: what happens to the optimiser --

        * ``"class"`` (default): the AdamW moments of class ``c`` alone are zeroed; the other classes keep their history (a class
          replaced or hot-loaded in the middle of training does not disturb its neighbours).  The optimiser step counter is one
          per trainer and keeps running.
        * ``"all"``: :meth:`reset_optimizer` -- a fresh AdamW for every class, step counter back to 0.  This is the reference's
          resume (it stores no optimiser state and builds a new AdamW, train.py:40,66-68): load every class, the last one
          with ``reset="all"`` (or call ``reset_optimizer()`` yourself).
        * ``None``: the optimiser state is left alone."""
        assert reset in ("class", "all", None)
        fc = d["FC_state_dict"]
        for th in (self.theta2[0], self.theta2[1]):
            v = self.lay.views(th)
            off = 0
            for n, o, i in TRUNK_LAYERS:
                v["trunk"][c, off:off + o * i].copy_(fc[n + ".weight"].reshape(-1)); off += o * i
                v["trunk"][c, off:off + o].copy_(fc[n + ".bias"]); off += o
            for k, n in enumerate(LATENT_LAYERS):
                v["latW"][c, k].copy_(fc[n + ".weight"])
                v["latb"][c, k].copy_(fc[n + ".bias"])
            v["B"][c].copy_(d["PE_state_dict"]["B_layer.weight"])
            v["shape"][c, :self.n_obj_list[c]].copy_(d["shape_code_state_dict"]["weight"])
            v["tex"][c, :self.n_obj_list[c]].copy_(d["texture_code_state_dict"]["weight"])
        if reset == "all":
            self.reset_optimizer()
        elif reset == "class":
            self.exp_avg[c].zero_()
            self.exp_avg_sq[c].zero_()

    def reset_optimizer(self):
        """A fresh AdamW for EVERY class: both moments zero and the device-side optimiser step counter back to 0 (the counter
        is one per trainer: bias correction with a large step on zeroed moments would mis-scale the first updates after a
        resume).  The reference saves no optimiser state and builds a new AdamW for all classes when it resumes
        (train.py:40,66-68)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.d_state2[:, 2] = 0
