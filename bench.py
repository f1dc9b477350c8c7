#!/usr/bin/env python3
"""Benchmark of the volumetric-rendering train step (BASELINE.json metric: rays/sec, train step).

One step = sample (a2-a6) -> latent rows (a7) -> PE + CodeNeRF trunk (a8, a9) -> composite (a11-a13) -> loss
(a14-a16) -> backward of all of it -> AdamW (a18), on synthetic random-pose ray pools resident in HBM
(SURVEY.md section 8(d)).  N = 1 runs BASELINE.json configs[1] (Replica room_0 shape: 1 category, 2048 rays x 64
samples, L = 256).  N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL; by default every GPU owns whole
categories (--shard class: one more category of the same shape per GPU, no gradient collective -- categories share nothing,
train.py:58-64 -- weak scaling; --scaling strong --classes 16 is BASELINE.json configs[2] on a fixed global batch);
--shard ray splits the rays of every category instead (one all-reduce of the flat gradient per step).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC: RCCL across the ranks of one node needs it on this driver; it must be in the environment before the HIP runtime
# starts, also when the ranks are started by somebody else's torch.distributed.run
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_wait_spin():
    """torch.cuda.synchronize() returns by polling instead of sleeping on an interrupt (hipDeviceScheduleSpin; must be set before the
    first HIP call of the process, and in the HIP runtime torch has loaded).  A 20-step timed region is 1.15 ms of GPU work
    between two synchronisations: the interrupt path's wake-up was ~25 us of it (tools/region_overhead.py: median region
    1204.6 -> 1181.3 us); the GPU work is the same."""
    import ctypes
    try:
        # the copy torch has mapped, by its path: a bare soname could load a second runtime from the system directories
        paths = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
        if len(paths) != 1:
            return False
        return ctypes.CDLL(paths.pop()).hipSetDeviceFlags(1) == 0
    except (OSError, AttributeError):
        return False

FLOP_PER_SAMPLE_STEP = 82140          # 3 x 13 648 MAC + 2 x 63 MAC, 2 FLOP/MAC
PEAK_MFMA_F16_TFLOPS = 2500.0         # MI355X dense bf16/f16 MFMA peak (MI355X_MICROARCH.md, chip-level table)


def _cpu_info():
    """CPU model string, physical cores (sockets x cores per socket) and logical CPUs of this host."""
    model, phys, sockets, cores = "unknown", None, set(), None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                sockets.add(v)
            elif k == "cpu cores" and cores is None:
                cores = int(v)
        if cores:
            phys = cores * max(len(sockets), 1)
    except OSError:
        pass
    return model, phys, os.cpu_count()


def cpu_baseline(cfg_shape, seconds=12.0):
    """The oracle's train step (reference op structure, fp32) on the host cores: rays/s on the same shape, with the
    per-phase split SURVEY.md section 8(d) asks for (sample / PE / MLP forward / loss / backward / optimiser)."""
    from oracle import ref_cpu as O
    C, R, n1, n2, L, n_obj = cfg_shape
    S = n1 + n2
    gen = torch.Generator().manual_seed(1234)
    mlp = {k: v.requires_grad_() for k, v in O.init_codenerf_params(C, 32, L, gen).items()}
    B = torch.tensor(O.UNIDIRS).view(21, 3).repeat(C, 1, 1).requires_grad_()
    sh = [O.init_codes(n_obj, L, gen).requires_grad_() for _ in range(C)]
    tx = [O.init_codes(n_obj, L, gen).requires_grad_() for _ in range(C)]
    params = list(mlp.values()) + [B] + sh + tx
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.013)
    import cnr_amd
    pool = cnr_amd.scene_cateogries.synthetic_pool(4 * R, n_obj, gen, "cpu")
    PH = ("sample", "pe", "mlp_fwd", "loss", "backward", "optim")

    def step(i, ph=None):
        t = [time.perf_counter()]
        mark = lambda: t.append(time.perf_counter())
        sl = slice((i % 3) * R, (i % 3) * R + R)
        rgbs, depth, dirs, T, idx = pool["rgbs"][sl], pool["depth"][sl], pool["dirs"][sl], pool["T_co"][sl], pool["indices"][sl]
        o, d = O.origin_dirs_O(T, dirs)
        u = torch.rand(R, S, generator=gen)
        g = torch.randn(R, n2, generator=gen) * (0.1 / 3)
        gt_rgb, gt_depth, mask, labels, pts, z = O.sample_3d_points(rgbs, depth, o, d, u, g, n1, n2, 0.1, 0.05)
        pts, z = pts[None].repeat(C, 1, 1, 1), z[None].repeat(C, 1, 1)
        idxc = idx[None].repeat(C, 1)
        mark()
        e = O.unidirs_embed(pts, B, 2.0)                                        # materialised (C,R,S,129), as the reference
        mark()
        cs = torch.stack([sh[c][idxc[c]][:, None, :] for c in range(C)])        # train.py:136-137
        ct = torch.stack([tx[c][idxc[c]][:, None, :] for c in range(C)])
        sig, col = O.codenerf_forward(mlp, e, cs, ct)
        mark()
        loss, _, _ = O.step_batch_loss(sig, col, gt_depth[None].repeat(C, 1), (gt_rgb / 255.0)[None].repeat(C, 1, 1),
                                       labels[None].repeat(C, 1), mask[None].repeat(C, 1), z)
        rs, rt = O.step_batch_loss_reg(sh, tx)
        loss = loss + 0.0005 * (rs + rt).sum()
        mark()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        mark()
        opt.step()
        mark()
        if ph is not None:
            for k, name in enumerate(PH):
                ph[name].append(t[k + 1] - t[k])
        return t[-1] - t[0]

    # pick the thread count that is fastest for THIS workload (a 128-thread box is slower with all threads on
    # these small GEMMs than with 16-32), on the median of three steps each: the baseline is the CPU's best
    model, phys, logical = _cpu_info()
    best = None
    for nt in sorted({8, 16, 32, 64, torch.get_num_threads()}):
        if nt > (logical or 1):
            continue
        torch.set_num_threads(nt)
        step(0)
        ts = sorted(step(1 + k) for k in range(3))
        if best is None or ts[1] < best[0]:
            best = (ts[1], nt)
    cores = best[1]
    torch.set_num_threads(cores)
    ph = {k: [] for k in PH}
    times, t_end, i = [], time.perf_counter() + seconds, 0
    while time.perf_counter() < t_end or len(times) < 3:
        times.append(step(i, ph))
        i += 1
    times.sort()
    med = times[len(times) // 2]
    phase_ms = {k: round(sorted(v)[len(v) // 2] * 1e3, 2) for k, v in ph.items()}
    return {"value": C * R / med, "unit": "rays/s", "cores": cores, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "phase_ms": phase_ms,
            "sample": f"{len(times)} steps of the same {C}x{R}x{S} (L={L}) train step, median {med * 1e3:.1f} ms, "
                      f"torch fp32 oracle with {cores} threads (best of 8/16/32/64/default on 3-step medians)"}


def fp8_leg(args):
    """configs[4] (8192 rays x 128 samples, fp8 weights on the CDNA4 fp8 MFMA): the field forward with 1 / 2 / 3 fp8 planes
    against the f16 kernel -- samples/s (HIP events, back to back) and rendered-occupancy error against the CPU oracle on a
    bounded sample.  The train step has no fp8 path: one plane is 60x outside north_star's 1e-3, the split that reaches it
    costs nine MFMAs where f16 needs one."""
    import cnr_amd
    from oracle import ref_cpu as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = torch.device("cuda", 0)
    R, S, L, n_obj = (8192, 128, args.latent, 4) if args.rays == 2048 else (args.rays, args.samples, args.latent, 4)
    gen = torch.Generator().manual_seed(1234)
    theta, lay = cnr_amd.fused.init_params(1, L, n_obj, gen, dev)
    v = lay.views(theta)
    trunk = v["trunk"].contiguous()
    packed = cnr_amd.ops.pack_weights(trunk)
    zl, brows = torch.empty(n_obj, 4, 32, device=dev), torch.empty(n_obj, 4, 32, device=dev)
    cnr_amd._C.call("cnr_latent_fwd", theta, lay.total, lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, 1, zl, brows)
    pts = (torch.rand(1, R, S, 3, generator=gen) * 2 - 1).to(dev)
    idx = torch.randint(0, n_obj, (1, R), generator=gen)
    ray_row = idx.to(torch.int32).to(dev)
    B = v["B"].contiguous()

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    legs = {}
    ms = timed(lambda: cnr_amd.ops.field_fwd(pts, B, packed, brows, ray_row, 2.0))
    outs = {"f16": cnr_amd.ops.field_fwd(pts, B, packed, brows, ray_row, 2.0)}
    legs["f16"] = {"ms": ms, "samples_per_s": R * S / (ms * 1e-3), "mfma_per_fragment": 1}
    for t in (1, 2, 3):
        p8 = torch.empty(1, int(cnr_amd._C.load().cnr_pack_fp8_bytes(t)), device=dev, dtype=torch.uint8)
        cnr_amd._C.call("cnr_pack_weights_fp8", trunk, p8, 1, t)
        sig, rgb = torch.empty(1, R, S, device=dev), torch.empty(1, R, S, 3, device=dev)
        run = lambda: cnr_amd._C.call("cnr_field_fwd_fp8", pts, B, packed, p8, brows, ray_row, 2.0, sig, rgb, 1, R, S, 0, t)
        ms = timed(run)
        outs[f"fp8x{t}"] = (sig.clone(), rgb.clone())
        legs[f"fp8_{t}_plane"] = {"ms": ms, "samples_per_s": R * S / (ms * 1e-3), "mfma_per_fragment": t * t}
    # error against the oracle on the first 256 rays
    n = min(R, 256)
    th = theta.cpu()
    vv = lay.views(th)
    mlp, off = {}, 0
    for name, o, i in cnr_amd.ops.TRUNK_LAYERS:
        mlp[name + ".weight"] = vv["trunk"][:, off:off + o * i].reshape(1, o, i); off += o * i
        mlp[name + ".bias"] = vv["trunk"][:, off:off + o]; off += o
    for k, name in enumerate(cnr_amd.ops.LATENT_LAYERS):
        mlp[name + ".weight"], mlp[name + ".bias"] = vv["latW"][:, k], vv["latb"][:, k]
    e = O.unidirs_embed(pts[:, :n].cpu(), vv["B"], 2.0)
    cs, ct = vv["shape"][0][idx[0, :n]][None, :, None], vv["tex"][0][idx[0, :n]][None, :, None]
    sig_ref, rgb_ref = O.codenerf_forward(mlp, e, cs, ct)
    occ_ref = torch.sigmoid(sig_ref.squeeze(-1))
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for k, (sg, rg) in outs.items():
        key = "f16" if k == "f16" else f"fp8_{k[-1]}_plane"
        legs[key]["occupancy_rel_l2_vs_oracle"] = rel(torch.sigmoid(sg[:, :n].cpu()), occ_ref)
        legs[key]["rgb_rel_l2_vs_oracle"] = rel(rg[:, :n].cpu(), rgb_ref)
    one = legs["fp8_1_plane"]
    print(json.dumps({"metric": "samples/sec (field forward only, fp8 MFMA) -- BASELINE.json configs[4]; NOT the train-step metric",
                      "value": one["samples_per_s"], "unit": "samples/s", "n_gpus": 1, "higher_is_better": True,
                      "dtype": "fp8", "data": "synthetic", "vs_baseline": None,
                      "config": {"workload": f"1 category x {n_obj} objects, {R} rays x {S} samples, latent {L}: PE + CodeNeRF forward, "
                                             "OCP e4m3 weights and activations on v_mfma_f32_32x32x16_fp8_fp8"},
                      "legs": legs,
                      "note": "no fp8 train step: one fp8 plane misses the 1e-3 parity bar by ~60x; three planes meet it at nine MFMAs per "
                              "fragment where the f16 kernel needs one of the same rate"}))


def dropin_vmap_flow(cnr_amd, dev, C, R, n1, n2, L, n_obj, steps=30, warmup=5):
    """ms per iteration of INTEGRATION.md section 1's flow: the reference's own loop (train.py:98-201) on the drop-in modules --
    ``get_training_samples`` per category, ``nn.Embedding`` code lookups, ``vmap(pe_model)`` / ``vmap(fc_model)`` over the
    ``update_vmap`` ensembles, ``loss.step_batch_loss`` + regulariser, ``backward``, ``torch.optim.AdamW.step``, the per-step
    copy-back.  This is the exact-fp32 modular tier (one kernel per reference function), NOT the benchmarked fused step."""
    from torch.func import vmap
    cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0 if L == 256 else 3.0,
                                       n_bins_cam2surface=n1, n_bins=n2)
    cls_dict = {}
    opt = torch.optim.AdamW([torch.zeros(1, device=dev, requires_grad=True)], lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    for c in range(C):
        pool = cnr_amd.scene_cateogries.synthetic_pool(16 * R, n_obj, torch.Generator().manual_seed(4321 + c), "cpu")
        sc = cnr_amd.scene_cateogries.sceneCategory.from_pool(cfg, c + 1, list(range(n_obj)), pool, seed=c)
        cls_dict[c + 1] = sc
        for params, lr, wd in ((sc.trainer.fc_occ_map.parameters(), cfg.learning_rate, cfg.weight_decay),
                               (sc.trainer.pe.parameters(), cfg.learning_rate, cfg.weight_decay),
                               (sc.trainer.shape_codes.parameters(), cfg.code_learning_rate, cfg.code_weight_decay),
                               (sc.trainer.texture_codes.parameters(), cfg.code_learning_rate, cfg.code_weight_decay)):
            opt.add_param_group({"params": params, "lr": lr, "weight_decay": wd})          # train.py:58-64
    fc_model, fc_param, fc_buffer = cnr_amd.utils.update_vmap([k.trainer.fc_occ_map for k in cls_dict.values()], opt)
    pe_model, pe_param, pe_buffer = cnr_amd.utils.update_vmap([k.trainer.pe for k in cls_dict.values()], opt)
    cls_ids = torch.arange(C, device=dev)

    def iteration():
        acc = [[] for _ in range(9)]
        for cls_k in cls_dict.values():
            gt_rgb, gt_depth, depth_mask, obj_mask, pcs, z, indices = cls_k.get_training_samples(R)
            for lst, v in zip(acc, (gt_depth, gt_rgb, depth_mask, obj_mask, pcs, z, indices,
                                    cls_k.trainer.shape_codes(indices)[:, None, :], cls_k.trainer.texture_codes(indices)[:, None, :])):
                lst.append(v)
        gt_depth, gt_rgb, depth_mask, obj_mask, pcs, z, _, cs, ct = [torch.stack(a) for a in acc]
        emb = vmap(pe_model)(pe_param, pe_buffer, pcs)
        alpha, color = vmap(fc_model)(fc_param, fc_buffer, emb, cs, ct)
        loss, _, _ = cnr_amd.loss.step_batch_loss(alpha, color, gt_depth.detach(), (gt_rgb / 255.).detach(), obj_mask.detach(),
                                                  depth_mask.detach(), z.detach())
        rs, rt = cnr_amd.loss.step_batch_loss_reg(cls_dict, cls_ids)
        loss = loss + 0.0005 * (rs + rt).sum()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():                                                               # train.py:196-201
            for m, cls_k in enumerate(cls_dict.values()):
                for i, prm in enumerate(cls_k.trainer.fc_occ_map.parameters()):
                    prm.copy_(fc_param[i][m])
                for i, prm in enumerate(cls_k.trainer.pe.parameters()):
                    prm.copy_(pe_param[i][m])
    for _ in range(warmup):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": f"{C} categor{'y' if C == 1 else 'ies'} x {n_obj} objects, {R} rays x {n1 + n2} samples: the reference's loop "
                        "(train.py:98-201: get_training_samples, vmap(pe), vmap(fc), step_batch_loss, backward, torch AdamW, copy-back) "
                        "on the drop-in modules = the exact-fp32 modular tier, eager",
            "steps": steps, "ms_per_step": dt * 1e3, "rays_per_s": C * R / dt}, cls_dict, cfg


def dropin_from_scene(cnr_amd, cls_dict, cfg, R, steps=2000):
    """... and the SAME sceneCategory objects handed to ``FullStepTrainer.from_scene``: the fused step behind the reference's
    objects (their pools, their modules' parameters), ``sync_to_modules()`` instead of the per-step copy-back."""
    full = cnr_amd.background.FullStepTrainer.from_scene(cls_dict, None, cfg, rays_per_step=R, seed=0)
    full.run(8)
    full.obj.prepare_graphs()
    full.run(2 * full.obj.unroll + 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    full.run(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t1 = time.perf_counter()
    full.sync_to_modules()
    torch.cuda.synchronize()
    return {"workload": "FullStepTrainer.from_scene(cls_dict, None, cfg) on the sceneCategory objects of dropin_vmap_flow: the fused "
                        "f16 step (3 launches, hipGraph) on their pools and parameters", "steps": steps, "ms_per_step": dt * 1e3,
            "rays_per_s": len(cls_dict) * R / dt, "sync_to_modules_ms": (time.perf_counter() - t1) * 1e3}


def strong_leg(cnr_amd, cfg, dev, pg, rank, world, n_cls, R, n_obj, steps, sync, timed_run):
    """north_star's own target, readable from ONE driver line: BASELINE.json configs[2]'s shape -- `n_cls` categories in total,
    whole categories per GPU, no gradient collective -- on all `world` ranks, and the same `n_cls` categories on rank 0's GPU alone,
    measured in the same run.  Both are exactly `steps` steps between barriers; the N-rank time is the max over ranks."""
    ids = cnr_amd.parallel.class_shard(n_cls, rank, world)
    mk = lambda cids, group, shard: cnr_amd.fused.FusedCategoryTrainer(
        cfg, len(cids), n_obj, [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, torch.Generator().manual_seed(1234 + 17 * (c + 1)), "cpu")
                                for c in cids], R, dev, seed=0, generator=torch.Generator().manual_seed(1234), process_group=group,
        shard=shard, n_cls_global=n_cls if shard else None, class_ids=cids if shard else None)

    def measure(tr):
        tr.run(4)
        tr.prepare_graphs()
        tr.run(2 * tr.unroll + 3)
        tr.run(max(steps // 2, 8))
        return timed_run(tr, steps)
    trN = mk(ids, pg, "class")
    dtN = measure(trN)
    del trN
    torch.cuda.empty_cache()
    dt1 = None
    if rank == 0:                      # the single-GPU reference point: all categories on this rank's GPU, no process group
        tr1 = mk(list(range(n_cls)), None, None)
        tr1.run(4)
        tr1.prepare_graphs()
        tr1.run(2 * tr1.unroll + 3)
        tr1.run(max(steps // 2, 8))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr1.run(steps)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        del tr1
    sync()
    rays = n_cls * R * steps
    return {"workload": f"BASELINE.json configs[2] shape: {n_cls} categories x {n_obj} objects x {R} rays in total, whole categories per GPU, "
                        f"no gradient collective; fixed global batch (strong scaling)", "classes_total": n_cls, "steps": steps,
            "n_rank_ms_per_step": dtN / steps * 1e3, "n_rank_rays_per_s": rays / dtN,
            "one_rank_ms_per_step": None if dt1 is None else dt1 / steps * 1e3,
            "one_rank_rays_per_s": None if dt1 is None else rays / dt1,
            "speedup_vs_one_rank_same_run": None if dt1 is None else dt1 / dtN}


def launch_ranks(n, argv, timeout_s=900.0):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script as a FRESH child process tree
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), pass rank 0's JSON
    line through and exit with the children's code.  The parent has made no HIP call (it never replaces itself either: the
    children are ordinary subprocesses), so an N-rank line or a failure are the only outcomes -- never a 1-rank line
    that says N."""
    import subprocess
    if not os.environ.get("CNR_SINGLE_DEVICE_REHEARSAL") and os.environ.get("CNR_DIST_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()                 # counts devices without initialising one
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s)")
    # The rendezvous port is picked by torch.distributed.run itself (--standalone: a c10d store on a free port, bound by the
    # agent that uses it), not by a bind()/close() here that another process could win the race for.
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    # A child tree of its own (new session = new process group) with a deadline: RCCL bring-up or a collective that never
    # returns must end in a message and a non-zero exit, not in a bench that blocks its driver forever.
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        import signal
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)        # exactly the group this call created
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        raise SystemExit(f"bench.py --gpus {n}: the {n}-rank launch did not finish within {timeout_s:.0f} s "
                         "(--launch-timeout); its process group was killed")
    line = None
    for ln in stdout.splitlines():
        if ln.startswith("{") and '"n_gpus"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        raise SystemExit(f"bench.py --gpus {n}: the {n}-rank launch failed (exit code {proc.returncode})")
    if line is None or json.loads(line).get("n_gpus") != n:
        raise SystemExit(f"bench.py --gpus {n}: the ranks produced no {n}-GPU result line")
    print(line)
    return 0


def launch_check(args, rank, local_rank, world, backend):
    """--launch-check: the process-group side of the bench alone.  Every rank joins the group (RCCL when the backend is
    nccl, each rank on its own GPU), a one is all-reduced, and rank 0 prints what the BACKEND says the world is."""
    got, name = 1, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            dev = torch.device("cpu")
            torch.distributed.init_process_group(backend)
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        got, name = int(one.item()), torch.distributed.get_backend()
        assert got == torch.distributed.get_world_size()
        torch.distributed.barrier()
    if got != args.gpus:
        raise SystemExit(f"launch check: {got} ranks answered, --gpus says {args.gpus}")
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": got, "backend": name, "value": None,
                          "metric": "launcher check only -- no kernel ran, not a benchmark result"}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--rays", type=int, default=2048, help="rays per class and step (per GPU with --scaling weak)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--classes", type=int, default=1, help="classes per GPU (weak) or in total (strong)")
    ap.add_argument("--latent", type=int, default=256)
    ap.add_argument("--shard", choices=("class", "ray"), default="class",
                    help="N > 1: 'class' = every GPU owns whole categories, no gradient collective (north_star's "
                         "per-category sharding); 'ray' = the rays of every category are split, one all-reduce per step")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--dtype", choices=("f16", "fp8"), default="f16",
                    help="fp8: BASELINE.json configs[4] -- forward-only throughput and error of the fp8-MFMA field forward "
                         "(there is no fp8 train step: see DESIGN.md section 3.4); prints its own JSON line")
    ap.add_argument("--bwd-blocks", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--unroll", type=int, default=0,
                    help="train steps per hipGraph launch (2 = one graph launch per two steps); 0 = --steps rounded to even, within "
                         "[16, 32]: a 20-step region then goes out as ONE launch (between two launches the GPU idles ~8 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--long-seconds", type=float, default=2.0, help="length of the long_run leg reported beside the timed region")
    ap.add_argument("--strong-classes", type=int, default=16,
                    help="N > 1: categories in total of the strong_configs2 leg (BASELINE.json configs[2]'s stand-in count); 0 = skip")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="seconds the self-started N-rank child tree may take before its process group is killed")
    ap.add_argument("--pre-warm-seconds", type=float, default=0.25,
                    help="seconds of untimed steps in FRONT of the W warm-up steps (reported as pre_warm_steps): the graph captures "
                         "before them are ~50 ms of host work with the GPU idle, and a 20-step region right behind that ran 3 us "
                         "per step slower than the same region in a running job; 0 = none")
    ap.add_argument("--host-wait", choices=("spin", "default"), default="spin",
                    help="how torch.cuda.synchronize() waits: spin = hipDeviceScheduleSpin (host_wait_spin), default = the runtime's")
    ap.add_argument("--launch-check", action="store_true",
                    help="bring the ranks up, all-reduce a one over the process group and print the world size the "
                         "backend reports -- no kernel runs (a check of the launcher, not a benchmark)")
    args = ap.parse_args()

    if args.dtype == "fp8":
        return fp8_leg(args)
    if args.unroll <= 0:
        args.unroll = min(32, max(16, args.steps // 2 * 2))
    # ---- N ranks: --gpus N is what decides; the launcher's environment must agree ---------------------------------
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                      # not under a launcher yet: become one (nothing has touched the GPU)
            return launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={os.environ['WORLD_SIZE']}: "
                         "the two must agree (the line would claim a GPU count that did not run)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("CNR_DIST_BACKEND", "nccl") if world > 1 else None   # nccl = RCCL over xGMI; gloo only for rehearsals
    if args.launch_check:
        return launch_check(args, rank, local_rank, world, backend)
    if os.environ.get("CNR_SINGLE_DEVICE_REHEARSAL"):           # N ranks on one card (gloo), 1-GPU boxes only
        local_rank = 0
    host_wait = "spin" if args.host_wait == "spin" and host_wait_spin() else "default"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD
        if torch.distributed.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {torch.distributed.get_world_size()} ranks, --gpus says {args.gpus}")

    dbg = (lambda *a: print(f"[bench rank {rank}]", *a, file=sys.stderr, flush=True)) if os.environ.get("CNR_BENCH_DEBUG") \
        else (lambda *a: None)
    dbg("process group up")
    import cnr_amd
    info = cnr_amd._C.device_info()
    S, L, n_obj = args.samples, args.latent, 4
    n1, n2 = S // 8, S - S // 8
    cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0 if L == 256 else 3.0,
                                       n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(1234)            # same parameters on every rank
    # ---- what this rank trains -----------------------------------------------------------------------------------
    shard = args.shard if world > 1 else None
    if shard == "class":
        # whole categories per GPU: classes share nothing (train.py:58-64), so NO gradient collective; the
        # any-class-empty flags cross ranks once per epoch.  weak: --classes per GPU; strong: --classes in total
        C_glob = args.classes * world if args.scaling == "weak" else args.classes
        if C_glob < world:
            raise SystemExit(f"--scaling strong with --shard class needs --classes >= {world}")
        ids = cnr_amd.parallel.class_shard(C_glob, rank, world)
        C, R, Rg = len(ids), args.rays, args.rays
    else:
        C_glob = args.classes
        ids, C = list(range(C_glob)), C_glob
        Rg = args.rays * world if args.scaling == "weak" else args.rays          # global rays per class and step
        R = Rg // world
        if R * world != Rg:
            raise SystemExit("--rays must divide by the number of GPUs with --shard ray --scaling strong")
    # pools: one per class, seeded by the GLOBAL class id -> ray shards hold identical copies, class shards their own
    pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * Rg, n_obj, torch.Generator().manual_seed(1234 + 17 * (c + 1)), "cpu")
             for c in ids]
    tr = cnr_amd.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=0, generator=gen,
                                            bwd_blocks=args.bwd_blocks, process_group=pg, use_graph=not args.no_graph,
                                            shard=shard, n_cls_global=C_glob, class_ids=ids if shard == "class" else None,
                                            unroll=args.unroll)
    rays_per_step_global = C_glob * Rg

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def timed_run(trainer, n_steps):
        sync()
        t0 = time.perf_counter()
        trainer.run(n_steps)   # = n_steps x step(); groups of trainer.unroll steps go out as one hipGraph launch each
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    timed = lambda n_steps: timed_run(tr, n_steps)

    dbg("trainer built")
    tr.run(4)                   # (the step's buffers exist after two steps)
    tr.prepare_graphs()         # every graph the timed region can need exists before it starts (captures are host work)
    for _ in range(2):          # ... and has been launched once, from either state parity (the first launch of a graph uploads it)
        for u in tr._group_sizes(tr.unroll):     # 16, 8, 4, 2: what run() sends out at an epoch's end
            tr.run(u)
        tr.run(1)
    # the W warm-up steps come LAST, directly in front of the timed region: the captures above are ~50 ms of host work with the GPU
    # idle, and the first region after them ran 2 us per step slower than every later one (tools/exp/region200.py)
    pre_warm_steps = 0
    if args.pre_warm_seconds > 0:      # the card back at its running-job clocks (the long_run leg below measures that state too)
        # a step COUNT, the same on every rank (a class-sharded epoch end is a collective: ranks must not run different numbers of steps)
        torch.cuda.synchronize()
        t_pw = time.perf_counter()
        tr.run(tr.unroll)
        torch.cuda.synchronize()
        est = (time.perf_counter() - t_pw) / tr.unroll
        n_pw = max(int(args.pre_warm_seconds / max(est, 1e-6)) // tr.unroll, 1) * tr.unroll
        if world > 1:
            tp = torch.tensor([n_pw], device=dev, dtype=torch.int64)
            torch.distributed.all_reduce(tp, op=torch.distributed.ReduceOp.MAX)
            n_pw = int(tp.item())
        for _ in range(n_pw // tr.unroll):     # one group at a time: 200 graph launches queued ahead of the GPU left the NEXT region's
            tr.run(tr.unroll)                  # launch slow (67-77 us per step over the 20 steps behind them, measured)
            torch.cuda.synchronize()
        pre_warm_steps = n_pw + tr.unroll
    tr.run(max(args.warmup, 4))
    dbg("warmup issued")
    dt = timed(args.steps)                      # EXACTLY --steps steps between barriers + synchronize, max over ranks
    dbg("timed region done")
    ms_per_step = dt / args.steps * 1e3
    rays_per_s = rays_per_step_global * args.steps / dt
    # the same measurement over >= 2 s of steps (a 20-step region at 0.06 ms per step is 1.2 ms long -- shorter than any utilisation
    # sampler's period): reported beside it, never instead of it
    n_long = max(args.steps, int(args.long_seconds / max(dt / args.steps, 1e-6)) + 1)
    if world > 1:
        tl = torch.tensor([n_long], device=dev, dtype=torch.int64)
        torch.distributed.all_reduce(tl, op=torch.distributed.ReduceOp.MAX)
        n_long = int(tl.item())
    dt_long = timed(n_long)
    long_run = {"steps": n_long, "seconds": dt_long, "ms_per_step": dt_long / n_long * 1e3,
                "value": rays_per_step_global * n_long / dt_long}

    # ---- roofline leg: the dominant kernel (fused backward) timed with HIP events on its launch stream ----
    bwd_name = "cnr_field_bwd_pipe"
    names = [bwd_name, "cnr_field_train", "cnr_field_fwd", "cnr_field_fwd_render", "cnr_step_prologue", "cnr_render_loss",
             "cnr_step_tail", "cnr_step_grad"]
    tr.use_graph = False                                           # eager so that events bracket single launches
    cnr_amd._C.enable_kernel_timing(names)
    for _ in range(min(args.steps, 50)):
        tr.step()
    tms = cnr_amd._C.kernel_timings_ms()
    tr.use_graph = not args.no_graph
    avg = {k: (sum(v) / len(v) if v else 0.0) for k, v in tms.items()}
    # the dominant call again, back to back (no host gaps between launches): this is the duration rocprofv3 reports
    # for its kernels (field kernel + reduce_records), and the one the roofline entry uses
    if tr._ft_blocks:        # the one-launch step body is the dominant call; the two-call form is not on this path
        bwd_name = "cnr_field_train"
        eager_bwd_ms = avg[bwd_name]
        avg[bwd_name] = tr.time_field_train(250)
    else:
        eager_bwd_ms = avg[bwd_name]
        avg[bwd_name] = tr.time_field_bwd(50)
    dom = max((bwd_name, "cnr_field_fwd", "cnr_field_fwd_render"), key=lambda k: avg[k])
    # the backward call = the field kernel(s) (pipe: one launch; split: texture + geometry launches) + reduce_records:
    # its duration is the sum of those (rocprof lists them separately, profiles/).
    # algorithmic FLOP of that call: fwd = 27 422 / sample; backward (recomputes the forward) or the one-launch forward +
    # backward = fwd + dX + dW = 82 140
    flop_per_sample = 82140 if dom == bwd_name else 27422
    achieved = C * R * S * flop_per_sample / (avg[dom] * 1e-3) / 1e12 if avg[dom] > 0 else 0.0
    # HBM bytes of that call from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes,
    # gfx950 corrections per MI355X_MICROARCH.md): measured offline on this shape, kept in profiles/
    traffic = None
    for tname in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if dom == bwd_name and (C, R, S) == (1, 2048, 64) and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(bwd_name + "_call_hbm_bytes")
            break
    roofline = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_MFMA_F16_TFLOPS, "traffic": traffic,
                "variant": "pipe8 (4 chain + 4 weight-gradient waves per workgroup, two per SIMD)",
                "kernel_ms": {k: round(v, 5) for k, v in avg.items()},
                "kernel_ms_note": "HIP events on the launch stream; %s back to back, median of five batches of 50 launches (one launch at a time in the "
                                  "eager step, gaps included: %.5f), the others one launch at a time" % (bwd_name, eager_bwd_ms),
                "step_tflops": rays_per_step_global * S * FLOP_PER_SAMPLE_STEP / (dt / args.steps) / 1e12}

    par = "1 GPU" if world == 1 else (
        f"{world} GPUs, whole categories per GPU ({C_glob} in total), no gradient collective, empty-mask flags all-reduced once per epoch"
        if shard == "class" else
        f"{world} GPUs, rays of every category split ({Rg} per category and step in total), one all-reduce of the flat gradient per step")
    out = {"metric": "rays/sec (train step) Replica room_0, 2048 rays x 64 samples, 1/2/4/8 GPU",
           "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
           "vs_baseline": None, "dtype": "f16", "data": "synthetic", "host_wait": host_wait, "pre_warm_steps": pre_warm_steps,
           "config": {"workload": f"Replica room_0 shape: {C_glob} categor{'y' if C_glob == 1 else 'ies'} x {n_obj} objects, "
                                  f"{Rg} rays x {S} samples per category and step, latent {L}, W=32 CodeNeRF, random-pose "
                                  f"synthetic pool, random init",
                      "rays_per_step_global": rays_per_step_global, "rays_per_gpu": C * R, "samples_per_ray": S,
                      "parallelism": par, "shard": shard,
                      "world_size": torch.distributed.get_world_size() if world > 1 else 1,
                      "dist_backend": torch.distributed.get_backend() if world > 1 else None,
                      "operand_dtype": "f16 MFMA operands, fp32 accumulate, where BASELINE.json configs[1] says bf16: bf16 operands "
                                       "miss north_star's 1e-3 parity bar (3.9e-3 / 5.1e-3, BASELINE.md section 4), f16 has the "
                                       "same MFMA rate on gfx950 and three more mantissa bits",
                      "hipgraph": (f"{tr.unroll} steps per graph launch (halving group sizes down to single steps at epoch ends)" if not tr.grad_exchange else "two graphs around the all-reduce, per state parity") if not args.no_graph else False,
                      "n_cu": info["n_cu"]},
           "long_run": long_run, "roofline": roofline}
    if world > 1:
        ranks = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(ranks)
        out["world_size"], out["dist_backend"], out["ranks_answering"] = world, torch.distributed.get_backend(), int(ranks.item())
        if args.strong_classes >= world and shard == "class":
            try:
                out["strong_configs2"] = strong_leg(cnr_amd, cfg, dev, pg, rank, world, args.strong_classes, args.rays, n_obj,
                                                    max(args.steps, 20), sync, timed_run)
            except Exception as e:          # never lose the main line over the extra leg
                out["strong_configs2"] = f"failed: {type(e).__name__}: {e}"
    if world > 1 and shard == "ray":   # outside the timed region: every rank must hold bitwise identical parameters after the run
        try:
            out["config"]["params_in_sync"] = bool(cnr_amd.parallel.params_in_sync(tr.theta.contiguous(), pg))
        except Exception as e:      # never lose the bench line over the check
            out["config"]["params_in_sync"] = f"check failed: {e}"
    if world == 1 and not args.no_extra_legs:
        # the reference's REAL batch shape (configs/Replica/config_replica_room0.json:25-28: 120 rays per object, 1 + 9
        # samples): rays padded to 16 sample slots, two per tile, on the same one-launch step body
        try:
            cfg2 = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=1, n_bins=9)
            R2 = 120 * n_obj
            pools2 = [cnr_amd.scene_cateogries.synthetic_pool(64 * R2, n_obj, torch.Generator().manual_seed(99), "cpu")]
            tr2 = cnr_amd.fused.FusedCategoryTrainer(cfg2, 1, n_obj, pools2, R2, dev, seed=1,
                                                     generator=torch.Generator().manual_seed(5))
            tr2.run(10)
            tr2.prepare_graphs()
            tr2.run(2 * tr2.unroll + 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr2.run(2000)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t0
            out["extra_legs"] = {"real_config_480x10": {"workload": "1 category x 4 objects, 120 rays per object x 10 samples "
                                                                    "(the reference's own batch shape), two 16-slot rays per 32-sample tile",
                                                        "steps": 2000, "ms_per_step": d2 / 2000 * 1e3, "rays_per_s": R2 * 2000 / d2}}
        except Exception as e:
            out["extra_legs"] = {"real_config_480x10": f"failed: {e}"}
        # the WHOLE iteration of train.py:113-184: background (1200 rays x 14 samples, OccupancyMap(128), the fused f16 step) + the
        # benchmarked category step, captured as one hipGraph per state parity
        try:
            cfg3 = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2)
            cfg3.n_bins_cam2surface_bg = 5
            g3 = torch.Generator().manual_seed(77)
            pools3 = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, g3, "cpu")]
            tr3 = cnr_amd.fused.FusedCategoryTrainer(cfg3, 1, n_obj, pools3, R, dev, seed=2, generator=g3)
            cfg_bg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=9)
            bg = cnr_amd.background.BackgroundStep(cfg_bg, cnr_amd.scene_cateogries.synthetic_pool(64 * 1200, 1, g3, "cpu"), 1200, dev,
                                                   precision="fused")
            full = cnr_amd.background.FullStepTrainer(tr3, bg)
            for _ in range(10):
                full.step()
            full.run(64)         # captures and first launches of the multi-iteration graphs, outside the timed region
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            full.run(500)        # = 500 x full.step(); groups of up to eight iterations per hipGraph launch
            torch.cuda.synchronize()
            d3 = time.perf_counter() - t0
            out["extra_legs"]["full_iteration_bg_plus_category"] = {
                "workload": f"background 1200 rays x 14 samples (OccupancyMap(128), fused f16 step: csrc/bg_fused.hip) + 1 category x {R} "
                            f"rays x {S} samples, the two chains of up to eight iterations as two free-running branches of one hipGraph (joined once per "
                            f"graph launch), the background step in four launches", "steps": 500, "ms_per_step": d3 / 500 * 1e3,
                "category_rays_per_s": R * 500 / d3, "all_rays_per_s": (R + 1200) * 500 / d3}
        except Exception as e:
            out["extra_legs"]["full_iteration_bg_plus_category"] = f"failed: {e}"
    if world == 1 and not args.no_extra_legs:
        # the two tiers behind the reference's call surface, side by side (VERDICT r03 row h): train.py's own loop on the drop-in
        # modules, and the fused step built from the same sceneCategory objects
        try:
            leg, cls_dict_d, cfg_d = dropin_vmap_flow(cnr_amd, dev, C, R, n1, n2, L, n_obj)
            out["extra_legs"]["dropin_vmap_flow"] = leg
            out["extra_legs"]["dropin_from_scene"] = dropin_from_scene(cnr_amd, cls_dict_d, cfg_d, R)
            del cls_dict_d
        except Exception as e:
            out["extra_legs"]["dropin_vmap_flow"] = f"failed: {type(e).__name__}: {e}"
        # the ray-sharded step's structure on ONE GPU, no collective: two graphs + the gap where the all-reduce goes, three host
        # calls per step -- what that form costs before any communication
        try:
            # (dp_world = 2 without a process group: rank 0 of a two-rank ray-sharded world whose gradient exchange is left out --
            #  front graph = prologue + field_train + step_grad, back graph = tail without the latent blocks, as with RCCL)
            trs = cnr_amd.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=0, generator=torch.Generator().manual_seed(1234),
                                                     shard="ray", dp_world=2, dp_rank=0)
            trs.run(50)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trs.run(1000)
            torch.cuda.synchronize()
            d4 = (time.perf_counter() - t0) / 1000
            out["extra_legs"]["split_graph_step_1gpu"] = {
                "workload": "rank 0 of a 2-rank ray-sharded step with the all-reduce left out: two hipGraphs per step (front: prologue + "
                            "field_train + step_grad; back: tail), three host calls per step, same rays per rank as the main line", "steps": 1000, "ms_per_step": d4 * 1e3,
                "rays_per_s": C * R / d4, "host_cost_vs_one_graph_ms": d4 * 1e3 - long_run["ms_per_step"]}
            del trs
        except Exception as e:
            out["extra_legs"]["split_graph_step_1gpu"] = f"failed: {type(e).__name__}: {e}"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline((C, R, n1, n2, L, n_obj), args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
