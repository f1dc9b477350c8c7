#!/usr/bin/env python3
"""Benchmark of the volumetric-rendering train step (BASELINE.json metric: rays/sec, train step).

One step = sample (a2-a6) -> latent rows (a7) -> PE + CodeNeRF trunk (a8, a9) -> composite (a11-a13) -> loss
(a14-a16) -> backward of all of it -> AdamW (a18), on synthetic random-pose ray pools resident in HBM
(SURVEY.md section 8(d)).  N = 1 runs BASELINE.json configs[1] (Replica room_0 shape: 1 category, 2048 rays x 64
samples, L = 256); N > 1 is launched by torch.distributed.run, every rank runs the same per-GPU workload on its
own rays of the same category and the flat gradient is all-reduced over RCCL (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_STEP = 82140          # 3 x 13 648 MAC + 2 x 63 MAC, 2 FLOP/MAC
PEAK_MFMA_F16_TFLOPS = 2500.0         # MI355X dense bf16/f16 MFMA peak (MI355X_MICROARCH.md, chip-level table)


def cpu_baseline(cfg_shape, seconds=12.0):
    """The oracle's train step (reference op structure, fp32) on the host cores: rays/s on the same shape."""
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

    def step(i):
        sl = slice((i % 3) * R, (i % 3) * R + R)
        rgbs, depth, dirs, T, idx = pool["rgbs"][sl], pool["depth"][sl], pool["dirs"][sl], pool["T_co"][sl], pool["indices"][sl]
        o, d = O.origin_dirs_O(T, dirs)
        u = torch.rand(R, S, generator=gen)
        g = torch.randn(R, n2, generator=gen) * (0.1 / 3)
        gt_rgb, gt_depth, mask, labels, pts, z = O.sample_3d_points(rgbs, depth, o, d, u, g, n1, n2, 0.1, 0.05)
        batch = dict(pts=pts[None].repeat(C, 1, 1, 1), z=z[None].repeat(C, 1, 1), gt_depth=gt_depth[None].repeat(C, 1),
                     gt_rgb=(gt_rgb / 255.0)[None].repeat(C, 1, 1), labels=labels[None].repeat(C, 1),
                     depth_mask=mask[None].repeat(C, 1), indices=idx[None].repeat(C, 1))
        loss, _ = O.forward_loss(mlp, B, 2.0, sh, tx, batch)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    # pick the thread count that is fastest for THIS workload (a 128-thread box is slower with all threads on
    # these small GEMMs than with 16-32): the baseline is the CPU's best, not its default
    best = None
    for nt in sorted({8, 16, 32, 64, torch.get_num_threads()}):
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        step(0)
        t0 = time.perf_counter()
        step(1)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    times, t_end, i = [], time.perf_counter() + seconds, 0
    while time.perf_counter() < t_end or len(times) < 3:
        t0 = time.perf_counter()
        step(i)
        times.append(time.perf_counter() - t0)
        i += 1
    times.sort()
    med = times[len(times) // 2]
    return {"value": C * R / med, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} steps of the same {C}x{R}x{S} (L={L}) train step, median {med * 1e3:.1f} ms, "
                      f"torch fp32 oracle with {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--classes", type=int, default=1)
    ap.add_argument("--latent", type=int, default=256)
    ap.add_argument("--bwd-blocks", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("CNR_SINGLE_DEVICE_REHEARSAL"):           # N ranks on one card (gloo), 1-GPU boxes only
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CNR_DIST_BACKEND", "nccl")     # nccl = RCCL over xGMI; gloo only for rehearsals
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    dbg = (lambda *a: print(f"[bench rank {rank}]", *a, file=sys.stderr, flush=True)) if os.environ.get("CNR_BENCH_DEBUG") \
        else (lambda *a: None)
    dbg("process group up")
    import cnr_amd
    info = cnr_amd._C.device_info()
    C, R, S, L, n_obj = args.classes, args.rays, args.samples, args.latent, 4
    n1, n2 = S // 8, S - S // 8
    cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0 if L == 256 else 3.0,
                                       n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(1234)            # same weights on every rank
    pgen = torch.Generator().manual_seed(1234 + 17 * (rank + 1))  # rank-private rays
    pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, pgen, "cpu") for _ in range(C)]
    tr = cnr_amd.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=rank, generator=gen,
                                            bwd_blocks=args.bwd_blocks, process_group=pg, use_graph=not args.no_graph)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    dbg("trainer built")
    for _ in range(max(args.warmup, 4)):
        tr.step()
    dbg("warmup issued")
    sync()
    dbg("warmup done")
    t0 = time.perf_counter()
    for i_step in range(args.steps):
        tr.step()
        if i_step % 8 == 0:
            dbg("timed step", i_step)
    dbg("timed loop issued")
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    dbg("timed region done")
    ms_per_step = dt / args.steps * 1e3
    rays_per_s = world * C * R * args.steps / dt

    # ---- roofline leg: the dominant kernel (fused backward) timed with HIP events on its launch stream ----
    bwd_name = "cnr_field_bwd" if cnr_amd.ops.FIELD_BWD_VARIANT == "split" else "cnr_field_bwd_pipe"
    names = [bwd_name, "cnr_field_fwd", "cnr_field_fwd_render", "cnr_step_prologue", "cnr_render_loss", "cnr_step_tail"]
    tr.use_graph = False                                           # eager so that events bracket single launches
    cnr_amd._C.enable_kernel_timing(names)
    for _ in range(min(args.steps, 50)):
        tr.step()
    tms = cnr_amd._C.kernel_timings_ms()
    tr.use_graph = not args.no_graph
    avg = {k: (sum(v) / len(v) if v else 0.0) for k, v in tms.items()}
    # the dominant call again, back to back (no host gaps between launches): this is the duration rocprofv3 reports
    # for its kernels (field kernel + reduce_records), and the one the roofline entry uses
    eager_bwd_ms = avg[bwd_name]
    avg[bwd_name] = tr.time_field_bwd(50)
    dom = max((bwd_name, "cnr_field_fwd", "cnr_field_fwd_render"), key=lambda k: avg[k])
    # the backward call = the field kernel(s) (pipe: one launch; split: texture + geometry launches) + reduce_records:
    # its duration is the sum of those (rocprof lists them separately, profiles/).
    # algorithmic FLOP of that call: fwd = 27 422 / sample, bwd (recompute fwd + dX + dW) = 82 140
    flop_per_sample = 82140 if dom == bwd_name else 27422
    achieved = C * R * S * flop_per_sample / (avg[dom] * 1e-3) / 1e12 if avg[dom] > 0 else 0.0
    # HBM bytes of that call from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes,
    # gfx950 corrections per MI355X_MICROARCH.md): measured offline on this shape, kept in profiles/
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if dom == bwd_name and (C, R, S) == (1, 2048, 64) and os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(bwd_name + "_call_hbm_bytes")
    roofline = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_MFMA_F16_TFLOPS, "traffic": traffic,
                "variant": cnr_amd.ops.FIELD_BWD_VARIANT,
                "kernel_ms": {k: round(v, 5) for k, v in avg.items()},
                "kernel_ms_note": "HIP events on the launch stream; %s back to back x50 (one launch at a time in the "
                                  "eager step, gaps included: %.5f), the others one launch at a time" % (bwd_name, eager_bwd_ms),
                "step_tflops": world * C * R * S * FLOP_PER_SAMPLE_STEP / (dt / args.steps) / 1e12}

    out = {"metric": "rays/sec (train step) Replica room_0, 2048 rays x 64 samples, 1/2/4/8 GPU",
           "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16", "data": "synthetic",
           "config": {"workload": f"Replica room_0 shape: {C} category x {n_obj} objects, {R} rays x {S} samples "
                                  f"per GPU and step, latent {L}, W=32 CodeNeRF, random-pose synthetic pool, random init",
                      "rays_per_gpu": C * R, "samples_per_ray": S, "parallelism": f"dp{world}",
                      "hipgraph": ("one graph per state parity" if world == 1 else "two graphs around the all-reduce, per state parity") if not args.no_graph else False, "n_cu": info["n_cu"]},
           "roofline": roofline}
    if world > 1:   # outside the timed region: every rank must hold bitwise identical parameters after the run
        try:
            out["config"]["params_in_sync"] = bool(cnr_amd.parallel.params_in_sync(tr.theta.contiguous(), pg))
        except Exception as e:      # never lose the bench line over the check
            out["config"]["params_in_sync"] = f"check failed: {e}"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline((C, R, n1, n2, L, n_obj), args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
