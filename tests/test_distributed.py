"""CPU, world_size 2, gloo: the N > 1 path -- rank-private ray shards, ONE all-reduce of the flat gradient, identical
AdamW on every rank (parameters stay bitwise equal).  Gradients come from the oracle here (no GPU); on the GPU box the
same parallel.allreduce_mean_ call sits in FusedCategoryTrainer._step_body over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import cnr_amd as cnr
    from oracle import ref_cpu as O
    torch.set_num_threads(2)
    r, lr, w, pg = cnr.parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and pg is not None
    R, n1, n2, L, n_obj = 32, 2, 14, 32, 4
    theta, lay = cnr.fused.init_params(1, L, n_obj, torch.Generator().manual_seed(0))      # same weights everywhere
    full = cnr.scene_cateogries.synthetic_pool(2 * 64, n_obj, torch.Generator().manual_seed(9), "cpu")
    pool = cnr.parallel.shard_pool(full, rank, world)                                        # rank-private rays
    m, v2 = torch.zeros_like(theta), torch.zeros_like(theta)
    for step in range(1, 3):
        th = theta.clone().requires_grad_()
        vv = lay.views(th)
        mlp, off = {}, 0
        for n, o, i in cnr.ops.TRUNK_LAYERS:
            mlp[n + ".weight"] = vv["trunk"][:, off:off + o * i].reshape(1, o, i); off += o * i
            mlp[n + ".bias"] = vv["trunk"][:, off:off + o]; off += o
        for k, n in enumerate(cnr.ops.LATENT_LAYERS):
            mlp[n + ".weight"], mlp[n + ".bias"] = vv["latW"][:, k], vv["latb"][:, k]
        gen = torch.Generator().manual_seed(5 + step)
        o_, d_ = O.origin_dirs_O(pool["T_co"][:R], pool["dirs"][:R])
        u, g = torch.rand(R, n1 + n2, generator=gen), torch.randn(R, n2, generator=gen) * (0.1 / 3)
        gt_rgb, gt_d, mask, lab, pts, z = O.sample_3d_points(pool["rgbs"][:R], pool["depth"][:R], o_, d_, u, g, n1, n2, 0.1, 0.05)
        batch = dict(pts=pts[None], z=z[None], gt_depth=gt_d[None], gt_rgb=(gt_rgb / 255.0)[None], labels=lab[None],
                     depth_mask=mask[None], indices=pool["indices"][:R][None])
        loss, _ = O.forward_loss(mlp, vv["B"], 2.0, [vv["shape"][0]], [vv["tex"][0]], batch)
        (loss / world).backward()                      # the fused trainer folds 1/world into the loss kernel
        grad = th.grad.clone()
        local = grad.clone()
        cnr.parallel.allreduce_mean_(grad, pg, prescaled=True)        # ONE collective on the flat buffer
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local, group=pg)
        assert torch.allclose(grad, sum(gathered), atol=1e-7)
        # identical AdamW on every rank
        p = theta.clone().requires_grad_(); p.grad = grad
        opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0.013)
        if step > 1:
            opt.state[p] = dict(step=torch.tensor(float(step - 1)), exp_avg=m, exp_avg_sq=v2)
        opt.step()
        m, v2 = opt.state[p]["exp_avg"], opt.state[p]["exp_avg_sq"]
        theta = p.detach()
        assert cnr.parallel.params_in_sync(theta, pg, atol=0.0)
    if rank == 0:
        torch.save(dict(theta=theta, loss=float(loss)), out)
    dist.destroy_process_group()


def test_two_rank_flat_allreduce_keeps_replicas_identical(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.isfinite(res["theta"]).all() and res["loss"] > 0
