"""Worker of tests/test_multigpu_gpu.py: one rank of a sharded FusedCategoryTrainer run (gloo group; every rank on
cuda:0 of a one-GPU box -- the multi-GPU launch differs in backend 'nccl' and LOCAL_RANK only).  Saves what the parent
compares against a single-process run."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(cnr, dev, mode, rank, world, pg, C, R, n_obj, L, graph, empty_class=None):
    """The trainer rank `rank` of `world` runs (world 1: the single-process reference)."""
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=4, n_bins=28)
    Rg = R                                              # global rays per class and step
    pools = []
    for c in range(C):
        p = cnr.scene_cateogries.synthetic_pool(6 * Rg, n_obj, torch.Generator().manual_seed(100 + c), "cpu")
        if c == empty_class:
            p["rgbs"][:, 3] = 0
        pools.append(p)
    gen = torch.Generator().manual_seed(42)             # same parameters on every rank
    # (the workgroup count per class is fixed here: its default follows the number of classes ON THE GPU -- 256 / C, for speed -- and
    #  the per-workgroup records are summed in order, so runs agree bit for bit when they split a class's rays the same way)
    kw = dict(seed=3, generator=gen, use_graph=graph, bwd_blocks=32)
    if world == 1:
        return cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, Rg, dev, **kw)
    if mode == "class":
        ids = cnr.parallel.class_shard(C, rank, world)
        return cnr.fused.FusedCategoryTrainer(cfg, len(ids), n_obj, [pools[i] for i in ids], Rg, dev, process_group=pg,
                                              shard="class", n_cls_global=C, class_ids=ids, dp_rank=rank, dp_world=world, **kw)
    return cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, Rg // world, dev, process_group=pg, shard="ray",
                                          dp_rank=rank, dp_world=world, **kw)


def main():
    mode, out, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    empty = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    import cnr_amd as cnr
    rank, _, world, pg = cnr.parallel.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tr = build(cnr, dev, mode, rank, world, pg, 4, 256, 4, 32, True, empty)
    hist = []
    for _ in range(steps):
        tr.step()
        hist.append(tr.loss_values().cpu())
    torch.cuda.synchronize()
    torch.save(dict(theta=tr.theta.cpu(), hist=torch.stack(hist), ids=tr.class_ids, flags=tr.flags.cpu(),
                    in_sync=cnr.parallel.params_in_sync(tr.theta.contiguous(), pg) if mode == "ray" else True),
               f"{out}.{rank}")
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
