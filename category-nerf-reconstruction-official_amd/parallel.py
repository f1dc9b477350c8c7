"""Multi-GPU layer of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference has no distributed code at all.  What shards here is the ray batch of each
class: every rank draws its own rays of the same categories from its own pool shard, runs the same per-GPU step,
and the ONE flat gradient buffer (all classes: trunk, latent layers, B, code tables -- 48 899 floats per class at
L = 256, n_obj = 4) is summed with ONE all-reduce per step; the loss kernel already scaled the per-rank gradient
by 1/world, so the sum is the mean over ranks and every rank applies the identical AdamW step (parameters stay
bitwise equal, no broadcast).  At 0.2 MB the collective is latency-bound on xGMI, which is why it is one call on
one contiguous buffer and not a call per tensor.  North-star wording ("all-reduce of the shared latent-code
gradients only") is a subset: the shared MLP / PE gradients must be in the buffer too, or replicas diverge.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run).  Returns
    (rank, local_rank, world, process_group or None)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, local_rank, world, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return rank, local_rank, world, dist.group.WORLD


def shard_rows(n_rows, rank, world):
    """Contiguous, near-equal shard [lo, hi) of a pool of n_rows rays for this rank."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_pool(pool, rank, world):
    """Row-shard every field of a ray pool dict (scene_cateogries.synthetic_pool layout)."""
    n = pool["depth"].shape[0]
    lo, hi = shard_rows(n, rank, world)
    return {k: v[lo:hi].contiguous() for k, v in pool.items()}


def allreduce_mean_(flat_grad, group, prescaled=True):
    """Sum the flat gradient over ranks (one collective).  prescaled: each rank's gradient already carries the
    1/world factor (the fused trainer folds it into the loss kernel), so the sum IS the mean."""
    if group is None:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    if not prescaled:
        flat_grad.div_(dist.get_world_size(group))
    return flat_grad


def allreduce_sum_(flat_grad, group):
    """ONE collective on the flat gradient of all classes: with ray shards every rank's gradient already carries
    1 / (global mask count), so the sum over ranks IS the gradient of the whole batch."""
    if group is not None:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def allreduce_any_(flags, group):
    """Logical OR over ranks of a 0/1 float table (the per-slice any-class-empty flags of render_rays.py:67-72 when
    classes live on different GPUs): once per epoch, a few hundred bytes."""
    if group is not None:
        t = flags.contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        flags.copy_(t)
    return flags


def class_shard(n_cls_global, rank, world):
    """Classes rank ``rank`` owns: ``rank, rank + world, ...`` (classes share nothing: no gradient ever crosses ranks)."""
    return list(range(rank, n_cls_global, world))


def params_in_sync(flat_params, group, atol=0.0):
    """Debug check: every rank holds the same parameters (max |p - p_rank0| <= atol)."""
    if group is None:
        return True
    ref = flat_params.clone()
    dist.broadcast(ref, src=0, group=group)
    bad = torch.tensor([float((flat_params - ref).abs().max() > atol)], device=flat_params.device)
    dist.all_reduce(bad, group=group)
    return bool(bad.item() == 0)


def mask_count_table(labels, depth_mask):
    """Host mirror of one entry of cnr_slice_maskcounts, for tables built outside the trainer and for CPU tests:
    labels (C,R) uint8, depth_mask (C,R) bool -> (C + 1, 4) floats: rows c = {#(depth & label != 0), #(label != 0),
    #(label != 2), 0}, row C = {1 where any class has a zero count, ..., ..., 0} (src/render_rays.py:66-72)."""
    mo, ms = labels != 0, labels != 2
    cnt = torch.stack([(depth_mask.bool() & mo).sum(-1), mo.sum(-1), ms.sum(-1), torch.zeros_like(mo.sum(-1))], -1).float()
    empty = torch.cat([(cnt[:, :3] == 0).any(0).float(), torch.zeros(1)])
    return torch.cat([cnt, empty[None]], 0)


def combine_ray_shard_tables(tables):
    """Per-rank tables of the ray shards of ONE global slice -> the table every rank must use: counts summed, empty flags
    recomputed from the sums (a rank-local count of zero means nothing)."""
    cnt = torch.stack([t[:-1] for t in tables]).sum(0)
    empty = torch.cat([(cnt[:, :3] == 0).any(0).float(), torch.zeros(1)])
    return torch.cat([cnt, empty[None]], 0)
