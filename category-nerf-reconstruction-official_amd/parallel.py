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


def params_in_sync(flat_params, group, atol=0.0):
    """Debug check: every rank holds the same parameters (max |p - p_rank0| <= atol)."""
    if group is None:
        return True
    ref = flat_params.clone()
    dist.broadcast(ref, src=0, group=group)
    bad = torch.tensor([float((flat_params - ref).abs().max() > atol)], device=flat_params.device)
    dist.all_reduce(bad, group=group)
    return bool(bad.item() == 0)
