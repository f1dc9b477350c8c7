"""Ensemble glue with the reference's signature (src/utils.py:24-28).

``update_vmap(models, optimiser)`` stacks the per-class modules' parameters into ``(C, ...)`` leaves,
registers them as a new param group, and returns ``(fmodel, params, buffers)`` such that
``functorch.vmap(fmodel)(params, buffers, *stacked_inputs)`` -- the exact call at train.py:154-155 --
lands in the class-batched HIP kernels through the Functions' ``vmap`` rules (ops.py).
"""
import torch
from torch.func import functional_call, stack_module_state


def combine_state_for_ensemble(models):
    """(fmodel, params, buffers) like functorch's helper: params/buffers are tuples of stacked
    tensors in ``named_parameters()`` / ``named_buffers()`` order."""
    params, buffers = stack_module_state(models)
    pnames, bnames = list(params.keys()), list(buffers.keys())
    base = models[0]

    def fmodel(p, b, *args, **kwargs):
        state = {n: t for n, t in zip(pnames, p)}
        state.update({n: t for n, t in zip(bnames, b)})
        return functional_call(base, state, args, kwargs)

    return fmodel, tuple(params[n] for n in pnames), tuple(buffers[n] for n in bnames)


def update_vmap(models, optimiser):
    fmodel, params, buffers = combine_state_for_ensemble(models)
    [p.requires_grad_() for p in params]
    optimiser.add_param_group({"params": params})
    return (fmodel, params, buffers)
