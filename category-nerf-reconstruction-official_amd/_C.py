"""ctypes binding of libcnr_hip.so (the C-ABI declared in include/cnr_hip.h).

There is NO CPU fallback: if the library is missing, or a tensor is not a contiguous device tensor of
the documented dtype, the call raises.  (tests/ may install a test double for host-logic tests on a
GPU-less box through :func:`install_test_double`; nothing in the product ever does.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CNR_HIP_LIB: another build of the same library, e.g. tools/exp's cycle-stamp build; never a different implementation)
LIB_PATH = os.environ.get("CNR_HIP_LIB") or os.path.join(_HERE, "libcnr_hip.so")

_vp, _i, _i64, _u64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float

# name -> argtypes ; every function returns int.  Mirrors include/cnr_hip.h one to one.
SIGNATURES = {
    "cnr_version": [],
    "cnr_device_info": [_vp, _vp, _vp],
    "cnr_camera_rays": [_vp, _i, _i, _f, _f, _f, _f, _vp],
    "cnr_sample_maxdepth": [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp],
    "cnr_step_advance": [_vp, _i64, _vp],
    "cnr_sample_rays": [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _f, _f, _f,
                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "cnr_latent_fwd": [_vp, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _vp, _vp, _vp],
    "cnr_latent_bwd": [_vp, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp],
    "cnr_pe_fwd": [_vp, _vp, _vp, _i, _i64, _f, _vp],
    "cnr_pe_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i64, _f, _vp],
    "cnr_mlp_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cnr_mlp_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "cnr_composite_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "cnr_composite_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "cnr_loss_fwd_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                         _i, _i, _vp],
    "cnr_adamw_step": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i64, _f, _vp, _vp],
    "cnr_pack_bytes": [],
    "cnr_pack_weights": [_vp, _vp, _i, _vp],
    "cnr_field_fwd": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i64, _vp, _vp],
    "cnr_pack_lo_bytes": [],
    "cnr_pack_weights_lo": [_vp, _vp, _i, _vp],
    "cnr_field_bwd_pipe": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _i64, _i64, _i64, _vp, _i, _vp, _vp],
    "cnr_field_bwd_workspace_bytes": [_i, _i],
    "cnr_field_bwd_pipe_blocks": [_i, _i, _i, _i],
    "cnr_gather_pool": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cnr_dense_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cnr_dense_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _i, _f, _vp],
    "cnr_dense_bwd_workspace_bytes": [_i, _i, _i],
    "cnr_step_prologue": [_vp, _vp],          # (const cnr_step_prologue_args*, stream): STRUCTS below
    "cnr_epoch_perm": [_vp, _i64, _i, _u64, _u64, _vp, _vp, _i64, _vp],
    "cnr_slice_maskcounts": [_vp, _vp, _vp, _i64, _i, _i, _i, _f, _vp, _vp],
    "cnr_slice_maxdepth": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp],
    "cnr_adamw_epilogue": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64,
                           _vp, _vp, _i, _i, _vp],
    "cnr_step_tail": [_vp, _vp],
    "cnr_step_grad": [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp],
    "cnr_field_fwd_render_blocks": [_i, _i],
    "cnr_field_fwd_render_workspace_bytes": [_i, _i, _i],
    "cnr_field_fwd_render": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                             _vp, _i, _i, _i, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "cnr_pack_fp8_bytes": [_i],
    "cnr_pack_weights_fp8": [_vp, _vp, _i, _i, _vp],
    "cnr_field_fwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i64, _i, _vp],
    "cnr_field_train_blocks": [_i, _i, _i, _i],
    "cnr_field_train_workspace_bytes": [_i, _i, _i, _i, _i],
    "cnr_field_train": [_vp, _vp],
    "cnr_param_prep": [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp],
    "cnr_render_loss_workspace_bytes": [_i, _i],
    "cnr_render_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp,
                        _i64, _vp, _vp, _vp],
    "cnr_render_loss_finish": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "cnr_step_epilogue": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _vp],
    "cnr_bg_pack_bytes": [],
    "cnr_bg_param_count": [],
    "cnr_bg_blocks": [_i],
    "cnr_bg_dw_chunks": [_i, _i],
    "cnr_bg_record_floats": [],
    "cnr_bg_pack": [_vp, _vp, _vp],
    "cnr_bg_forward": [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "cnr_bg_backward": [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "cnr_bg_dw": [_vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp],
    "cnr_bg_backward_render_workspace_bytes": [_i],
    "cnr_bg_backward_render": [_vp, _vp],
    "cnr_bg_tail_sample": [_vp, _vp],
    "cnr_bg_tail": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _f, _f, _f, _f, _vp, _i64, _vp, _vp, _i, _vp, _vp, _vp],
}
# The three launches of the fused trainer's step take ONE versioned struct (include/cnr_hip.h: struct_size and abi_version
# first, then these fields in this order).  call_struct() wants every field by NAME: a missing, misspelt or surplus argument
# raises here instead of shifting 40 positional values by one.
ABI_VERSION = 3
_u32, _i32 = ctypes.c_uint32, ctypes.c_int32
STRUCTS = {
    "cnr_step_prologue": [
        ("theta", _vp), ("class_stride", _i64), ("off_trunk", _i64), ("off_latW", _i64), ("off_latb", _i64), ("off_shape", _i64),
        ("off_tex", _i64), ("L", _i32), ("n_obj", _i32), ("C", _i32), ("packed", _vp), ("packed_lo", _vp), ("zl", _vp),
        ("biasrows", _vp), ("zero_buf", _vp), ("zero_count", _i64), ("rgbs", _vp), ("depth", _vp), ("dirs_c", _vp), ("T", _vp),
        ("u", _vp), ("g", _vp), ("seed", _u64), ("offset", _u64), ("d_state", _vp), ("pool_rows", _i64), ("max_bound", _vp),
        ("world_frame", _i32), ("R", _i32), ("n1", _i32), ("n2", _i32), ("eps", _f), ("stop_eps", _f), ("min_bound", _f),
        ("z", _vp), ("pts", _vp), ("origins", _vp), ("dirs_o", _vp), ("gt_rgb", _vp), ("gt_depth", _vp), ("depth_mask", _vp),
        ("labels", _vp), ("pool_indices", _vp), ("ray_row", _vp), ("perm", _vp), ("max_bound_slices", _i32), ("rng_c0", _i32),
        ("rng_cstride", _i32), ("rng_R", _i32), ("rng_r0", _i32)],
    "cnr_step_tail": [
        ("theta_in", _vp), ("theta_out", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("class_stride", _i64),
        ("off_B", _i64), ("off_latW", _i64), ("off_latb", _i64), ("off_shape", _i64), ("off_tex", _i64), ("L", _i32),
        ("n_obj", _i32), ("C", _i32), ("zl", _vp), ("dbiasrows", _vp), ("reg_scale", _f), ("do_latent", _i32), ("lr", _f),
        ("beta1", _f), ("beta2", _f), ("eps", _f), ("weight_decay", _f), ("state_cur", _vp), ("state_next", _vp),
        ("add_rows", _i64), ("rl_workspace", _vp), ("losses", _vp), ("flags", _vp), ("depth", _vp), ("pool_rows", _i64),
        ("perm", _vp), ("next_max_bound", _vp), ("R", _i32), ("records", _vp), ("nwg", _i32), ("rows_fix", _vp),
        ("rl_blocks", _i32), ("clamp_flags", _vp), ("n_obj_cls", _vp), ("code_lr", _f), ("code_weight_decay", _f)],
    "cnr_field_train": [
        ("pts", _vp), ("B", _vp), ("packed", _vp), ("packed_lo", _vp), ("biasrows", _vp), ("ray_row", _vp), ("scale", _f),
        ("z", _vp), ("gt_depth", _vp), ("gt_rgb", _vp), ("labels", _vp), ("depth_mask", _vp), ("counts_tab", _vp),
        ("d_state", _vp), ("color_scaling", _f), ("opacity_scaling", _f), ("loss_scale", _f), ("grad_scale", _f),
        ("depth", _vp), ("var", _vp), ("rgb", _vp), ("opacity", _vp), ("C", _i32), ("R", _i32), ("S", _i32),
        ("rows_per_class", _i32), ("max_blocks", _i32), ("records", _vp), ("records_bytes", _i64), ("loss_workspace", _vp),
        ("loss_workspace_bytes", _i64), ("B_stride", _i64), ("rows_fix", _vp), ("clamp_flags", _vp)],
    "cnr_bg_tail_sample": [
        ("theta", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("partials", _vp), ("chunks", _i32), ("records", _vp),
        ("nrec", _i32), ("grad_scale", _f), ("lr", _f), ("beta1", _f), ("beta2", _f), ("adam_eps", _f), ("weight_decay", _f),
        ("packed", _vp), ("rl_workspace", _vp), ("rl_R", _i32), ("losses", _vp), ("flags", _vp), ("rgbs", _vp), ("depth", _vp),
        ("dirs_c", _vp), ("T", _vp), ("seed", _u64), ("offset", _u64), ("d_state", _vp), ("pool_rows", _i64), ("max_bound", _vp),
        ("max_bound_slices", _i32), ("world_frame", _i32), ("R", _i32), ("n1", _i32), ("n2", _i32), ("eps", _f), ("stop_eps", _f),
        ("min_bound", _f), ("perm", _vp), ("z", _vp), ("pts", _vp), ("origins", _vp), ("dirs_o", _vp), ("gt_rgb", _vp),
        ("gt_depth", _vp), ("depth_mask", _vp), ("labels", _vp)],
    "cnr_bg_backward_render": [
        ("pts", _vp), ("theta", _vp), ("packed", _vp), ("scale", _f), ("R", _i32), ("S", _i32), ("sigma", _vp), ("rgb", _vp),
        ("z", _vp), ("gt_depth", _vp), ("gt_rgb", _vp), ("labels", _vp), ("depth_mask", _vp), ("counts_tab", _vp),
        ("d_state", _vp), ("color_scaling", _f), ("opacity_scaling", _f), ("grad_scale", _f), ("act", _vp), ("dpre", _vp),
        ("records", _vp), ("depth", _vp), ("var", _vp), ("rgb_render", _vp), ("opacity", _vp), ("d_sigma", _vp), ("d_rgb", _vp), ("loss_workspace", _vp),
        ("loss_workspace_bytes", _i64)],
}
_struct_types = {}


def struct_type(name):
    """ctypes.Structure of entry point `name`'s argument block (natural C alignment, like the header's typedef)."""
    if name not in _struct_types:
        fields = [("struct_size", _u32), ("abi_version", _u32)] + STRUCTS[name]
        _struct_types[name] = type(name + "_args", (ctypes.Structure,), {"_fields_": fields})
    return _struct_types[name]


_RESTYPE64 = {"cnr_pack_bytes", "cnr_pack_lo_bytes", "cnr_field_bwd_workspace_bytes", "cnr_render_loss_workspace_bytes",
              "cnr_dense_bwd_workspace_bytes", "cnr_field_fwd_render_workspace_bytes", "cnr_field_train_workspace_bytes", "cnr_pack_fp8_bytes", "cnr_bg_pack_bytes", "cnr_bg_backward_render_workspace_bytes"}

_lib = None
_double = None


class CnrError(RuntimeError):
    pass


def load():
    """Load libcnr_hip.so; raise loudly when it is absent (build with __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CnrError(f"{LIB_PATH} not found: the HIP extension is not built "
                           "(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU path")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int64 if name in _RESTYPE64 else ctypes.c_int
        _lib = lib
    return _lib


def install_test_double(obj):
    """tests/ only: route calls to `obj.<name>(*tensors_and_scalars)` instead of the HIP library."""
    global _double
    _double = obj


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise CnrError("required tensor is None")
    if not t.is_cuda:
        raise CnrError("cnr kernels take device tensors only (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise CnrError("cnr kernels take contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise CnrError(f"expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """args: torch tensors (passed as device pointers), None (NULL) or python scalars."""
    if _double is not None:
        rc = getattr(_double, name)(*args)
        if rc:
            raise CnrError(f"{name} (test double) returned {rc}")
        return
    lib = load()
    conv = []
    for a in args:
        if torch.is_tensor(a):
            conv.append(_ptr(a))
        else:
            conv.append(a)
    # every parameter of the header's prototype, in order (optional pointers as an explicit None): a short or long argument
    # list is a caller bug, never padded
    if name in STRUCTS:
        raise CnrError(f"{name} takes a versioned argument block: use call_struct({name!r}, field=value, ...)")
    types = SIGNATURES[name][:-1]
    if len(conv) != len(types):
        raise CnrError(f"{name}: {len(conv)} arguments for {len(types)} parameters (include/cnr_hip.h)")
    rc = getattr(lib, name)(*conv, _stream())
    if rc != 0:
        raise CnrError(f"{name} failed with code {rc}" + (" (argument error)" if rc < 0 else " (hipError_t)"))


def call_struct(name, **fields):
    """Entry points with a versioned argument block: every field of STRUCTS[name] by keyword (tensors as device pointers,
    None = NULL).  Unknown or missing names raise."""
    _invoke_struct(name, _prepare_struct(name, fields))


def _prepare_struct(name, fields):
    spec = STRUCTS[name]
    names = [n for n, _ in spec]
    missing, unknown = [n for n in names if n not in fields], [n for n in fields if n not in names]
    if missing or unknown:
        raise CnrError(f"{name}: missing fields {missing}, unknown fields {unknown}")
    if _double is not None:
        return dict(fields)
    load()
    st = struct_type(name)()
    st.struct_size, st.abi_version = ctypes.sizeof(st), ABI_VERSION
    for n, t in spec:
        v = fields[n]
        if t is _vp:
            v = _ptr(v) if torch.is_tensor(v) else v
            if v is not None and not isinstance(v, int):
                raise CnrError(f"{name}.{n}: expected a device tensor or None, got {type(fields[n]).__name__}")
        elif torch.is_tensor(v) or v is None:
            raise CnrError(f"{name}.{n}: expected a number, got {type(v).__name__}")
        elif t is not _f:
            v = int(v)
        setattr(st, n, v)
    return st


def _invoke_struct(name, st):
    if _double is not None:
        rc = getattr(_double, name)(**st)
        if rc:
            raise CnrError(f"{name} (test double) returned {rc}")
        return
    rc = getattr(load(), name)(ctypes.byref(st), _stream())
    if rc != 0:
        raise CnrError(f"{name} failed with code {rc}" + (" (argument error)" if rc < 0 else " (hipError_t)"))


def version():
    return load().cnr_version()


def pack_bytes():
    return int(load().cnr_pack_bytes())


def field_bwd_workspace_bytes(C, max_blocks):
    """bytes of the per-workgroup gradient records of cnr_field_bwd_pipe / cnr_field_train"""
    return int(load().cnr_field_bwd_workspace_bytes(int(C), int(max_blocks)))


def render_loss_workspace_bytes(C, R):
    return int(load().cnr_render_loss_workspace_bytes(int(C), int(R)))


def device_info():
    n_cu, lds, is950 = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    rc = load().cnr_device_info(ctypes.byref(n_cu), ctypes.byref(lds), ctypes.byref(is950))
    if rc != 0:
        raise CnrError(f"cnr_device_info failed with {rc}")
    return {"n_cu": n_cu.value, "lds_bytes": lds.value, "gfx950": bool(is950.value)}


# ---- optional per-kernel HIP-event timing (bench.py's roofline leg) -------------------------------------
_timing = None  # {name: [(start_event, end_event), ...]} when enabled


def enable_kernel_timing(names):
    """Record a HIP event pair (on the launch stream) around every call of the named entry points."""
    global _timing
    _timing = {n: [] for n in names}


def kernel_timings_ms():
    """-> {name: [ms, ...]} ; synchronises.  Disables timing."""
    global _timing
    torch.cuda.synchronize()
    out = {n: [a.elapsed_time(b) for a, b in evs] for n, evs in (_timing or {}).items()}
    _timing = None
    return out


_raw_call = call


def call(name, *args):  # noqa: F811
    if _timing is not None and name in _timing:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _raw_call(name, *args)
        b.record()
        _timing[name].append((a, b))
    else:
        _raw_call(name, *args)


def call_struct(name, **fields):  # noqa: F811
    st = _prepare_struct(name, fields)       # (the argument block is filled BEFORE the first event: ~40 us of host work that an
    if _timing is not None and name in _timing:      # idle GPU would otherwise show as kernel time)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _invoke_struct(name, st)
        b.record()
        _timing[name].append((a, b))
    else:
        _invoke_struct(name, st)
