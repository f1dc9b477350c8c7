"""CPU: the C-ABI library loads without a GPU, exports every symbol include/cnr_hip.h declares, the ctypes
signature table mirrors the header, and argument errors come back as codes (never exit(), never a crash)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "cnr_hip.h")
LIB = os.path.join(ROOT, "category-nerf-reconstruction-official_amd", "libcnr_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(cnr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",")]
        out[m.group(2)] = [a for a in args if a and a != "void"]
    return out


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_hot_path():
    fns = declared_functions()
    for name in ("cnr_sample_rays", "cnr_pe_fwd", "cnr_pe_bwd", "cnr_mlp_fwd_f32", "cnr_mlp_bwd_f32",
                 "cnr_composite_fwd", "cnr_composite_bwd", "cnr_loss_fwd_bwd", "cnr_adamw_step", "cnr_pack_weights",
                 "cnr_field_fwd", "cnr_field_bwd_pipe", "cnr_render_loss", "cnr_render_loss_finish", "cnr_step_epilogue", "cnr_param_prep", "cnr_step_prologue", "cnr_adamw_epilogue", "cnr_step_tail", "cnr_slice_maxdepth", "cnr_step_grad", "cnr_field_fwd_render", "cnr_gather_pool", "cnr_dense_fwd", "cnr_dense_bwd", "cnr_latent_fwd", "cnr_latent_bwd", "cnr_step_advance", "cnr_field_train", "cnr_slice_maskcounts", "cnr_field_fwd_fp8",
                 "cnr_pack_weights_fp8"):
        assert name in fns, name


def test_every_declared_symbol_is_exported(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in cnr_hip.h but not exported by libcnr_hip.so"


def test_ctypes_table_matches_header():
    import cnr_amd
    fns = declared_functions()
    sig = cnr_amd._C.SIGNATURES
    assert set(sig) == set(fns), (set(sig) ^ set(fns))
    for name, args in fns.items():
        assert len(sig[name]) == len(args), (name, len(sig[name]), len(args))
        for ct, decl in zip(sig[name], args):
            if "*" in decl:
                assert ct is ctypes.c_void_p, (name, decl)
            elif decl.startswith("float"):
                assert ct is ctypes.c_float, (name, decl)
            elif decl.startswith(("int64_t", "uint64_t")):
                assert ct in (ctypes.c_int64, ctypes.c_uint64), (name, decl)
            elif decl.startswith("int"):
                assert ct is ctypes.c_int, (name, decl)


def declared_structs():
    """typedef struct NAME_args { type field; ... } NAME_args;  ->  {NAME: [(field, c type text), ...]}"""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(cnr_\w+)_args\s*\{(.*?)\}\s*\1_args\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if decl:
                typ, name = decl.rsplit(" ", 1)
                if name.startswith("*"):
                    typ, name = typ + "*", name.lstrip("*")
                fields.append((name, typ))
        out[m.group(1)] = fields
    return out


def test_argument_blocks_match_the_header():
    """The versioned argument structs: same fields, order and C types in include/cnr_hip.h and in the ctypes table, and the
    same size as the C compiler gives the header's typedef."""
    import subprocess
    import tempfile
    import cnr_amd
    _C = cnr_amd._C
    structs = declared_structs()
    assert set(structs) == set(_C.STRUCTS) and len(structs) == 5
    ctype_of = {"float": ctypes.c_float, "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64,
                "uint64_t": ctypes.c_uint64}
    for name, fields in structs.items():
        assert fields[0] == ("struct_size", "uint32_t") and fields[1] == ("abi_version", "uint32_t"), name
        assert [f for f, _ in fields[2:]] == [f for f, _ in _C.STRUCTS[name]], name
        for (f, typ), (_, ct) in zip(fields[2:], _C.STRUCTS[name]):
            assert ct is (ctypes.c_void_p if "*" in typ else ctype_of[typ]), (name, f, typ)
    src = '#include <stdio.h>\n#include "cnr_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %d", sizeof(cnr_step_prologue_args), ' \
          'sizeof(cnr_step_tail_args), sizeof(cnr_field_train_args), sizeof(cnr_bg_backward_render_args), ' \
          'sizeof(cnr_bg_tail_sample_args), CNR_ABI_VERSION);return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.dirname(HEADER), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        a, b, c, e, f5, ver = (int(v) for v in subprocess.run([os.path.join(d, "s")], capture_output=True, text=True).stdout.split())
    assert (a, b, c, e, f5) == tuple(ctypes.sizeof(_C.struct_type(n)) for n in ("cnr_step_prologue", "cnr_step_tail", "cnr_field_train",
                                                                                  "cnr_bg_backward_render", "cnr_bg_tail_sample"))
    assert ver == _C.ABI_VERSION


def test_argument_blocks_of_another_revision_are_refused(lib):
    """struct_size / abi_version that do not match this library: CNR_E_ARG before anything else is read."""
    import cnr_amd
    _C = cnr_amd._C
    for name in _C.STRUCTS:
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int
        st = _C.struct_type(name)()
        st.struct_size, st.abi_version = ctypes.sizeof(st) - 8, _C.ABI_VERSION
        assert fn(ctypes.byref(st), None) == -1, name
        st.struct_size, st.abi_version = ctypes.sizeof(st), _C.ABI_VERSION + 1
        assert fn(ctypes.byref(st), None) == -1, name
        assert fn(None, None) == -1, name
        st.struct_size, st.abi_version = ctypes.sizeof(st), _C.ABI_VERSION      # right revision, all-NULL fields: still an error code
        assert fn(ctypes.byref(st), None) == -1, name


def test_binding_refuses_short_misspelt_and_surplus_arguments():
    """_C.call wants exactly the header's parameter list (round 2 padded missing trailing arguments with NULL / 0);
    _C.call_struct wants every field of the argument block by name."""
    import torch
    import cnr_amd
    _C = cnr_amd._C
    with pytest.raises(_C.CnrError, match="arguments for"):
        _C.call("cnr_pe_fwd", None, None, None, 1, 10)                     # scale and the output left out
    with pytest.raises(_C.CnrError, match="arguments for"):
        _C.call("cnr_step_advance", None, 1, 2, 3)
    with pytest.raises(_C.CnrError, match="versioned argument block"):
        _C.call("cnr_step_tail", *([None] * 42))
    good = {n: (None if t is ctypes.c_void_p else 0) for n, t in _C.STRUCTS["cnr_step_tail"]}
    with pytest.raises(_C.CnrError, match="missing fields \\['code_lr'\\]"):
        _C.call_struct("cnr_step_tail", **{k: v for k, v in good.items() if k != "code_lr"})
    with pytest.raises(_C.CnrError, match="unknown fields \\['learning_rate'\\]"):
        _C.call_struct("cnr_step_tail", learning_rate=1e-3, **good)
    with pytest.raises(_C.CnrError, match="expected a number"):
        _C.call_struct("cnr_step_tail", **dict(good, lr=None))
    with pytest.raises(_C.CnrError):                                          # a host tensor where a device pointer belongs
        _C.call_struct("cnr_step_tail", **dict(good, grad=torch.zeros(4)))


def test_argument_errors_are_return_codes(lib):
    """NULL pointers / bad sizes -> CNR_E_ARG (-1) before anything touches the device."""
    import cnr_amd
    for name, argtypes in cnr_amd._C.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name in cnr_amd._C._RESTYPE64 else ctypes.c_int
    assert lib.cnr_version() >= 100
    assert lib.cnr_pack_bytes() == 62464
    assert lib.cnr_pe_fwd(None, None, None, 1, 10, 2.0, None) == -1
    assert lib.cnr_composite_fwd(None, None, None, None, None, None, None, None, 4, 8, 0, None) == -1
    assert lib.cnr_adamw_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None, None) == -1
    assert lib.cnr_step_advance(None, 1, None) == -1
    rec_entries = (13892 + 126 + 15 * 128 + 255) // 256 * 256     # trunk | two dB halves | up to 15 object rows x 128
    assert lib.cnr_field_bwd_workspace_bytes(2, 0) == 2 * 256 * rec_entries * 2       # bf16 entries


def test_missing_library_fails_loudly(monkeypatch):
    import cnr_amd
    monkeypatch.setattr(cnr_amd._C, "_lib", None)
    monkeypatch.setattr(cnr_amd._C, "LIB_PATH", "/nonexistent/libcnr_hip.so")
    with pytest.raises(cnr_amd._C.CnrError):
        cnr_amd._C.load()


def test_cpu_tensors_are_rejected():
    """No CPU fallback: handing a host tensor to a kernel wrapper raises."""
    import torch
    import cnr_amd
    with pytest.raises(cnr_amd._C.CnrError):
        cnr_amd.ops.UniDirsEmbedFn.apply(torch.zeros(4, 3), torch.zeros(21, 3), 2.0)


def test_kernels_contain_no_instruction_emitting_inline_asm():
    """hipcc pads no hazards around an inline-asm instruction and may hand its output a register an MFMA issued just before
    is still reading ("a hazard worth recording": run-to-run different gradients that the small fixtures
    never showed).  The kernels therefore use asm statements only as optimisation barriers: every asm template in csrc/ must
    be the empty string -- or an assembler COMMENT ("; ..."), which emits nothing either (the phase marks of tools/isa_mix.py,
    compiled in only with -DCNR_ISA_MARKS)."""
    csrc = os.path.join(ROOT, "category-nerf-reconstruction-official_amd", "csrc")
    bad = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(\s*\"([^\"]*)\"", src):
            if m.group(1).strip() and not m.group(1).strip().startswith(";"):
                bad.append((name, m.group(1)))
    assert not bad, bad
