#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's --save-temps assembly (gfx950).

    hipcc --offload-arch=gfx950 -O3 ... --save-temps -c file.hip        # leaves file-hip-amdgcn-amd-amdhsa-gfx950.s
    python tools/isa_mix.py file-hip-amdgcn-amd-amdhsa-gfx950.s '<substring of the mangled kernel name>' [--blocks N] [--phases]

Prints the kernel's instruction classes per basic block (largest first) and, with --phases, per stretch between two
`s_barrier`s / `; CNRMARK` comments inside each loop body, in program order.  Classes: mfma, trans (quarter-rate: v_sin / v_cos /
v_exp / v_log / v_rcp / v_rsq / v_sqrt), cvt (v_cvt_*), pk (v_pk_*), valu (every other v_*), lds (ds_*), vmem, salu, wait.
A wave-instruction occupies the SIMD's vector unit for 4 cycles (16 for a transcendental): `issue` = 4 (valu + cvt + pk) + 16 trans.
"""
import re
import sys
from collections import Counter, OrderedDict

TRANS = ("v_sin_", "v_cos_", "v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_cvt_"):
        return "cvt"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_lines(path, needle):
    out, on = [], False
    for ln in open(path):
        if not on:
            if ln.startswith("_Z") and needle in ln and ln.rstrip().split(":")[0].endswith(ln.split(":")[0]):
                on = True
            continue
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            break
        out.append(ln.rstrip("\n"))
    return out


def blocks_of(lines):
    """[(label, [ops])] in program order"""
    blocks, cur, name = [], [], "entry"
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) and not s.startswith("; CNRMARK"):
            continue
        if s.startswith("; CNRMARK"):
            cur.append(("mark", s))
            continue
        op = s.split()[0]
        cur.append((op, s))
    blocks.append((name, cur))
    return blocks


def mix(ops):
    c = Counter(classify(op) for op, _ in ops if op != "mark")
    c["issue"] = 4 * (c["valu"] + c["cvt"] + c["pk"]) + 16 * c["trans"]
    return c


COLS = ("mfma", "valu", "cvt", "pk", "trans", "lds", "vmem", "salu", "wait", "barrier", "issue")


def fmt(c):
    return " ".join(f"{k}={c[k]:<5d}" for k in COLS)


def main():
    path, needle = sys.argv[1], sys.argv[2]
    nblocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 12
    lines = kernel_lines(path, needle)
    if not lines:
        sys.exit(f"no kernel matching {needle!r}")
    blocks = blocks_of(lines)
    total = Counter()
    for _, ops in blocks:
        total.update(mix(ops))
    print(f"kernel: {len(lines)} lines, {len(blocks)} basic blocks")
    print("total  ", fmt(total))
    big = sorted(blocks, key=lambda b: -len(b[1]))[:nblocks]
    print("\nlargest basic blocks:")
    for name, ops in big:
        print(f"  {name:<14s} n={len(ops):<6d}", fmt(mix(ops)))
    if "--phases" in sys.argv:
        print("\nstretches between barriers / marks, blocks in program order (blocks with >= 40 instructions):")
        for name, ops in blocks:
            if len(ops) < 40:
                continue
            print(f"  block {name} ({len(ops)} instructions)")
            cur, k, tag = [], 0, "start"
            for op, s in ops:
                if op == "s_barrier" or op == "mark":
                    print(f"    {k:>2d} {tag:<28s}", fmt(mix(cur)))
                    cur, k = [], k + 1
                    tag = s[2:].strip() if op == "mark" else "s_barrier"
                else:
                    cur.append((op, s))
            print(f"    {k:>2d} {tag:<28s}", fmt(mix(cur)))


if __name__ == "__main__":
    main()
