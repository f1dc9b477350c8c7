"""Probe the 8-wave backward at growing sizes, one synchronised launch at a time (progress goes to stdout)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import ops, _C
if os.environ.get("CNR_TEST_LIB"):
    _C.LIB_PATH = os.environ["CNR_TEST_LIB"]
dev = torch.device("cuda:0")
L, n_obj = 256, 4
variant = sys.argv[1] if len(sys.argv) > 1 else "pipe4"
for (C, R, S, blocks) in [(1, 64, 96, 1), (1, 64, 96, 2), (1, 512, 64, 16), (1, 2048, 64, 16), (1, 2048, 64, 128),
                          (1, 2048, 64, 256), (1, 2048, 64, 0), (1, 8192, 128, 0), (2, 2048, 64, 0)]:
    gen = torch.Generator().manual_seed(0)
    theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
    v = lay.views(theta)
    trunk = v["trunk"].contiguous()
    packed = ops.pack_weights(trunk)
    pts = (torch.rand(C, R, S, 3, device=dev) * 2 - 1)
    B = v["B"].contiguous()
    brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
    ray_row = (torch.randint(0, n_obj, (C, R), device=dev) + torch.arange(C, device=dev)[:, None] * n_obj).to(torch.int32)
    dsig = torch.randn(C, R, S, device=dev) * 1e-3
    drgb = torch.randn(C, R, S, 3, device=dev) * 1e-3
    wsp = torch.empty(_C.field_bwd_workspace_bytes(C, 0), device=dev, dtype=torch.uint8)
    outs = {}
    for var in ("pipe3",) + tuple(variant.split(",")):
        dtrunk = torch.zeros(C, 13892, device=dev); dB = torch.zeros(C, 21, 3, device=dev); dbr = torch.zeros_like(brows)
        print(f"C{C} R{R} S{S} blocks{blocks} {var}: launch", flush=True)
        ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S, n_obj, blocks,
                      wsp, variant=var)
        torch.cuda.synchronize()
        outs[var] = torch.cat([dtrunk.flatten(), dB.flatten(), dbr.flatten()]).clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S, n_obj,
                          blocks, wsp, variant=var)
        b.record(); torch.cuda.synchronize()
        print(f"   done, {a.elapsed_time(b) / 10 * 1e3:.1f} us", flush=True)
    err = max(((outs[k] - outs["pipe3"]).norm() / outs["pipe3"].norm()).item() for k in outs)
    print(f"   rel diff vs pipe3 {err:.2e}", flush=True)
