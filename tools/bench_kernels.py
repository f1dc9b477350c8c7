"""Time the fused kernels in isolation (back-to-back launches, HIP events)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import ops, _C
dev = torch.device("cuda:0")
L, n_obj = 256, 4


def run(C, R, S, blocks, iters=50):
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
    dtrunk = torch.zeros(C, 13892, device=dev); dB = torch.zeros(C, 21, 3, device=dev); dbr = torch.zeros_like(brows)
    wsp = torch.empty(_C.field_bwd_workspace_bytes(C, 0, C * n_obj), device=dev, dtype=torch.uint8)
    sig = torch.empty(C, R, S, device=dev); rgb = torch.empty(C, R, S, 3, device=dev)

    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3
    fwd = t(lambda: _C.call("cnr_field_fwd", pts, B, packed, brows, ray_row, 2.0, sig, rgb, C, R, S, 0, None))
    n = C * R * S
    line = f"C{C} R{R} S{S} blocks{blocks}: fwd {fwd:8.1f} us ({n*27422/fwd/1e6:7.1f} TF)"
    ref = None
    for variant in ("split", "pipe4"):
        dtrunk.zero_(); dB.zero_(); dbr.zero_()
        ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S, n_obj, blocks,
                      wsp, variant=variant)
        got = torch.cat([dtrunk.flatten(), dB.flatten(), dbr.flatten()]).clone()
        if ref is None: ref = got
        err = ((got - ref).norm() / ref.norm()).item()
        bwd = t(lambda: ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S,
                                      n_obj, blocks, wsp, variant=variant))
        line += f" | {variant} {bwd:8.1f} us ({n*82140/bwd/1e6:6.1f} TF) d={err:.1e}"
    print(line, flush=True)


for (C, R, S, blocks) in [(1, 2048, 64, 0), (1, 2048, 64, 228), (1, 2048, 64, 171), (1, 2048, 64, 128), (1, 2048, 64, 64), (1, 8192, 128, 0), (1, 8192, 128, 128), (1, 64, 32, 0), (1, 64, 32, 1)]:
    run(C, R, S, blocks)
