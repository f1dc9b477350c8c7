"""cnr_render_loss vs the three-call sequence, back-to-back launches timed with HIP events."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import _C
dev = torch.device("cuda:0")


def t(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for C, R, S in [(1, 2048, 64), (1, 8192, 128), (8, 4096, 64)]:
    f = lambda *s: torch.empty(*s, device=dev)
    sig, col, z = torch.randn(C, R, S, device=dev), torch.rand(C, R, S, 3, device=dev), torch.rand(C, R, S, device=dev).sort(-1).values
    gt_d, gt_c = torch.rand(C, R, device=dev), torch.rand(C, R, 3, device=dev)
    labels = torch.randint(0, 3, (C, R), device=dev).to(torch.uint8)
    dmask = (torch.rand(C, R, device=dev) > 0.2).to(torch.uint8)
    depth, var, rgb, opa, dd, dr, do = f(C, R), f(C, R), f(C, R, 3), f(C, R), f(C, R), f(C, R, 3), f(C, R)
    losses, flags = f(3, C), torch.empty(C, device=dev, dtype=torch.int32)
    dsig, dcol = f(C, R, S), f(C, R, S, 3)
    ws = torch.zeros(_C.render_loss_workspace_bytes(C, R), device=dev, dtype=torch.uint8)

    def three():
        _C.call("cnr_composite_fwd", sig, col, z, None, depth, var, rgb, opa, C * R, S, 0)
        _C.call("cnr_loss_fwd_bwd", depth, var, rgb, opa, gt_d, gt_c, labels, dmask, 5.0, 10.0, 0.5, losses, flags, dd, dr, do, C, R)
        _C.call("cnr_composite_bwd", sig, col, z, dd, dr, do, None, dsig, dcol, C * R, S, 0)

    def one():
        _C.call("cnr_render_loss", sig, col, z, gt_d, gt_c, labels, dmask, 5.0, 10.0, 0.5, dsig, dcol,
                depth, var, rgb, opa, C, R, S, ws, ws.numel(), None, None)
        _C.call("cnr_render_loss_finish", ws, losses, flags, C, R, 0)
    print(f"C{C} R{R} S{S}: three calls {t(three):7.1f} us   one launch {t(one):7.1f} us", flush=True)
