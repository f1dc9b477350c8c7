"""cnr_field_fwd_render at a multi-class shape (ScanNet row of BASELINE.json on one GPU: 8 classes x 4096 x 128)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, cnr_amd
from cnr_amd import ops, _C
dev = torch.device("cuda:0")
L, n_obj = 256, 4
for C, R, S in ((8, 4096, 128), (4, 2048, 64), (1, 8192, 128)):
    gen = torch.Generator().manual_seed(0)
    theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
    v = lay.views(theta)
    packed = ops.pack_weights(v["trunk"].contiguous())
    B = v["B"].contiguous()
    brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
    pts = torch.rand(C, R, S, 3, device=dev) * 2 - 1
    z = torch.rand(C, R, S, device=dev).sort(-1).values
    ray_row = (torch.randint(0, n_obj, (C, R), device=dev) + torch.arange(C, device=dev)[:, None] * n_obj).to(torch.int32)
    f = lambda *s: torch.empty(*s, device=dev)
    gt_d, gt_c = f(C, R).zero_(), f(C, R, 3).zero_()
    lab, dm = torch.ones(C, R, device=dev, dtype=torch.uint8), torch.ones(C, R, device=dev, dtype=torch.uint8)
    ds, dc, d1, v1, r1, o1 = f(C, R, S), f(C, R, S, 3), f(C, R), f(C, R), f(C, R, 3), f(C, R)
    ws = torch.zeros(int(_C.load().cnr_field_fwd_render_workspace_bytes(C, R, S)), device=dev, dtype=torch.uint8)
    sig, rgb = f(C, R, S), f(C, R, S, 3)
    ws2 = torch.zeros(_C.render_loss_workspace_bytes(C, R), device=dev, dtype=torch.uint8)
    one = lambda: _C.call("cnr_field_fwd_render", pts, B, packed, brows, ray_row, 2.0, z, gt_d, gt_c, lab, dm, 5.0, 10.0,
                          1.0, ds, dc, d1, v1, r1, o1, C, R, S, 0, ws, ws.numel(), None, None, None)
    def two():
        _C.call("cnr_field_fwd", pts, B, packed, brows, ray_row, 2.0, sig, rgb, C, R, S, 0, None)
        _C.call("cnr_render_loss", sig, rgb, z, gt_d, gt_c, lab, dm, 5.0, 10.0, 1.0, ds, dc, d1, v1, r1, o1, C, R, S, ws2, ws2.numel(), None, None)
    for name, fn in (("one launch", one), ("two launches", two)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        b.record(); torch.cuda.synchronize()
        print(C, R, S, name, f"{a.elapsed_time(b)/50*1e3:.1f} us", flush=True)
