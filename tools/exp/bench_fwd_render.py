import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, cnr_amd
from cnr_amd import ops, _C
dev = torch.device("cuda:0")
C, L, n_obj = 1, 256, 4
gen = torch.Generator().manual_seed(0)
theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
v = lay.views(theta)
trunk = v["trunk"].contiguous()
packed, lo = ops.pack_weights(trunk), ops.pack_weights_lo(trunk)
B = v["B"].contiguous()
brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
for R, S in ((2048, 64), (8192, 128)):
    pts = torch.rand(C, R, S, 3, device=dev) * 2 - 1
    z = torch.rand(C, R, S, device=dev).sort(-1).values
    ray_row = torch.randint(0, n_obj, (C, R), device=dev).to(torch.int32)
    f = lambda *s: torch.empty(*s, device=dev)
    gt_d, gt_c = f(C, R).zero_(), f(C, R, 3).zero_()
    lab, dm = torch.ones(C, R, device=dev, dtype=torch.uint8), torch.ones(C, R, device=dev, dtype=torch.uint8)
    ds, dc, d1, v1, r1, o1 = f(C, R, S), f(C, R, S, 3), f(C, R), f(C, R), f(C, R, 3), f(C, R)
    ws = torch.zeros(int(_C.load().cnr_field_fwd_render_workspace_bytes(C, R, S)), device=dev, dtype=torch.uint8)
    for name, l in (("f16", None), ("split", lo)):
        fn = lambda: _C.call("cnr_field_fwd_render", pts, B, packed, brows, ray_row, 2.0, z, gt_d, gt_c, lab, dm, 5.0, 10.0,
                             1.0, ds, dc, d1, v1, r1, o1, C, R, S, 0, ws, ws.numel(), l, None, None)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): fn()
        b.record(); torch.cuda.synchronize()
        print(R, S, name, f"{a.elapsed_time(b)/200*1e3:.1f} us", flush=True)
