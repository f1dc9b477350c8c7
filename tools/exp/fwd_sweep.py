import sys, os
sys.path.insert(0, "/root/repo")
import torch, cnr_amd
from cnr_amd import ops, _C
dev = torch.device("cuda:0")
C, L, n_obj = 1, 256, 4
gen = torch.Generator().manual_seed(0)
theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
v = lay.views(theta)
packed = ops.pack_weights(v["trunk"].contiguous()); B = v["B"].contiguous()
brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
for R, S in ((2048, 64), (8192, 128)):
    pts = torch.rand(C, R, S, 3, device=dev) * 2 - 1
    ray_row = torch.randint(0, n_obj, (C, R), device=dev).to(torch.int32)
    sig = torch.empty(C, R, S, device=dev); rgb = torch.empty(C, R, S, 3, device=dev)
    fn = lambda: _C.call("cnr_field_fwd", pts, B, packed, brows, ray_row, 2.0, sig, rgb, C, R, S, 0, None)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): fn()
    b.record(); torch.cuda.synchronize()
    print(os.environ.get("CNR_FWD_BLOCKS"), R, S, f"{a.elapsed_time(b)/200*1e3:.1f} us", flush=True)
