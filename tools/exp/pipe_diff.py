"""Where does a backward variant differ from pipe3?  Per-segment relative differences of one call."""
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
C, R, S, blocks = 1, 2048, 64, 0
variant = sys.argv[1] if len(sys.argv) > 1 else "pipe4"
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
for var in ("pipe3", variant, variant):
    dtrunk = torch.zeros(C, 13892, device=dev); dB = torch.zeros(C, 21, 3, device=dev); dbr = torch.zeros_like(brows)
    ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S, n_obj, blocks, wsp, variant=var)
    torch.cuda.synchronize()
    outs.setdefault(var, []).append((dtrunk.clone(), dB.clone(), dbr.clone()))
ref = outs["pipe3"][0]
for k, got in enumerate(outs[variant]):
    off = 0
    print("call", k)
    for n, o, i in ops.TRUNK_LAYERS:
        for kind, cnt in (("weight", o * i), ("bias", o)):
            a, b = got[0][0, off:off + cnt], ref[0][0, off:off + cnt]
            d = ((a - b).norm() / (b.norm() + 1e-30)).item()
            if d > 1e-4: print(f"  {n}.{kind}: {d:.2e}")
            off += cnt
    print("  dB", ((got[1] - ref[1]).norm() / ref[1].norm()).item(), "dbr", ((got[2] - ref[2]).norm() / ref[2].norm()).item())
a, b = outs[variant]
print("run-to-run:", [float((x - y).abs().max()) for x, y in zip(a, b)])
