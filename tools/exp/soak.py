"""Soak: many captured steps across many epochs (losses finite, no flags), and many repeats of the backward call at the
benchmark size (bitwise identical every time) -- a rare scheduling-dependent fault shows up here, not in short tests."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cnr_amd as cnr
from cnr_amd import ops, _C
dev = torch.device("cuda:0")
torch.manual_seed(0)
C, R, S, L, n_obj = 1, 2048, 64, 256, 4
cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
pools = [cnr.scene_cateogries.synthetic_pool(16 * R, n_obj, torch.Generator().manual_seed(5), "cpu") for _ in range(C)]
tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=0, generator=torch.Generator().manual_seed(1), use_graph=True)
bad = 0
for it in range(20000):
    tr.step()
    if it % 500 == 499:
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(tr.losses).all()) and int(tr.flags.abs().sum()) == 0
        bad += 0 if ok else 1
        print("step", it + 1, "losses", [round(float(x), 4) for x in tr.losses.flatten()], "flags", tr.flags.tolist(), flush=True)
print("soak steps: bad checkpoints =", bad)
# ---- backward repeatability
theta, lay = cnr.fused.init_params(C, L, n_obj, torch.Generator().manual_seed(3), dev)
v = lay.views(theta)
packed = ops.pack_weights(v["trunk"].contiguous()); B = v["B"].contiguous()
pts = torch.rand(C, R, S, 3, device=dev) * 2 - 1
brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
ray_row = (torch.randint(0, n_obj, (C, R), device=dev)).to(torch.int32)
dsig, drgb = torch.randn(C, R, S, device=dev) * 1e-3, torch.randn(C, R, S, 3, device=dev) * 1e-3
wsp = torch.empty(_C.field_bwd_workspace_bytes(C, 0, C * n_obj), device=dev, dtype=torch.uint8)
def run():
    dtrunk = torch.zeros(C, 13892, device=dev); dB = torch.zeros(C, 21, 3, device=dev); dbr = torch.zeros_like(brows)
    ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S, n_obj, 0, wsp)
    return torch.cat([dtrunk.flatten(), dB.flatten(), dbr.flatten()])
ref = run(); diff = 0
for i in range(2000):
    if not torch.equal(run(), ref): diff += 1
torch.cuda.synchronize()
print("backward repeats differing from the first:", diff, "of 2000")
