"""Launch the fused kernels a few times for rocprofv3 --kernel-trace / --pmc: `prof_one.py R S train` = ten eager steps of the
trainer (cnr_step_prologue | cnr_field_train | cnr_step_tail); `prof_one.py R S [blocks]` = the stand-alone forward and backward."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import ops, _C
dev = torch.device("cuda:0")
if len(sys.argv) > 3 and sys.argv[3] == "train":
    # the trainer's own step (param_prep | field forward + composite + backward in one launch | tail), eager, ten times
    R, S = int(sys.argv[1]), int(sys.argv[2])
    cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
    gen = torch.Generator().manual_seed(0)
    pools = [cnr_amd.scene_cateogries.synthetic_pool(16 * R, 4, gen, "cpu")]
    tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, 4, pools, R, dev, seed=0, generator=gen, use_graph=False)
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    sys.exit(0)
L, n_obj = 256, 4
C, R, S = 1, int(sys.argv[1]), int(sys.argv[2])
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
gen = torch.Generator().manual_seed(0)
theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
v = lay.views(theta)
packed = ops.pack_weights(v["trunk"].contiguous())
pts = (torch.rand(C, R, S, 3, device=dev) * 2 - 1)
B = v["B"].contiguous()
brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
ray_row = (torch.randint(0, n_obj, (C, R), device=dev) + torch.arange(C, device=dev)[:, None] * n_obj).to(torch.int32)
dsig = torch.randn(C, R, S, device=dev) * 1e-3
drgb = torch.randn(C, R, S, 3, device=dev) * 1e-3
dtrunk = torch.zeros(C, 13892, device=dev); dB = torch.zeros(C, 21, 3, device=dev); dbr = torch.zeros_like(brows)
wsp = torch.empty(_C.field_bwd_workspace_bytes(C, 0), device=dev, dtype=torch.uint8)
sig = torch.empty(C, R, S, device=dev); rgb = torch.empty(C, R, S, 3, device=dev)
for _ in range(10):
    _C.call("cnr_field_fwd", pts, B, packed, brows, ray_row, 2.0, sig, rgb, C, R, S, 0, None)
    ops.field_bwd(pts, B, packed, brows, ray_row, 2.0, dsig, drgb, 2048.0, dtrunk, dB, dbr, C, R, S, n_obj, blocks, wsp)
torch.cuda.synchronize()
