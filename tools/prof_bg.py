"""The background step alone (1200 rays x 14 samples, OccupancyMap(128)), 60 captured steps, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=9)     # as bench.py's extra leg
gen = torch.Generator().manual_seed(3)
bg = cnr_amd.background.BackgroundStep(cfg, cnr_amd.scene_cateogries.synthetic_pool(64 * 1200, 1, gen, "cpu"), 1200, dev,
                                       precision=prec)
for _ in range(60):
    bg.step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200):
    bg.step()
torch.cuda.synchronize()
print("bg step ms", (time.perf_counter() - t0) / 200 * 1e3)
