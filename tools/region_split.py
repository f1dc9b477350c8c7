"""Does a short first graph hide the launch cost of a long one?  20-step regions as one graph of 20 against 2 + 18 and 4 + 16."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd

dev = torch.device("cuda:0")
R, S = 2048, 64
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, gen, "cpu")]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, 4, pools, R, dev, seed=1, generator=gen, unroll=20)
tr.run(4)
for u in (20, 18, 16):
    tr.prepare_graphs(u)
plans = {"20": [20], "2+18": [2, 18], "4+16": [4, 16], "2+2+16": [2, 2, 16]}
for name, plan in plans.items():
    for _ in range(3):
        for u in plan:
            tr.run(u, unroll=u)
torch.cuda.synchronize()
for rep in range(2):
    for name, plan in plans.items():
        ts = []
        for _ in range(25):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for u in plan:
                tr.run(u, unroll=u)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(name, "median %.1f us  min %.1f  max %.1f  -> %.2f us/step" % (ts[12] * 1e6, ts[0] * 1e6, ts[-1] * 1e6, ts[12] / 20 * 1e6))
