"""A 20-step timed region with an epoch end inside it (the reshuffle is host-launched: the region's graphs split there):
python tools/region_epoch_end.py [steps_before_the_end]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda:0")
R, S = 2048, 64
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, gen, "cpu")]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, 4, pools, R, dev, seed=1, generator=gen, unroll=20)
tr.run(4)
tr.prepare_graphs()
for _ in range(2):
    for u in tr._group_sizes(tr.unroll):
        tr.run(u)
    tr.run(1)
for _ in range(60):
    tr.run(20)
torch.cuda.synchronize()
res = {"inside": [], "clear": []}
for rep in range(12):
    for name, kk in (("inside", k), ("clear", 40)):
        left = -(-(tr.pool_rows - tr.Rg - tr.cursor) // tr.Rg)
        while left != kk:          # walk to kk steps in front of the epoch end
            tr.run(1)
            left = -(-(tr.pool_rows - tr.Rg - tr.cursor) // tr.Rg)
            if left <= 0:
                tr.run(1)
                left = -(-(tr.pool_rows - tr.Rg - tr.cursor) // tr.Rg)
        tr.run(20 if name == "clear" else 0) if False else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.run(20)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e6)
for name, v in res.items():
    v.sort()
    print("20-step region, epoch end %s: median %.2f us per step (min %.2f, max %.2f)" % (name, v[len(v) // 2], v[0], v[-1]))
