"""What an epoch end costs: FusedCategoryTrainer._reshuffle() alone (GPU time between two events, host ahead), and the step time
over whole epochs against the step time inside an epoch.  python tools/exp/time_reshuffle.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cnr_amd
dev = torch.device("cuda:0")
R, S = 2048, 64
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=56)
gen = torch.Generator().manual_seed(1)
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, 4, [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, gen, "cpu")], R, dev, seed=1, generator=gen)
tr.run(200)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    tr._reshuffle()
e1.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    tr._reshuffle()
torch.cuda.synchronize()
print("reshuffle: %.1f us of GPU stream time each, %.1f us wall each (host-bound if larger)" % (e0.elapsed_time(e1) / 50 * 1e3, (time.perf_counter() - t0) / 50 * 1e6))
for n in (63 * 40, 63 * 40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.run(n); torch.cuda.synchronize()
    print("run(%d) over whole epochs: %.2f us per step" % (n, (time.perf_counter() - t0) / n * 1e6))
