"""Fixed cost of a timed region: python tools/region_overhead.py [spin] -- the bench's own 20-step region (sync | run(20) | sync)
repeated, against the 200-step region; `spin` sets hipDeviceScheduleSpin before the first HIP call."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spin = len(sys.argv) > 1 and sys.argv[1] == "spin"
if spin:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
import torch
import cnr_amd

dev = torch.device("cuda:0")
R, S = 2048, 64
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, gen, "cpu")]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, 4, pools, R, dev, seed=1, generator=gen, unroll=20)
tr.run(4)
tr.prepare_graphs()
for _ in range(3):
    tr.run(20)
torch.cuda.synchronize()


def region(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.run(n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for n in (20, 200):
    ts = sorted(region(n) for _ in range(15))
    print("spin" if spin else "default", "steps", n, "median region %.1f us = %.2f us/step (min %.2f)" % (ts[7] * 1e6, ts[7] / n * 1e6, ts[0] / n * 1e6))
t = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    t.append(time.perf_counter() - t0)
print("empty synchronize: %.1f us" % (sorted(t)[25] * 1e6))
