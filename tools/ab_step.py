"""A/B of two library builds on ONE box: CNR_HIP_LIB=<lib> python tools/ab_step.py [R S [C [n_obj]]] prints the one-launch kernel's
back-to-back time and the graph step time (medians of 5 x 400 steps)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_obj = int(sys.argv[4]) if len(sys.argv) > 4 else 4
L = int(sys.argv[5]) if len(sys.argv) > 5 else 256
dev = torch.device("cuda:0")
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=max(S // 8, 1), n_bins=S - max(S // 8, 1))
gen = torch.Generator().manual_seed(1)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, gen, "cpu") for _ in range(C)]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=1, generator=gen)
for _ in range(200):
    tr.step()
torch.cuda.synchronize()
ks = sorted(tr.time_field_train(100) for _ in range(5))
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        tr.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 400)
ts.sort()
print(os.environ.get("CNR_HIP_LIB", "default"), "R", R, "S", S, "C", C, "kernel_us %.2f (min %.2f)" % (ks[2] * 1e3, ks[0] * 1e3),
      "step_us %.2f (min %.2f)" % (ts[2] * 1e6, ts[0] * 1e6), "Mrays/s %.2f" % (C * R / ts[2] / 1e6), "losses", [round(float(v), 5) for v in torch.as_tensor(tr.loss_values()).flatten().tolist()])
