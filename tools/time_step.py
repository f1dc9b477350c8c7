import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
dev = torch.device("cuda:0")
R, S, L = 2048, 64, 256
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, gen, "cpu")]
for use_graph in (False, True):
    tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, 4, pools, R, dev, seed=0, generator=gen, use_graph=use_graph)
    for _ in range(10): tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): tr.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph={use_graph}: host issue {1e3*(t1-t0)/200:.3f} ms/step, total {1e3*(t2-t0)/200:.3f} ms/step")
