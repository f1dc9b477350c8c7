"""Whole iteration (background + one category) timing: python tools/time_full.py [concurrent 0/1] [R S]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnr_amd
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 64)
dev = torch.device("cuda:0")
torch.manual_seed(0)     # module initialisation and epoch permutations draw from the default generators
n1, n2 = max(S // 8, 1), S - max(S // 8, 1)
cfg3 = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=n1, n_bins=n2)
cfg3.n_bins_cam2surface_bg = 5
g3 = torch.Generator().manual_seed(77)
tr3 = cnr_amd.fused.FusedCategoryTrainer(cfg3, 1, 4, [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, g3, "cpu")], R, dev, seed=2, generator=g3)
cfg_bg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=n1, n_bins=9)
bg = cnr_amd.background.BackgroundStep(cfg_bg, cnr_amd.scene_cateogries.synthetic_pool(64 * 1200, 1, g3, "cpu"), 1200, dev, precision="fused")
full = cnr_amd.background.FullStepTrainer(tr3, bg, concurrent=bool(conc))
for _ in range(10):
    full.step()
full.run(64, unroll=int(os.environ.get('CNR_FULL_UNROLL', '8')))
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    if os.environ.get("CNR_FULL_SINGLE"):
        for _ in range(500):
            full.step()
    else:
        full.run(500, unroll=int(os.environ.get('CNR_FULL_UNROLL', '8')))
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 500)
print(f"concurrent {conc} R {R} S {S}: whole iteration {sorted(ts)[2] * 1e6:.1f} us (min {min(ts) * 1e6:.1f}); category losses {[round(float(v), 4) for v in tr3.losses.flatten()]} bg {[round(float(v), 4) for v in bg.losses]}")
