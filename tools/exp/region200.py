"""How long do 200 steps between two synchronisations take, region after region?  (bench.py's timed region against its long run)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnr_amd
R, S, n_obj = 2048, 64, 4
dev = torch.device("cuda:0")
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1234)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, torch.Generator().manual_seed(1251), "cpu")]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, n_obj, pools, R, dev, seed=0, generator=gen)
tr.run(20); tr.prepare_graphs()
for _ in range(2):
    for u in tr._group_sizes(tr.unroll):
        tr.run(u)
    tr.run(1)
torch.cuda.synchronize()
out = []
for n in [200] * 8 + [63, 63, 126, 1000, 200, 200]:
    torch.cuda.synchronize()
    c0 = tr.cursor // tr.Rg
    t0 = time.perf_counter(); tr.run(n); th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append(f"{n} steps from slice {c0}: {dt / n * 1e6:.2f} us/step (host done after {th * 1e6:.0f} us of {dt * 1e6:.0f})")
print("\n".join(out))
