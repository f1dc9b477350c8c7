"""Step / kernel time of the fused trainer at one shape: python tools/quick_step.py [R S n_obj precise]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnr_amd
R, S, n_obj, precise = (int(v) for v in (sys.argv[1:5] + ["2048", "64", "4", "1"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1234)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, torch.Generator().manual_seed(1251), "cpu")]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, n_obj, pools, R, dev, seed=0, generator=gen, precise_geometry=bool(precise))
tr.run(20); tr.prepare_graphs(); tr.run(34); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 4000; tr.run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
k = tr.time_field_train(100) if tr._ft_blocks else float("nan")
print(f"R {R} S {S} n_obj {n_obj} precise {precise}: step {dt / n * 1e6:.2f} us = {R * n / dt / 1e6:.2f} M rays/s; cnr_field_train {k * 1e3:.2f} us")
