"""Soak of the multi-step graphs: two trainers from the same seeds, one stepping with step(), one with run(): parameters must
stay bitwise equal over tens of thousands of steps and many epochs (any rare scheduling-dependent fault in the one-launch
kernel or a stale pointer in a captured graph shows up as a divergence here)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cnr_amd as cnr
dev = torch.device("cuda:0")
for (C, R, S, L, n_obj, total, every) in ((1, 2048, 64, 256, 4, 20000, 2000), (2, 480, 10, 32, 5, 6000, 1000),
                                           (1, 8192, 128, 256, 4, 1500, 500), (3, 1000, 32, 64, 9, 3000, 1000)):
    n1 = max(S // 8, 1)
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0, n_bins_cam2surface=n1, n_bins=S - n1)
    trs = []
    for k in range(2):
        pools = [cnr.scene_cateogries.synthetic_pool(16 * R, n_obj, torch.Generator().manual_seed(5 + c), "cpu") for c in range(C)]
        trs.append(cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=0, generator=torch.Generator().manual_seed(1)))
    a, b = trs
    bad = 0
    for it in range(0, total, every):
        for _ in range(every):
            a.step()
        b.run(every)
        torch.cuda.synchronize()
        same = torch.equal(a.theta, b.theta) and torch.equal(a.exp_avg_sq, b.exp_avg_sq) and torch.equal(a.losses, b.losses)
        fin = bool(torch.isfinite(a.losses).all())
        bad += 0 if (same and fin) else 1
        print(f"C{C} {R}x{S} L{L} n_obj{n_obj} step {it + every}: equal {same} finite {fin} losses "
              f"{[round(float(x), 4) for x in a.losses.flatten()][:6]} flags {a.check_flags().tolist()} {b.check_flags().tolist()}", flush=True)
    print(f"C{C} {R}x{S}: bad checkpoints = {bad}", flush=True)
