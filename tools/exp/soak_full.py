"""Soak of the whole-iteration graphs: two FullStepTrainers from the same seeds, one stepping with step() (one iteration per graph,
a join per iteration), one with run() (up to eight iterations per graph, the background chain and the category chain as two
free-running branches, the next background batch drawn by the previous step's last launch): the background parameters, the
category parameters and both loss vectors must stay bitwise equal over tens of thousands of iterations and every epoch end of
either pool.  python tools/exp/soak_full.py [iterations]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cnr_amd
total = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")


def make(R, S, n_obj):
    torch.manual_seed(5)
    n1 = max(S // 8, 1)
    cfg3 = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=n1, n_bins=S - n1)
    g3 = torch.Generator().manual_seed(77)
    tr3 = cnr_amd.fused.FusedCategoryTrainer(cfg3, 1, n_obj, [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, g3, "cpu")], R, dev,
                                             seed=2, generator=g3)
    cfg_bg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=n1, n_bins=9)
    cfg_bg.n_bins_cam2surface_bg = 5
    bg = cnr_amd.background.BackgroundStep(cfg_bg, cnr_amd.scene_cateogries.synthetic_pool(37 * 1200, 1, g3, "cpu"), 1200, dev, precision="fused")
    return cnr_amd.background.FullStepTrainer(tr3, bg, concurrent=True)


for (R, S, n_obj, every) in ((2048, 64, 4, 2000), (480, 10, 4, 2000)):
    a, b = make(R, S, n_obj), make(R, S, n_obj)
    bad = 0
    for it in range(0, total, every):
        for _ in range(every):
            a.step()
        b.run(every)
        torch.cuda.synchronize()
        same = (torch.equal(a.bg.flat, b.bg.flat) and torch.equal(a.obj.theta, b.obj.theta) and torch.equal(a.bg.losses, b.bg.losses)
                and torch.equal(a.obj.losses, b.obj.losses) and torch.equal(a.bg.d_state, b.bg.d_state))
        fin = bool(torch.isfinite(a.bg.losses).all() and torch.isfinite(a.obj.losses).all())
        bad += 0 if (same and fin) else 1
        print(f"{R}x{S} + 1200x14 iteration {it + every}: equal {same} finite {fin} category losses "
              f"{[round(float(x), 4) for x in a.obj.losses.flatten()]} background {[round(float(x), 4) for x in a.bg.losses]}", flush=True)
    print(f"{R}x{S}: bad checkpoints = {bad}", flush=True)
