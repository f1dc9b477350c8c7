"""Is the fused background step bitwise repeatable over many steps -- alone, and inside FullStepTrainer (sequential / concurrent,
single-iteration / multi-iteration graphs)?  python tools/exp/bg_determinism.py [steps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
dev = torch.device("cuda:0")


def make(full, conc):
    torch.manual_seed(5)
    R, S = 2048, 64
    cfg3 = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=56)
    g3 = torch.Generator().manual_seed(77)
    tr3 = cnr_amd.fused.FusedCategoryTrainer(cfg3, 1, 4, [cnr_amd.scene_cateogries.synthetic_pool(64 * R, 4, g3, "cpu")], R, dev, seed=2, generator=g3)
    cfg_bg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=9)
    bg = cnr_amd.background.BackgroundStep(cfg_bg, cnr_amd.scene_cateogries.synthetic_pool(64 * 1200, 1, g3, "cpu"), 1200, dev, precision="fused")
    return (cnr_amd.background.FullStepTrainer(tr3, bg, concurrent=conc) if full else None), tr3, bg


def go(mode):
    full, tr, bg = make(mode != "alone", mode.startswith("conc"))
    if mode == "alone":
        for _ in range(n):
            bg.step()
    elif mode.endswith("run"):
        full.run(n)
    else:
        for _ in range(n):
            full.step()
    torch.cuda.synchronize()
    return bg.flat.clone(), tr.theta.clone(), bg.losses.clone()


res = {}
for mode in ("alone", "alone", "seq_step", "seq_step", "conc_step", "conc_step", "conc_run", "conc_run"):
    r = go(mode)
    if mode in res:
        print(mode, "repeat: bg bitwise", torch.equal(r[0], res[mode][0]), "category bitwise", torch.equal(r[1], res[mode][1]),
              "bg losses", r[2].tolist(), res[mode][2].tolist())
    else:
        res[mode] = r
for m in ("seq_step", "conc_step", "conc_run"):
    print(m, "vs alone: bg bitwise", torch.equal(res[m][0], res["alone"][0]), "max diff", float((res[m][0] - res["alone"][0]).abs().max()))
