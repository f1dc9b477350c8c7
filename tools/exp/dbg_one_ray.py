"""one ray, both step forms: where does the gradient differ?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnr_amd as cnr
dev = torch.device("cuda:0")
C, R, n1, n2, n_obj, L = 1, 1, 8, 56, 1, 32
res = {}
for name, one in (("two", False), ("one", True)):
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(5)
    pools = [cnr.scene_cateogries.synthetic_pool(max(4 * R, 8), n_obj, gen, "cpu") for _ in range(C)]
    tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=2, generator=gen, use_graph=False, one_launch=one,
                                        precise_geometry=False)
    print(name, "perm", tr.perm.tolist(), "pool depth", tr.pool["depth"].flatten().tolist(), "state", tr.pool["rgbs"][0, :, 3].tolist())
    tr.step(); torch.cuda.synchronize()
    b = tr.bufs
    print(name, "losses", tr.losses.flatten().tolist(), "flags", tr.flags.tolist(), "depth", b["depth"].tolist(), "var", b["var"].tolist(), "opa", b["opa"].tolist(),
          "label", b["labels"].tolist(), "dmask", b["depth_mask"].tolist(), "gt_depth", b["gt_depth"].tolist())
    res[name] = tr.grad.clone(), tr.lay
g1, lay = res["one"]; g2, _ = res["two"]
v1, v2 = lay.views(g1), lay.views(g2)
for k in v1:
    a, b = v1[k].double().flatten(), v2[k].double().flatten()
    print(k, "norms %.4e %.4e" % (a.norm(), b.norm()), "rel diff %.3e" % ((a - b).norm() / (b.norm() + 1e-30)))
