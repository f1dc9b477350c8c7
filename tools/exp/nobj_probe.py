import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, cnr_amd as cnr
dev = torch.device("cuda:0")
for n_obj, C, R, n1, n2, L in ((6, 2, 256, 8, 56, 64), (1, 1, 128, 4, 28, 32), (5, 1, 200, 3, 13, 256)):
    torch.manual_seed(0)
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2)
    gen = torch.Generator().manual_seed(1)
    pools = [cnr.scene_cateogries.synthetic_pool(8 * R, n_obj, gen, "cpu") for _ in range(C)]
    res = []
    for graph in (False, False, True):
        torch.manual_seed(0)
        tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=1, generator=torch.Generator().manual_seed(2), use_graph=graph)
        for _ in range(8): tr.step()
        torch.cuda.synchronize()
        res.append((tr.losses.clone(), tr.theta.clone()))
    print(n_obj, C, R, "losses", res[0][0].flatten().tolist(), "eager==eager", torch.equal(res[0][1], res[1][1]), "graph==eager", torch.equal(res[0][1], res[2][1]),
          "max rel diff", float(((res[0][1] - res[2][1]).abs().max() / res[0][1].abs().max())),
          "fused_tail", tr.fused_tail, flush=True)
