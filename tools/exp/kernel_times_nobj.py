import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cnr_amd
dev = torch.device("cuda:0")
for n_obj in (4, 7, 12):
    cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=56)
    gen = torch.Generator().manual_seed(1)
    pools = [cnr_amd.scene_cateogries.synthetic_pool(16 * 2048, n_obj, gen, "cpu")]
    tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, n_obj, pools, 2048, dev, seed=1, generator=gen, use_graph=False)
    for _ in range(5): tr.step()
    names = ["cnr_field_train", "cnr_step_prologue", "cnr_step_tail"]
    cnr_amd._C.enable_kernel_timing(names)
    for _ in range(30): tr.step()
    t = cnr_amd._C.kernel_timings_ms()
    print(n_obj, {k: round(sum(v) / len(v) * 1e3, 1) for k, v in t.items() if v})
