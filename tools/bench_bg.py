"""Background branch of one train step: 1200 rays x 14 samples, OccupancyMap(128), eager modules."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
dev = torch.device("cuda:0")
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256)
cfg.n_bins = 9
gen = torch.Generator().manual_seed(0)
pool = cnr_amd.scene_cateogries.synthetic_pool(64 * 1200, 1, gen, "cpu")
sc = cnr_amd.scene_cateogries.sceneCategory.from_pool(cfg, 0, [0], pool)
params = list(sc.trainer.fc_occ_map.parameters()) + list(sc.trainer.pe.parameters())
opt = torch.optim.AdamW(params, lr=cfg.learning_rate, weight_decay=cfg.weight_decay, fused=True)
R = cfg.n_per_optim_bg


def step():
    gt_rgb, gt_depth, dmask, labels, pts, z, _ = sc.get_training_samples(R)
    alpha, color = sc.trainer.fc_occ_map(sc.trainer.pe(pts))
    loss, _, _ = cnr_amd.loss.step_batch_loss(alpha[None], color[None], gt_depth[None], gt_rgb[None] / 255.0, labels[None],
                                              dmask[None], z[None])
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    return loss


for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 50
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"background step {R} rays x {cfg.n_bins_cam2surface_bg + cfg.n_bins} samples, hidden {cfg.hidden_feature_size_bg}: "
      f"{dt*1e3:.3f} ms/step eager ({R/dt/1e6:.2f} M rays/s), loss {float(l):.4f}")
cnr_amd._C.enable_kernel_timing(["cnr_dense_fwd", "cnr_dense_bwd"])
for _ in range(20): step()
tm = cnr_amd._C.kernel_timings_ms()
for k, v in tm.items():
    print(k, f"{sum(v)/len(v)*1e3:.1f} us avg over {len(v)} calls, {sum(v)/20*1e3:.1f} us per step")
