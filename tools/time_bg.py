"""Background step timing: python tools/time_bg.py [precision] -> per-step time (graph replay) and per-kernel HIP-event times"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnr_amd
prec = sys.argv[1] if len(sys.argv) > 1 else "fused"
dev = torch.device("cuda:0")
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=9)
cfg.n_bins_cam2surface_bg = 5
g3 = torch.Generator().manual_seed(77)
bg = cnr_amd.background.BackgroundStep(cfg, cnr_amd.scene_cateogries.synthetic_pool(64 * 1200, 1, g3, "cpu"), 1200, dev, precision=prec)
for _ in range(10):
    bg.step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 500
for _ in range(n):
    bg.step()
torch.cuda.synchronize()
print(prec, "background step %.1f us" % ((time.perf_counter() - t0) / n * 1e6), "losses", [round(float(v), 4) for v in bg.losses])
if prec == "fused":
    names = ["cnr_sample_maxdepth", "cnr_sample_rays", "cnr_bg_pack", "cnr_bg_forward", "cnr_render_loss", "cnr_render_loss_finish",
             "cnr_bg_backward", "cnr_bg_backward_render", "cnr_bg_dw", "cnr_bg_tail", "cnr_bg_tail_sample"]
    cnr_amd._C.enable_kernel_timing(names)
    for _ in range(30):
        bg.step(use_graph=False)
    tm = cnr_amd._C.kernel_timings_ms()
    print({k: round(sum(v) / max(len(v), 1) * 1e3, 1) for k, v in tm.items()})
