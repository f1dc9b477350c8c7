"""Step time of the process-group path on ONE GPU (RCCL world of one rank): what the N > 1 step costs before the
collective has any peer to talk to (launch structure: front graph, eager all-reduce, back graph)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import cnr_amd as cnr
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for name, pg in (("single-GPU path", None), ("process-group path", dist.group.WORLD)):
    torch.manual_seed(0)
    C, R, S, L, n_obj = 1, 2048, 64, 256, 4
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
    gen = torch.Generator().manual_seed(1234)
    pools = [cnr.scene_cateogries.synthetic_pool(64 * R, n_obj, torch.Generator().manual_seed(5), "cpu") for _ in range(C)]
    tr = cnr.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=0, generator=gen, process_group=pg, use_graph=True)
    for _ in range(20): tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{name}: {dt * 1e3:.4f} ms/step, losses {tr.losses.flatten().tolist()}", flush=True)
dist.destroy_process_group()
