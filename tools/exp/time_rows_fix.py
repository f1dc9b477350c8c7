"""cnr_field_train back to back with and without the fixed-point row table (its int64 atomics), per object count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import _C
dev = torch.device("cuda:0")
for n_obj in (4, 7, 12):
    R, S = 2048, 64
    cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=8, n_bins=56)
    gen = torch.Generator().manual_seed(1)
    pools = [cnr_amd.scene_cateogries.synthetic_pool(16 * R, n_obj, gen, "cpu")]
    tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, n_obj, pools, R, dev, seed=1, generator=gen)
    for _ in range(6):
        tr.step()
    torch.cuda.synchronize()
    o, b, lay, C = tr.bufs, tr.bufs, tr.lay, tr.C
    Bc = tr.theta[0, lay.B[0]:lay.B[1]]
    clamp = torch.zeros_like(tr.clamp); st = tr.d_state2[tr.parity]
    for name, fix in (("with table", torch.zeros_like(tr.rows_fix)), ("without", None)):
        args = tr._field_train_args(b, o, Bc, st, 1.0, fix, clamp)
        run = lambda: _C.call_struct("cnr_field_train", **args)
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): run()
        e1.record(); torch.cuda.synchronize()
        print("n_obj", n_obj, name, "%.2f us" % (e0.elapsed_time(e1) * 10))
