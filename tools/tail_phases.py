"""Which block type of tail_kernel finishes last?  Needs tools/libs/libcnr_stamps.so = the library with tail.hip
compiled with -DCNR_TAIL_STAMPS (earliest start / latest end per block type, 100 MHz realtime counter)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import _C
_C.LIB_PATH = os.path.join(ROOT, "tools/libs/libcnr_stamps.so")
dev = torch.device("cuda:0")
C, R, S, L, n_obj = 1, 2048, 64, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=L, obj_scale=2.0, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1234)
pools = [cnr_amd.scene_cateogries.synthetic_pool(64 * R, n_obj, torch.Generator().manual_seed(5), "cpu") for _ in range(C)]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, C, n_obj, pools, R, dev, seed=0, generator=gen, use_graph=False)
lib = _C.load()
lib.cnr_tail_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.cnr_prep_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for it in range(6):
    for _ in range(100):   # keep the card warm (clocks, L2): the stamped step follows ~100 us behind a run of queued steps
        tr.step()
    torch.cuda.synchronize()
    lib.cnr_tail_stamps(None, 1)
    lib.cnr_prep_stamps(None, 1)
    tr.step()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.cnr_tail_stamps(buf, 0)
    t0 = buf[6]
    names = ("latent + AdamW blocks", "record reduce + AdamW blocks", "epilogue block")
    print("step", it, "tail:", " | ".join(f"{n}: first start {(buf[2*i]-t0)/100:.2f} us, last end {(buf[2*i+1]-t0)/100:.2f} us" for i, n in enumerate(names)))
    pb = (ctypes.c_ulonglong * 10)()
    lib.cnr_prep_stamps(pb, 0)
    t0 = pb[8]
    print("       prologue:", " | ".join(f"{n}: {(pb[2*i]-t0)/100:.2f} .. {(pb[2*i+1]-t0)/100:.2f} us" for i, n in enumerate(("pack", "latent fwd", "zero fill", "sample rays"))))

