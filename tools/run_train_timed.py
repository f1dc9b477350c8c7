"""Cycle stamps of the one-launch step body (cnr_field_train) inside a real trainer step; build the stamped library with
tools/build_timed_lib.sh and run with CNR_HIP_LIB pointing at it.  Prints, per wave of workgroup 0 (its last iteration):
work between barriers and wait at barriers."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cnr_amd

R, S = int(sys.argv[1]), int(sys.argv[2])
one = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256, n_bins_cam2surface=S // 8, n_bins=S - S // 8)
gen = torch.Generator().manual_seed(1)
NOBJ = int(sys.argv[4]) if len(sys.argv) > 4 else 4
pools = [cnr_amd.scene_cateogries.synthetic_pool(8 * R, NOBJ, gen, "cpu")]
tr = cnr_amd.fused.FusedCategoryTrainer(cfg, 1, NOBJ, pools, R, dev, seed=1, generator=gen, use_graph=False, one_launch=bool(one))
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
lib = cnr_amd._C.load()
buf = (ctypes.c_longlong * (8 * 64))()
lib.cnr_pipe8_read_stamps.argtypes = [ctypes.c_void_p]
lib.cnr_pipe8_read_stamps(buf)
b = list(buf)
# barriers per iteration: [chain only: the previous iteration's "images consumed"], [composite exchange when a ray spans waves],
# one per layer step (9), [dW: "images consumed"]
nb = 11 if (one and S > 32) else 10
base = min(b[w * 64] for w in range(8))
for w in range(8):
    st = b[w * 64: w * 64 + 2 * nb + 1]
    work = [st[1] - st[0]] + [st[2 * k + 1] - st[2 * k] for k in range(1, nb)]
    wait = [st[2 * k + 2] - st[2 * k + 1] for k in range(nb)]
    print("wave", w, "chain" if w < 4 else "dW", "t0", st[0] - base, "total", st[2 * nb] - st[0], "work", sum(work), "wait", sum(wait))
    print("    work", work)
    print("    wait", wait)
    mk = b[w * 64 + 44: w * 64 + 56]
    if w < 4 and one:
        x1 = st[4]      # after the exchange barrier (the iteration's second)
        print("    forward (cycles, the stamps themselves wait for outstanding LDS reads): top barrier -> encoding in registers", mk[8] - st[2],
              "| -> xyz + cat[e1] products issued", mk[9] - mk[8], "| -> four hidden layers", mk[10] - mk[9], "| -> sigma head + sigma scans", mk[11] - mk[10],
              "| -> colour branch", mk[7] - mk[11])
        print("    marks: fwd end->exch barrier", st[3] - mk[7], "| exch->sums", mk[0] - x1, "| var", mk[1] - mk[0], "| loss scalars", mk[2] - mk[1],
              "| suffix carry", mk[3] - mk[2], "| scan+docc", mk[4] - mk[3], "| dsg..DWS", mk[5] - mk[4], "| R2 stage+mfma", mk[6] - mk[5])
    its = [v for v in b[w * 64 + 30: w * 64 + 42] if v]
    print("    iteration starts (from loop start)", [v - b[w * 64 + 57] for v in its], "lengths", [its[i + 1] - its[i] for i in range(len(its) - 1)])
    ph = b[w * 64 + 56: w * 64 + 64]
    print("    behind the loop (from kernel start): loop left", ph[3] - ph[0], "| +", ph[4] - ph[3], "(chain: last barrier; dW: record blocks) | +", ph[6] - ph[4],
          "(chain: last tile's PE backward; dW: rgb.2 + row-sum block) | +", ph[7] - ph[6], "(partial sums) | barrier +", ph[2] - ph[7])
    print("    phases: weights in LDS", ph[1] - ph[0], "| loops", ph[2] - ph[1], "| flush", ph[5] - ph[2], "| total", ph[5] - ph[0])
