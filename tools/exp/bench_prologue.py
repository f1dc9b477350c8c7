import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, cnr_amd
from cnr_amd import _C
dev = torch.device("cuda:0")
C, L, n_obj = 1, 256, 4
gen = torch.Generator().manual_seed(0)
theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
zl, br = torch.empty(C * n_obj, 4, 32, device=dev), torch.empty(C * n_obj, 4, 32, device=dev)
pk = torch.empty(C, _C.pack_bytes(), device=dev, dtype=torch.uint8)
zb = torch.empty(60000, device=dev)
args = (lay.total, lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, C)
def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("latent_fwd %.1f us" % t(lambda: _C.call("cnr_latent_fwd", theta, *args, zl, br)))
print("pack       %.1f us" % t(lambda: _C.call("cnr_pack_weights", theta[:, :13892].contiguous() if C > 1 else theta[:, :13892], pk, C)))
print("param_prep %.1f us" % t(lambda: _C.call("cnr_param_prep", theta, lay.total, lay.trunk[0], lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, C, pk, zl, br, zb, zb.numel())))
print("zero only  %.1f us" % t(lambda: zb.zero_()))
