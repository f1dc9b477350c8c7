import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cnr_amd
from conftest import Golden, rel_l2
import test_fused_gpu as T
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "s0_c1_r120_s10_l256"
gs = float(sys.argv[2]) if len(sys.argv) > 2 else 1024.0
g = Golden(name, dev)
out = T._fused_step(cnr_amd, g, dev, grad_scale=gs)
P, B, shape, tex, sig, rgb = T._emulated_f16_step(cnr_amd, g, dev)
loss = T._torch_loss(sig, rgb, g)
loss.backward()
print("loss", float(out["loss"]), float(loss))
off = 0
for n, o, i in cnr_amd.ops.TRUNK_LAYERS:
    for kind, cnt, shp in (("weight", o * i, (g.C, o, i)), ("bias", o, (g.C, o))):
        got = out["trunk"].grad[:, off:off + cnt].reshape(shp)
        ref = P[n + "." + kind].grad
        ref = torch.zeros_like(got) if ref is None else ref
        print(f"{n+'.'+kind:32s} {rel_l2(got, ref):.3e} |ref| {ref.norm().item():.3e}")
        off += cnt
print("B", rel_l2(out["B"].grad, B.grad))
