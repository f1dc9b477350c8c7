import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cnr_amd
from conftest import Golden, rel_l2
from test_parity_gpu import _build_reference_style_step
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "s1_c1_r64_s16_l256"
g = Golden(name, dev)
trainers, opt, names, fc_param, pe_param, alpha, color, loss, ld = _build_reference_style_step(cnr_amd, g, dev)
loss.backward()
for n, p in zip(names, fc_param):
    ref = g.t("grad." + n)
    print(f"{n:36s} {rel_l2(p.grad, ref):.3e}  |ref|={ref.norm().item():.3e}")
print("B", rel_l2(pe_param[0].grad, g.t("grad_B")))
w = dict(zip(names, fc_param))["encoding_viewdir.0.weight"].grad[0]
ref = g.t("grad.encoding_viewdir.0.weight")[0]
d = (w - ref).abs()
print("viewdir err per column block:", [d[:, a:b].max().item() for a, b in ((0, 32), (32, 64), (64, 74))])
print("viewdir err per row max:", d.max(dim=1).values.tolist())
print("ref row norms:", ref.norm(dim=1).tolist())
