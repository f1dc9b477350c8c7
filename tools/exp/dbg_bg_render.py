"""cnr_bg_backward_render against cnr_render_loss + cnr_bg_backward on a reference fixture: where do they differ?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cnr_amd as cnr
from conftest import Golden
from test_bg_fused_gpu import _flat
_C = cnr._C
lib = _C.load()
dev = torch.device("cuda:0")
g = Golden("bg_r240_s14_h128", dev)
R, S = g.R, g.S
M = R * S
theta = _flat(g, "mlp.").contiguous()
f = lambda *sh, dt=torch.float32: torch.zeros(*sh, device=dev, dtype=dt)
packed = f(int(lib.cnr_bg_pack_bytes()), dt=torch.uint8)
sigma, rgbs = f(1, R, S), f(1, R, S, 3)
act, eimg, dpre = f(5, M, 128, dt=torch.float16), f(M, 144, dt=torch.float16), f(5, M, 128, dt=torch.float16)
nblk = int(lib.cnr_bg_blocks(M))
records = f(nblk, int(lib.cnr_bg_record_floats()))
pts = g.t("pts").contiguous()
_C.call("cnr_bg_pack", theta, packed)
_C.call("cnr_bg_forward", pts, theta, packed, g.scale, M, sigma, rgbs, act, eimg)
gscale = 256.0
ws = torch.zeros(_C.render_loss_workspace_bytes(1, R), device=dev, dtype=torch.uint8)
dsig, drgb = f(1, R, S), f(1, R, S, 3)
depth, var, rgb, opa = f(1, R), f(1, R), f(1, R, 3), f(1, R)
args = (g.t("z").contiguous(), g.t("gt_depth").contiguous(), g.t("gt_rgb").contiguous(), g.t("labels").contiguous(),
        g.t("depth_mask").to(torch.uint8).contiguous())
_C.call("cnr_render_loss", sigma, rgbs, *args, 5.0, 10.0, gscale, dsig, drgb, depth, var, rgb, opa, 1, R, S, ws, ws.numel(), None, None)
_C.call("cnr_bg_backward", pts, theta, packed, g.scale, M, dsig, drgb, rgbs, act, dpre, records, None, 0)
pool_rgbs = torch.zeros(1, R, 4, device=dev, dtype=torch.uint8)
pool_rgbs[0, :, 3] = g.t("labels").reshape(-1).to(torch.uint8)
pool_depth = torch.where(g.t("depth_mask").reshape(1, R).bool(), torch.ones(1, R, device=dev), torch.zeros(1, R, device=dev))
tab = f(1, 2, 4)
_C.call("cnr_slice_maskcounts", pool_rgbs, pool_depth, None, R, 1, R, 1, 0.0, tab)
print("tab", tab.flatten().tolist(), "hdr", ws.view(torch.float32)[-4:].tolist())
# the two-call form WITH the table
dsig_t, drgb_t = f(1, R, S), f(1, R, S, 3)
_C.call("cnr_render_loss", sigma, rgbs, *args, 5.0, 10.0, gscale, dsig_t, drgb_t, depth, var, rgb, opa, 1, R, S, ws, ws.numel(), tab, None)
print("render_loss table vs counted: dsig equal", torch.equal(dsig_t, dsig), "drgb equal", torch.equal(drgb_t, drgb))
ws2 = torch.zeros(int(lib.cnr_bg_backward_render_workspace_bytes(M)), device=dev, dtype=torch.uint8)
dpre2, records2 = torch.zeros_like(dpre), torch.zeros_like(records)
depth2, var2, rgb2, opa2 = f(1, R), f(1, R), f(1, R, 3), f(1, R)
dsig2, drgb2 = f(1, R, S), f(1, R, S, 3)
_C.call_struct("cnr_bg_backward_render", pts=pts, theta=theta, packed=packed, scale=g.scale, R=R, S=S, sigma=sigma, rgb=rgbs,
               z=args[0], gt_depth=args[1], gt_rgb=args[2], labels=args[3], depth_mask=args[4], counts_tab=tab, d_state=None,
               color_scaling=5.0, opacity_scaling=10.0, grad_scale=gscale, act=act, dpre=dpre2, records=records2, depth=depth2,
               var=var2, rgb_render=rgb2, opacity=opa2, d_sigma=dsig2, d_rgb=drgb2, loss_workspace=ws2, loss_workspace_bytes=ws2.numel())
torch.cuda.synchronize()
print("dsig equal", torch.equal(dsig2, dsig), float((dsig2-dsig).abs().max()), "drgb equal", torch.equal(drgb2, drgb), float((drgb2-drgb).abs().max()))
for m in (1276, 1314, 1583):
    r = rgbs.view(-1, 3)[m]; d = drgb.view(-1, 3)[m]
    v1 = (d * r) * (1 - r); v2 = torch.addcmul(d * r, -(d * r), r)
    print(m, "drgb", d.tolist(), "rgb", r.tolist(), "plain f16", v1.half().tolist(), "fma-form f16", v2.half().tolist(), "dsig", float(dsig.view(-1)[m]), float(dsig2.view(-1)[m]))
print("renders equal", torch.equal(depth2, depth), torch.equal(var2, var), torch.equal(rgb2, rgb), torch.equal(opa2, opa))
for l in range(5):
    d = (dpre2[l].float() - dpre[l].float()).abs()
    bad = (d > 0).any(1).nonzero().flatten()
    print("layer", l, "max diff", float(d.max()), "ref max", float(dpre[l].float().abs().max()), "rows differing", bad.numel(), bad[:12].tolist())
d = (records2 - records).abs()
print("records max diff", float(d.max()), "of", float(records.abs().max()))
# which of the two is the f16 chain's own arithmetic?  dPre5 = (a5 > 0) * (OC @ f16(W_oc)), OC = f16(d rgb * rgb (1 - rgb))
Woc = g.t("mlp.out_color.weight").half().float()          # (3, 128)
r3, d3 = rgbs.view(-1, 3), drgb.view(-1, 3)
v = (d3 * r3) * (1 - r3)
oc_rne = v.half().float()
emu = (oc_rne @ Woc) * (act[4].float() > 0)
for m in (1276, 1314, 1583):
    a_, b_, e_ = dpre[4][m].float(), dpre2[4][m].float(), emu[m].half().float()
    idx = (a_ != b_).nonzero().flatten()[:6]
    print(m, "cols", idx.tolist(), "two-call", a_[idx].tolist(), "one-call", b_[idx].tolist(), "emulated", e_[idx].tolist(), "fp32", emu[m][idx].tolist())
