"""GPU: the benchmarked FusedCategoryTrainer step at BASELINE sizes against the oracle, tensor by tensor, and direct
checks of the latent backward / record reduction / fixed-point row sums that the step's last launch performs.

Bars (also in DESIGN.md section 3.3):
  * losses of the step: 2e-3 of the oracle's (f16 forward);
  * every gradient tensor against autograd of a torch restatement of the SAME f16 pipeline (same ReLU masks up to fp32
    summation order): 1e-2 per tensor, 1e-3 on the whole trunk (measured at configs[1]: 4.4e-3 worst tensor, 1e-4
    trunk) -- this is the bar that proves the kernel computes the gradient of what it evaluates, at full size;
  * every gradient tensor against the oracle's fp32 autograd: 0.06 relative L2 per tensor, cosine of the whole
    gradient > 0.9995 (measured at configs[1]: 0.032 worst tensor, cosine 0.99992) -- the distance between an f16 and
    an fp32 forward (flipped ReLU units), not a kernel error: the convergence test (tests/test_convergence_gpu.py) shows
    it is harmless to training;
  * AdamW: exp_avg = 0.1 g and exp_avg_sq = 0.001 g^2 of the kernel's own gradient to 1e-6, the parameter update
    against the oracle's AdamW where the gradient is not rounding noise.
"""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_cpu as O
from test_fused_gpu import _emulated_f16_step, _torch_loss
from test_trainer_gpu import _oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr(dev):
    import cnr_amd
    return cnr_amd


class _Batch:
    """The fields test_fused_gpu's torch restatements read from a Golden, served from a trainer's sampled batch."""

    def __init__(self, cnr, tr, theta0, b, idx, dev):
        self.C, self.n_obj, self.scale = tr.C, tr.n_obj, tr.scale
        v = tr.lay.views(theta0.to(dev))
        self._t = dict(pts=b["pts"], z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"], labels=b["labels"],
                       depth_mask=b["depth_mask"].bool(), indices=idx.to(dev), B=v["B"], shape_codes=v["shape"],
                       texture_codes=v["tex"])
        mlp, off = {}, 0
        for n, o, i in cnr.ops.TRUNK_LAYERS:
            mlp[n + ".weight"] = v["trunk"][:, off:off + o * i].reshape(tr.C, o, i); off += o * i
            mlp[n + ".bias"] = v["trunk"][:, off:off + o]; off += o
        for k, n in enumerate(cnr.ops.LATENT_LAYERS):
            mlp[n + ".weight"], mlp[n + ".bias"] = v["latW"][:, k], v["latb"][:, k]
        self._mlp = mlp

    def t(self, k):
        return self._t[k]

    def mlp(self):
        return {k: v.clone() for k, v in self._mlp.items()}


def _grad_tensors(cnr, tr, flat):
    """flat (C,P) gradient-like buffer -> {reference tensor name: tensor}"""
    v = tr.lay.views(flat)
    out, off = {}, 0
    for n, o, i in cnr.ops.TRUNK_LAYERS:
        out[n + ".weight"] = v["trunk"][:, off:off + o * i].reshape(tr.C, o, i); off += o * i
        out[n + ".bias"] = v["trunk"][:, off:off + o]; off += o
    for k, n in enumerate(cnr.ops.LATENT_LAYERS):
        out[n + ".weight"], out[n + ".bias"] = v["latW"][:, k], v["latb"][:, k]
    out["B"], out["shape_codes"], out["texture_codes"] = v["B"], v["shape"], v["tex"]
    return out


@pytest.mark.parametrize("C,R,n1,n2,L", [(1, 2048, 8, 56, 256), (2, 4096, 16, 112, 32), (1, 8192, 16, 112, 256)])
def test_full_size_train_step_against_oracle(cnr, dev, C, R, n1, n2, L):
    """configs[1] (1 x 2048 x 64, L = 256), the ScanNet shape (2 x 4096 x 128, L = 32) and configs[4]'s shape (8192 x 128, L = 256;
    f16 operands -- there is no fp8 train step): ONE step of the benchmarked trainer; losses, every gradient tensor and the AdamW
    update against the oracle on the batch the kernels sampled."""
    torch.manual_seed(4321)
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=n1, n_bins=n2,
                                   obj_scale=2.0 if L == 256 else 3.0)
    gen = torch.Generator().manual_seed(77)
    pools = [cnr.scene_cateogries.synthetic_pool(4 * R, 4, gen, "cpu") for _ in range(C)]
    tr = cnr.fused.FusedCategoryTrainer(cfg, C, 4, pools, R, dev, seed=5, generator=gen, use_graph=False)
    theta0 = tr.theta.clone()
    rows = tr.perm[:, :R].long().cpu()
    tr.step()
    torch.cuda.synchronize()
    bd = {k: v for k, v in tr.bufs.items() if torch.is_tensor(v)}
    b = {k: v.cpu() for k, v in bd.items()}
    idx = torch.stack([pools[c]["indices"][rows[c]] for c in range(C)])
    assert torch.equal((b["ray_row"].long() - torch.arange(C)[:, None] * 4), idx)
    got = {k: v.cpu() for k, v in _grad_tensors(cnr, tr, tr.grad).items()}

    # ---- oracle (fp32, CPU) on exactly that batch ---------------------------------------------------------------------
    mlp, B, shape, tex = _oracle_params(cnr, tr, theta0)
    mlp = {k: v.requires_grad_() for k, v in mlp.items()}
    B.requires_grad_()
    sh = [shape[c].clone().requires_grad_() for c in range(C)]
    tx = [tex[c].clone().requires_grad_() for c in range(C)]
    batch = dict(pts=b["pts"], z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"], labels=b["labels"],
                 depth_mask=b["depth_mask"].bool(), indices=idx)
    loss, aux = O.forward_loss(mlp, B, cfg.obj_scale, sh, tx, batch)
    loss.backward()
    losses = tr.losses.cpu()
    for k, name in enumerate(("loss_depth", "loss_color", "loss_opacity")):
        assert rel_l2(losses[k], aux[name]) < 2e-3, (name, rel_l2(losses[k], aux[name]))
    for k, key in (("depth", "depth"), ("rgb", "rgb"), ("opa", "opacity")):
        assert rel_l2(b[k], aux[key]) < 1e-3, (k, rel_l2(b[k], aux[key]))       # the north-star quantities
    ref = {k: (torch.zeros_like(v) if v.grad is None else v.grad) for k, v in mlp.items()}
    ref["B"], ref["shape_codes"], ref["texture_codes"] = B.grad, torch.stack([s.grad for s in sh]), torch.stack([s.grad for s in tx])
    worst, dot, na, nb = 0.0, 0.0, 0.0, 0.0
    report = []
    for k in got:
        if float(ref[k].abs().sum()) == 0.0:
            assert float(got[k].abs().sum()) == 0.0, k
            continue
        e = rel_l2(got[k], ref[k])
        report.append(f"{k}={e:.3f}")
        worst = max(worst, e)
        assert e < 0.06, (k, e)
        a_, b_ = got[k].double().reshape(-1), ref[k].double().reshape(-1)
        dot += float(a_ @ b_); na += float(a_ @ a_); nb += float(b_ @ b_)
    cos = dot / (na * nb) ** 0.5
    print(f"full-size step C{C} R{R} S{n1 + n2} L{L} vs fp32 oracle: cos={cos:.5f} worst={worst:.3f}  " + " ".join(report))
    assert cos > 0.9995

    # ---- the plain-f16 pipeline restated in torch (device): the kernel must return ITS gradient.  The restatement rounds
    # every operand to f16, so it is compared with the trainer's plain-f16 form (precise_geometry=False; same seeds, same
    # batch, same backward) -- the default's geometry forward is deliberately NOT that arithmetic
    torch.manual_seed(4321)
    gen2 = torch.Generator().manual_seed(77)
    pools2 = [cnr.scene_cateogries.synthetic_pool(4 * R, 4, gen2, "cpu") for _ in range(C)]
    tr_p = cnr.fused.FusedCategoryTrainer(cfg, C, 4, pools2, R, dev, seed=5, generator=gen2, use_graph=False,
                                          precise_geometry=False)
    assert torch.equal(tr_p.theta, theta0)
    tr_p.step()
    torch.cuda.synchronize()
    assert torch.equal(tr_p.bufs["pts"], bd["pts"]) and torch.equal(tr_p.bufs["z"], bd["z"])
    got_default = got
    got = {k: v.cpu() for k, v in _grad_tensors(cnr, tr_p, tr_p.grad).items()}
    g = _Batch(cnr, tr, theta0, bd, idx, dev)
    P, Be, she, txe, sig, rgb = _emulated_f16_step(cnr, g, dev)
    le = _torch_loss(sig, rgb, g) + 0.0005 * sum(torch.norm(she[c], dim=-1).sum() + torch.norm(txe[c], dim=-1).sum()
                                                for c in range(C))
    le.backward()
    emu = {k: (torch.zeros_like(v) if v.grad is None else v.grad).cpu() for k, v in P.items()}
    emu["B"], emu["shape_codes"], emu["texture_codes"] = Be.grad.cpu(), she.grad.cpu(), txe.grad.cpu()
    num = den = 0.0
    rep2 = []
    trunk_names = {n + s for n, _, _ in cnr.ops.TRUNK_LAYERS for s in (".weight", ".bias")}
    for k in got:
        if float(emu[k].abs().sum()) == 0.0:
            continue
        e = rel_l2(got[k], emu[k])
        rep2.append(f"{k}={e:.4f}")
        assert e < 1e-2, (k, e)
        if k in trunk_names:
            num += float((got[k] - emu[k]).double().pow(2).sum()); den += float(emu[k].double().pow(2).sum())
    print(f"   vs emulated f16 pipeline: trunk={(num / den) ** 0.5:.4f}  " + " ".join(rep2))
    assert (num / den) ** 0.5 < 1e-3
    got = got_default

    # ---- AdamW (step 1): moments of the kernel's own gradient, update against the oracle's optimiser ------------------
    g_flat = tr.grad.cpu()
    assert rel_l2(tr.exp_avg.cpu(), 0.1 * g_flat) < 1e-6
    assert rel_l2(tr.exp_avg_sq.cpu(), 0.001 * g_flat * g_flat) < 1e-4    # (1 - beta2 is not exactly 1e-3 in fp32)
    params = list(mlp.values()) + [B] + sh + tx
    opt = torch.optim.AdamW(params, lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    opt.step()
    new = _grad_tensors(cnr, tr, tr.theta.cpu())
    old = _grad_tensors(cnr, tr, theta0.cpu())
    ref_new = dict(mlp)
    ref_new["B"], ref_new["shape_codes"], ref_new["texture_codes"] = B, torch.stack(list(sh)), torch.stack(list(tx))
    agree, total = 0, 0
    for k in got:
        gr = ref[k].reshape(-1).abs()
        if float(gr.max()) == 0.0:
            continue
        clear = gr > 0.02 * gr.max()      # first AdamW step = -lr * g / (|g| + eps): compare where g is not noise
        du_ref = (ref_new[k].detach() - old[k]).reshape(-1)[clear]
        du_got = (new[k] - old[k]).reshape(-1)[clear]
        agree += int((torch.sign(du_ref) == torch.sign(du_got)).sum()); total += int(clear.sum())
        assert rel_l2(new[k], ref_new[k].detach()) < 5e-3, k   # parameters after the step (each entry moved by +-lr)
    assert agree / total > 0.98, agree / total


@pytest.mark.parametrize("C,n_obj,L", [(1, 4, 256), (2, 3, 32), (1, 7, 64), (1, 1, 32), (2, 12, 32)])
def test_latent_bwd_against_autograd(cnr, dev, C, n_obj, L):
    """cnr_latent_fwd / cnr_latent_bwd (a7 + the four latent layers + the code regulariser, src/model.py:38-51,
    src/loss.py:5-15) against torch autograd on given bias-row gradients: <= 1e-5; trunk entries are ADDED."""
    _C = cnr._C
    gen = torch.Generator().manual_seed(100 * C + n_obj)
    theta, lay = cnr.fused.init_params(C, L, n_obj, gen, dev)
    args = (lay.total, lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, C)
    zl, br = torch.empty(C * n_obj, 4, 32, device=dev), torch.empty(C * n_obj, 4, 32, device=dev)
    _C.call("cnr_latent_fwd", theta, *args, zl, br)
    dbr = torch.randn(C * n_obj, 4, 32, generator=gen).to(dev)
    g0 = torch.randn(C, lay.total, generator=gen).to(dev)          # what the field backward left in the buffer
    grad = g0.clone()
    reg = 0.0005
    _C.call("cnr_latent_bwd", theta, *args, zl, dbr, reg, grad, None)
    th = theta.clone().double().requires_grad_()
    v = lay.views(th)
    zs = []
    for k in range(4):
        code = v["tex"] if k == 3 else v["shape"]
        zs.append(torch.relu(torch.einsum("col,cnl->cno", v["latW"][:, k], code) + v["latb"][:, k][:, None, :]))
    zref = torch.stack(zs, dim=2)                                   # (C, n_obj, 4, 32)
    rows = cnr.ops.bias_rows(v["trunk"], zref)
    assert rel_l2(zl, zref.reshape(C * n_obj, 4, 32)) < 1e-6 and rel_l2(br, rows.reshape(C * n_obj, 4, 32)) < 1e-6
    obj = (rows.reshape(C * n_obj, 4, 32) * dbr.double()).sum()
    if n_obj > 1:
        obj = obj + reg * (torch.norm(v["shape"], dim=-1).sum() + torch.norm(v["tex"], dim=-1).sum())
    obj.backward()
    want = th.grad
    gv, wv, g0v = lay.views(grad), lay.views(want), lay.views(g0)
    for k in ("latW", "latb", "shape", "tex"):
        assert rel_l2(gv[k], wv[k]) < 1e-5, (k, rel_l2(gv[k], wv[k]))
    assert rel_l2(gv["trunk"] - g0v["trunk"], wv["trunk"]) < 1e-5
    assert torch.equal(gv["B"], g0v["B"])                           # not touched


@pytest.mark.parametrize("C,n_obj,L,nwg", [(1, 4, 256, 256), (2, 3, 32, 37), (1, 7, 64, 64), (2, 15, 32, 50)])
def test_step_grad_and_tail_reduce_records_and_fixed_point_rows(cnr, dev, C, n_obj, L, nwg):
    """The gradient half of the step's last launch on synthetic inputs: per-workgroup records (fixed-order float sums),
    the int64 2^-40 fixed-point table of the per-object bias-row sums (8 copies), latent backward.  cnr_step_grad
    against a float64 torch evaluation <= 1e-5; cnr_step_tail(records) leaves the same gradient bit for bit and its
    AdamW equals torch.optim.AdamW on that gradient."""
    _C = cnr._C
    gen = torch.Generator().manual_seed(nwg)
    theta, lay = cnr.fused.init_params(C, L, n_obj, gen, dev)
    args = (lay.total, lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L, n_obj, C)
    zl, br = torch.empty(C * n_obj, 4, 32, device=dev), torch.empty(C * n_obj, 4, 32, device=dev)
    _C.call("cnr_latent_fwd", theta, *args, zl, br)
    rec_floats = _C.field_bwd_workspace_bytes(1, 1) // 2            # entries of one record (bf16 each since round 4)
    TR = 13892
    recs = (torch.randn(C, nwg, rec_floats, generator=gen) * 1e-2).to(torch.bfloat16)
    ws = torch.zeros(_C.field_bwd_workspace_bytes(C, nwg) // 2, device=dev, dtype=torch.bfloat16)
    ws[: recs.numel()] = recs.reshape(-1).to(dev)
    recs = recs.float()                                              # (the expectation below sums exactly what the records hold)
    rows = torch.randn(C * n_obj, 4, 32, generator=gen, dtype=torch.float64) * 0.3
    parts = torch.rand(8, *rows.shape, generator=gen, dtype=torch.float64)
    parts = parts / parts.sum(0, keepdim=True) * rows                # eight addends per entry
    fix = torch.round(parts * 2.0 ** 40).to(torch.int64).to(dev).contiguous()
    rows_q = (fix.sum(0).double() * 2.0 ** -40)                      # what the table holds exactly
    reg = 0.0005

    grad = torch.full((C, lay.total), 7.0, device=dev)               # every entry must be overwritten
    dbr_out = torch.empty(C * n_obj, 4, 32, device=dev)
    _C.call("cnr_step_grad", theta, grad, lay.total, lay.B[0], lay.latW[0], lay.latb[0], lay.shape[0], lay.tex[0], L,
            n_obj, C, zl, dbr_out, reg, ws, nwg, fix, None)
    assert rel_l2(dbr_out, rows_q) < 1e-6
    # ---- float64 expectation
    th = theta.clone().double().requires_grad_()
    v = lay.views(th)
    zs = []
    for k in range(4):
        code = v["tex"] if k == 3 else v["shape"]
        zs.append(torch.relu(torch.einsum("col,cnl->cno", v["latW"][:, k], code) + v["latb"][:, k][:, None, :]))
    rws = cnr.ops.bias_rows(v["trunk"], torch.stack(zs, dim=2)).reshape(C * n_obj, 4, 32)
    obj = (rws * rows_q).sum()
    if n_obj > 1:
        obj = obj + reg * (torch.norm(v["shape"], dim=-1).sum() + torch.norm(v["tex"], dim=-1).sum())
    obj.backward()
    want = th.grad.clone()
    rs = recs.double().sum(1).to(want.device)                        # (C, rec_floats)
    wv = lay.views(want)
    trunk_rec = rs[:, :TR].clone()
    for off in (3840, 8736, 4896, 13281):                            # biases of the latent-conditioned layers: rows only
        trunk_rec[:, off:off + 32] = 0.0
    wv["trunk"] += trunk_rec
    wv["B"] += (rs[:, TR:TR + 63] + rs[:, TR + 63:TR + 126]).reshape(C, 21, 3)
    gv = lay.views(grad)
    for k in ("trunk", "B", "latW", "latb", "shape", "tex"):
        assert rel_l2(gv[k], wv[k]) < 1e-5, (k, rel_l2(gv[k], wv[k]))

    # ---- the one-launch tail: same gradient bits, AdamW on it
    th2 = torch.stack([theta, theta.clone()])
    grad2 = torch.full((C, lay.total), 3.0, device=dev)
    m, vv = torch.zeros_like(theta), torch.zeros_like(theta)
    state = torch.zeros(2, 3, device=dev, dtype=torch.int64)
    R = 64
    rl_ws = torch.zeros(_C.render_loss_workspace_bytes(C, R), device=dev, dtype=torch.uint8)
    losses, flags = torch.zeros(3, C, device=dev), torch.zeros(C, device=dev, dtype=torch.int32)
    _C.call_struct("cnr_step_tail", theta_in=th2[0], theta_out=th2[1], grad=grad2, exp_avg=m, exp_avg_sq=vv,
                   class_stride=lay.total, off_B=lay.B[0], off_latW=lay.latW[0], off_latb=lay.latb[0], off_shape=lay.shape[0],
                   off_tex=lay.tex[0], L=L, n_obj=n_obj, C=C, zl=zl, dbiasrows=torch.empty_like(dbr_out), reg_scale=reg,
                   do_latent=1, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.013, state_cur=state[0],
                   state_next=state[1], add_rows=R, rl_workspace=rl_ws, losses=losses, flags=flags, depth=None,
                   pool_rows=8 * R, perm=None, next_max_bound=None, R=R, records=ws, nwg=nwg, rows_fix=fix, rl_blocks=0,
                   clamp_flags=None, n_obj_cls=None, code_lr=0.0, code_weight_decay=0.0)
    assert torch.equal(grad2, grad)
    p = theta.clone().requires_grad_()
    p.grad = grad.clone()
    torch.optim.AdamW([p], lr=1e-3, weight_decay=0.013).step()
    assert rel_l2(th2[1], p.detach()) < 1e-6
    assert state[1].tolist() == [R, 1, 1]
