"""CPU: the oracle (oracle/ref_cpu.py) against every committed golden vector of the reference."""
import pytest
import torch

from conftest import Golden, bg_golden_names, golden_names, rel_l2
from oracle import ref_cpu as O


@pytest.mark.parametrize("name", golden_names())
def test_sampling(name):
    g = Golden(name)
    for c in range(g.C):
        T, dirs = g.t("pool_T")[c], g.t("pool_dirs")[c]
        o, d = (O.origin_dirs_W if g.single_obj else O.origin_dirs_O)(T, dirs)
        assert torch.equal(o, g.t("origins")[c]) and torch.equal(d, g.t("dirs_o")[c])
        out = O.sample_3d_points(g.t("pool_rgbs")[c], g.t("pool_depth")[c], o, d, g.t("u")[c], g.t("g")[c],
                                 g.n1, g.n2, g.eps, g.stop_eps)
        assert torch.equal(out[5], g.t("z")[c])
        assert torch.equal(out[4], g.t("pts")[c])
        assert torch.equal(out[2], g.t("depth_mask")[c])
        assert torch.equal(out[3], g.t("labels")[c])


@pytest.mark.parametrize("name", golden_names())
def test_forward_loss_and_grads(name):
    g = Golden(name)
    mlp = {k: v.clone().requires_grad_() for k, v in g.mlp().items()}
    B = g.t("B").clone().requires_grad_()
    sh = [g.t("shape_codes")[c].clone().requires_grad_() for c in range(g.C)]
    tx = [g.t("texture_codes")[c].clone().requires_grad_() for c in range(g.C)]
    batch = dict(pts=g.t("pts"), z=g.t("z"), gt_depth=g.t("gt_depth"), gt_rgb=g.t("gt_rgb"),
                 labels=g.t("labels"), depth_mask=g.t("depth_mask"), indices=g.t("indices"))
    loss, aux = O.forward_loss(mlp, B, g.scale, sh, tx, batch)
    loss.backward()
    if "emb" in g:
        assert rel_l2(aux["emb"], g.t("emb")) < 2e-6
    for k in ("sigmas", "rgbs", "occ", "term", "depth", "var", "rgb", "opacity"):
        assert rel_l2(aux[k], g.t(k)) < 2e-6, k
    for k in ("loss_depth", "loss_color", "loss_opacity", "reg_shape", "reg_texture"):
        assert rel_l2(aux[k], g.t(k)) < 2e-6, k
    assert rel_l2(loss, g.t("loss")) < 2e-6
    for k, p in mlp.items():
        ref = g.t("grad." + k)
        got = torch.zeros_like(ref) if p.grad is None else p.grad
        assert rel_l2(got, ref) < 2e-5, k
    assert rel_l2(B.grad, g.t("grad_B")) < 2e-5
    assert rel_l2(torch.stack([s.grad for s in sh]), g.t("grad_shape_codes")) < 2e-5
    assert rel_l2(torch.stack([s.grad for s in tx]), g.t("grad_texture_codes")) < 2e-5


def test_adamw_step_matches_reference_update():
    """One torch.optim.AdamW step on the fixture's params+grads reproduces the reference's new params."""
    g = Golden("s0_c2_r64_s16_l32")
    names = list(g.mlp().keys())
    ps = [g.t("mlp." + n).clone().requires_grad_() for n in names] + [g.t("B").clone().requires_grad_()]
    gs = [g.t("grad." + n) for n in names] + [g.t("grad_B")]
    opt = torch.optim.AdamW(ps, lr=1e-3, weight_decay=0.013)
    for p, gr in zip(ps, gs):
        p.grad = gr.clone()
    opt.step()
    for p, n in zip(ps, ["new." + n for n in names] + ["new_B"]):
        assert rel_l2(p, g.t(n)) < 1e-6, n


def test_empty_mask_quirk():
    """render_rays.py:67-72: one class with an empty mask zeroes that term for every class."""
    g = Golden("edge_empty_mask")
    assert float(g.t("loss_depth").abs().sum()) == 0.0 and float(g.t("loss_color").abs().sum()) == 0.0
    assert float(g.t("loss_opacity").abs().sum()) > 0.0


def _bg_oracle_step(g):
    mlp = {k: v.clone().requires_grad_() for k, v in g.mlp().items()}
    B = g.t("B").clone().requires_grad_()
    emb = O.unidirs_embed(g.t("pts"), B, g.scale)[0]
    alpha, color = O.occupancy_map_forward(mlp, emb)
    loss, aux, _ = O.step_batch_loss(alpha[None], color[None], g.t("gt_depth"), g.t("gt_rgb"), g.t("labels"),
                                     g.t("depth_mask"), g.t("z"))
    loss.backward()
    return mlp, B, alpha, color, loss, aux


@pytest.mark.parametrize("name", bg_golden_names())
def test_background_branch(name):
    """the background step (world-frame rays, OccupancyMap) of the oracle against the reference's
    vectors: sampling bit-exact, forward <= 2e-6, gradients <= 2e-5."""
    g = Golden(name)
    o, d = O.origin_dirs_W(g.t("pool_T")[0], g.t("pool_dirs")[0])
    out = O.sample_3d_points(g.t("pool_rgbs")[0], g.t("pool_depth")[0], o, d, g.t("u")[0], g.t("g")[0], g.n1, g.n2,
                             g.eps, g.stop_eps)
    assert torch.equal(out[5], g.t("z")[0]) and torch.equal(out[4], g.t("pts")[0])
    assert torch.equal(out[2], g.t("depth_mask")[0]) and torch.equal(out[3], g.t("labels")[0])
    mlp, B, alpha, color, loss, aux = _bg_oracle_step(g)
    assert rel_l2(alpha[None], g.t("sigmas")) < 2e-6 and rel_l2(color[None], g.t("rgbs")) < 2e-6
    assert rel_l2(loss, g.t("loss")) < 2e-6
    for k in ("depth", "color", "opacity"):
        assert rel_l2(aux[k], g.t("loss_" + k)) < 2e-6
    for n, p in mlp.items():
        assert rel_l2(p.grad, g.t("grad." + n)) < 2e-5, n
    assert rel_l2(B.grad, g.t("grad_B")[0]) < 2e-5
