"""GPU: BASELINE.json configs[2] at the size the SCALE run uses -- 16 categories x 4 objects, 2048 rays x 64 samples per
category and step ("all categories" stands for C = 16) -- on ONE GPU, one step of the benchmarked
trainer against the oracle:

  * every category's renders (depth / rgb / opacity per ray) and its three loss terms against the oracle's forward on the batch
    the kernels sampled (all 16 x 2048 x 64 = 2.1 M samples);
  * the complete gradient (trunk, latent layers, B, both code tables) and the AdamW update of TWO categories against the
    oracle's backward; categories share nothing (train.py:58-64), so a category's gradient is its own loss's;
  * properties of all 16: finite, no explode / clamp flag, every category's gradient non-zero, categories differ, the step is
    bitwise repeatable, and 8-category shards (what two GPUs would hold) reproduce their halves of the 16-category step bit
    for bit -- the class-shard claim of DESIGN.md section 5 at the benchmark's own shape."""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_cpu as O
from test_trainer_gpu import _oracle_params

pytestmark = pytest.mark.gpu
C, N_OBJ, R, N1, N2, L = 16, 4, 2048, 8, 56, 256
CHECK_BACKWARD = (3, 12)


def _trainer(cnr, dev, ids):
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L, n_bins_cam2surface=N1, n_bins=N2)
    pools = [cnr.scene_cateogries.synthetic_pool(4 * R, N_OBJ, torch.Generator().manual_seed(1234 + 17 * (c + 1)), "cpu")
             for c in ids]
    kw = {} if len(ids) == C else dict(shard="class", n_cls_global=C, class_ids=list(ids), dp_rank=ids[0] % 2, dp_world=2)
    # every trainer gives a class the same number of workgroups: the fixed-order record sum is then the same sum
    tr = cnr.fused.FusedCategoryTrainer(cfg, len(ids), N_OBJ, pools, R, dev, seed=0, generator=torch.Generator().manual_seed(1234),
                                        use_graph=False, bwd_blocks=16, **kw)
    return cfg, tr


def test_sixteen_categories_one_step_against_the_oracle(dev):
    import cnr_amd as cnr
    cfg, tr = _trainer(cnr, dev, list(range(C)))
    theta0 = tr.theta.clone()
    tr.step()
    torch.cuda.synchronize()
    fl = tr.check_flags()
    assert int(fl.max()) == 0, fl.tolist()                       # no explode, no clamp, no empty mask
    b = {k: v.cpu() for k, v in tr.bufs.items() if torch.is_tensor(v)}
    idx = b["ray_row"].long() - torch.arange(C)[:, None] * N_OBJ
    assert int(idx.min()) >= 0 and int(idx.max()) < N_OBJ
    batch = dict(pts=b["pts"], z=b["z"], gt_depth=b["gt_depth"], gt_rgb=b["gt_rgb"], labels=b["labels"],
                 depth_mask=b["depth_mask"].bool(), indices=idx)
    mlp, B, shape, tex = _oracle_params(cnr, tr, theta0)
    # ---- forward of ALL categories ---------------------------------------------------------------------------------------
    with torch.no_grad():
        _, aux = O.forward_loss(mlp, B, cfg.obj_scale, [shape[c] for c in range(C)], [tex[c] for c in range(C)], batch)
    for c in range(C):
        for k, key in (("depth", "depth"), ("rgb", "rgb"), ("opa", "opacity")):
            e = rel_l2(b[k][c], aux[key][c])
            assert e < 1e-3, (c, k, e)
    losses = tr.losses.cpu()
    for k, name in enumerate(("loss_depth", "loss_color", "loss_opacity")):
        assert torch.allclose(losses[k], aux[name], rtol=2e-3, atol=1e-6), (name, losses[k], aux[name])
    # ---- backward + AdamW of two categories --------------------------------------------------------------------------------
    grad = tr.grad.cpu()
    theta1 = tr.theta.cpu()
    for c in CHECK_BACKWARD:
        th = theta0[c:c + 1].cpu().clone().requires_grad_()
        v = tr.lay.views(th)
        m1, off = {}, 0
        for n, o, i in cnr.ops.TRUNK_LAYERS:
            m1[n + ".weight"] = v["trunk"][:, off:off + o * i].reshape(1, o, i); off += o * i
            m1[n + ".bias"] = v["trunk"][:, off:off + o]; off += o
        for k, n in enumerate(cnr.ops.LATENT_LAYERS):
            m1[n + ".weight"], m1[n + ".bias"] = v["latW"][:, k], v["latb"][:, k]
        loss, _ = O.forward_loss(m1, v["B"], cfg.obj_scale, [v["shape"][0]], [v["tex"][0]], {k: t[c:c + 1] for k, t in batch.items()})
        loss.backward()
        g_ref, g = th.grad[0], grad[c]
        cos = float((g.double() @ g_ref.double()) / (g.double().norm() * g_ref.double().norm()))
        print(f"category {c}: gradient cosine vs oracle {cos:.6f}, rel {rel_l2(g, g_ref):.4f}")
        assert cos > 0.9995 and rel_l2(g, g_ref) < 0.03, (c, cos)
        p = theta0[c].cpu().clone().requires_grad_()
        p.grad = g_ref.clone()
        torch.optim.AdamW([p], lr=cfg.learning_rate, weight_decay=cfg.weight_decay).step()
        clear = g_ref.abs() > 0.02 * g_ref.abs().max()         # first AdamW step = lr * sign(g): compare where g is not noise
        assert rel_l2((theta1[c] - theta0[c].cpu())[clear], (p.detach() - theta0[c].cpu())[clear]) < 0.02
    # ---- properties of all sixteen -----------------------------------------------------------------------------------------
    assert torch.isfinite(grad).all() and torch.isfinite(theta1).all() and torch.isfinite(losses).all()
    gn = grad.norm(dim=1)
    assert float(gn.min()) > 0 and float(gn.max() / gn.min()) < 50, gn.tolist()
    assert len({float(v) for v in losses[1]}) == C             # sixteen different categories, not one repeated
    # bitwise repeatable, and 8-category shards reproduce their halves bit for bit (no gradient collective between categories)
    _, tr_b = _trainer(cnr, dev, list(range(C)))
    tr_b.step()
    torch.cuda.synchronize()
    assert torch.equal(tr_b.theta, tr.theta) and torch.equal(tr_b.grad, tr.grad)
    for ids in (list(range(0, C, 2)), list(range(1, C, 2))):
        _, tr_s = _trainer(cnr, dev, ids)
        tr_s.step()
        torch.cuda.synchronize()
        assert torch.equal(tr_s.theta, tr.theta[ids]), ids
        assert torch.equal(tr_s.losses, tr.losses[:, ids]), ids
