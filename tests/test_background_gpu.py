"""GPU: the background branch as a captured step, and the whole iteration of train.py:113-184 (background + categories) as
one hipGraph.  (The background's arithmetic itself is pinned to the reference's bg_*.npz vectors by
tests/test_parity_gpu.py::test_background_step_against_reference.)"""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr(dev):
    import cnr_amd
    return cnr_amd


def _bg(cnr, dev, R=300, hidden=128):
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=9)
    cfg.hidden_feature_size_bg = hidden
    torch.manual_seed(3)                                            # module initialisation
    pool = cnr.scene_cateogries.synthetic_pool(6 * R, 1, torch.Generator().manual_seed(12), "cpu")
    return cnr.background.BackgroundStep(cfg, pool, R, dev, seed=5)


def _flat(bg):
    return torch.cat([p.detach().reshape(-1) for p in bg.params])


def test_background_graph_replay_equals_eager_and_follows_the_epochs(cnr, dev):
    """Twelve steps over a pool of six slices (two reshuffles): three eager steps then nine replays of ONE captured graph
    against twelve eager steps -- same samples (device cursor, in-kernel Philox), parameters equal to fp32 rounding (the PE
    backward adds dB with float atomics), losses finite and the cursor where the host thinks it is."""
    eager, graph = _bg(cnr, dev), _bg(cnr, dev)
    assert torch.equal(_flat(eager), _flat(graph))
    for it in range(12):
        eager.step(use_graph=False)
        graph.step(use_graph=True)
        torch.cuda.synchronize()
        assert torch.equal(eager.bufs["z"], graph.bufs["z"]) and torch.equal(eager.bufs["labels"], graph.bufs["labels"]), it
        assert rel_l2(graph.losses, eager.losses) < 1e-3, (it, graph.losses, eager.losses)
    assert graph.graph is not None and torch.isfinite(graph.losses).all()
    assert rel_l2(_flat(graph), _flat(eager)) < 1e-4
    assert int(graph.d_state[0]) == graph.cursor and int(graph.d_state[2]) == 12
    assert float((_flat(graph) - _flat(_bg(cnr, dev))).abs().max()) > 1e-4     # it did train


def test_whole_iteration_in_one_graph_equals_the_two_steps_run_separately(cnr, dev):
    """FullStepTrainer: background + fused category step captured together, per state parity, against the two trainers
    stepped on their own: the category parameters bit for bit (the two branches share no parameter and no buffer), the
    background to fp32 rounding."""
    def make():
        cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=32, n_bins_cam2surface=4, n_bins=28)
        gen = torch.Generator().manual_seed(21)
        pools = [cnr.scene_cateogries.synthetic_pool(6 * 256, 4, gen, "cpu") for _ in range(2)]
        tr = cnr.fused.FusedCategoryTrainer(cfg, 2, 4, pools, 256, dev, seed=1, generator=gen)
        return tr, _bg(cnr, dev)
    tr_a, bg_a = make()
    tr_b, bg_b = make()
    full = cnr.background.FullStepTrainer(tr_a, bg_a)
    for it in range(11):
        full.step()
        tr_b.step()
        bg_b.step()
    torch.cuda.synchronize()
    assert len(full.graphs) == 2
    assert torch.equal(tr_a.theta, tr_b.theta) and torch.equal(tr_a.losses, tr_b.losses)
    assert rel_l2(_flat(bg_a), _flat(bg_b)) < 1e-4 and rel_l2(bg_a.losses, bg_b.losses) < 1e-3
    assert tr_a.steps_done == 11 and bg_a.steps_done == 11


@pytest.mark.parametrize("M,K,N,relu", [(16800, 215, 128, True), (1000, 87, 128, True), (513, 128, 3, False)])
def test_f16_dense_mode_against_torch_emulation(cnr, dev, M, K, N, relu):
    """cnr_dense_fwd / cnr_dense_bwd with f16_operands: both operands rounded to f16, fp32 accumulation -- against torch on
    f16-rounded operands in fp64 (forward 1e-5: same products, fp32 summation order), and against the exact layer at the f16
    level; the gradient operand of dx carries a 2^10 loss scale so that small gradients do not fall into f16 subnormals; the
    weight gradient is formed from the fp32 operands in both tiers."""
    gen = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(N, K, generator=gen) * 0.2).to(dev).requires_grad_()
    b = (torch.randn(N, generator=gen) * 0.1).to(dev).requires_grad_()
    xg = x.clone().requires_grad_()
    y = cnr.ops.DenseFn.apply(xg, W, b, relu, True)
    q = lambda t: t.half().double()
    pre = q(x) @ q(W.detach()).T + b.detach().double()
    want = torch.relu(pre) if relu else pre
    assert rel_l2(y, want) < 1e-5
    exact = torch.nn.functional.linear(x, W.detach(), b.detach())
    assert rel_l2(y, torch.relu(exact) if relu else exact) < 2e-3
    dy = (torch.randn(M, N, generator=gen) * 1e-4).to(dev)            # small upstream gradients, like 1 / rays
    y.backward(dy)
    dpre = dy.double() * ((pre > 0).double() if relu else 1.0)
    s = cnr.ops.DENSE_GRAD_SCALE
    dq = (dpre * s).half().double() / s
    assert rel_l2(xg.grad, dq @ q(W.detach())) < 1e-4
    # dW / db: fp32 operands straight from memory in both tiers (dense_dw_kernel) -> the exact gradient of the masked dy
    assert rel_l2(W.grad, dpre.T @ x.double()) < 1e-5 and rel_l2(b.grad, dpre.sum(0)) < 1e-5


def test_background_f16_tier_tracks_the_fp32_tier(cnr, dev):
    """BackgroundStep(precision="f16"): same samples, same losses to the f16 level over the first steps, and faster."""
    import time
    a, b = _bg(cnr, dev, R=1200), _bg(cnr, dev, R=1200)
    b.trainer.fc_occ_map.half = True
    for it in range(4):
        a.step(use_graph=False)
        b.step(use_graph=False)
        torch.cuda.synchronize()
        assert torch.equal(a.bufs["z"], b.bufs["z"])
        assert rel_l2(b.losses, a.losses) < (2e-3 if it == 0 else 2e-2), (it, a.losses, b.losses)
    out = {}
    for name, bg in (("fp32", a), ("f16", b)):
        for _ in range(6):
            bg.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            bg.step()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 50 * 1e3
    print(f"background step 1200 x 13, hidden 128, captured: fp32 tier {out['fp32']:.3f} ms, f16 tier {out['f16']:.3f} ms")
    assert out["f16"] < out["fp32"]
