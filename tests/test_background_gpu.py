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
