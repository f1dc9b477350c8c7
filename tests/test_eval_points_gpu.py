"""GPU: Trainer.eval_points (src/trainer.py:125-151), the forward-only consumer of the hot path (meshing queries up to
256^3 points in 500 k chunks) against the oracle on 10^5 points: CodeNeRF on the fused f16 forward <= 1e-3 (the
north-star bar), the background OccupancyMap on the exact-fp32 kernels <= 2e-5; chunking must not change a value."""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr(dev):
    import cnr_amd
    return cnr_amd


@pytest.mark.parametrize("L", [32, 256])
def test_eval_points_codenerf_against_oracle(cnr, dev, L):
    torch.manual_seed(L)
    cfg = cnr.cfg.synthetic_config(device=str(dev), latent_dim=L)
    ids = [4, 9, 11]
    t = cnr.trainer.Trainer(cfg, 2, ids)
    with torch.no_grad():
        t.pe.B_layer.weight.add_(0.01 * torch.randn(21, 3, device=dev))
    N = 100_000
    pts = torch.rand(N, 3) * 2 - 1
    occ, col = t.eval_points(pts.to(dev), inst_id=9, chunk_size=500000)
    occ2, col2 = t.eval_points(pts.to(dev), inst_id=9, chunk_size=33333)     # ragged chunks: same values
    assert torch.equal(occ, occ2) and torch.equal(col, col2)
    assert occ.shape == (N,) and col.shape == (N, 3)
    mlp = {k: v.detach().cpu()[None] for k, v in t.fc_occ_map.state_dict().items()}
    e = O.unidirs_embed(pts[None, None], t.pe.B_layer.weight.detach().cpu()[None], cfg.obj_scale)   # (1,1,N,129)
    row = t.inst_id_to_index[9]
    cs = t.shape_codes.weight.detach().cpu()[row].view(1, 1, 1, L)
    ct = t.texture_codes.weight.detach().cpu()[row].view(1, 1, 1, L)
    sig, rgb = O.codenerf_forward(mlp, e, cs, ct)
    occ_ref = O.occupancy_activation(sig.reshape(-1))
    e_occ, e_col = rel_l2(occ, occ_ref), rel_l2(col, rgb.reshape(-1, 3))
    print(f"eval_points CodeNeRF L={L}: occ {e_occ:.2e} colour {e_col:.2e}")
    assert e_occ < 1e-3 and e_col < 1e-3
    # another object of the category gives other values (the code row is really used)
    occ3, _ = t.eval_points(pts.to(dev), inst_id=4)
    assert rel_l2(occ3, occ_ref) > 1e-2


@pytest.mark.parametrize("hidden", [32, 128])
def test_eval_points_occupancy_map_against_oracle(cnr, dev, hidden):
    torch.manual_seed(hidden)
    cfg = cnr.cfg.synthetic_config(device=str(dev))
    cfg.hidden_feature_size = hidden
    cfg.obj_scale = 5.0
    t = cnr.trainer.Trainer(cfg, 0, [0])
    N = 100_000
    pts = torch.rand(N, 3) * 6 - 3
    occ, col = t.eval_points(pts.to(dev), chunk_size=40000)
    mlp = {k: v.detach().cpu() for k, v in t.fc_occ_map.state_dict().items()}
    e = O.unidirs_embed(pts[None, None], t.pe.B_layer.weight.detach().cpu()[None], 5.0)[0, 0]
    alpha, color = O.occupancy_map_forward(mlp, e)
    assert rel_l2(occ, O.occupancy_activation(alpha.reshape(-1))) < 2e-5
    assert rel_l2(col, color) < 2e-5


def test_eval_points_background_on_the_fused_forward(cnr, dev):
    """``Trainer.eval_precision = "fused"``: the background's meshing query on the f16-MFMA forward of csrc/bg_fused.hip (cnr_bg_forward
    with nothing kept for a backward) -- occupancy and colour inside north_star's 1e-3 of the oracle, chunking changes no value,
    and the default stays the exact fp32 modules."""
    import time
    torch.manual_seed(128)
    cfg = cnr.cfg.synthetic_config(device=str(dev))
    cfg.hidden_feature_size, cfg.obj_scale = 128, 5.0
    t = cnr.trainer.Trainer(cfg, 0, [0])
    assert t.eval_precision == "fp32"
    N = 200_000
    pts = (torch.rand(N, 3) * 6 - 3)
    exact = t.eval_points(pts.to(dev), chunk_size=50000)
    t.eval_precision = "fused"
    occ, col = t.eval_points(pts.to(dev), chunk_size=50000)
    occ2, col2 = t.eval_points(pts.to(dev), chunk_size=33333)       # ragged chunks (not multiples of the 32-sample tile)
    assert torch.equal(occ, occ2) and torch.equal(col, col2)
    mlp = {k: v.detach().cpu() for k, v in t.fc_occ_map.state_dict().items()}
    e = O.unidirs_embed(pts[None, None], t.pe.B_layer.weight.detach().cpu()[None], 5.0)[0, 0]
    alpha, color = O.occupancy_map_forward(mlp, e)
    e_occ, e_col = rel_l2(occ, O.occupancy_activation(alpha.reshape(-1))), rel_l2(col, color)
    assert e_occ < 1e-3 and e_col < 1e-3, (e_occ, e_col)
    assert rel_l2(occ, exact[0]) < 1e-3
    big = torch.rand(4_000_000, 3, device=dev) * 6 - 3
    times = {}
    for prec in ("fp32", "fused"):
        t.eval_precision = prec
        t.eval_points(big[:100000])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.eval_points(big)
        torch.cuda.synchronize()
        times[prec] = time.perf_counter() - t0
    print(f"eval_points background, 4 M points: fp32 modules {times['fp32'] * 1e3:.1f} ms, fused forward {times['fused'] * 1e3:.1f} ms; "
          f"errors vs oracle: occupancy {e_occ:.1e}, colour {e_col:.1e}")
    assert times["fused"] < times["fp32"]
