"""GPU: get_uncertainty_fields (src/category_registration.py:58-177), the forward-only consumer of the hot path inside
category registration, as a PRODUCT function (cnr_amd.category_registration) against the oracle's restatement on the same
weights, centres, radii and jitter draws."""
import numpy as np
import pytest
import torch

from conftest import Golden, bg_golden_names, rel_l2
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


class _Cloud:
    def __init__(self, pts):
        self.points = pts


@pytest.mark.parametrize("use_reliability", [True, False])
def test_get_uncertainty_fields_against_the_oracle(dev, use_reliability):
    import cnr_amd as cnr
    from types import SimpleNamespace
    name = [n for n in bg_golden_names() if n.endswith("h32")][0]
    g = Golden(name)
    rng = np.random.default_rng(3)
    tg = torch.Generator().manual_seed(21)
    cls_id, obj_ids = 7, [11, 12, 15]
    inst_dict, pe_dict, fc_dict, states = {cls_id: {}}, {cls_id: {}}, {cls_id: {}}, {}
    for k, oid in enumerate(obj_ids):
        ext = np.array([0.5, 0.4, 0.05]) * (1 + 0.3 * k)
        pts = (rng.random((300, 3)) - 0.5) * ext + np.array([0.2 * k, -0.1, 0.05 * k])
        inst_dict[cls_id][oid] = {"pcs": _Cloud(pts)}
        sd = {n: v + (0.05 * k) * torch.randn(v.shape, generator=tg) * v.abs().mean() for n, v in g.mlp().items()}
        B = g.t("B")[0] + 0.01 * k * torch.randn(21, 3, generator=tg)
        fc = cnr.model.OccupancyMap(87, 42, hidden_size=g.L)
        fc.load_state_dict(sd)
        pe = cnr.embedding.UniDirsEmbed(max_deg=5, scale=g.scale)
        with torch.no_grad():
            pe.B_layer.weight.copy_(B)
        fc_dict[cls_id][oid], pe_dict[cls_id][oid] = fc.to(dev), pe.to(dev)
        states[oid] = (sd, B)
    cfg = SimpleNamespace(data_device=str(dev))
    count = {}
    gen = torch.Generator(device=dev).manual_seed(77)
    cnr.category_registration.get_uncertainty_fields(inst_dict, {}, count, pe_dict, fc_dict, cfg, name="replica",
                                                     use_reliability=use_reliability, generator=gen)
    assert set(count[cls_id].keys()) == set(obj_ids)
    # the oracle on the same draws (the product draws (10000, 96) uniforms per object, in object order)
    gen2 = torch.Generator(device=dev).manual_seed(77)
    ents, metrics = [], []
    for oid in obj_ids:
        u = torch.rand(10000, 96, device=dev, generator=gen2).cpu()
        p = np.asarray(inst_dict[cls_id][oid]["pcs"].points)
        half = torch.from_numpy((np.maximum(p.max(0) - p.min(0), 0.10) / 2).astype(np.float32))
        r = float(1.2 * torch.sqrt(torch.square(half).sum()))
        center = torch.from_numpy(((p.max(0) + p.min(0)) / 2).astype(np.float32))
        sd, B = states[oid]
        term_o, ent_o, metric_o = O.uncertainty_probe(sd, B[None], g.scale, center, r, u)
        # the product's probe on the same z
        z = O.stratified_bins(0.0, 2 * r, 96, 10000, u)
        term, ent, opa = cnr.category_registration.uncertainty_probe(pe_dict[cls_id][oid], fc_dict[cls_id][oid], center, r, dev, z_vals=z)
        assert rel_l2(term, term_o) < 1e-4 and rel_l2(ent, ent_o) < 1e-4 and rel_l2(opa, term_o.sum(-1)) < 1e-4
        ents.append(ent_o)
        metrics.append(metric_o if use_reliability else ent_o)
    thr = 0.5 if use_reliability else 0.8 * min(float(e.max()) for e in ents)
    for oid, m in zip(obj_ids, metrics):
        want = int((m < thr).sum())
        near = int(((m - thr).abs() < 2e-4 * max(1.0, abs(thr))).sum())      # rays within rounding of the threshold
        assert abs(count[cls_id][oid] - want) <= near + 2, (oid, count[cls_id][oid], want, near)
    print("uncertainty counts", count, "threshold", thr)
