"""GPU: BASELINE.json configs[4], the field forward on the gfx950 fp8 matrix instruction (cnr_field_fwd_fp8).

Two questions, answered with numbers (also DESIGN.md section 3.4):
  1. does the kernel compute the quantised network it claims to?  -> against a torch restatement of the same quantisation
     (OCP e4m3 planes of 64 W and 16 x, fp32 accumulation): the two agree far better than either agrees with the oracle;
  2. how far is that network from the reference?  -> rendered occupancy / depth / rgb against the fp32 oracle, for 1, 2 and
     3 residual planes.  One plane (fp8 weights and activations on the fp8 MFMA, what configs[4] names) misses north_star's
     1e-3 by ~60x; three planes reach it at nine MFMAs per fragment -- the f16 kernel needs one of the same rate."""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_cpu as O
from test_fused_gpu import latent_rows, trunk_blob

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def _planes(x, scale, terms):
    r, out = x * scale, torch.zeros_like(x)
    for _ in range(terms):
        t = r.to(F8).float()
        out, r = out + t, r - t
    return out / scale


def _emulated(mlp, e, cs, ct, terms):
    wq = lambda w: _planes(w, 64.0, terms)
    xq = lambda x: _planes(x, 16.0, terms)
    p = {k: (wq(v) if k.endswith("weight") and "latent" not in k and k != "sigma.0.weight" else v) for k, v in mlp.items()}
    lin = lambda n, x: torch.matmul(x, p[n + ".weight"].transpose(-1, -2)[:, None]) + p[n + ".bias"][:, None, None, :]
    lat = lambda n, c: torch.relu(torch.matmul(c, mlp[n + ".weight"].transpose(-1, -2)[:, None]) + mlp[n + ".bias"][:, None, None, :])
    fold = lambda n, z, cols=32: torch.matmul(z, mlp[n + ".weight"][:, :, :cols].transpose(-1, -2)[:, None])   # fp32, as the bias rows
    e1, e2 = xq(e[..., :87]), xq(e[..., 87:])
    y = xq(torch.relu(lin("encoding_xyz.0", e1)))
    y = xq(torch.relu(lin("shape_layer_1.0", y) + fold("shape_layer_1.0", lat("shape_latent_layer_1.0", cs))))
    y = xq(torch.relu(lin("cat_layer.0", torch.cat((y, e1), -1)) + fold("cat_layer.0", lat("cat_latent_layer.0", cs))))
    y = xq(torch.relu(lin("shape_layer_2.0", y) + fold("shape_layer_2.0", lat("shape_latent_layer_2.0", cs))))
    y4 = lin("encoding_shape", y)
    sig = lin("sigma.0", y4) * 10.0
    y = xq(torch.relu(lin("encoding_viewdir.0", torch.cat((xq(y4), e2), -1))))
    y = xq(torch.relu(lin("texture_layer_1.0", y) + fold("texture_layer_1.0", lat("texture_latent_layer_1.0", ct))))
    y = xq(torch.relu(lin("rgb.0", y)))
    return sig, torch.sigmoid(lin("rgb.2", y))


@pytest.mark.parametrize("R,S", [(512, 64), (8192, 128)], ids=["512x64", "configs4_8192x128"])
@pytest.mark.parametrize("terms", [1, 2, 3])
def test_fp8_forward_is_the_quantised_network_and_how_far_that_is_from_the_reference(dev, terms, R, S):
    """(8192 x 128 is BASELINE.json configs[4]'s own shape: the whole batch against the oracle, 1 M samples.)"""
    import cnr_amd as cnr
    C, L, n_obj = 1, 256, 4
    gen = torch.Generator().manual_seed(4321)
    mlp = O.init_codenerf_params(C, 32, L, gen)
    B = torch.tensor(O.UNIDIRS).view(21, 3).repeat(C, 1, 1) + 0.01 * torch.randn(C, 21, 3, generator=gen)
    pts = torch.rand(C, R, S, 3, generator=gen) * 2 - 1
    z = torch.sort(torch.rand(C, R, S, generator=gen) * 3 + 0.5, dim=-1).values
    shape = torch.randn(C, n_obj, L, generator=gen) / (L / 2) ** 0.5
    tex = torch.randn(C, n_obj, L, generator=gen) / (L / 2) ** 0.5
    idx = torch.randint(0, n_obj, (C, R), generator=gen)
    cs = torch.stack([shape[c][idx[c]][:, None] for c in range(C)])
    ct = torch.stack([tex[c][idx[c]][:, None] for c in range(C)])
    e = O.unidirs_embed(pts, B, 2.0)
    sig_ref, rgb_ref = O.codenerf_forward(mlp, e, cs, ct)
    ref = O.composite(sig_ref.squeeze(-1), rgb_ref, z)
    sig_e, rgb_e = _emulated(mlp, e, cs, ct, terms)
    emu = O.composite(sig_e.squeeze(-1), rgb_e, z)
    d = lambda t: t.to(dev)
    mlp_d = {k: d(v) for k, v in mlp.items()}
    trunk = trunk_blob(cnr, mlp_d)
    packed = cnr.ops.pack_weights(trunk)
    brows = cnr.ops.bias_rows(trunk, latent_rows(cnr, mlp_d, d(shape), d(tex))).reshape(C * n_obj, 4, 32)
    ray_row = (d(idx) + torch.arange(C, device=dev)[:, None] * n_obj).to(torch.int32).contiguous()
    sig, rgb = cnr.ops.field_fwd_fp8(d(pts), d(B), trunk, packed, brows, ray_row, 2.0, terms=terms)
    got = O.composite(sig.cpu(), rgb.cpu(), z)
    names = ("occ", "term", "depth", "var", "rgb", "opacity")
    err_ref = {n: rel_l2(got[i], ref[i]) for i, n in enumerate(names) if n in ("occ", "depth", "rgb")}
    err_emu = {n: rel_l2(got[i], emu[i]) for i, n in enumerate(names) if n in ("occ", "depth", "rgb")}
    emu_ref = {n: rel_l2(emu[i], ref[i]) for i, n in enumerate(names) if n in ("occ", "depth", "rgb")}
    print(f"fp8 forward {R}x{S}, {terms} plane(s): kernel vs oracle {err_ref} | kernel vs torch emulation {err_emu} | emulation vs oracle {emu_ref}")
    for n in err_ref:
        # (1) the kernel IS that quantised network: closer to its restatement than the restatement is to the reference (a
        # hardware sine that differs in the last bit moves an fp8 rounding now and then: 6 % of one feature)
        assert err_emu[n] < 0.6 * emu_ref[n] + 1e-4, (n, err_emu, emu_ref)
        assert 0.4 * emu_ref[n] < err_ref[n] < 2.5 * emu_ref[n] + 1e-4, (n, err_ref, emu_ref)
    # (2) how far from the reference: one plane misses the 1e-3 bar by more than an order of magnitude; three reach it
    if terms == 1:
        assert err_ref["occ"] > 2e-2
    if terms == 3:
        assert max(err_ref.values()) < 1e-3
