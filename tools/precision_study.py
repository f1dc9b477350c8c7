"""Offline study (CPU, torch): which operand split of the f16 forward meets north_star's 1e-3 on TRAINED weights.

Trains the oracle (fp32) on the analytic scene of tests/scene_synth.py, then evaluates the geometry branch of
CodeNeRF.forward (src/model.py:56-75) with emulated operand rounding -- f16 weights / activations / PE features, each
either plain (hi) or split (hi + lo, the product's cross terms as separate MFMAs) -- against fp32.  Accumulation is fp32 in
every variant, the latent rows are folded into an fp32 bias as in the kernels.  Prints relative L2 of occupancy,
depth, opacity per variant.  Not part of the product or the tests: a design tool."""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_cpu as O          # noqa: E402
from scene_synth import analytic_pool    # noqa: E402


def q16(x):
    return x.half().float()


def split(x, terms):
    """x -> list of f16-representable pieces whose sum approximates x (terms = 1: plain rounding)."""
    out, r = [], x
    for _ in range(terms):
        p = q16(r)
        out.append(p)
        r = r - p
    return out


def mm(xs, ws, cross):
    """sum of products of operand pieces; cross = which (i, j) piece pairs are computed (an MFMA each)."""
    acc = 0
    for (i, j) in cross:
        if i < len(xs) and j < len(ws):
            acc = acc + torch.matmul(xs[i], ws[j].transpose(-1, -2))
    return acc


def geometry_forward(p, e1, cs, scheme):
    """scheme: dict(w=terms, x=terms, pe=terms, cross=list of (xi, wj))  -> sigma logits (R,S)"""
    lat = lambda n: torch.relu(torch.matmul(cs, p[n + ".weight"].transpose(-1, -2)) + p[n + ".bias"])   # (R,1,32) fp32
    def layer(n, x_pieces, extra_in=None):
        W, b = p[n + ".weight"], p[n + ".bias"]
        return W, b
    W = lambda n: p[n + ".weight"]
    b = lambda n: p[n + ".bias"]
    cross = scheme["cross"]
    pe = split(e1, scheme["pe"])
    only = scheme.get("layers")
    def lin(n, xs, z=None, cols=None):
        w = W(n) if cols is None else W(n)[:, cols[0]:cols[1]]
        if only is not None and n not in only:
            return mm(xs[:1], split(w, 1), [(0, 0)])
        ws = split(w, scheme["w"])
        if xs is pe and "pe_cross" in scheme:
            return mm(xs, ws, scheme["pe_cross"])
        return mm(xs, ws, cross)
    # encoding_xyz
    y = torch.relu(lin("encoding_xyz.0", pe) + b("encoding_xyz.0"))
    z1, zc, z2 = lat("shape_latent_layer_1.0"), lat("cat_latent_layer.0"), lat("shape_latent_layer_2.0")
    fold = lambda n, z, cols=None: torch.matmul(z, (W(n) if cols is None else W(n)[:, cols[0]:cols[1]]).transpose(-1, -2)) + b(n)
    y = torch.relu(lin("shape_layer_1.0", split(y, scheme["x"])) + fold("shape_layer_1.0", z1))
    cat_e1 = lin("cat_layer.0", pe, cols=(32, 119)) if scheme.get("cat_e1", True) else \
        mm(pe[:1], split(W("cat_layer.0")[:, 32:119], 1), [(0, 0)])
    y = torch.relu(lin("cat_layer.0", split(y, scheme["x"]), cols=(0, 32)) + cat_e1
                   + fold("cat_layer.0", zc, cols=(0, 32)))
    y = torch.relu(lin("shape_layer_2.0", split(y, scheme["x"])) + fold("shape_layer_2.0", z2))
    y = lin("encoding_shape", split(y, scheme["x"])) + b("encoding_shape")
    return (torch.matmul(y, W("sigma.0").transpose(-1, -2)) + b("sigma.0")).squeeze(-1) * 10.0    # fp32 VALU head


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--n1", type=int, default=4)
    ap.add_argument("--n2", type=int, default=28)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--load", default=None, help="a .pt written by tests/test_trained_parity_gpu.py (theta + batch) instead of training")
    ap.add_argument("--save", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    n_obj, R, n1, n2, L = 4, args.rays, args.n1, args.n2, args.latent
    S = n1 + n2
    if args.load:
        d = torch.load(args.load)
        mlp, B, sh, tx, batch = d["mlp"], d["B"], d["shape"], d["tex"], d["batch"]
        if B.dim() == 3:        # class-stacked (C = 1) dump
            mlp = {k: v[0] for k, v in mlp.items()}
            B, sh, tx = B[0], sh[0], tx[0]
            batch = {k: v[0] for k, v in batch.items()}
    else:
        gen = torch.Generator().manual_seed(11)
        pool = analytic_pool(64 * R, n_obj, gen)
        mlpC = {k: v.requires_grad_() for k, v in O.init_codenerf_params(1, 32, L, gen).items()}
        Bc = torch.tensor(O.UNIDIRS).view(1, 21, 3).clone().requires_grad_()
        shl = [O.init_codes(n_obj, L, gen).requires_grad_()]
        txl = [O.init_codes(n_obj, L, gen).requires_grad_()]
        opt = torch.optim.AdamW(list(mlpC.values()) + [Bc] + shl + txl, lr=1e-3, weight_decay=0.013)
        N = pool["depth"].shape[0]

        def sample(i):
            perm = torch.randperm(N, generator=gen)[:R]
            o, dd = O.origin_dirs_O(pool["T_co"][perm], pool["dirs"][perm])
            u = torch.rand(R, S, generator=gen)
            g = torch.randn(R, n2, generator=gen) * (0.1 / 3)
            gt_rgb, gt_d, mask, lab, pts, z = O.sample_3d_points(pool["rgbs"][perm], pool["depth"][perm], o, dd, u, g, n1, n2, 0.1, 0.05)
            return dict(pts=pts[None], z=z[None], gt_depth=gt_d[None], gt_rgb=(gt_rgb / 255.0)[None], labels=lab[None],
                        depth_mask=mask[None], indices=pool["indices"][perm][None])
        for it in range(args.steps):
            bt = sample(it)
            loss, aux = O.forward_loss(mlpC, Bc, 2.0, shl, txl, bt)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if it % 100 == 0 or it == args.steps - 1:
                print(f"step {it}: depth {float(aux['loss_depth']):.4f} colour {float(aux['loss_color']):.4f} opacity {float(aux['loss_opacity']):.4f}", flush=True)
        bt = sample(-1)
        mlp = {k: v.detach()[0] for k, v in mlpC.items()}
        B, sh, tx = Bc.detach()[0], shl[0].detach(), txl[0].detach()
        batch = {k: v[0] for k, v in bt.items()}
        if args.save:
            torch.save(dict(mlp=mlp, B=B, shape=sh, tex=tx, batch=batch), args.save)
    with torch.no_grad():
        e = O.unidirs_embed(batch["pts"][None], B[None], 2.0)[0]
        e1 = e[..., :87]
        cs = sh[batch["indices"]][:, None, :]
        ct = tx[batch["indices"]][:, None, :]
        mlpC1 = {k: v[None] for k, v in mlp.items()}
        sig_ref, col_ref = O.codenerf_forward(mlpC1, e[None], cs[None], ct[None])
        sig_ref = sig_ref[0].squeeze(-1)
        occ_ref, term_ref, depth_ref, _, _, opa_ref = O.composite(sig_ref, col_ref[0], batch["z"])
        rel = lambda a, bb: float((a.double() - bb.double()).norm() / bb.double().norm())
        print(f"logit |max| {float(sig_ref.abs().max()):.1f} rms {float(sig_ref.pow(2).mean().sqrt()):.1f}; "
              f"occ in (0.02, 0.98): {float(((occ_ref > 0.02) & (occ_ref < 0.98)).float().mean()):.3f}, > 0.5: {float((occ_ref > 0.5).float().mean()):.3f}")
        full = [(0, 0), (1, 0), (0, 1)]
        schemes = {
            "fp32 check": dict(w=3, x=3, pe=3, cross=[(i, j) for i in range(3) for j in range(3)]),
            "f16 (shipped)": dict(w=1, x=1, pe=1, cross=[(0, 0)]),
            "split weights only (2 MFMA)": dict(w=2, x=1, pe=1, cross=[(0, 0), (0, 1)]),
            "split activations + PE only (2 MFMA)": dict(w=1, x=2, pe=2, cross=[(0, 0), (1, 0)]),
            "split PE only, weights split (PE layers 3, others 2)": dict(w=2, x=1, pe=2, cross=full),
            "split all (3 MFMA)": dict(w=2, x=2, pe=2, cross=full),
            "split all (4 MFMA)": dict(w=2, x=2, pe=2, cross=full + [(1, 1)]),
        }
        for ln in ("encoding_xyz.0", "shape_layer_1.0", "cat_layer.0", "shape_layer_2.0", "encoding_shape"):
            schemes["split all, ONLY " + ln] = dict(w=2, x=2, pe=2, cross=full, layers={ln})
            schemes["split all, all BUT " + ln] = dict(w=2, x=2, pe=2, cross=full, layers={"encoding_xyz.0", "shape_layer_1.0", "cat_layer.0", "shape_layer_2.0", "encoding_shape"} - {ln})
        hid = {"shape_layer_1.0", "cat_layer.0", "shape_layer_2.0", "encoding_shape"}
        schemes["split hidden-input layers (s1, cat[y], s2, es); PE products plain"] = dict(w=2, x=2, pe=1, cross=full, layers=hid, cat_e1=False)
        schemes["same, activations split only (2 MFMA)"] = dict(w=1, x=2, pe=1, cross=[(0, 0), (1, 0)], layers=hid, cat_e1=False)
        schemes["same, weights split only (2 MFMA)"] = dict(w=2, x=1, pe=1, cross=[(0, 0), (0, 1)], layers=hid, cat_e1=False)
        schemes["hidden-input layers + cat[e1] split (xyz plain)"] = dict(w=2, x=2, pe=2, cross=full, layers=hid, cat_e1=True)
        schemes["hidden full split; PE products: PE lo only (Wh el), 2 MFMA"] = dict(w=2, x=2, pe=2, cross=full, pe_cross=[(0, 0), (1, 0)])
        schemes["hidden full split; PE products: W lo only (Wl eh), 2 MFMA"] = dict(w=2, x=2, pe=2, cross=full, pe_cross=[(0, 0), (0, 1)])
        for name, sc in schemes.items():
            sig = geometry_forward(mlp, e1, cs, sc)
            occ, term, depth, _, _, opa = O.composite(sig, col_ref[0], batch["z"])
            print(f"{name:55s} logit {rel(sig, sig_ref):.2e}  occ {rel(occ, occ_ref):.2e}  depth {rel(depth, depth_ref):.2e}  opacity {rel(opa, opa_ref):.2e}")


if __name__ == "__main__":
    main()
