"""TEST DOUBLE for the C-ABI (tests only): implements the modular entry points on CPU tensors with the oracle so
that the host logic (autograd Functions, vmap rules, modules, loss assembly) can be exercised on a GPU-less box.
Installed explicitly by tests through cnr_amd._C.install_test_double; the product never does that."""
import torch

from oracle import ref_cpu as O

E1, E = 87, 129
TRUNK = [("encoding_xyz.0", 32, 87), ("shape_layer_1.0", 32, 32), ("shape_layer_2.0", 32, 32), ("cat_layer.0", 32, 119),
         ("encoding_shape", 32, 32), ("sigma.0", 1, 32), ("encoding_viewdir.0", 32, 74), ("texture_layer_1.0", 32, 32),
         ("rgb.0", 16, 32), ("rgb.2", 3, 16)]


def _unpack(trunk):
    C, p, off = trunk.shape[0], {}, 0
    for n, o, i in TRUNK:
        p[n + ".weight"] = trunk[:, off:off + o * i].reshape(C, o, i); off += o * i
        p[n + ".bias"] = trunk[:, off:off + o]; off += o
    return p


def _trunk_forward(p, e, zlat):
    lin = lambda n, x: torch.matmul(x, p[n + ".weight"].transpose(-1, -2)[:, None]) + p[n + ".bias"][:, None, None, :]
    z = lambda k: zlat[:, :, None, k, :]
    e1, e2 = e[..., :E1], e[..., E1:]
    y = torch.relu(lin("encoding_xyz.0", e1)) + z(0)
    y = torch.relu(lin("shape_layer_1.0", y)) + z(1)
    y = torch.relu(lin("cat_layer.0", torch.cat((y, e1), -1))) + z(2)
    y = torch.relu(lin("shape_layer_2.0", y))
    y = lin("encoding_shape", y)
    sig = lin("sigma.0", y).squeeze(-1) * 10.0
    y = torch.relu(lin("encoding_viewdir.0", torch.cat((y, e2), -1))) + z(3)
    y = torch.relu(lin("texture_layer_1.0", y))
    y = torch.relu(lin("rgb.0", y))
    return sig, torch.sigmoid(lin("rgb.2", y))


def _with_grad(fn):
    def wrapped(*a, **k):
        with torch.enable_grad():      # these run inside autograd.Function.backward, where grad mode is off
            return fn(*a, **k)
    return wrapped


class Double:
    def cnr_pe_fwd(self, x, B, e, C, N, scale):
        e.copy_(O.unidirs_embed(x.reshape(C, N, 1, 3), B.reshape(C, 21, 3), scale).reshape(e.shape))

    @_with_grad
    def cnr_pe_bwd(self, x, B, de, dB, dx, C, N, scale):
        Bv = B.reshape(C, 21, 3).detach().clone().requires_grad_()
        xv = x.reshape(C, N, 1, 3).detach().clone().requires_grad_()
        O.unidirs_embed(xv, Bv, scale).backward(de.reshape(C, N, 1, E))
        dB.add_(Bv.grad.reshape(dB.shape))
        if dx is not None:
            dx.copy_(xv.grad.reshape(dx.shape))

    def cnr_mlp_fwd_f32(self, e, zlat, trunk, sig, rgb, C, R, S):
        s, c = _trunk_forward(_unpack(trunk), e.reshape(C, R, S, E), zlat.reshape(C, R, 4, 32))
        sig.copy_(s.reshape(sig.shape)); rgb.copy_(c.reshape(rgb.shape))

    @_with_grad
    def cnr_mlp_bwd_f32(self, e, zlat, trunk, dsig, drgb, de, dz, dtrunk, C, R, S):
        ev = e.reshape(C, R, S, E).detach().clone().requires_grad_()
        zv = zlat.reshape(C, R, 4, 32).detach().clone().requires_grad_()
        tv = trunk.detach().clone().requires_grad_()
        s, c = _trunk_forward(_unpack(tv), ev, zv)
        torch.autograd.backward([s, c], [dsig.reshape(s.shape), drgb.reshape(c.shape)])
        de.copy_(ev.grad.reshape(de.shape)); dz.add_(zv.grad.reshape(dz.shape)); dtrunk.add_(tv.grad)

    def cnr_composite_fwd(self, alpha, color, z, term, depth, var, rgb, opa, NR, S, in_is_occ):
        a = alpha.reshape(NR, S)
        occ = a if in_is_occ else torch.sigmoid(a)
        t = O.occupancy_to_termination(occ)
        if term is not None: term.copy_(t.reshape(term.shape))
        if depth is not None:
            zz = z.reshape(NR, S); d = (t * zz).sum(-1)
            depth.copy_(d.reshape(depth.shape)); var.copy_((t * (zz - d[:, None]) ** 2).sum(-1).reshape(var.shape))
        if rgb is not None: rgb.copy_((t[..., None] * color.reshape(NR, S, 3)).sum(-2).reshape(rgb.shape))
        if opa is not None: opa.copy_(t.sum(-1).reshape(opa.shape))

    @_with_grad
    def cnr_composite_bwd(self, alpha, color, z, dd, dr, do, dt, d_alpha, d_color, NR, S, in_is_occ):
        a = alpha.reshape(NR, S).detach().clone().requires_grad_()
        occ = a if in_is_occ else torch.sigmoid(a)
        t = O.occupancy_to_termination(occ)
        tot = torch.zeros(())
        col = None
        if color is not None:
            col = color.reshape(NR, S, 3).detach().clone().requires_grad_()
        if dd is not None: tot = tot + ((t * z.reshape(NR, S)).sum(-1) * dd.reshape(NR)).sum()
        if dr is not None: tot = tot + ((t[..., None] * col).sum(-2) * dr.reshape(NR, 3)).sum()
        if do is not None: tot = tot + (t.sum(-1) * do.reshape(NR)).sum()
        if dt is not None: tot = tot + (t * dt.reshape(NR, S)).sum()
        tot.backward()
        d_alpha.copy_(a.grad.reshape(d_alpha.shape))
        if d_color is not None:
            d_color.copy_((col.grad if col.grad is not None else torch.zeros_like(col)).reshape(d_color.shape))

    @_with_grad
    def cnr_loss_fwd_bwd(self, depth, var, rgb, opa, gt_d, gt_rgb, labels, dmask, cs, os_, gs, losses, flags, dd, dr, do, C, R):
        d = depth.detach().clone().requires_grad_(); c = rgb.detach().clone().requires_grad_(); o = opa.detach().clone().requires_grad_()
        mo, ms = labels != 0, labels != 2
        md = dmask.bool() & mo
        ld = O.reduce_batch_loss((d - gt_d).abs() * md, var=var, mask=md)
        lc = O.reduce_batch_loss((c - gt_rgb).abs().sum(-1) * mo, mask=mo)
        lo = O.reduce_batch_loss((o - mo.float()).abs() * ms, mask=ms)
        losses.copy_(torch.stack([ld, lc, lo]).detach())
        total = (ld + cs * lc + os_ * lo).sum() * gs
        if total.requires_grad:
            total.backward()
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
        dd.copy_(z(d)); dr.copy_(z(c)); do.copy_(z(o))
        fl = torch.zeros(C, dtype=torch.int32)
        fl |= ((ld > 1e5) | (lc > 1e5) | (lo > 1e5)).int()
        fl |= 2 * int((md.sum(-1) == 0).any()) | 4 * int((mo.sum(-1) == 0).any()) | 8 * int((ms.sum(-1) == 0).any())
        flags.copy_(fl)
