"""CodeNeRF with the reference's parameter names / registration order (src/model.py:22-84) so
state_dicts and ``combine_state_for_ensemble`` stacking interchange.  The ten per-sample layers run in
the HIP trunk kernel; the four latent layers are per-RAY (really per-object) work and stay in PyTorch
(plain library GEMMs), producing the (R,4,32) ``zlat`` rows the kernel adds in."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ops import CodeNeRFTrunkFn, LATENT_LAYERS, TRUNK_LAYERS


def init_weights(m, init_fn=torch.nn.init.xavier_normal_):
    """src/model.py:4-6: xavier-normal weights, biases keep nn.Linear's default."""
    if type(m) == torch.nn.Linear:
        init_fn(m.weight)


class CodeNeRF(nn.Module):
    def __init__(self, emb_size1, emb_size2, shape_blocks=2, texture_blocks=1, W=32, latent_dim=32):
        super().__init__()
        if (emb_size1, emb_size2, shape_blocks, texture_blocks, W) != (87, 42, 2, 1, 32):
            raise NotImplementedError("HIP trunk kernels are built for emb 87/42, 2 shape blocks, "
                                      "1 texture block, W=32 (every shipped config)")
        self.shape_blocks, self.texture_blocks = shape_blocks, texture_blocks
        self.embedding_size1, self.embedding_size2 = emb_size1, emb_size2
        self.encoding_xyz = nn.Sequential(nn.Linear(emb_size1, W), nn.ReLU())
        for j in range(shape_blocks):
            setattr(self, f"shape_latent_layer_{j+1}", nn.Sequential(nn.Linear(latent_dim, W), nn.ReLU()))
            setattr(self, f"shape_layer_{j+1}", nn.Sequential(nn.Linear(W, W), nn.ReLU()))
        self.cat_layer = nn.Sequential(nn.Linear(W + emb_size1, W), nn.ReLU())
        self.cat_latent_layer = nn.Sequential(nn.Linear(latent_dim, W), nn.ReLU())
        self.encoding_shape = nn.Linear(W, W)
        self.sigma = nn.Sequential(nn.Linear(W, 1))
        self.encoding_viewdir = nn.Sequential(nn.Linear(W + emb_size2, W), nn.ReLU())
        for j in range(texture_blocks):
            setattr(self, f"texture_layer_{j+1}", nn.Sequential(nn.Linear(W, W), nn.ReLU()))
            setattr(self, f"texture_latent_layer_{j+1}", nn.Sequential(nn.Linear(latent_dim, W), nn.ReLU()))
        self.rgb = nn.Sequential(nn.Linear(W, W // 2), nn.ReLU(), nn.Linear(W // 2, 3), nn.Sigmoid())

    def _linear(self, name):
        m = self
        for part in name.split("."):
            m = getattr(m, part) if not part.isdigit() else m[int(part)]
        return m

    def latent_rows(self, shape_latent, texture_latent):
        """(R,1,L) codes -> zlat (R,4,32): ReLU(latent layer) per slot (src/model.py:61,66,83)."""
        zs = []
        for i, name in enumerate(LATENT_LAYERS):
            lin = self._linear(name)
            code = texture_latent if i == 3 else shape_latent
            zs.append(F.relu(F.linear(code, lin.weight, lin.bias)))  # (R,1,32)
        return torch.cat(zs, dim=-2)                                  # (R,4,32)

    def forward(self, x, shape_latent, texture_latent, noise_std=None, do_cat=True):
        if noise_std is not None or not do_cat:
            raise NotImplementedError("the train path uses noise_std=None, do_cat=True (train.py:155)")
        zlat = self.latent_rows(shape_latent, texture_latent)
        params = []
        for name, _, _ in TRUNK_LAYERS:
            lin = self._linear(name)
            params += [lin.weight, lin.bias]
        return CodeNeRFTrunkFn.apply(x, zlat, *params)


def fc_block(in_f, out_f):
    """src/model.py:8-12 (Linear + ReLU); kept as a Sequential so that state_dict keys match ('<name>.0.weight')."""
    return nn.Sequential(nn.Linear(in_f, out_f), nn.ReLU())


class OccupancyMap(nn.Module):
    """vMAP-style background / pretrained per-object field (src/model.py:86-155), same constructor, parameter names
    and forward signature.  Every Linear (+ ReLU) runs on cnr_dense_fwd / cnr_dense_bwd (exact fp32 MFMA); the
    concatenations are views/copies in torch."""

    def __init__(self, emb_size1, emb_size2, hidden_size=256, do_color=True, hidden_layers_block=1):
        super().__init__()
        self.half = False     # True: the dense layers run on the f16-operand MFMA form (fp32 accumulate); not a parameter
        self.grad_out = {}    # id(Linear) -> (dW, db) tensors the backward writes into directly (background.BackgroundStep)
        self.do_color = do_color
        self.embedding_size1, self.embedding_size2 = emb_size1, emb_size2
        self.in_layer = fc_block(emb_size1, hidden_size)
        self.mid1 = nn.Sequential(*[fc_block(hidden_size, hidden_size) for _ in range(hidden_layers_block)])
        self.cat_layer = fc_block(hidden_size + emb_size1, hidden_size)
        self.mid2 = nn.Sequential(*[fc_block(hidden_size, hidden_size) for _ in range(hidden_layers_block)])
        self.out_alpha = nn.Linear(hidden_size, 1)
        if do_color:
            self.color_linear = fc_block(emb_size2 + hidden_size, hidden_size)
            self.out_color = nn.Linear(hidden_size, 3)

    def _fc(self, block, x):       # fc_block: Linear + ReLU in one kernel
        return ops.DenseFn.apply(x, block[0].weight, block[0].bias, True, self.half, self.grad_out.get(id(block[0])))

    def forward(self, x, noise_std=None, do_alpha=True, do_color=True, do_cat=True):
        e1 = x[..., :self.embedding_size1]
        fc2 = self._fc(self.in_layer, e1)
        for blk in self.mid1:
            fc2 = self._fc(blk, fc2)
        fc3 = self._fc(self.cat_layer, torch.cat((fc2, e1), dim=-1)) if do_cat else fc2
        fc4 = fc3
        for blk in self.mid2:
            fc4 = self._fc(blk, fc4)
        alpha = None
        if do_alpha:
            raw = ops.DenseFn.apply(fc4, self.out_alpha.weight, self.out_alpha.bias, False, False,
                                    self.grad_out.get(id(self.out_alpha)))                    # the x10 logit stays fp32
            if noise_std is not None:
                raw = raw + torch.randn(raw.shape, device=x.device) * noise_std
            alpha = raw * 10.0                         # unisurf scaling, src/model.py:142
        color = None
        if self.do_color and do_color:
            fc5 = self._fc(self.color_linear, torch.cat((fc4, x[..., self.embedding_size1:]), dim=-1))
            color = torch.sigmoid(ops.DenseFn.apply(fc5, self.out_color.weight, self.out_color.bias, False, self.half,
                                                    self.grad_out.get(id(self.out_color))))
        return alpha, color
