"""CodeNeRF with the reference's parameter names / registration order (src/model.py:22-84) so
state_dicts and ``combine_state_for_ensemble`` stacking interchange.  The ten per-sample layers run in
the HIP trunk kernel; the four latent layers are per-RAY (really per-object) work and stay in PyTorch
(plain library GEMMs), producing the (R,4,32) ``zlat`` rows the kernel adds in."""
import torch
import torch.nn as nn
import torch.nn.functional as F

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
