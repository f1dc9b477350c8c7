"""Positional encoding module, same interface and state_dict as the reference's
``embedding.UniDirsEmbed`` (src/embedding.py:43-92): trainable ``B_layer.weight (21,3)``, persistent
buffer ``scale``, non-persistent ``frequency_bands``; forward = HIP kernel cnr_pe_fwd."""
import torch

from .ops import UniDirsEmbedFn

# the 21 icosahedral unit directions of the method (src/embedding.py:51-73)
_G = 0.8506508
_H = 0.5257311
UNIDIRS = [
    [_G, 0, _H], [0.809017, 0.5, 0.309017], [_H, _G, 0], [1, 0, 0], [0.809017, 0.5, -0.309017],
    [_G, 0, -_H], [0.309017, 0.809017, -0.5], [0, _H, -_G], [0.5, 0.309017, -0.809017], [0, 1, 0],
    [-_H, _G, 0], [-0.309017, 0.809017, -0.5], [0, _H, _G], [-0.309017, 0.809017, 0.5],
    [0.309017, 0.809017, 0.5], [0.5, 0.309017, 0.809017], [0.5, -0.309017, 0.809017], [0, 0, 1],
    [-0.5, 0.309017, 0.809017], [-0.809017, 0.5, 0.309017], [-0.809017, 0.5, -0.309017],
]


class UniDirsEmbed(torch.nn.Module):
    def __init__(self, min_deg=0, max_deg=2, scale=2.):
        super().__init__()
        if min_deg != 0 or max_deg != 5:
            # the kernels are built for the 6-band / 129-d layout every shipped config uses
            # (n_unidir_funcs = 5, src/trainer.py:20-21)
            raise NotImplementedError("UniDirsEmbed HIP kernels support min_deg=0, max_deg=5 only")
        self.min_deg, self.max_deg = min_deg, max_deg
        self.n_freqs = max_deg - min_deg + 1
        self._scale = float(scale)
        self.B_layer = torch.nn.Linear(3, 21, bias=False)
        self.B_layer.weight.data = torch.tensor(UNIDIRS, dtype=torch.float32)
        self.register_buffer("frequency_bands", 2.0 ** torch.linspace(min_deg, max_deg, self.n_freqs),
                             persistent=False)
        self.register_buffer("scale", torch.tensor(float(scale)), persistent=True)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "scale"
        if key in state_dict:
            self._scale = float(state_dict[key])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x):
        return UniDirsEmbedFn.apply(x, self.B_layer.weight, self._scale)
