"""MI355X-native volumetric-rendering train-step hot path for category-level neural fields.

Mirrors the reference's call surface (flat modules of /src): ``trainer``, ``scene_cateogries``,
``embedding``, ``model``, ``render_rays``, ``loss``, ``utils.update_vmap`` -- backed by hand-written
gfx950 HIP kernels behind a C-ABI (include/cnr_hip.h, libcnr_hip.so).  ``fused`` holds the
single-launch fast path, ``background`` the capturable background branch / whole-iteration graph, ``parallel`` the
multi-GPU sharding and ``category_registration`` the forward-only uncertainty probe.  Import as ``cnr_amd`` (see
cnr_amd.py at the repo root; the directory name carries a hyphen).
"""
from . import _C, ops  # noqa: F401
from . import cfg, embedding, model, render_rays, loss, trainer, utils, scene_cateogries, fused, parallel, background, category_registration  # noqa: F401

__version__ = "0.1.0"
