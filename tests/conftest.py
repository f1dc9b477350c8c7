import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _all_golden():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def golden_names():
    """object-branch fixtures (CodeNeRF train step)"""
    return [n for n in _all_golden() if not n.startswith(("bg_", "pool_", "ckpt_", "cam_"))]


def pool_golden_names():
    """ray-pool construction fixtures (frames + the pools the reference's sceneCategory.__init__ built)"""
    return [n for n in _all_golden() if n.startswith("pool_")]


def bg_golden_names():
    """background-branch fixtures (OccupancyMap train step; meta[3] = hidden size)"""
    return [n for n in _all_golden() if n.startswith("bg_")]


class Golden:
    """One committed fixture (generated from the reference by tests/golden/gen_golden.py)."""

    def __init__(self, name, device="cpu"):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.device = device
        m = self.z["meta"]
        self.C, self.R, self.S, self.L, self.n_obj, self.n1, self.n2 = [int(v) for v in m[:7]]
        self.single_obj = bool(m[7])
        self.scale = float(self.z["scale"])
        self.eps, self.stop_eps = float(self.z["eps"]), float(self.z["stop_eps"])

    def __contains__(self, k):
        return k in self.z.files

    def t(self, k):
        return torch.from_numpy(self.z[k]).to(self.device)

    def mlp(self, prefix="mlp."):
        return {k[len(prefix):]: self.t(k) for k in self.z.files if k.startswith(prefix)}


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


@pytest.fixture(autouse=True)
def _seed_every_test(request):
    """Every test starts from a seed derived from its id: the default CPU generator AND the device generators
    (torch.manual_seed seeds both), so that a draw made without an explicit generator -- torch.rand(..., device=dev) --
    is the same in every run and a red run reproduces."""
    import zlib
    torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF)
    yield


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
