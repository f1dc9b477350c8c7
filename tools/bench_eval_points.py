#!/usr/bin/env python3
"""Throughput of Trainer.eval_points (src/trainer.py:125-151; the reference's only stated number for this path is the
comment at :134, "2s/it 1000000 pts" per 500 k-point chunk pair on an unstated CUDA GPU): 256^3-scale query in 500 k
chunks through the fused forward (CodeNeRF) and the fp32 dense kernels (background OccupancyMap).  Writes one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnr_amd  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    out = {}
    for name, cls_id, hidden in (("codenerf_L256", 3, 32), ("occupancy_map_h128", 0, 128)):
        cfg = cnr_amd.cfg.synthetic_config(device=str(dev), latent_dim=256)
        cfg.hidden_feature_size = hidden
        t = cnr_amd.trainer.Trainer(cfg, cls_id, [0, 1, 2, 3] if cls_id else [0])
        pts = torch.rand(n, 3, device=dev) * 2 - 1
        t.eval_points(pts[:500000], inst_id=1 if cls_id else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.eval_points(pts, inst_id=1 if cls_id else None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"points": n, "seconds": dt, "points_per_s": n / dt, "s_per_1e6_points": dt / n * 1e6}
    out["reference_comment"] = "src/trainer.py:134: '2s/it 1000000 pts' (unstated CUDA GPU, fp32 PyTorch)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
