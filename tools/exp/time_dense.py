"""Per-call times of the background model's dense products (M = 16800 samples)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import _C
dev = torch.device("cuda:0")
M = 16800
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, N in ((87, 128), (128, 128), (215, 128), (170, 128), (128, 1), (128, 3)):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    wsb = int(_C.load().cnr_dense_bwd_workspace_bytes(M, K, N)); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    for half in (0, 1):
        f = t(lambda: _C.call("cnr_dense_fwd", x, W, b, y, M, K, N, 1, half))
        bx = t(lambda: _C.call("cnr_dense_bwd", x, W, y, dy, dx, dW, db, M, K, N, 1, ws, wsb, half, 1024.0))
        bw = t(lambda: _C.call("cnr_dense_bwd", x, W, y, dy, None, dW, db, M, K, N, 1, ws, wsb, half, 1024.0))
        print(f"K {K:4d} N {N:4d} half {half}: fwd {f:6.1f} us | bwd (dx + dW + reduce) {bx:6.1f} us | dW + reduce alone {bw:6.1f} us")
