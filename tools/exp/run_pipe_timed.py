"""Cycle stamps of the pipelined field backward (build: see tools/exp/build_pipe_timed.sh)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cnr_amd
from cnr_amd import ops, _C
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpipetimed" + (sys.argv[4] if len(sys.argv) > 4 else "") + ".so"))
dev = torch.device("cuda:0")
L, n_obj = 256, 4
C, R, S, NCH = 1, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen = torch.Generator().manual_seed(0)
theta, lay = cnr_amd.fused.init_params(C, L, n_obj, gen, dev)
v = lay.views(theta)
packed = ops.pack_weights(v["trunk"].contiguous())
pts = (torch.rand(C, R, S, 3, device=dev) * 2 - 1)
B = v["B"].contiguous()
brows = torch.randn(C * n_obj, 4, 32, device=dev) * 0.1
ray_row = (torch.randint(0, n_obj, (C, R), device=dev)).to(torch.int32)
torch.cuda.synchronize()
dsig = torch.randn(C, R, S, device=dev) * 1e-3
drgb = torch.randn(C, R, S, 3, device=dev) * 1e-3
dtrunk = torch.zeros(C, 13892, device=dev); dB = torch.zeros(C, 21, 3, device=dev); dbr = torch.zeros_like(brows)
vp = ctypes.c_void_p
fn = lib.cnr_field_bwd_pipe
fn.argtypes = [vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, ctypes.c_float, vp, vp, vp] + [ctypes.c_int] * 6 + [vp] + [ctypes.c_int64] * 4 + [vp, ctypes.c_int, vp]
wsp = torch.empty(_C.field_bwd_workspace_bytes(C, 0), device=dev, dtype=torch.uint8)
for it in range(3):
    rc = fn(pts.data_ptr(), B.data_ptr(), packed.data_ptr(), brows.data_ptr(), ray_row.data_ptr(), 2.0, dsig.data_ptr(), drgb.data_ptr(),
            2048.0, dtrunk.data_ptr(), dB.data_ptr(), dbr.data_ptr(), C, R, S, n_obj, 0, NCH, wsp.data_ptr(), wsp.numel(), 0, 0, 0, None, 0, None)
    torch.cuda.synchronize()
    if NCH == 4:
        buf = (ctypes.c_longlong * (8 * 64))()
        lib.cnr_pipe8_read_stamps(buf)
        base = min(buf[w * 64] for w in range(8))
        for w in range(8):
            st = list(buf)[w * 64: w * 64 + 37]
            # stamps: 0 = iteration start, then (before barrier, after barrier) x 18 (A, B per layer step)
            work = [st[1] - st[0]] + [st[2 * k + 1] - st[2 * k] for k in range(1, 18)]
            wait = [st[2 * k + 2] - st[2 * k + 1] for k in range(18)]
            print(rc, "wave", w, "t0", st[0] - base, "total", st[36] - st[0], "sum work", sum(work), "sum wait", sum(wait))
            print("      work", work)
            print("      wait", wait)
            ph = list(buf)[w * 64 + 56: w * 64 + 62]
            print("      phases: start->weights in LDS", ph[1] - ph[0], "loops", ph[2] - ph[1], "partial sums -> record", ph[3] - ph[2],
                  "blocks -> LDS image", ph[4] - ph[3], "image -> record", ph[5] - ph[4], "| total", ph[5] - ph[0])
        continue
    buf = (ctypes.c_longlong * 160)()
    lib.cnr_pipe_read_stamps(buf)
    base = min(buf[w * 32] for w in range(4))
    for w in range(4):
        st = list(buf)[w * 32: w * 32 + 21]
        # stamps: 0 = iteration start, then (before barrier, after barrier) x 10
        work = [st[1] - st[0]] + [st[2 * k + 1] - st[2 * k] for k in range(1, 10)]
        wait = [st[2 * k + 2] - st[2 * k + 1] for k in range(10)]
        print(rc, "wave", w, "t0", st[0] - base, "work", work, "wait", wait, "total", st[20] - st[0])
    f = list(buf)[128:134]
    print("   wave 0 fwd: start->loads/pe", f[0] - buf[0], "pe+imgs", f[1] - f[0], "xyz", f[2] - f[1], "s1,cat,s2", f[3] - f[2],
          "es", f[4] - f[3], "vd,t1,r0,r2", f[5] - f[4], "to barrier", buf[1] - f[5])
