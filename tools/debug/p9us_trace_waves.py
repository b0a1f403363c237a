"""per-wave cycle stamps of S stages 1 and 2 of one workgroup of the iconv forward kernel (library built with -DP9S_TRACE
-DP9US_TRACE_WAVES): which wave arrives late at the stage-end barrier.  Columns per wave: stage start, patch stored, after barrier 1,
step loop issued -- for stage 1, then stage 2 (cycles relative to the earliest stamp)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jperceiver_amd import ops, _lib
from jperceiver_amd.ops import Var, Tape, recording
g = torch.Generator().manual_seed(1)
N, H, W, Cr, Cx, Cout = 8, 256, 256, 256, 256, 256
r, xh, d = (torch.randn(N, Cr, H, W, generator=g).cuda(), torch.randn(N, Cx, H // 2, W // 2, generator=g).cuda(),
            torch.randn(N, 1, H, W, generator=g).cuda())
w = (torch.randn(Cout, Cr + Cx + 1, 3, 3, generator=g) * (9 * (Cr + Cx + 1)) ** -0.5).cuda()
wv = Var(w)
for it in range(3):
    with recording(Tape()):
        y = ops.conv2d(None, wv, None, 1, 1, 1, 0, srcs=[(Var(r), 0), (Var(xh), 1), (Var(d), 0)])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
f = _lib.lib().cdll.dbg_p9s_trace
f.argtypes = [ctypes.c_void_p]
assert f(ctypes.cast(buf, ctypes.c_void_p)) == 0
t = list(buf)
t0 = min(v for v in t if v)
print("wave (wm, py, px) | S1: start stored barrier1 issued | S2: start stored barrier1 issued")
for wv_ in range(8):
    row = [t[wv_ * 8 + i] - t0 if t[wv_ * 8 + i] else -1 for i in range(8)]
    print(f"  wave {wv_} ({wv_ >> 2}, {(wv_ >> 1) & 1}, {wv_ & 1}) | " + " ".join(f"{v:7d}" for v in row[:4]) + " | " + " ".join(f"{v:7d}" for v in row[4:]))
