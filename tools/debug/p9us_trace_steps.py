"""per-step cycle stamps of S stage 2 of one workgroup of the iconv forward kernel (library built with -DP9S_TRACE -DP9US_TRACE_STEPS):
waves 0, 1 (the older wave of their SIMD) and 4, 5 (the younger).  Stamps: stage start, patch stored, after barrier 1, start of
steps 0..8, step loop issued, after the stage-end barrier."""
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
# round-3 kernel (JP_P9US2=0): stage start, patch stored, after barrier 1, steps, loop issued, after the stage-end barrier;
# P9US2: stage start (after the younger half's staging burst), steps, loop issued, after the barrier
V2 = os.environ.get("JP_P9US2", "1") != "0"
names = (["start"] if V2 else ["start", "stored", "bar1"]) + [f"u{i}" for i in range(9)] + ["issued", "bar2"]
print("wave | " + " ".join(f"{n:>7s}" for n in names))
for slot, wv_ in enumerate((0, 1, 4, 5)):
    row = [t[slot * 16 + i] - t0 if t[slot * 16 + i] else -1 for i in range(len(names))]
    print(f"  w{wv_} | " + " ".join(f"{v:7d}" for v in row))
    print("       " + " ".join(f"{(row[i] - row[i - 1]) if i else 0:7d}" for i in range(len(names))))
