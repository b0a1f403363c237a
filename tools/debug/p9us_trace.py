"""cycle stamps of wave 0 of one workgroup of the iconv forward kernel (igemm_p9us.h; library built with -DP9S_TRACE, JP_LIB_PATH) at
the decoder's largest iconv: cat(skip 256, up2x(x 256), disp 1) -> 256, reflect, 8 x 256 x 256.  Stamp order: start | S stages 0..3:
(start, patch stored, step loop issued) | S stages NS0-2, NS0-1: start | U stages 0..3: (start, stored, issued) | K loop done | end."""
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
t = [v for v in buf if v]
names = ["start"]
for s in range(4):
    names += [f"S{s} start", f"S{s} patch stored", f"S{s} step loop issued (9 steps)"]
names += ["S(last-1) start", "S(last) start"]
for s in range(4):
    names += [f"U{s} start", f"U{s} patch stored", f"U{s} step loop issued (4 steps)"]
names += ["K loop done (incl. D stage)", "epilogue done"]
prev = t[0]
for i, v in enumerate(t):
    print(f"  [{i:2d}] {v - t[0]:8d}  (+{v - prev:6d})  {names[i] if i < len(names) else ''}")
    prev = v
