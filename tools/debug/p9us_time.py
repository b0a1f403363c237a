"""time of the iconv forward kernel alone (8 x [256 skip + 256 up + 1] -> 256 @256^2), library's own HIP-event profile"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jperceiver_amd import ops
from jperceiver_amd.ops import Var, Tape, recording
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from conv_bench import profiled
g = torch.Generator().manual_seed(1)
N, H, W, Cr, Cx, Cout = 8, 256, 256, 256, 256, 256
r, xh, d = (torch.randn(N, Cr, H, W, generator=g).cuda(), torch.randn(N, Cx, H // 2, W // 2, generator=g).cuda(),
            torch.randn(N, 1, H, W, generator=g).cuda())
w = (torch.randn(Cout, Cr + Cx + 1, 3, 3, generator=g) * (9 * (Cr + Cx + 1)) ** -0.5).cuda()
wv = Var(w)
def step():
    with recording(Tape()):
        ops.conv2d(None, wv, None, 1, 1, 1, 0, srcs=[(Var(r), 0), (Var(xh), 1), (Var(d), 0)])
for it in range(6):
    recs = profiled(step)
    if it >= 2:
        print(os.environ.get("JP_LIB_PATH", "default")[-24:], [(n[:40], round(ms, 3), round(fl / ms / 1e9)) for n, fl, ms in recs if "p9us" in n])
