"""numerics of the 1x1 patch kernels under the current JP_P1_TILE: 256->256 @64^2 and @128^2 forward + dgrad vs ATen on the CPU"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from jperceiver_amd import ops
from jperceiver_amd.ops import Var, Tape, recording
for N, C, H, W, Co in ((8, 256, 64, 64, 256), (8, 256, 128, 128, 256), (8, 512, 64, 64, 256)):
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(N, C, H, W, generator=g), torch.randn(Co, C, 1, 1, generator=g) * C ** -0.5
    gy = torch.randn(N, Co, H, W, generator=g)
    xv, wv = Var(x.cuda(), True), Var(w.cuda(), True, torch.zeros_like(w).cuda())
    t = Tape()
    with recording(t):
        y = ops.conv2d(xv, wv, None, 1, 0, 0, 0)
    y.g = gy.cuda(); t.backward()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr); yr.backward(gy)
    e = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    print(f"JP_P1_TILE={os.environ.get('JP_P1_TILE','0')} {C}->{Co} @{H}: fwd {e(y.t, yr.detach()):.2e} dgrad {e(xv.g, xr.grad):.2e} wgrad {e(wv.g, wr.grad):.2e}", flush=True)
