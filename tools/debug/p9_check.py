"""numerics of the 3x3 patch kernels under the current JP_P9_TILE vs ATen on the CPU (forward + dgrad, zero and reflect padding)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from jperceiver_amd import ops
from jperceiver_amd.ops import Var, Tape, recording
for N, C, H, W, Co, pm in ((8, 256, 128, 128, 256, 1), (8, 128, 128, 128, 128, 0), (8, 256, 64, 64, 256, 0)):
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(N, C, H, W, generator=g), torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5
    gy = torch.randn(N, Co, H, W, generator=g)
    xv, wv = Var(x.cuda(), True), Var(w.cuda(), True, torch.zeros_like(w).cuda())
    t = Tape()
    with recording(t):
        y = ops.conv2d(xv, wv, None, 1, 1, pm, 0)
    y.g = gy.cuda(); t.backward()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr) if pm else F.conv2d(xr, wr, None, 1, 1)
    yr.backward(gy)
    e = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    print(f"JP_P9_TILE={os.environ.get('JP_P9_TILE','0')} {C}->{Co} @{H} pad_mode {pm}: fwd {e(y.t, yr.detach()):.2e} dgrad {e(xv.g, xr.grad):.2e}", flush=True)
