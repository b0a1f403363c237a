import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jperceiver_amd._lib import call
N, Cin, H, W, Cout, K = 8, 256, 256, 256, 256, 3
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
y = torch.empty(N, Cout, H, W, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
ws = torch.empty((9 * 256 + 256) * 256, device="cuda")
for _ in range(3):
    call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, K, 1, 1, 1, 0, ws)
    call("jp_conv2d_dgrad", dy, w, dx, N, Cin, H, W, Cout, K, 1, 1, 1, 0, ws)
    call("jp_conv2d_wgrad", x, dy, dw, N, Cin, H, W, Cout, K, 1, 1, 1, 0)
torch.cuda.synchronize()
