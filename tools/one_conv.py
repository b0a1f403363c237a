"""GPU-box aid: time the three conv kernels of one layer shape (events on the launch stream)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jperceiver_amd._lib import call, lib
aws = torch.empty(int(lib().fn["jp_conv2d_amax_ws_floats"]()), device="cuda")     # scratch for the operand magnitudes the call reduces itself
N, Cin, H, W, Cout, K = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (8, 256, 256, 256, 256, 3))]
mode = int(sys.argv[7]) if len(sys.argv) > 7 else 1   # 1 = reflect pad, 0 = zero pad
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
y = torch.empty(N, Cout, H, W, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
wsf = torch.empty(int(lib().fn["jp_conv2d_ws_floats"](Cin, Cout, K, 0)), device="cuda")
wsd = torch.empty(int(lib().fn["jp_conv2d_ws_floats"](Cin, Cout, K, 1)), device="cuda")
nws = int(lib().fn["jp_conv2d_wgrad_ws_floats"](N, Cin, H, W, Cout, K, 1, K // 2))
wsw = torch.empty(max(nws, 1), device="cuda")
print("wgrad scratch floats", nws)
fl = 2e-9 * N * H * W * Cout * Cin * K * K
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
p = K // 2
for name, f in (("fwd", lambda: call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, K, 1, p, mode, 0, wsf, 0, None, None, None, None, aws, None, None)),
                ("dgrad", lambda: call("jp_conv2d_dgrad", dy, w, dx, N, Cin, H, W, Cout, K, 1, p, mode, 0, wsd, 0, None, None, None, None, aws)),
                ("wgrad", lambda: call("jp_conv2d_wgrad", x, dy, dw, N, Cin, H, W, Cout, K, 1, p, mode, 0, wsw, nws, None, None, aws))):
    ms = t(f)
    print("%-6s %d->%d k%d @%d  %.3f ms  %.1f TF" % (name, Cin, Cout, K, H, ms, fl / ms))
