"""GPU-box aid: TFLOP/s of conv fwd / dgrad / wgrad on the step's main layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jperceiver_amd._lib import call

SHAPES = [  # N, Cin, H, W, Cout, K, stride, pad, pad_mode
    (8, 256, 256, 256, 256, 3, 1, 1, 1),
    (8, 512, 128, 128, 256, 3, 1, 1, 1),
    (8, 256, 256, 256, 256, 1, 1, 0, 0),
    (8, 64, 256, 256, 64, 3, 1, 1, 0),
    (8, 128, 128, 128, 128, 3, 1, 1, 0),
    (8, 256, 64, 64, 256, 3, 1, 1, 0),
    (8, 512, 32, 32, 512, 3, 1, 1, 0),
    (8, 64, 256, 256, 128, 3, 2, 1, 0),
    (8, 256, 512, 512, 1, 3, 1, 1, 1),
    (8, 3, 1024, 1024, 64, 7, 2, 3, 0),
]
only = sys.argv[1:] 
def p32(c): return (c + 31) // 32 * 32
for (N, Cin, H, W, Cout, K, s, p, pm) in SHAPES:
    OH, OW = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    y = torch.empty(N, Cout, OH, OW, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    wsf = torch.empty(K * K * Cout * p32(Cin), device="cuda") if Cin >= 32 else None
    wsd = torch.empty(K * K * Cin * p32(Cout), device="cuda") if Cout >= 32 else None
    flops = 2.0 * N * OH * OW * Cout * Cin * K * K
    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    tf = t(lambda: call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, K, s, p, pm, 0, wsf, 0, None))
    td = t(lambda: call("jp_conv2d_dgrad", dy, w, dx, N, Cin, H, W, Cout, K, s, p, pm, 0, wsd, 0, None))
    tw = t(lambda: call("jp_conv2d_wgrad", x, dy, dw, N, Cin, H, W, Cout, K, s, p, pm, 0, None, 0))
    print(f"{Cin:4d}->{Cout:4d} k{K} s{s} @{H}x{W}: fwd {tf:7.3f} ms {flops/tf/1e9:6.1f} TF | dgrad {td:7.3f} ms {flops/td/1e9:6.1f} TF | wgrad {tw:7.3f} ms {flops/tw/1e9:6.1f} TF", flush=True)
