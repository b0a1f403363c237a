"""GPU-box aid: TFLOP/s of conv fwd / dgrad / wgrad on the step's main layer shapes (weights pre-packed: ws_state 1).
   JP_P9=0 python tools/conv_bench.py   vs   JP_P9=1 python tools/conv_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jperceiver_amd._lib import call, lib

SHAPES = [  # N, Cin, H, W, Cout, K, stride, pad, pad_mode
    (8, 256, 256, 256, 256, 3, 1, 1, 1),
    (8, 256, 128, 128, 256, 3, 1, 1, 1),
    (8, 256, 64, 64, 256, 3, 1, 1, 1),
    (8, 512, 32, 32, 256, 3, 1, 1, 1),
    (8, 128, 128, 128, 128, 3, 1, 1, 0),
    (8, 256, 64, 64, 256, 3, 1, 1, 0),
    (8, 512, 32, 32, 512, 3, 1, 1, 0),
    (8, 64, 256, 256, 64, 3, 1, 1, 0),
    (8, 256, 256, 256, 256, 1, 1, 0, 0),
]
L = lib()
for (N, Cin, H, W, Cout, K, s, p, pm) in SHAPES:
    OH, OW = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    y = torch.empty(N, Cout, OH, OW, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    wsf = torch.empty(int(L.fn["jp_conv2d_ws_floats"](Cin, Cout, K, 0)), device="cuda")
    wsd = torch.empty(int(L.fn["jp_conv2d_ws_floats"](Cin, Cout, K, 1)), device="cuda")
    nws = int(L.fn["jp_conv2d_wgrad_ws_floats"](N, Cin, H, W, Cout, K, s, p))
    wsw = torch.empty(max(nws, 1), device="cuda")
    flops = 2.0 * N * OH * OW * Cout * Cin * K * K
    def t(fn, n=5):
        fn(0); fn(1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn(1)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    tf = t(lambda st: call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, K, s, p, pm, 2, wsf, st, None))
    td = t(lambda st: call("jp_conv2d_dgrad", dy, w, dx, N, Cin, H, W, Cout, K, s, p, pm, 0, wsd, st, None))
    tw = t(lambda st: call("jp_conv2d_wgrad", x, dy, dw, N, Cin, H, W, Cout, K, s, p, pm, 0, wsw if nws else None, nws))
    print(f"{Cin:4d}->{Cout:4d} k{K} s{s} pm{pm} @{H}x{W}: fwd {tf:7.3f} ms {flops/tf/1e9:6.1f} TF | dgrad {td:7.3f} ms {flops/td/1e9:6.1f} TF | wgrad {tw:7.3f} ms {flops/tw/1e9:6.1f} TF", flush=True)
