import os, sys
sys.path.insert(0, "/root/repo")
import torch
from jperceiver_amd._lib import call, lib
aws = torch.empty(int(lib().fn["jp_conv2d_amax_ws_floats"]()), device="cuda")     # scratch for the operand magnitudes the call reduces itself
L = lib()
for (N, Cin, H, W, Cout) in [(8,256,256,256,256),(8,256,256,256,128),(8,256,256,256,64),(8,512,128,128,256),(8,128,256,256,256)]:
    K,s,p,pm=1,1,0,0
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    y = torch.empty(N, Cout, H, W, device="cuda")
    wsf = torch.empty(int(L.fn["jp_conv2d_ws_floats"](Cin, Cout, K, 0)), device="cuda")
    flops = 2.0 * N * H * W * Cout * Cin
    def t(fn, n=8):
        fn(0); fn(1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn(1)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    tf = t(lambda st: call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, K, s, p, pm, 2, wsf, st, None, None, None, None, aws, None, None))
    gb = (x.numel()+y.numel())*4/1e9
    print(f"{Cin}->{Cout} @{H}: {tf:.3f} ms {flops/tf/1e9:.1f} TF, min HBM {gb:.2f} GB -> {gb/tf:.2f} TB/s", flush=True)
