"""GPU-box aid: run ONE conv layer forward a few times (for rocprofv3 --pmc passes).  args: N Cin H W Cout pad_mode reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jperceiver_amd._lib import call, lib
aws = torch.empty(int(lib().fn["jp_conv2d_amax_ws_floats"]()), device="cuda")     # scratch for the operand magnitudes the call reduces itself
N, Cin, H, W, Cout, pm, reps = [int(a) for a in sys.argv[1:8]] if len(sys.argv) > 7 else (8, 256, 128, 128, 256, 1, 5)
L = lib()
x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
y = torch.empty(N, Cout, H, W, device="cuda")
ws = torch.empty(int(L.fn["jp_conv2d_ws_floats"](Cin, Cout, 3, 0)), device="cuda")
call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, 3, 1, 1, pm, 2, ws, 0, None, None, None, None, aws, None, None)
for _ in range(reps):
    call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, 3, 1, 1, pm, 2, ws, 1, None, None, None, None, aws, None, None)
torch.cuda.synchronize()
print("done")
