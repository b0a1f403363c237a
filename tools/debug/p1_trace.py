"""cycle stamps of wave 0 of one workgroup of the 1x1 split-bf16 patch kernel (library built with -DP9S_TRACE, JP_LIB_PATH):
where a 256->256 @256^2 tile spends its time.  Stamps: 0 start, 1 prologue done (first patch + weights requested); per stage s:
2+4s before lstore, 3+4s after lstore (= the stage's input has arrived and is split), 4+4s after barrier 1, 5+4s after the
step loop was issued; 34 K loop done, 35 epilogue done."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jperceiver_amd import ops, _lib
from jperceiver_amd.ops import Var, Tape, recording
g = torch.Generator().manual_seed(1)
N, C, H, W = 8, 256, 256, 256
x, w = torch.randn(N, C, H, W, generator=g).cuda(), (torch.randn(C, C, 1, 1, generator=g) * C ** -0.5).cuda()
wv = Var(w, True, torch.zeros_like(w))
for it in range(3):
    with recording(Tape()):
        y = ops.conv2d(Var(x), wv, None, 1, 0, 0, 0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
f = _lib.lib().cdll.dbg_p9s_trace
f.argtypes = [ctypes.c_void_p]
assert f(ctypes.cast(buf, ctypes.c_void_p)) == 0
t = list(buf)
print("JP_P1_KGS =", os.environ.get("JP_P1_KGS", "2"))
print("raw deltas from start (cycles of the shader clock counter):")
names = {0: "start", 1: "prologue issued", 34: "K loop done", 35: "epilogue done"}
prev = t[0]
for i, v in enumerate(t):
    if v == 0:
        continue
    if 2 <= i < 34:
        s, k = divmod(i - 2, 4)
        nm = f"stage {s} " + ["before lstore", "after lstore (input arrived + split)", "after barrier 1", "step loop issued"][k]
    else:
        nm = names.get(i, "")
    print(f"  [{i:2d}] {v - t[0]:8d}  (+{v - prev:6d})  {nm}")
    prev = v
