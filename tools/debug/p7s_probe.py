"""Debug aid: one-hot weight probes of the 7x7 stem kernel (which (c, ky, kx) taps / output positions disagree with ATen)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from jperceiver_amd import ops
from jperceiver_amd.ops import Var, Tape, recording
Cin = int(os.environ.get("CIN", "3"))
g = torch.Generator().manual_seed(1)
x = torch.randn(4, Cin, 256, 512, generator=g)
bad = []
for c in range(Cin):
    for ky in range(0, 7, 3):
        for kx in range(0, 7, 2):
            w = torch.zeros(64, Cin, 7, 7)
            w[(c * 49 + ky * 7 + kx) % 64, c, ky, kx] = 1.0
            with recording(Tape()):
                y = ops.conv2d(Var(x.cuda()), Var(w.cuda()), None, 2, 3, 0, 0).t.cpu()
            ref = F.conv2d(x, w, None, 2, 3)
            e = (y - ref).abs()
            if float(e.max()) > 1e-5:
                idx = (e > 1e-5).nonzero()
                bad.append((c, ky, kx, float(e.max()), idx[:3].tolist(), int((e > 1e-5).sum())))
print("bad taps:", len(bad))
for b in bad[:30]:
    print(b)
