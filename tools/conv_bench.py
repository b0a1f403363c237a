#!/usr/bin/env python
"""Per-layer timing of ops.conv2d forward + backward at the step's shapes, per kernel instantiation (the library's own
HIP-event profile, jp_profile_*), and the error of every result against a float64 CPU reference on a small shape.

  python tools/conv_bench.py [--iters 5] [--acc]          (A/B: JP_P9S=0 python tools/conv_bench.py ...)
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from jperceiver_amd import ops, _lib  # noqa: E402
from jperceiver_amd.ops import Var, Tape, recording  # noqa: E402
import bench  # noqa: E402

CASES = [
    # label, N, Cin, H, W, Cout, K, stride, pad, pad_mode
    ("merge 256->256 3x3 refl @256^2", 8, 256, 256, 256, 256, 3, 1, 1, 1),
    ("merge 256->256 3x3 refl @128^2", 8, 256, 128, 128, 256, 3, 1, 1, 1),
    ("merge 256->256 3x3 refl @64^2", 8, 256, 64, 64, 256, 3, 1, 1, 1),
    ("CRP 256->256 1x1 @256^2", 8, 256, 256, 256, 256, 1, 1, 0, 0),
    ("CRP 256->256 1x1 @128^2", 8, 256, 128, 128, 256, 1, 1, 0, 0),
    ("layer1 64->64 3x3 @256^2", 8, 64, 256, 256, 64, 3, 1, 1, 0),
    ("layer2 128->128 3x3 @128^2", 8, 128, 128, 128, 128, 3, 1, 1, 0),
    ("layer3 256->256 3x3 @64^2", 8, 256, 64, 64, 256, 3, 1, 1, 0),
    ("layer4 512->512 3x3 @32^2", 8, 512, 32, 32, 512, 3, 1, 1, 0),
    ("pose layer1 64->64 3x3 @48x160 (N = 16)", 16, 64, 48, 160, 64, 3, 1, 1, 0),
    ("layer2.0 64->128 3x3 stride 2 @256^2", 8, 64, 256, 256, 128, 3, 2, 1, 0),
    ("layer3.0 128->256 3x3 stride 2 @128^2", 8, 128, 128, 128, 256, 3, 2, 1, 0),
    ("layer4.0 256->512 3x3 stride 2 @64^2", 8, 256, 64, 64, 512, 3, 2, 1, 0),
    ("stem 3->64 7x7 stride 2 @1024^2", 8, 3, 1024, 1024, 64, 7, 2, 3, 0),
    # small maps of the step (conv_p9sm.hip; A/B with JP_P9SM=0 / 1 / 2)
    ("small: pose layer2 128->128 3x3 @24x80 N16", 16, 128, 24, 80, 128, 3, 1, 1, 0),
    ("small: pose layer3 256->256 3x3 @12x40 N16", 16, 256, 12, 40, 256, 3, 1, 1, 0),
    ("small: pose layer4 512->512 3x3 @6x20 N16", 16, 512, 6, 20, 512, 3, 1, 1, 0),
    ("small: pose squeeze 512->256 1x1 @6x20 N16", 16, 512, 6, 20, 256, 1, 1, 0, 0),
    ("small: pose 256->256 3x3 @6x20 N16", 16, 256, 6, 20, 256, 3, 1, 1, 0),
    ("small: layout conv1 512->128 3x3 refl @32^2", 8, 512, 32, 32, 128, 3, 1, 1, 1),
    ("small: 512->256 3x3 refl @32^2", 8, 512, 32, 32, 256, 3, 1, 1, 1),
    ("small: 256->256 3x3 refl @32^2", 8, 256, 32, 32, 256, 3, 1, 1, 1),
    ("small: 128->128 3x3 refl @16^2", 8, 128, 16, 16, 128, 3, 1, 1, 1),
    ("small: 256->128 3x3 @16^2", 8, 256, 16, 16, 128, 3, 1, 1, 0),
    ("small: 128->64 3x3 @32^2", 8, 128, 32, 32, 64, 3, 1, 1, 0),
    ("small: 64->32 3x3 @64^2", 8, 64, 64, 64, 32, 3, 1, 1, 0),
    ("small: 128->256 3x3 @8^2", 8, 128, 8, 8, 256, 3, 1, 1, 0),
    ("small: 256->256 1x1 @32^2", 8, 256, 32, 32, 256, 1, 1, 0, 0),
    ("small: 128->128 1x1 @8^2", 8, 128, 8, 8, 128, 1, 1, 0, 0),
    # 1x1 layers the patch kernels leave to the generic engine (JP_P9SM=2 takes them)
    ("mode2: reduce 64->256 1x1 @256^2", 8, 64, 256, 256, 256, 1, 1, 0, 0),
    ("mode2: reduce 128->256 1x1 @128^2", 8, 128, 128, 128, 256, 1, 1, 0, 0),
    ("mode2: reduce 256->256 1x1 @64^2", 8, 256, 64, 64, 256, 1, 1, 0, 0),
    ("mode2: 512->256 1x1 @32^2", 8, 512, 32, 32, 256, 1, 1, 0, 0),
]


def profiled(fn):
    L = _lib.lib()
    assert L.fn["jp_profile_begin"](512) == 0
    fn()
    torch.cuda.synchronize()
    n = L.fn["jp_profile_end"]()
    buf, fl, ms = ctypes.create_string_buffer(512), ctypes.c_double(), ctypes.c_float()
    out = []
    for i in range(n):
        L.fn["jp_profile_get"](i, ctypes.cast(buf, ctypes.c_void_p), 512, ctypes.cast(ctypes.pointer(fl), ctypes.c_void_p),
                               ctypes.cast(ctypes.pointer(ms), ctypes.c_void_p))
        out.append((bench._kernel_name(buf.value.decode()), fl.value, ms.value))
    return out


def run_case(c, iters):
    label, N, Cin, H, W, Cout, K, s, p, pm = c
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, K, K, generator=g) * (Cin * K * K) ** -0.5).cuda()
    gy = torch.randn(N, Cout, (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1, generator=g).cuda()
    wv = Var(w, True, torch.zeros_like(w))
    wv.p = torch.nn.Parameter(w)          # persistent pack, as in the model
    wv.t = wv.p.data
    agg = {}
    for it in range(iters + 2):
        xv = Var(x, True)
        tape = Tape()

        def step():
            with recording(tape):
                y = ops.conv2d(xv, wv, None, s, p, pm, 0)
            y.g = gy
            tape.backward()
        recs = profiled(step)
        if it < 2:
            continue
        for name, fl, ms in recs:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += fl
    alg = 2.0 * N * ((H + 2 * p - K) // s + 1) * ((W + 2 * p - K) // s + 1) * Cout * Cin * K * K
    print(f"== {label}: algorithmic {alg / 1e9:.1f} GFLOP per pass")
    for name, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        per = ms / iters
        print(f"   {per:8.3f} ms/step  {alg / per / 1e9 if per > 0 else 0:7.1f} alg-TF (if one pass)  exec {fl / iters / per / 1e9:7.1f} TF  x{n // iters}  {name[:110]}")


def accuracy():
    """error vs float64 of fwd / dgrad / wgrad on 8x128x64x64 -> 128 (3x3 zero pad and 1x1): a shape the patch kernels take"""
    for K, p in ((3, 1), (1, 0)):
        g = torch.Generator().manual_seed(3)
        N, Cin, H, W, Cout = 8, 128, 64, 64, 128
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, K, K, generator=g) * (Cin * K * K) ** -0.5
        gy = torch.randn(N, Cout, H, W, generator=g)
        xv, wv = Var(x.cuda(), True), Var(w.cuda(), True, torch.zeros_like(w).cuda())
        tape = Tape()

        def step():
            with recording(tape):
                step.y = ops.conv2d(xv, wv, None, 1, p, 0, 0)
            step.y.g = gy.cuda()
            tape.backward()
        print("   kernels:", sorted({r[0] for r in profiled(step)}))
        y = step.y
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yd = F.conv2d(xd, wd, None, 1, p)
        yd.backward(gy.double())
        xf, wf = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yf = F.conv2d(xf, wf, None, 1, p)
        yf.backward(gy)
        for nm, got, ref, cpu in (("fwd", y.t, yd, yf), ("dgrad", xv.g, xd.grad, xf.grad), ("wgrad", wv.g, wd.grad, wf.grad)):
            e = float((got.cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
            ec = float((cpu.detach().double() - ref.detach()).abs().max() / ref.detach().abs().max())
            er = float((got.cpu().double() - ref.detach()).pow(2).mean().sqrt() / ref.detach().pow(2).mean().sqrt())
            ecr = float((cpu.detach().double() - ref.detach()).pow(2).mean().sqrt() / ref.detach().pow(2).mean().sqrt())
            print(f"   K={K} {nm:5s}: max err / max |ref| = {e:.3e} (CPU ATen fp32: {ec:.3e});  rms rel {er:.3e} (CPU {ecr:.3e})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--acc", action="store_true")
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    print("JP_P9S =", os.environ.get("JP_P9S", "(default)"), " JP_P9SM =", os.environ.get("JP_P9SM", "(default)"))
    if a.acc:
        accuracy()
    for c in CASES:
        if a.only and not any(pat in c[0] for pat in a.only.split(",")):
            continue
        run_case(c, a.iters)
