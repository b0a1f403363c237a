#!/usr/bin/env python
"""Telemetry for the fp16 two-way split arithmetic: how far below its tensor's largest magnitude
does the data of every convolution operand of the REAL training step sit?

One power-of-two scale per operand tensor carries an element to 2^-22 relative while it is within 2^-17 of the tensor's largest
magnitude and to 2^-40 OF THAT LARGEST below (include/jperceiver_hip.h), so the numbers that matter per operand tensor are
    max / median of the non-zero magnitudes,
    the share of non-zero elements below 2^-17 max  (carried with fewer than 22 bits),
    the share below 2^-29 max                       (carried at fp16 grade or worse: < 11 bits),
weighted by how much of the operand's energy (sum of squares) those elements hold -- what they can contribute to an output.
Recorded on the benchmark's own step (configs[1], B = 8, 1024^2, random-init weights) at iteration 1 and after `--train` Adam
iterations on the synthetic batches (the gradients' spread changes once the loss terms move), for every tensor handed to a
jp_conv2d_* entry point: activations (forward / weight gradient), output gradients (dgrad / weight gradient), weights.
usage (GPU box): python tools/operand_ranges.py [--batch 8] [--hw 1024] [--train 40] > profiles/r06_operand_ranges.md"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                   # noqa: E402
from jperceiver_amd import ops, ops_loss, runtime as rt                        # noqa: E402
from jperceiver_amd.model import net as netmod, modules as mods                # noqa: E402
from jperceiver_amd import _lib                                                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--hw", type=int, default=1024)
ap.add_argument("--train", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[1]
B, HW, frames = args.batch, args.hw, cfg["frames"]
optd = bench.make_opt(B, HW, HW, frames, cfg["type"], cfg["split"], loss_sum=cfg["loss_sum"])
runner, batch = bench.build_runner(optd, dev, 1, 0, dict(B=B, height=HW, width=HW, frame_ids=frames, occ=HW // 4,
                                                         full_hw=cfg["full_hw"], split=cfg["split"], seed=1))

# operand tensors of the conv entry points: (role, argument index) per entry point, ABI argument order
ROLES = {
    "jp_conv2d_fwd_src3": [("x", 0), ("x", 3), ("x", 6), ("w", 9)],
    "jp_conv2d_dgrad": [("dy", 0), ("w", 1)],
    "jp_conv2d_dgrad_src3": [("dy", 0), ("w", 1)],
    "jp_conv2d_wgrad_src3": [("x", 0), ("x", 3), ("x", 6), ("dy", 9)],
}
seen, rows = set(), []


def stats(t):
    a = t.detach().abs().flatten().float()
    a = a[torch.isfinite(a)]
    nz = a[a > 0]
    if nz.numel() == 0:
        return None
    mx = float(nz.max())
    samp = nz[:: max(1, nz.numel() // 2_000_000)]
    med = float(samp.median())
    e = nz.double().pow(2)
    tot = float(e.sum())
    lo17, lo29 = nz < mx * 2.0 ** -17, nz < mx * 2.0 ** -29
    return dict(n=a.numel(), zeros=1.0 - nz.numel() / a.numel(), max=mx, ratio=mx / med, f17=float(lo17.double().mean()),
                f29=float(lo29.double().mean()), e17=float(e[lo17].sum()) / tot, e29=float(e[lo29].sum()) / tot)


orig = _lib.call


def spy(name, *a):
    if name in ROLES:
        for role, i in ROLES[name]:
            t = a[i]
            if isinstance(t, torch.Tensor) and t.numel() >= 4096:
                key = (role, t.data_ptr(), tuple(t.shape))
                if key not in seen:
                    seen.add(key)
                    s = stats(t)
                    if s is not None:
                        rows.append((role, tuple(t.shape), s))
    return orig(name, *a)


def record(title):
    seen.clear()
    rows.clear()
    for m in (ops, ops_loss, netmod, rt, mods):
        if hasattr(m, "call"):
            m.call = spy
    try:
        runner.train_iter(batch)
        torch.cuda.synchronize()
    finally:
        for m in (ops, ops_loss, netmod, rt, mods):
            if hasattr(m, "call"):
                m.call = orig
    print(f"\n## {title}: {len(rows)} operand tensors\n")
    for role in ("x", "dy", "w"):
        rr = [r for r in rows if r[0] == role]
        if not rr:
            continue
        w17 = max(rr, key=lambda r: r[2]["f17"])
        w29 = max(rr, key=lambda r: r[2]["f29"])
        we = max(rr, key=lambda r: r[2]["e17"])
        wr = max(rr, key=lambda r: r[2]["ratio"])
        print(f"* **{role}** ({len(rr)} tensors): largest max/median {wr[2]['ratio']:.3g} {wr[1]}; largest share of non-zero elements below "
              f"2^-17 max {w17[2]['f17']:.2e} {w17[1]} / below 2^-29 max {w29[2]['f29']:.2e} {w29[1]}; largest share of the tensor's ENERGY "
              f"held by elements below 2^-17 max {we[2]['e17']:.2e} {we[1]}")
    print("\n| role | shape | zeros | max | max / median | share < 2^-17 max | share < 2^-29 max | energy < 2^-17 max | energy < 2^-29 max |")
    print("|---|---|---|---|---|---|---|---|---|")
    for role, shape, s in sorted(rows, key=lambda r: -r[2]["ratio"])[:40]:
        print(f"| {role} | {'x'.join(map(str, shape))} | {s['zeros']:.2f} | {s['max']:.3g} | {s['ratio']:.3g} | {s['f17']:.2e} | {s['f29']:.2e} | "
              f"{s['e17']:.2e} | {s['e29']:.2e} |")


print(f"# Operand ranges of the convolution operands, bench step (configs[1], B = {B}, {HW}^2), library arithmetic scheme {ops.split_scheme()}")
print("\n(tools/operand_ranges.py; one row per distinct operand tensor of one training iteration, the 40 with the largest max / median; "
      "`share` = of the non-zero elements, `energy` = of the tensor's sum of squares)")
runner.train_iter(batch)            # iteration 0: first-use packs etc.
record("iteration 1 (random-init weights)")
for _ in range(args.train):
    runner.train_iter(batch)
record(f"iteration {args.train + 2} (after {args.train + 1} Adam steps on the synthetic batch)")
