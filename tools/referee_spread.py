#!/usr/bin/env python
"""How far apart are two fp32 evaluations of the SAME step's gradients?  ("settle the gradient referee".)

The step-parity tests compare every device gradient with the fp32 CPU oracle inside a 2 % band and hand the misses to a
float64 referee.  This tool measures, on the CPU alone, the spread of fp32 evaluations around the float64 result: the
oracle is run in float64 once and in fp32 under several summation orders --

    * intra-op thread counts (ATen's conv / reduction kernels split their sums differently),
    * oneDNN on / off (another conv algorithm altogether),
    * every convolution with its input channels permuted (x[:, p], w[:, p]: same mathematical sum, other order)

-- all with the same discrete selections (automask arg-min, CCT arg-max) forced, so only rounding differs.  Per parameter
group it prints each draw's relative distance to float64; tests/golden/referee_spread.json (written with --write) is what
tests/test_step_parity_gpu.py derives its referee bound from.

    python tools/referee_spread.py argo_both_1024_b1 [--write]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
import torch.nn.functional as F                                 # noqa: E402
from tests.golden_util import load_case, run_oracle, run_oracle_f64   # noqa: E402
from oracle import jp_oracle as J                               # noqa: E402

GROUPS = {   # the parameters VERDICT r03 names (scale-3 decoder group) and two controls
    "decoder_scale3": ("DepthDecoder.crp3", "DepthDecoder.merge3", "DepthDecoder.disp3", "DepthDecoder.iconv3", "DepthDecoder.reduce3"),
    "decoder_scale0": ("DepthDecoder.crp1", "DepthDecoder.merge1", "DepthDecoder.disp1", "DepthDecoder.iconv1"),
    "depth_encoder": ("DepthEncoder.",),
    "pose": ("PoseEncoder.", "PoseDecoder."),
}


def main():
    case = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "argo_both_1024_b1"
    g, meta = load_case(case)
    t0 = time.time()
    free = run_oracle(meta)
    force = {("min_index", s): free["out"][("min_index", s)] for s in range(4)}
    for tag in ("road", "car"):
        force["cv_argmax_" + tag] = free["out"]["cv_argmax_" + tag]
        force["cm_argmax_" + tag] = free["out"]["cm_argmax_" + tag]
    label = J.scale_label_both(free["opt"], free["inp"])
    print(f"free-running fp32 oracle: {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    g64 = run_oracle_f64(meta, force, label)
    print(f"float64 oracle: {time.time() - t0:.1f} s", flush=True)

    orig_conv = F.conv2d
    draws = {}

    def run(tag):
        t = time.time()
        o = run_oracle(meta, force=force)
        draws[tag] = {n: p.grad.detach().clone() for n, p in o["P"].items() if p.grad is not None}
        print(f"draw {tag}: {time.time() - t:.1f} s", flush=True)

    nthr = torch.get_num_threads()
    for k in sorted({nthr, max(1, nthr // 2), 3}):
        torch.set_num_threads(k)
        run(f"threads{k}")
    torch.set_num_threads(nthr)
    with torch.backends.mkldnn.flags(enabled=False):
        run("onednn_off")
    for seed in range(1, 9):
        gen = torch.Generator().manual_seed(seed)

        def permuted(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            if groups != 1 or x.shape[1] < 2:
                return orig_conv(x, w, b, stride, padding, dilation, groups)
            p = torch.randperm(x.shape[1], generator=gen)
            return orig_conv(x[:, p], w[:, p], b, stride, padding, dilation, groups)
        F.conv2d = permuted
        try:
            run(f"perm{seed}")
        finally:
            F.conv2d = orig_conv

    def dist(ga, names):
        num = sum(float((ga[n].double() - g64[n]).pow(2).sum()) for n in names)
        den = sum(float(g64[n].pow(2).sum()) for n in names)
        return (num / max(den, 1e-300)) ** 0.5

    report = {"case": case, "draws": list(draws), "groups": {}, "per_parameter": {}}
    for gname, prefixes in GROUPS.items():
        names = [n for n in g64 if n.startswith(prefixes)]
        if names:
            report["groups"][gname] = {t: dist(d, names) for t, d in draws.items()}
    # per parameter: the worst and the median draw, and the spread between two fp32 draws themselves
    for n in g64:
        es = sorted(dist(d, [n]) for d in draws.values())
        report["per_parameter"][n] = {"min": es[0], "median": es[len(es) // 2], "max": es[-1]}
    for gname, d in report["groups"].items():
        print(gname, " ".join(f"{t}={v:.4f}" for t, v in d.items()))
    worst = sorted(report["per_parameter"].items(), key=lambda kv: -kv[1]["max"])[:12]
    for n, d in worst:
        print(f"{n}: min {d['min']:.4f} median {d['median']:.4f} max {d['max']:.4f}  (max/min {d['max'] / max(d['min'], 1e-30):.2f})")
    if "--write" in sys.argv:
        out = os.path.join(ROOT, "tests", "golden", f"referee_spread_{case}.json")
        json.dump(report, open(out, "w"), indent=0)
        print("wrote", out)


if __name__ == "__main__":
    main()
