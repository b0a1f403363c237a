#!/usr/bin/env python
"""tools/referee_spread.py for the config-step cases of tests/test_config_steps_gpu.py (no golden file: inputs and weights are
generated from the case's flags): the spread of fp32 CPU evaluations of ONE step around its float64 evaluation, per parameter.
The discrete selections (automask arg-min, CCT arg-max) of the free-running fp32 oracle are forced in every evaluation and the
oracle's own scale label is used, so only the summation order differs between draws.

    python tools/referee_spread_cfg.py cfg1_full_B8_1024 [--draws 8] [--write]   -> tests/golden/referee_spread_<case>.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
import torch.nn.functional as F                                 # noqa: E402
from jperceiver_amd import synthetic as syn                     # noqa: E402
from jperceiver_amd.model import MONO                           # noqa: E402
from oracle import jp_oracle as J                               # noqa: E402

CONFIGS = {   # (tests/test_config_steps_gpu.py::CONFIGS)
    "cfg1_full_B8_1024": dict(HW=1024, B=8, FR=[0, -1, 1], type="static", split="odometry", loss_sum=3, full_hw=(375, 1242), seed=1),
    "cfg0_odometry_B1_1024": dict(HW=1024, B=1, FR=[0, -1], type="static", split="odometry", loss_sum=3, full_hw=(375, 1242), seed=21),
    "cfg1_odometry_1024_20": dict(HW=512, B=3, FR=[0, -1, 1], type="static", split="odometry", loss_sum=3, full_hw=(375, 1242), seed=22),
}


def main():
    case = sys.argv[1]
    ndraw = int(sys.argv[sys.argv.index("--draws") + 1]) if "--draws" in sys.argv else 8
    c = CONFIGS[case]
    opt = J.default_opt(frame_ids=c["FR"], imgs_per_gpu=c["B"], height=c["HW"], width=c["HW"], occ_map_size=c["HW"] // 4,
                        type=c["type"], split=c["split"], loss_sum=c["loss_sum"])
    HW, B, FR = c["HW"], c["B"], c["FR"]
    state = syn.synth_state_dict(MONO.module_dict["Baseline"](opt).state_dict(), seed=0)
    inp = syn.make_batch(B, HW, HW, FR, HW // 4, c["full_hw"], c["split"], seed=c["seed"])
    masks = syn.make_dropout_masks(B, HW, HW, seed=c["seed"])
    noise = syn.make_automask_noise(B, HW, HW, 4, len(FR) - 1, seed=c["seed"])
    label = torch.nan_to_num(J.scale_label_static(opt, inp, True)[0], nan=0.0, posinf=0.0, neginf=0.0)
    shapes = J.state_shapes(HW // 4)

    def run32(force):
        P, Bf = J.make_params(shapes, state)
        out, L = J.forward(P, Bf, opt, inp, True, masks, noise, label, force)
        J.total_loss(L).backward()
        return out, {n: p.grad.detach().clone() for n, p in P.items() if p.grad is not None}

    t0 = time.time()
    out, _ = run32(None)
    force = {("min_index", s): out[("min_index", s)] for s in range(4)}
    for tag in ("road", "car"):
        force["cv_argmax_" + tag], force["cm_argmax_" + tag] = out["cv_argmax_" + tag], out["cm_argmax_" + tag]
    print(f"free-running fp32 oracle: {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    P64, B64 = {}, {}
    for n in shapes:
        t = state[n].clone()
        if J.is_buffer(n):
            B64[n] = t.double() if t.dtype == torch.float32 else t
        else:
            P64[n] = t.double().requires_grad_(True)
    inp64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in inp.items()}
    torch.set_default_dtype(torch.float64)
    try:
        _, L = J.forward(P64, B64, opt, inp64, True, tuple(m.double() for m in masks), [[z.double() for z in per] for per in noise],
                         label.double(), force)
        J.total_loss(L).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    g64 = {n: p.grad for n, p in P64.items() if p.grad is not None}
    del P64, B64, inp64
    print(f"float64 oracle: {time.time() - t0:.1f} s", flush=True)

    draws, orig_conv = {}, F.conv2d

    def run(tag):
        t = time.time()
        draws[tag] = run32(force)[1]
        print(f"draw {tag}: {time.time() - t:.1f} s", flush=True)
    nthr = torch.get_num_threads()
    for k in sorted({nthr, max(1, nthr // 2)}):
        torch.set_num_threads(k)
        run(f"threads{k}")
    torch.set_num_threads(nthr)
    with torch.backends.mkldnn.flags(enabled=False):
        run("onednn_off")
    for seed in range(1, max(1, ndraw - 3) + 1):
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

    def dist(ga, n):
        return float((ga[n].double() - g64[n]).norm() / max(float(g64[n].norm()), 1e-300))
    report = {"case": case, "draws": list(draws), "per_parameter": {}}
    for n in g64:
        es = sorted(dist(d, n) for d in draws.values())
        report["per_parameter"][n] = {"min": es[0], "median": es[len(es) // 2], "max": es[-1]}
    worst = sorted(((n, d) for n, d in report["per_parameter"].items() if float(g64[n].norm()) > 0), key=lambda kv: -kv[1]["max"])[:16]
    for n, d in worst:
        print(f"{n}: min {d['min']:.4f} median {d['median']:.4f} max {d['max']:.4f}  (max/min {d['max'] / max(d['min'], 1e-30):.2f})")
    if "--write" in sys.argv:
        out_f = os.path.join(ROOT, "tests", "golden", f"referee_spread_{case}.json")
        json.dump(report, open(out_f, "w"), indent=0)
        print("wrote", out_f)


if __name__ == "__main__":
    main()
