"""Accuracy of the split-product kernels (igemm_p9s.h / igemm_w9s.h / igemm_p9us2.h / igemm_p9sd.h / igemm_w4s.h / igemm_p9s2*.h)
against FLOAT64.

The patch convolutions form every fp32 product on the bf16 matrix pipe as 6 bf16 products of exact three-way bf16 splits
of both operands, accumulated in fp32.  The claim that makes this "fp32 arithmetic" rather than reduced precision is
checked here, per kernel family, on well-scaled AND on wide-dynamic-range operands:

  (a) against the float64 result, the split kernel's error is no larger than the EXACT-fp32 MFMA kernel's (the same library
      with JP_P9S=0 JP_W9S=0 JP_P9US=0: `v_mfma_f32_32x32x2_f32`, bit-identical to an fmaf chain) -- rms within 1.25x, max
      within 2x (the max of ~10^6 samples fluctuates);
  (b) absolutely: |error| <= 2^-19 * sum_k |a_k| |b_k| for every output element (an fp32 FMA chain of K terms is bounded by
      ~K * 2^-24 of that sum; the layers here have K = 256 ... 4617; measured worst cases on the wide-range operands: split
      1.3e-6, exact-fp32 FMA chain 1.7e-6).

Each variant runs in its own process (the library reads the JP_* switches once)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _emit():
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn.functional as F
    from jperceiver_amd import ops
    from jperceiver_amd.ops import Var, Tape, recording
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    res = {}

    def data(shape, g, wide):
        t = torch.randn(shape, generator=g)
        if wide:        # per-element random power-of-two scale: 2^-6 .. 2^6
            t = t * torch.pow(2.0, torch.randint(-6, 7, shape, generator=g).float())
        return t

    def metrics(got, ref64, bound64):
        e = (got.double() - ref64).abs()
        return dict(rms=float(e.pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt()), max=float(e.max() / ref64.abs().max()),
                    rel_bound=float((e / bound64.clamp_min(1e-300)).max()))

    def conv_case(name, N, Cin, H, W, Cout, K, pad, pm, wide, stride=1):
        g = torch.Generator().manual_seed(11)
        x, w = data((N, Cin, H, W), g, wide), data((Cout, Cin, K, K), g, wide) * (Cin * K * K) ** -0.5
        gy = data((N, Cout, (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1), g, wide)
        xv, wv = Var(x.cuda(), True), Var(w.cuda(), True, torch.zeros_like(w).cuda())
        tape = Tape()
        with recording(tape):
            y = ops.conv2d(xv, wv, None, stride, pad, pm, 0)
        y.g = gy.cuda()
        tape.backward()
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        pad_ = (lambda t: F.pad(t, (pad,) * 4, mode="reflect")) if (pm == 1 and pad) else (lambda t: t)
        yd = F.conv2d(pad_(xd), wd, None, stride, 0 if pm == 1 else pad)
        yd.backward(gy.double())
        # sum |a||b| bounds of the three GEMMs
        xa, wa, ga = x.double().abs().requires_grad_(True), w.double().abs().requires_grad_(True), gy.double().abs()
        ya = F.conv2d(pad_(xa), wa, None, stride, 0 if pm == 1 else pad)
        ya.backward(ga)
        res[name + "/fwd"] = metrics(y.t.cpu(), yd.detach(), ya.detach())
        res[name + "/dgrad"] = metrics(xv.g.cpu(), xd.grad, xa.grad)
        res[name + "/wgrad"] = metrics(wv.g.cpu(), wd.grad, wa.grad)

    for wide in (False, True):
        tag = "wide" if wide else "unit"
        conv_case(f"3x3_zero_128_{tag}", 8, 128, 64, 64, 128, 3, 1, 0, wide)
        conv_case(f"3x3_reflect_256_{tag}", 8, 256, 64, 64, 256, 3, 1, 1, wide)
        conv_case(f"1x1_256_{tag}", 8, 256, 64, 64, 256, 1, 0, 0, wide)
        conv_case(f"3x3_stride2_64_128_{tag}", 8, 64, 128, 128, 128, 3, 1, 0, wide, stride=2)      # P9S2F / P9S2D / W9S2<2>
        conv_case(f"3x3_stride2_128_256_{tag}", 8, 128, 64, 64, 256, 3, 1, 0, wide, stride=2)      # ... / W9S2<1>
        conv_case(f"7x7_stem_{tag}", 4, 3, 256, 256, 64, 7, 3, 0, wide, stride=2)                  # P7S
        if not wide:    # the stems at the shapes the step issues (VERDICT r03 "weak" 1): depth 8x3x1024^2, pose pairs 8x6x192x640
            conv_case("7x7_stem_bench_depth_unit", 8, 3, 1024, 1024, 64, 7, 3, 0, False, stride=2)
            conv_case("7x7_stem_bench_pose_unit", 8, 6, 192, 640, 64, 7, 3, 0, False, stride=2)
        # iconv (P9US): cat(skip 64, up2x(x 96), disp 1) -> 256, reflect
        g = torch.Generator().manual_seed(12)
        N, H, W, Cr, Cx, Cout = 2, 64, 128, 64, 96, 256
        r, xh, d = data((N, Cr, H, W), g, wide), data((N, Cx, H // 2, W // 2), g, wide), data((N, 1, H, W), g, wide)
        w = data((Cout, Cr + Cx + 1, 3, 3), g, wide) * (9 * (Cr + Cx + 1)) ** -0.5
        with recording(Tape()):
            y = ops.conv2d(None, Var(w.cuda()), None, 1, 1, 1, 0, srcs=[(Var(r.cuda()), 0), (Var(xh.cuda()), 1), (Var(d.cuda()), 0)])
        cat = torch.cat((r.double(), F.interpolate(xh.double(), scale_factor=2, mode="nearest"), d.double()), 1)
        yd = F.conv2d(F.pad(cat, (1, 1, 1, 1), mode="reflect"), w.double())
        ya = F.conv2d(F.pad(cat.abs(), (1, 1, 1, 1), mode="reflect"), w.double().abs())
        res[f"iconv_{tag}/fwd"] = metrics(y.t.cpu(), yd, ya)
        # iconv backward of the upsampled segment (P9SD dgrad at half resolution, W4S weight gradient): cat(skip 128, up2x(x 128))
        g = torch.Generator().manual_seed(13)
        N, H, W, Cr, Cx, Cout = 4, 128, 128, 128, 128, 256
        r, xh = data((N, Cr, H, W), g, wide), data((N, Cx, H // 2, W // 2), g, wide)
        w = data((Cout, Cr + Cx, 3, 3), g, wide) * (9 * (Cr + Cx)) ** -0.5
        gy = data((N, Cout, H, W), g, wide)
        rv_, xv_, wv_ = Var(r.cuda(), True), Var(xh.cuda(), True), Var(w.cuda(), True, torch.zeros_like(w).cuda())
        tape = Tape()
        with recording(tape):
            y = ops.conv2d(None, wv_, None, 1, 1, 1, 0, srcs=[(rv_, 0), (xv_, 1)])
        y.g = gy.cuda()
        tape.backward()
        xd, wd = xh.double().requires_grad_(True), w.double().requires_grad_(True)
        cat = torch.cat((r.double(), F.interpolate(xd, scale_factor=2, mode="nearest")), 1)
        F.conv2d(F.pad(cat, (1, 1, 1, 1), mode="reflect"), wd).backward(gy.double())
        xa, wa = xh.double().abs().requires_grad_(True), w.double().abs().requires_grad_(True)
        cata = torch.cat((r.double().abs(), F.interpolate(xa, scale_factor=2, mode="nearest")), 1)
        F.conv2d(F.pad(cata, (1, 1, 1, 1), mode="reflect"), wa).backward(gy.double().abs())
        res[f"iconv_up_{tag}/dgrad"] = metrics(xv_.g.cpu(), xd.grad, xa.grad)
        res[f"iconv_up_{tag}/wgrad"] = metrics(wv_.g.cpu()[:, Cr:], wd.grad[:, Cr:], wa.grad[:, Cr:])
    print("JSON" + json.dumps(res))


def _emit_edges():
    """Range edges of the three-way split (`jp_split3`, igemm_p9s.h:33-45): non-finite and out-of-bf16-range inputs, tiny inputs."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn.functional as F
    from jperceiver_amd import ops
    from jperceiver_amd.ops import Var, Tape, recording
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    res = {}
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 8, 128, 64, 64             # the shape of the accuracy cases above: runs jp_igemm_p9s_kernel / jp_igemm_p9_kernel
    w = torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5
    from tests.test_bench_shapes_gpu import kernel_tags

    def fwd(x):
        with recording(Tape()):
            return ops.conv2d(Var(x.cuda()), Var(w.cuda()), None, 1, 1, 0, 0).t.cpu()
    x0 = torch.randn((N, C, H, W), generator=g)
    with kernel_tags() as kt:
        y0 = fwd(x0)
    res["kernels"] = sorted(set(kt.names))
    # 3.40e38: above the rounding threshold to the largest bf16 (0x7F7F8000 = 3.3962e38), below FLT_MAX (3.4028e38)
    for name, val in (("pinf", float("inf")), ("ninf", float("-inf")), ("nan", float("nan")), ("over_bf16", 3.40e38)):
        x = x0.clone()
        x[1, 7, 10, 20] = val
        y = fwd(x)
        touched = torch.zeros((N, 1, H, W), dtype=torch.bool)
        touched[1, 0, 9:12, 19:22] = True
        touched = touched.expand(N, C, H, W)
        res[name] = dict(nonfinite_touched=int((~torch.isfinite(y[touched])).sum()), n_touched=int(touched.sum()),
                         nonfinite_elsewhere=int((~torch.isfinite(y[~touched])).sum()),
                         elsewhere_bit_identical=bool(torch.equal(y[~touched], y0[~touched])),
                         n_posinf=int((y[touched] == float("inf")).sum()), n_neginf=int((y[touched] == float("-inf")).sum()),
                         n_nan=int(torch.isnan(y[touched]).sum()))
    for name, e in (("tiny_2^-100", -100), ("tiny_2^-120", -120)):
        x = x0 * 2.0 ** e
        y = fwd(x)
        yd = F.conv2d(x.double(), w.double(), None, 1, 1)
        ya = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
        res[name] = dict(rel_bound=float(((y.double() - yd).abs() / ya).max()))
    print("JSON" + json.dumps(res))


def _emit_outliers():
    """Where the fp16 two-way split with ONE power-of-two scale per tensor differs from fp32 arithmetic: operand
    tensors whose largest magnitude sits far above their typical one.  Per output element, against float64:
        ratio  = |err| / sum_k |a_k| |b_k|                                   (fp32 arithmetic: <= ~K 2^-24 whatever the data)
        ratioT = |err| / (sum|a||b| + 2^-19 (A sum_k|b_k| + B sum_k|a_k|))    A, B = the operand tensors' largest magnitudes
    The arithmetic guarantees ratioT <= 2^-19 (an element is carried to 2^-22 relative OR 2^-40 of its tensor's largest, whichever is
    larger); ratio <= 2^-19 holds while the typical element is within ~2^20 of the largest."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn.functional as F
    from jperceiver_amd import ops
    from jperceiver_amd.ops import Var, Tape, recording
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    res = {}
    N, C, H, W, K = 8, 128, 64, 64, 3

    def stats(got, ref, sab, sa_B, sb_A, keep=None):
        e = (got.double() - ref).abs()
        if keep is not None:
            e, sab, sa_B, sb_A, ref = e[keep], sab[keep], sa_B[keep], sb_A[keep], ref[keep]
        ratio = e / sab.clamp_min(1e-300)
        ratioT = e / (sab + 2.0 ** -19 * (sa_B + sb_A)).clamp_min(1e-300)
        return dict(max_ratio=float(ratio.max()), frac_over=float((ratio > 2.0 ** -19).double().mean()),
                    max_ratioT=float(ratioT.max()), rms=float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))

    def run(name, x, w, gy, keep_fwd=None, keep_wg=None):
        xv, wv = Var(x.cuda(), True), Var(w.cuda(), True, torch.zeros_like(w).cuda())
        tape = Tape()
        with recording(tape):
            y = ops.conv2d(xv, wv, None, 1, 1, 0, 0)
        y.g = gy.cuda()
        tape.backward()
        xd, wd, gd = x.double(), w.double(), gy.double()
        A, Bw, G = float(x.abs().max()), float(w.abs().max()), float(gy.abs().max())
        one_x, one_w, one_g = torch.ones_like(xd), torch.ones_like(wd), torch.ones_like(gd)
        conv = lambda a, b: F.conv2d(a, b, None, 1, 1)
        dgr = lambda g_, w_: torch.nn.grad.conv2d_input(xd.shape, w_, g_, 1, 1)
        wgr = lambda x_, g_: torch.nn.grad.conv2d_weight(x_, wd.shape, g_, 1, 1)
        res[name + "/fwd"] = stats(y.t.cpu(), conv(xd, wd), conv(xd.abs(), wd.abs()), conv(xd.abs(), one_w) * Bw, conv(one_x, wd.abs()) * A,
                                   keep_fwd)
        res[name + "/dgrad"] = stats(xv.g.cpu(), dgr(gd, wd), dgr(gd.abs(), wd.abs()), dgr(gd.abs(), one_w) * Bw, dgr(one_g, wd.abs()) * G)
        res[name + "/wgrad"] = stats(wv.g.cpu(), wgr(xd, gd), wgr(xd.abs(), gd.abs()), wgr(xd.abs(), one_g) * G, wgr(one_x, gd.abs()) * A,
                                     keep_wg)

    g = torch.Generator().manual_seed(21)
    w = torch.randn((C, C, K, K), generator=g) * (9 * C) ** -0.5
    base = torch.randn((N, C, H, W), generator=g).clamp_min(0)          # ReLU activations: half zeros, median of the rest 0.67
    gy0 = torch.randn((N, C, H, W), generator=g)
    for tag, mag in (("1e4", 1e4), ("1e6", 1e6)):
        x = base.clone()
        x[3, 17, 30, 40] = mag * 0.67
        kf = torch.ones((N, C, H, W), dtype=torch.bool)
        kf[3, :, 29:32, 39:42] = False                                  # outputs whose 3x3 window holds the outlier
        kw = torch.ones((C, C, K, K), dtype=torch.bool)
        kw[:, 17] = False                                               # weight gradients of the outlier's input channel
        run(f"act_outlier_{tag}", x, w, gy0, kf, kw)
    for sig in (3.0, 4.0):
        gy = torch.exp(sig * torch.randn((N, C, H, W), generator=g)) * torch.sign(torch.randn((N, C, H, W), generator=g))
        run(f"grad_lognormal_sigma{sig:.0f}", base + 0.01, w, gy)
    x = base.clone() + 0.01
    x[:, 5] *= 2.0 ** -20
    kw = torch.zeros((C, C, K, K), dtype=torch.bool)
    kw[:, 5] = True
    run("channel_2^-20_below/others", x, w, gy0)
    run("channel_2^-20_below/that_channel", x, w, gy0, None, kw)
    # a weight tensor with one filter 2^12 above the rest (the per-tensor weight scale costs the other filters bits)
    w2 = w.clone()
    w2[9] *= 4096.0
    kf = torch.ones((N, C, H, W), dtype=torch.bool)
    kf[:, 9] = False
    run("filter_2^12_above", base + 0.01, w2, gy0, kf)
    print("JSON" + json.dumps(res))


def _run(env_extra, mode="--emit"):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=1800, env=env,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")][-1]
    return json.loads(line[4:])


def test_split_product_accuracy_vs_float64():
    split = _run(dict(JP_P9S="1", JP_W9S="1", JP_P9US="1", JP_P9SD="1", JP_P9S2="1"))
    exact = _run(dict(JP_P9S="0", JP_W9S="0", JP_P9US="0", JP_P9SD="0", JP_P9S2="0"))
    assert set(split) == set(exact)
    report = []
    for k in sorted(split):
        s, e = split[k], exact[k]
        report.append(f"{k}: rms {s['rms']:.2e} (exact fp32 {e['rms']:.2e})  max {s['max']:.2e} ({e['max']:.2e})  "
                      f"max err/sum|a||b| {s['rel_bound']:.2e} ({e['rel_bound']:.2e})")
    print("\n".join(report))
    for k in sorted(split):
        s, e = split[k], exact[k]
        assert s["rms"] <= 1.25 * e["rms"] + 1e-9, (k, s, e)
        assert s["max"] <= 2.0 * e["max"] + 1e-9, (k, s, e)
        assert s["rel_bound"] <= 2.0 ** -19, (k, s)
        assert e["rel_bound"] <= 2.0 ** -19, (k, e)


def test_split_range_edges_match_documented_behaviour():
    """VERDICT r03 "weak" 2.  What include/jperceiver_hip.h states for the split-product convolutions, next to the
    exact-fp32 kernels on the same inputs.  (Written for the six-product bf16 scheme; the fp16 two-way scheme of round 5 meets the
    same assertions for different reasons: Inf / NaN / |x| >= 2^100 inputs are kept out of the operand scale and overflow fp16 ->
    NaN for the outputs that read them, everything else bit-identical; tiny tensors are lifted by their scale.)
      * a NaN / +-Inf input poisons exactly the outputs whose window contains it, in both; the exact kernel yields +-Inf (sign
        of the weight) for an Inf input, the split kernel NaN (Inf - bf16(Inf) = NaN in the residual splits) -- non-finite
        either way, everything else bit-identical to the unpoisoned run;
      * a finite input above the largest bf16 (3.3895e38 < |x| <= FLT_MAX, the top 0.4 % of the fp32 range) rounds to Inf in
        its high split: non-finite outputs in the split kernel where the exact one may stay finite;
      * tiny inputs: full accuracy while every split is a normal bf16 (|x| >= 2^-109); below that the low splits flush and
        the relative accuracy degrades towards the first split's 2^-9 -- the absolute error stays below 2^-126 |w|."""
    split = _run(dict(JP_P9S="1"), "--edges")
    exact = _run(dict(JP_P9S="0"), "--edges")
    print(json.dumps(dict(split=split, exact=exact), indent=1))
    assert any("jp_igemm_p9s_kernel" in k for k in split.pop("kernels")), "the split-bf16 kernel did not run"
    assert any("jp_igemm_p9_kernel" in k for k in exact.pop("kernels")), "the exact-fp32 kernel did not run"
    for k in ("pinf", "ninf", "nan", "over_bf16"):
        for r in (split[k], exact[k]):
            assert r["nonfinite_elsewhere"] == 0 and r["elsewhere_bit_identical"], (k, r)
    for k in ("pinf", "ninf", "nan"):
        assert split[k]["nonfinite_touched"] == split[k]["n_touched"] == exact[k]["nonfinite_touched"], (k, split[k], exact[k])
    assert exact["pinf"]["n_posinf"] + exact["pinf"]["n_neginf"] == exact["pinf"]["n_touched"]      # fp32: +-Inf by weight sign
    assert split["pinf"]["n_nan"] == split["pinf"]["n_touched"]                                     # split: NaN (documented)
    assert exact["over_bf16"]["nonfinite_touched"] == 0                                             # 3.396e38 * |w| <= 0.1 stays finite
    assert split["over_bf16"]["nonfinite_touched"] == split["over_bf16"]["n_touched"]               # documented range limit
    assert split["tiny_2^-100"]["rel_bound"] <= 2.0 ** -19 and exact["tiny_2^-100"]["rel_bound"] <= 2.0 ** -19
    assert split["tiny_2^-120"]["rel_bound"] <= 2.0 ** -7, split["tiny_2^-120"]


def test_split_accuracy_where_one_scale_per_tensor_differs_from_fp32():
    """activation outliers of 10^4 / 10^6 x the median, log-normal gradients (sigma 3 / 4), one channel 2^20 below the
    rest, one filter 2^12 above the rest -- on the GPU kernels, per output element, next to the exact-fp32 MFMA kernels on the same data
    (an fp32 accumulation of K = 1152 heavy-tailed terms is itself 3.4e-6 of sum|a||b| off: the element-wise yardstick is the exact
    kernel's own figure, not a constant).  Measured (profiles/r06_split_outliers.md): the three-product kernels are CLOSER to float64 than
    the exact-fp32 kernels in 17 of the 21 (case, pass) rows, including both log-normal cases; they are further in exactly the rows the
    arithmetic predicts, and those are the documented limits asserted here (include/jperceiver_hip.h):
      * one activation 10^6 x the median: the outputs that do NOT read it carry 1.6 x (forward) / 2.7 x (weight gradient) the rms error
        of the fp32 accumulation (bound: 3.5 x);
      * a channel whose activations all sit 2^20 below their tensor's largest: ITS weight gradients are carried to 7e-6 relative rms
        (fp32: 3e-7; bound 2^-16), 23 x the exact kernel's -- every other output is unaffected.
    Everywhere: |err| <= max(2 x the exact kernel's worst, 2^-21) x sum|a||b| per output element, and the bound the arithmetic guarantees
    relative to the operand TENSORS' largest magnitudes (ratioT <= 2^-18)."""
    split = _run(dict(JP_P9S="1", JP_W9S="1"), "--outliers")
    exact = _run(dict(JP_P9S="0", JP_W9S="0"), "--outliers")
    rows = ["| case / pass | max err / sum|a||b| (exact fp32) | tensor-relative | rms vs float64 (exact fp32) |", "|---|---|---|---|"]
    for k in sorted(split):
        s, e = split[k], exact[k]
        rows.append(f"| {k} | {s['max_ratio']:.2e} ({e['max_ratio']:.2e}) | {s['max_ratioT']:.2e} | {s['rms']:.2e} ({e['rms']:.2e}) |")
    print("\n".join(rows))
    limits = {"act_outlier_1e6/fwd": ("x", 3.5), "act_outlier_1e6/wgrad": ("x", 3.5),
              "channel_2^-20_below/that_channel/wgrad": ("abs", 2.0 ** -16)}
    for k, s in split.items():
        e = exact[k]
        assert s["max_ratioT"] <= 2.0 ** -18, (k, s)
        assert s["max_ratio"] <= max(2.0 * e["max_ratio"], 2.0 ** -21), (k, s, e)
        kind, lim = limits.get(k, ("x", 1.25))
        if kind == "x":
            assert s["rms"] <= lim * e["rms"] + 1e-9, (k, s, e)
        else:
            assert s["rms"] <= lim, (k, s)


if __name__ == "__main__":
    if "--emit" in sys.argv:
        _emit()
    if "--edges" in sys.argv:
        _emit_edges()
    if "--outliers" in sys.argv:
        _emit_outliers()
