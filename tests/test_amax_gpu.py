"""Operand magnitudes of the fp16 split kernels (csrc/scale.hip) on a real MI355X, through the C ABI:
  * jp_amax / jp_amax_into: the largest ORDINARY magnitude of a tensor (Inf / NaN / |x| >= 2^100 take no part), in a 32-way slot;
  * jp_amax_hint: a convolution that is handed its operand's magnitude computes bit-for-bit what it computes when it reduces it itself;
  * jp_amax_out: the producers that fold the reduction into their kernel (patch-kernel conv forward, BatchNorm forward / backward,
    activation backward) report exactly max |what they wrote|; an entry point that cannot leaves the request untaken;
  * the host mirror (ops.conv2d / batchnorm_train) ends up with the same step whether magnitudes come from producers or reductions."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from jperceiver_amd import ops                                                 # noqa: E402
from jperceiver_amd._lib import call, lib                                      # noqa: E402
from jperceiver_amd.ops import Var, Tape, recording                            # noqa: E402

DEV = "cuda"
needs_scales = pytest.mark.skipif(False, reason="")


def _slot():
    return torch.zeros(int(lib().fn["jp_amax_slot_floats"]()), device=DEV)


def _val(slot):
    return float(slot.max())


def _scheme2():
    if ops.split_scheme() != 2:
        pytest.skip("this build of the library uses the bf16 three-way split: no operand scales")


def test_amax_is_the_largest_ordinary_magnitude():
    g = torch.Generator().manual_seed(1)
    for n in (1, 5, 1023, 4096 + 3, 3 * 1000 * 1000 + 1):
        x = torch.randn(n + 3, generator=g).to(DEV)
        for off in (0, 1, 3):                       # unaligned starts take the scalar head / tail of the kernel
            v = x[off:off + n]
            s = _slot()
            call("jp_amax", v, n, s)
            assert _val(s) == float(v.abs().max()), (n, off)
    x = torch.randn(100000, generator=g).to(DEV)
    ref = float(x.abs().max())
    for bad in (float("inf"), float("-inf"), float("nan"), 2.0 ** 100, -3.0e38):
        y = x.clone()
        y[777] = bad
        s = _slot()
        call("jp_amax", y, y.numel(), s)
        assert _val(s) == ref, bad                  # not part of the scale: the kernels turn them into NaN for the outputs that read them
    y = x.clone()
    y[5] = 2.0 ** 99
    s = _slot()
    call("jp_amax", y, y.numel(), s)
    assert _val(s) == 2.0 ** 99
    s = _slot()                                     # jp_amax_into extends what the slot holds
    call("jp_amax_into", x[:50000], 50000, s)
    call("jp_amax_into", x[50000:], 50000, s)
    assert _val(s) == ref
    z = torch.zeros(4096, device=DEV)
    s = _slot()
    call("jp_amax", z, z.numel(), s)
    assert _val(s) == 0.0


def _conv_direct(x, w, hint=None, want_out=False):
    N, Cin, H, W = x.shape
    Cout, _, KH, _ = w.shape
    L = lib()
    y = torch.empty(N, Cout, H, W, device=DEV)
    ws = torch.empty(int(L.fn["jp_conv2d_ws_floats"](Cin, Cout, KH, 0)), device=DEV)
    nsp = int(L.fn["jp_conv2d_fwd_split_floats"](N, Cin, H, W, Cout, KH, 1, KH // 2))
    sp = torch.empty(nsp, device=DEV) if nsp else None
    out_slot = _slot() if want_out else None
    if hint is not None:
        assert L.fn["jp_amax_hint"](x.data_ptr(), hint.data_ptr()) == 0
    if want_out:
        assert L.fn["jp_amax_out"](out_slot.data_ptr()) == 0
    try:
        call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, KH, 1, KH // 2, 0, 0, ws, 0, sp)
    finally:
        L.fn["jp_amax_hint_clear"]()
        done = L.fn["jp_amax_out_done"]()
    return y, out_slot, done


@pytest.mark.parametrize("K", [3, 1])
def test_hinted_convolution_equals_self_reduced_and_reports_its_output(K):
    _scheme2()
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(2, 128, 128, 128, generator=g) * 3.0).to(DEV)     # (>= 192 tiles: the patch kernels of launch_p9s, which report)
    w = (torch.randn(128, 128, K, K, generator=g) * 0.05).to(DEV)
    y0, _, _ = _conv_direct(x, w)
    s = _slot()
    call("jp_amax", x, x.numel(), s)
    y1, out, done = _conv_direct(x, w, hint=s, want_out=True)
    assert torch.equal(y0, y1)                      # the same scale either way -> the same bits
    if K == 3:
        assert done == 1                            # the 3x3 patch kernel reports; (the 128-channel 1x1 layer runs the generic engine: no report)
    if done:
        assert _val(out) == float(y1.abs().max())
    else:
        assert _val(out) == 0.0
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=K // 2)
    assert float((y1.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # a hint that is too LARGE only costs precision (16 x: four of the seventeen spare bits)
    s16 = s * 16.0
    y2, _, _ = _conv_direct(x, w, hint=s16)
    assert float((y2.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_amax_out_is_left_alone_by_kernels_that_cannot_report():
    _scheme2()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)        # 7x7 stem: not a reporting kernel
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).to(DEV)
    L = lib()
    y = torch.empty(2, 64, 32, 32, device=DEV)
    ws = torch.empty(int(L.fn["jp_conv2d_ws_floats"](3, 64, 7, 0)), device=DEV)
    slot = _slot()
    L.fn["jp_amax_out"](slot.data_ptr())
    call("jp_conv2d_fwd", x, w, None, y, 2, 3, 64, 64, 64, 7, 2, 3, 0, 0, ws, 0, None)
    assert L.fn["jp_amax_out_done"]() == 0 and _val(slot) == 0.0
    # ... and the dropped request does not reach a later entry point
    d = torch.empty_like(y)
    call("jp_act_bwd", y, y, d, y.numel(), 1)
    assert L.fn["jp_amax_out_done"]() == 0 and _val(slot) == 0.0


def test_batchnorm_and_activation_backward_report_what_they_write():
    g = torch.Generator().manual_seed(4)
    x = Var((torch.randn(4, 64, 32, 48, generator=g) * 2.0 + 0.3).to(DEV), True)
    gamma, beta = Var(torch.rand(64, generator=g).to(DEV) + 0.5, True, torch.zeros(64, device=DEV)), Var(
        torch.randn(64, generator=g).to(DEV), True, torch.zeros(64, device=DEV))
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    tape = Tape()
    with recording(tape):
        y = ops.batchnorm_train(x, gamma, beta, rm, rv, relu=True)
    if ops.split_scheme() == 2:
        assert y.amax is not None and _val(y.amax) == float(y.t.abs().max())
    y.g = torch.randn(4, 64, 32, 48, generator=g).to(DEV)
    tape.backward()
    if ops.split_scheme() == 2:
        assert x.gamax is not None and _val(x.gamax) == float(x.g.abs().max())
        x.add_grad(torch.ones_like(x.t))            # anything added to the gradient drops the reported magnitude
        assert x.gamax is None
    L = lib()
    dy, yy = torch.randn(2, 32, 40, 40, generator=g).to(DEV), torch.randn(2, 32, 40, 40, generator=g).to(DEV)
    for name, extra in (("jp_act_bwd", None), ("jp_act_bwd_bias", torch.zeros(32, device=DEV))):
        d, slot = torch.empty_like(dy), _slot()
        L.fn["jp_amax_out"](slot.data_ptr())
        if extra is None:
            call(name, dy, yy, d, dy.numel(), 2)
        else:
            call(name, dy, yy, d, extra, 2, 32, 1600, 2)
        assert L.fn["jp_amax_out_done"]() == 1 and _val(slot) == float(d.abs().max()), name


def test_conv_chain_uses_reported_magnitudes_and_matches_the_reduced_ones():
    """conv -> conv -> loss with the magnitudes reported by the producers (default) and with every one reduced separately
    (producer reports and cached bounds switched off): same forward bits, same gradients."""
    _scheme2()
    g = torch.Generator().manual_seed(5)
    xt = torch.randn(2, 64, 128, 128, generator=g).to(DEV)
    w1 = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(DEV)
    w2 = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).to(DEV)
    b1 = torch.randn(128, generator=g).to(DEV)
    up = torch.randn(2, 64, 128, 128, generator=g).to(DEV)

    def run(reported):
        saved = ops._amax_out.__init__
        if not reported:
            def off(self, dev, on=True):
                saved(self, dev, False)
            ops._amax_out.__init__ = off
        try:
            x = Var(xt.clone(), True)
            v1, v2, vb = (Var(t.clone(), True, torch.zeros_like(t)) for t in (w1, w2, b1))
            tape = Tape()
            with recording(tape):
                h = ops.conv2d(x, v1, vb, 1, 1, ops.PAD_REFLECT, ops.ACT_LEAKY)
                if reported:
                    assert h.amax is not None
                y = ops.conv2d(h, v2, None, 1, 1, ops.PAD_ZERO, ops.ACT_NONE)
            y.g = up.clone()
            tape.backward()
            torch.cuda.synchronize()
            return y.t.clone(), x.g.clone(), v1.g.clone(), v2.g.clone(), vb.g.clone()
        finally:
            ops._amax_out.__init__ = saved

    a, b = run(True), run(False)
    for p, q in zip(a[:4], b[:4]):
        assert torch.equal(p, q)
    # (the bias gradient is a sum that meets in float atomics -- jp_act_bwd_bias -- and differs in its last bits from run to run)
    assert torch.allclose(a[4], b[4], rtol=1e-5, atol=1e-4)
