"""Operand magnitudes of the fp16 split kernels (csrc/scale.hip) on a real MI355X, through the C ABI (version 3: every
magnitude is an explicit argument of the entry point that reads or writes it):
  * jp_amax / jp_amax_into: the largest ORDINARY magnitude of a tensor (Inf / NaN / |x| >= 2^100 take no part), in a 32-way slot;
  * amax_x: a convolution that is handed its operand's magnitude computes bit-for-bit what it computes when it reduces it itself into
    the caller's amax_ws; with neither the call is refused;
  * amax_y / amax_dx: the producers that fold the reduction into their kernel (patch-kernel conv forward, BatchNorm forward / backward,
    activation backward, 5x5 max-pool backward) report exactly max |what they wrote|; a conv kernel that cannot says so (*amax_y_done);
  * the host mirror (ops.conv2d / batchnorm_train) ends up with the same step whether magnitudes come from producers or reductions."""
import ctypes
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


def _amax_ws():
    return torch.full((int(lib().fn["jp_conv2d_amax_ws_floats"]()),), float("nan"), device=DEV)     # (need not be initialised)


def _conv_direct(x, w, hint=None, want_out=False, amax_ws=True):
    N, Cin, H, W = x.shape
    Cout, _, KH, _ = w.shape
    L = lib()
    y = torch.empty(N, Cout, H, W, device=DEV)
    ws = torch.empty(int(L.fn["jp_conv2d_ws_floats"](Cin, Cout, KH, 0)), device=DEV)
    nsp = int(L.fn["jp_conv2d_fwd_split_floats"](N, Cin, H, W, Cout, KH, 1, KH // 2))
    sp = torch.empty(nsp, device=DEV) if nsp else None
    out_slot = _slot() if want_out else None
    done = ctypes.c_int(-1)
    call("jp_conv2d_fwd", x, w, None, y, N, Cin, H, W, Cout, KH, 1, KH // 2, 0, 0, ws, 0, sp, hint, out_slot, ctypes.addressof(done),
         _amax_ws() if amax_ws else None, None, None)
    return y, out_slot, done.value


@pytest.mark.parametrize("K", [3, 1])
def test_hinted_convolution_equals_self_reduced_and_reports_its_output(K):
    _scheme2()
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(2, 128, 128, 128, generator=g) * 3.0).to(DEV)     # (>= 192 tiles: the patch kernels of launch_p9s, which report)
    w = (torch.randn(128, 128, K, K, generator=g) * 0.05).to(DEV)
    y0, _, _ = _conv_direct(x, w)
    s = _slot()
    call("jp_amax", x, x.numel(), s)
    y1, out, done = _conv_direct(x, w, hint=s, want_out=True, amax_ws=False)      # every operand magnitude given: no scratch needed
    assert torch.equal(y0, y1)                      # the same scale either way -> the same bits
    if K == 3:
        assert done == 1                            # the 3x3 patch kernel reports; (the 128-channel 1x1 layer runs the generic engine: no report)
    if done:
        assert _val(out) == float(y1.abs().max())
    else:
        assert done == 0 and _val(out) == 0.0
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=K // 2)
    assert float((y1.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # a magnitude that is too LARGE only costs precision (16 x: four of the seventeen spare bits)
    s16 = s * 16.0
    y2, _, _ = _conv_direct(x, w, hint=s16)
    assert float((y2.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # neither the operand's magnitude nor scratch to reduce it into: refused, nothing launched
    from jperceiver_amd._lib import JPerceiverHipError
    with pytest.raises(JPerceiverHipError, match="amax_ws"):
        _conv_direct(x, w, amax_ws=False)


def test_conv_entry_points_keep_no_state_between_calls():
    """Two different tensors at the SAME address with different magnitudes, back to back: each call sees only its own arguments
    (ABI version 2 kept hints by address in thread-local state; a stale one would mis-scale the second call)."""
    _scheme2()
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(2, 128, 64, 128, generator=g)).to(DEV)
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(DEV)
    s = _slot()
    call("jp_amax", x, x.numel(), s)
    y_small, _, _ = _conv_direct(x, w, hint=s)
    x.mul_(4096.0)                                  # same address, 2^12 larger: a stale magnitude would overflow fp16
    y_big, _, _ = _conv_direct(x, w)                # un-hinted: must reduce afresh
    assert bool(torch.isfinite(y_big).all())
    assert torch.equal(y_big, y_small * 4096.0)     # power-of-two scaling commutes with the whole arithmetic


def test_amax_out_is_left_alone_by_kernels_that_cannot_report():
    _scheme2()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)        # 7x7 stem: not a reporting kernel
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).to(DEV)
    L = lib()
    y = torch.empty(2, 64, 32, 32, device=DEV)
    ws = torch.empty(int(L.fn["jp_conv2d_ws_floats"](3, 64, 7, 0)), device=DEV)
    slot = _slot()
    done = ctypes.c_int(-1)
    call("jp_conv2d_fwd", x, w, None, y, 2, 3, 64, 64, 64, 7, 2, 3, 0, 0, ws, 0, None, None, slot, ctypes.addressof(done), _amax_ws(), None, None)
    assert done.value == 0 and _val(slot) == 0.0
    # ... and nothing of the request reaches a later entry point
    d = torch.empty_like(y)
    call("jp_act_bwd", y, y, d, y.numel(), 1, None)
    assert _val(slot) == 0.0


def test_three_source_forward_folds_the_given_magnitudes():
    """iconv layer cat(skip, up2x(x), disparity): with the three sources' slots given the kernel's scale comes from their maximum
    (one 64-lane fold launch, no re-read of the sources) and the output equals the self-reduced call's bit for bit."""
    _scheme2()
    g = torch.Generator().manual_seed(13)
    N, H, W = 4, 64, 128                            # (128 workgroups: the P9US2 kernel takes the layer)
    x0 = torch.randn(N, 64, H, W, generator=g).to(DEV)
    x1 = (torch.randn(N, 64, H // 2, W // 2, generator=g) * 5.0).to(DEV)          # the largest magnitude sits in the upsampled source
    x2 = torch.rand(N, 1, H, W, generator=g).to(DEV)
    w = (torch.randn(128, 129, 3, 3, generator=g) * 0.05).to(DEV)
    L = lib()
    ws = torch.empty(int(L.fn["jp_conv2d_ws_floats"](129, 128, 3, 0)), device=DEV)

    def run(am):
        y = torch.empty(N, 128, H, W, device=DEV)
        call("jp_conv2d_fwd_src3", x0, 64, 0, x1, 64, 1, x2, 1, 0, w, None, y, N, H, W, 128, 3, 1, 1, 1, 2, ws, 0, None,
             *am, None, None, _amax_ws(), None, None)
        return y

    slots = []
    for t in (x0, x1, x2):
        s = _slot()
        call("jp_amax", t, t.numel(), s)
        slots.append(s)
    a, b, c = run((None, None, None)), run(tuple(slots)), run((slots[0], None, slots[2]))
    assert torch.equal(a, b) and torch.equal(a, c)


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
    dy, yy = torch.randn(2, 32, 40, 40, generator=g).to(DEV), torch.randn(2, 32, 40, 40, generator=g).to(DEV)
    for name, extra in (("jp_act_bwd", None), ("jp_act_bwd_bias", torch.zeros(32, device=DEV))):
        d, slot = torch.empty_like(dy), _slot()
        if extra is None:
            call(name, dy, yy, d, dy.numel(), 2, slot)
        else:
            call(name, dy, yy, d, extra, 2, 32, 1600, 2, slot, None)
        assert _val(slot) == float(d.abs().max()), name
    # one Inf in the tensor a producer writes: every ordinary value still counts (the filter is per element)
    dy2 = dy.clone()
    dy2[1, 3, 7, 9] = float("inf")
    d, slot = torch.empty_like(dy2), _slot()
    call("jp_act_bwd", dy2, yy, d, dy2.numel(), 2, slot)
    fin = d[torch.isfinite(d)]
    assert _val(slot) == float(fin.abs().max())
    # 5x5 stride-1 max-pool backward (+ addend): the row kernel reports max |dx|
    xx = torch.randn(2, 32, 64, 64, generator=g).to(DEV)
    yp, ip = torch.empty_like(xx), torch.empty(2, 32, 64, 64, dtype=torch.uint8, device=DEV)
    call("jp_maxpool_fwd", xx, yp, ip, 64, 64, 64, 5, 1, 2)
    gy, add = torch.randn(2, 32, 64, 64, generator=g).to(DEV), torch.randn(2, 32, 64, 64, generator=g).to(DEV)
    for ad in (None, add):
        dx, slot = torch.empty_like(xx), _slot()
        call("jp_maxpool_bwd", gy, ip, dx, ad, 64, 64, 64, 5, 1, 2, slot)
        assert _val(slot) == float(dx.abs().max())
    xs = torch.randn(2, 8, 30, 50, generator=g).to(DEV)          # a shape the row kernel does not take: reduction pass behind the kernel
    ys, is_ = torch.empty(2, 8, 15, 25, device=DEV), torch.empty(2, 8, 15, 25, dtype=torch.uint8, device=DEV)
    call("jp_maxpool_fwd", xs, ys, is_, 16, 30, 50, 2, 2, 0)
    dxs, slot = torch.empty_like(xs), _slot()
    call("jp_maxpool_bwd", torch.randn(2, 8, 15, 25, generator=g).to(DEV), is_, dxs, None, 16, 30, 50, 2, 2, 0, slot)
    assert _val(slot) == float(dxs.abs().max())


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
        saved = ops._out_slot
        if not reported:
            ops._out_slot = lambda dev, on=True: None
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
            ops._out_slot = saved

    a, b = run(True), run(False)
    for p, q in zip(a, b):          # (round 6: the bias gradient's partial sums are folded in a fixed order too)
        assert torch.equal(p, q)


@pytest.mark.parametrize("second", ["add", "view"])
def test_reported_gradient_magnitude_is_dropped_by_every_accumulation_path(second):
    """conv -> {BatchNorm, add / view} fan-out.  BatchNorm backward reports max |dx| for the conv output's
    gradient; the second consumer then accumulates a gradient 2^20 x larger into the same buffer.  A stale (too small) magnitude
    would overflow fp16 in the conv's dgrad / wgrad (Inf / NaN gradients); the accumulation paths of ops.add and ops_loss.view drop it."""
    _scheme2()
    import torch.nn.functional as F
    from jperceiver_amd import ops_loss
    g = torch.Generator().manual_seed(6)
    N, C, H, W = 8, 64, 64, 64
    xt = torch.randn(N, C, H, W, generator=g)
    wt = torch.randn(C, C, 3, 3, generator=g) * 0.05
    up_bn = torch.randn(N, C, H, W, generator=g)
    up_big = torch.randn(N, C, H, W, generator=g) * 2.0 ** 20
    x = Var(xt.to(DEV), True)
    w = Var(wt.to(DEV), True, torch.zeros(C, C, 3, 3, device=DEV))
    gamma = Var(torch.ones(C, device=DEV), True, torch.zeros(C, device=DEV))
    beta = Var(torch.zeros(C, device=DEV), True, torch.zeros(C, device=DEV))
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    tape = Tape()
    with recording(tape):
        h = ops.conv2d(x, w, None, 1, 1, ops.PAD_ZERO, ops.ACT_NONE)
        # recorded first = replayed last: the second consumer's backward runs AFTER BatchNorm's and accumulates into h.g
        z = ops.add(h, Var(torch.zeros_like(h.t), False)) if second == "add" else ops_loss.view(h, (N, C, H * W))
        y = ops.batchnorm_train(h, gamma, beta, rm, rv)
    y.g = up_bn.to(DEV)
    z.g = up_big.to(DEV).view(z.t.shape).clone()
    tape.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x.g).all()) and bool(torch.isfinite(w.g).all())
    xd, wd = xt.double().requires_grad_(True), wt.double().requires_grad_(True)
    hd = F.conv2d(xd, wd, None, 1, 1)
    yd = F.batch_norm(hd, None, None, torch.ones(C, dtype=torch.float64), torch.zeros(C, dtype=torch.float64), True)
    (yd * up_bn.double()).sum().backward(retain_graph=True)
    hd.backward(up_big.double())
    assert float((x.g.cpu().double() - xd.grad).abs().max() / xd.grad.abs().max()) < 1e-5
    assert float((w.g.cpu().double() - wd.grad).abs().max() / wd.grad.abs().max()) < 1e-5


@pytest.mark.parametrize("pm", [ops.PAD_ZERO, ops.PAD_REFLECT])
def test_dgrad_reports_the_gradient_it_writes(pm):
    """jp_conv2d_dgrad amax_dx: max |dx| out of the patch kernel's epilogue; for reflection-padded layers the border fold, which
    changes the border pixels afterwards, folds its final values into the same slot (an upper bound of max |dx|, exact when the
    largest element is not a border pixel the fold lowered)."""
    _scheme2()
    g = torch.Generator().manual_seed(8)
    x = Var(torch.randn(8, 128, 64, 64, generator=g).to(DEV), True)
    w = Var((torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(DEV), False)
    tape = Tape()
    with recording(tape):
        y = ops.conv2d(x, w, None, 1, 1, pm, ops.ACT_NONE)
    y.g = torch.randn(8, 128, 64, 64, generator=g).to(DEV)
    tape.backward()
    torch.cuda.synchronize()
    assert x.gamax is not None, "the dgrad patch kernel did not report"
    got, ref = _val(x.gamax), float(x.g.abs().max())
    assert got >= ref and got <= ref * 1.5, (got, ref)
    if pm == ops.PAD_ZERO:
        assert got == ref


def test_sum_and_stem_pool_report_what_they_write():
    g = torch.Generator().manual_seed(9)
    a = [torch.randn(2, 32, 40, 40, generator=g).to(DEV) for _ in range(5)]
    out, slot = torch.empty_like(a[0]), _slot()
    call("jp_sum_n", *a, out, out.numel(), slot)
    assert torch.equal(out, a[0] + a[1] + a[2] + a[3] + a[4]) and _val(slot) == float(out.abs().max())
    x = Var((torch.randn(4, 64, 64, 96, generator=g) * 3.0).to(DEV), True)
    gamma, beta = Var(torch.rand(64, generator=g).to(DEV) + 0.5, True, torch.zeros(64, device=DEV)), Var(
        torch.randn(64, generator=g).to(DEV), True, torch.zeros(64, device=DEV))
    with recording(Tape()):
        y = ops.bn_relu_maxpool_train(x, gamma, beta, torch.zeros(64, device=DEV), torch.ones(64, device=DEV))
    if ops.split_scheme() == 2:
        assert y.amax is not None and _val(y.amax) == float(y.t.abs().max())
