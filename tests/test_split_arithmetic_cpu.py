"""The error model include/jperceiver_hip.h states for the fp16 two-way split arithmetic (csrc/igemm_p9s.h: jp_split2h, jp_scale_exp,
jp_amag), checked on the CPU against a numpy restatement of those three functions -- no GPU, no library call.  (The kernels themselves
are held to float64 on the GPU by tests/test_split_accuracy_gpu.py; this file pins the ARITHMETIC the header documents.)"""
import numpy as np


def amag(x):
    """jp_amag: |x| as a magnitude that takes part in the scale -- Inf / NaN / |x| >= 2^100 do not."""
    u = np.abs(x.astype(np.float32)).view(np.uint32)
    return np.where(u >= np.uint32(227 << 23), np.uint32(0), u).view(np.float32)


def scale_exp(amax):
    """jp_scale_exp: k with 2^k * amax in [2^14, 2^15); 0 for an all-zero tensor; clamped to 126."""
    u = int(np.float32(amax).view(np.uint32)) & 0x7FFFFFFF
    return 0 if u == 0 else min(126, 14 - ((u >> 23) - 127))


def split2h(x, k):
    xs = (x.astype(np.float32) * np.float32(2.0 ** k)).astype(np.float32)
    h0 = xs.astype(np.float16)
    h1 = (xs - h0.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h0, h1


def test_scale_puts_the_largest_magnitude_below_the_fp16_limit():
    rng = np.random.default_rng(0)
    for e in list(range(-120, 100, 7)) + [-126, 99]:
        for m in (1.0, 1.2345, 1.999999):
            a = np.float32(m * 2.0 ** e)
            k = scale_exp(a)
            s = float(a) * 2.0 ** k
            if e >= -112:
                assert 2.0 ** 14 <= s < 2.0 ** 15, (e, m, s)
            else:
                assert s < 2.0 ** 15 and k == 126                   # tiny tensors: the clamp, still far above the fp16 subnormals' floor
            assert np.isfinite(np.float16(s))
    assert scale_exp(0.0) == 0
    x = rng.standard_normal(1000).astype(np.float32)
    x[3], x[5], x[7], x[9] = np.inf, -np.inf, np.nan, 2.0 ** 100
    assert float(amag(x).max()) == float(np.abs(np.delete(x, [3, 5, 7, 9])).max())
    y = x.copy()
    y[9] = 2.0 ** 99
    assert float(amag(y).max()) == 2.0 ** 99


def test_two_way_split_carries_the_documented_bits():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp(rng.standard_normal(200000) * 4)).astype(np.float32)        # magnitudes over ~14 decades
    amax = float(amag(x).max())
    k = scale_exp(amax)
    h0, h1 = split2h(x, k)
    xs = x.astype(np.float64) * 2.0 ** k
    err = np.abs(xs - (h0.astype(np.float64) + h1.astype(np.float64)))
    top = np.abs(xs) >= amax * 2.0 ** k * 2.0 ** -17
    assert np.all(err[top] <= 2.0 ** -22 * np.abs(xs[top]))           # within 2^-17 of the largest: h1 is a normal fp16
    assert np.all(err <= 2.0 ** -25 + 2.0 ** -22 * np.abs(xs))       # everywhere: the fp16 subnormal grid, 2^-40 of the scaled largest
    assert 2.0 ** -25 / (amax * 2.0 ** k) <= 2.0 ** -39
    assert np.all(np.isfinite(h0.astype(np.float32))) and np.all(np.isfinite(h1.astype(np.float32)))


def test_three_products_stay_below_the_fp32_accumulation_error():
    """a0 b0 + a0 b1 + a1 b0 (exact in fp32: 11 x 11 significand bits) against float64, next to what an fp32 accumulator rounded once
    per 16 products -- the MFMA -- loses on the same data: the claim of DESIGN 4.6b / tools/split_study.py, as an assertion."""
    rng = np.random.default_rng(2)
    M, K, N = 32, 1152, 128
    A = (rng.standard_normal((M, K)) * 0.05).astype(np.float32)
    for B in (np.maximum(rng.standard_normal((K, N)), 0).astype(np.float32),
              (rng.standard_normal((K, N)) * 1e-6 * np.exp(rng.standard_normal((K, N)) * 2)).astype(np.float32)):
        ref = A.astype(np.float64) @ B.astype(np.float64)
        den = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        ka, kb = scale_exp(float(amag(A).max())), scale_exp(float(amag(B).max()))
        a0, a1 = (t.astype(np.float64) for t in split2h(A, ka))
        b0, b1 = (t.astype(np.float64) for t in split2h(B, kb))
        for t in (a0, b1):                                             # every product is exact in fp32
            assert np.all(t == t.astype(np.float32))
        got = (a0 @ b0 + a0 @ b1 + a1 @ b0) * 2.0 ** -(ka + kb)
        acc = np.zeros((M, N), np.float32)
        for j in range(0, K, 16):
            acc = (acc + (A[:, j:j + 16].astype(np.float64) @ B[j:j + 16].astype(np.float64)).astype(np.float32)).astype(np.float32)
        e_split = np.sqrt((((got - ref) / den) ** 2).mean())
        e_acc = np.sqrt((((acc.astype(np.float64) - ref) / den) ** 2).mean())
        assert e_split < 0.6 * e_acc, (e_split, e_acc)
        assert np.abs(got - ref).max() <= 2.0 ** -21 * den.max()
