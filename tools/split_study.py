"""Which operand-split schemes on the 16-bit matrix pipe are 'fp32-grade'?  A CPU study (numpy, float64 reference).

Each scheme's products are exact in fp32, so its error against float64 is (operand representation + the product terms it leaves out);
the sums below are formed in float64 to isolate exactly that part.  It is compared with the rounding error of the fp32 ACCUMULATION
every scheme shares (an fp32 accumulator rounded once per 16 reduction elements, as the MFMA does) on layer-shaped data:
activations with outliers, log-normally spread gradients, weight gradients.  Errors are relative to sum |a_i| |b_i|.

  bf16x3 / 6 products   the round-3..5 kernels: exact operands, a1 b2 + a2 b1 + a2 b2 left out
  bf16x3 / 3 products   a0 b0 + a0 b1 + a1 b0 only ("bf16x3-lite", ~16 significant bits): NOT fp32-grade, listed for contrast
  fp16x2 / 3 products   operands scaled by a power of two so that the tensor's largest magnitude lands in [2^14, 2^15), h0 = fp16(s x),
                        h1 = fp16(s x - h0); a0 b0 + a0 b1 + a1 b0 in ONE accumulator (igemm_p9s.h:jp_split2h)
Run: python tools/split_study.py
"""
import numpy as np

rng = np.random.default_rng(1)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split_bf16(x, n=3):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(r)
        out.append(p.astype(np.float64))
        r = (r - p).astype(np.float32)
    return out


def split_f16(x):
    m = np.abs(x).max()
    s = 2.0 ** (14 - np.floor(np.log2(m)))          # s * m in [2^14, 2^15)
    xs = x.astype(np.float32) * np.float32(s)
    h0 = xs.astype(np.float16)
    h1 = (xs - h0.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h0.astype(np.float64), h1.astype(np.float64), s


def run(name, A, B):
    M, K = A.shape
    N = B.shape[1]
    ref = A.astype(np.float64) @ B.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)

    def err(C):
        e = np.abs(C - ref) / den
        return e.max(), np.sqrt((e ** 2).mean())

    acc = np.zeros((M, N), np.float32)
    for k in range(0, K, 16):
        acc = (acc + (A[:, k:k + 16].astype(np.float64) @ B[k:k + 16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    a, b = split_bf16(A), split_bf16(B)
    b6 = a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0] + a[1] @ b[1] + a[0] @ b[2] + a[2] @ b[0]
    b3 = a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0]
    a0, a1, sa = split_f16(A)
    b0, b1, sb = split_f16(B)
    h3 = (a0 @ b0 + a0 @ b1 + a1 @ b0) / (sa * sb)
    spread = lambda T: np.abs(T).max() / np.median(np.abs(T[T != 0]))
    print(f"{name}   (largest / median magnitude: A {spread(A):.1e}, B {spread(B):.1e})")
    for nm, C in (("fp32 accumulation alone (exact products)", acc.astype(np.float64)), ("bf16x3, 6 products", b6), ("bf16x3, 3 products", b3),
                  ("fp16x2, 3 products, one accumulator", h3)):
        mx, rms = err(C)
        print(f"   {nm:42s} max {mx:.2e}   rms {rms:.2e}")


if __name__ == "__main__":
    M, K, N = 64, 2304, 512
    w = (rng.standard_normal((M, K)) * 0.02).astype(np.float32)
    x = np.maximum(rng.standard_normal((K, N)), 0).astype(np.float32)
    run("forward: weights x relu activations", w, x)
    for o in (1e4, 1e6):
        xo = x.copy()
        xo[0, 0] = o
        run(f"forward, one activation outlier of {o:.0e}", w, xo)
    for sg in (2, 4):
        g = (rng.standard_normal((K, N)) * 1e-6 * np.exp(rng.standard_normal((K, N)) * sg)).astype(np.float32)
        run(f"dgrad: weights x log-normal gradients (sigma {sg})", w, g)
    g2 = (rng.standard_normal((M, K)) * 1e-7 * np.exp(rng.standard_normal((M, K)) * 3)).astype(np.float32)
    run("wgrad: log-normal gradients (sigma 3) x activations", g2, x)
