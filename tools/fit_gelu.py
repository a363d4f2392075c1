"""Fit of the erf-GELU form used by the row-local feed-forward kernel (gligen_amd/csrc/ffn.hip, FFR_GELU_C0..C3):

    gelu(x) = x Phi(x) = max(x, 0) - |x| h(|x|),   h(a) = erfc(a / sqrt 2) / 2 = 2^L(a),   L a cubic in a.

One v_exp_f32 per value instead of the rcp + exp of the Abramowitz-Stegun erf (common.h gelu_erf_f), 7 VALU instead of 16.
L is fitted to log2 h on [0, 9] by iteratively re-weighted least squares towards the minimax of |a (2^L - h)|, the absolute error
of gelu. All four coefficients come out negative, so h -> 0 monotonically beyond the fitted range (no clamp needed).
Prints the coefficients and the error of an fp32 evaluation over [-40, 40]; tests/test_host_cpu.py pins the shipped constants."""
import numpy as np
from scipy.special import erf, erfc


def fit(deg=3, hi=9.0, iters=200):
    a = np.linspace(0, hi, 200001)
    h = 0.5 * erfc(a / np.sqrt(2))
    L = np.log2(np.maximum(h, 1e-300))
    w = h * np.log(2) + 1e-6
    for _ in range(iters):
        c = np.polyfit(a, L, deg, w=w)
        err = np.abs(2.0 ** np.polyval(c, a) - h) * np.maximum(a, 0.3)
        w = w * (1 + 2 * err / err.max())
    return c


def gelu_fit(x, c):
    x = x.astype(np.float32)
    a = np.abs(x)
    L = np.float32(c[0])
    for k in c[1:]:
        L = L * a + np.float32(k)
    return np.maximum(x, 0) - a * np.exp2(L.astype(np.float64)).astype(np.float32)


if __name__ == "__main__":
    c = fit()
    print("coefficients, highest power first:", [float(v) for v in c])
    x = np.linspace(-40, 40, 2000001)
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    e = np.abs(gelu_fit(x, c).astype(np.float64) - ref)
    print(f"max |gelu error| {e.max():.3e} at x = {x[e.argmax()]:.3f}")
