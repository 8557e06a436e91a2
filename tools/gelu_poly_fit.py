"""Coefficients of the polynomial erf-GELU of csrc/common.h (bf16 MFMA epilogues only; the fp32 parity kernels keep erff).

    Phi(x)   - 1/2 = x * P(s)      gelu(x)  = x * Phi(x)
    gelu'(x) - 1/2 = x * Q(s)      (Phi - 1/2 and x * phi(x) are both odd)
with x clamped to [-R, R] and s = 2 x^2 / R^2 - 1 in [-1, 1] (Chebyshev interpolation, monomial form in s: coefficients <= 0.16 in
magnitude, Horner in fp32 is well conditioned).  Prints the C arrays and the maximum absolute error of an fp32 Horner evaluation
over [-8, 8]."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def gelud(x):
    return Phi(x) + x * np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def fit(f, R, deg):
    def g(s):
        x = np.sqrt(np.maximum((s + 1) * R * R / 2, 1e-30))
        x = np.maximum(x, 1e-8)
        return (f(x) - 0.5) / x
    return C.cheb2poly(C.chebinterpolate(g, deg))


def horner32(mono, x, R):
    xc = np.clip(np.float32(x), np.float32(-R), np.float32(R))
    s = xc * xc * np.float32(2 / (R * R)) + np.float32(-1)
    r = np.full_like(s, np.float32(mono[-1]))
    for c in mono[-2::-1]:
        r = r * s + np.float32(c)
    return np.float32(0.5) + xc * r


if __name__ == "__main__":
    x = np.linspace(-8, 8, 800001)
    for name, f, R, deg in (("GELU_P", Phi, 4.5, 10), ("GELU_Q", gelud, 5.0, 12)):
        m = fit(f, R, deg)
        a = horner32(m, x, R).astype(np.float64)
        err = np.abs(a - f(x)).max()
        extra = f", gelu abs err {np.abs(x * a - x * f(x)).max():.2e}" if name == "GELU_P" else ""
        print(f"// {name}: R = {R}, degree {deg} in s, max abs err {err:.2e}{extra}")
        print(f"constexpr float {name}[{deg + 1}] = {{" + ", ".join(f"{c:.9e}f" for c in m) + "};")
