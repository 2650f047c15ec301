"""Derive the polynomial coefficients of the fp32 pow spec ("S6 pow").

log2(m) = r*Q(r), r = m-1, m in [sqrt(1/2), sqrt(2));  exp2(g) = 1 + g*R(g), g in [-1/2, 1/2].
Weighted least squares on Chebyshev nodes in float64; coefficients are then rounded to fp32 and
frozen as literals in oracle/tbrm_oracle.c and tbraymarcherplugin_amd/csrc/tbrm_device_math.h.
The two files must carry identical literals; tests/test_oracle_math.py checks accuracy vs libm.
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(fun, a, b, deg, n=4000):
    x = cheb_nodes(a, b, n)
    y = fun(x)
    V = np.vander(x, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V, y, rcond=None)
    return c

def q_fun(r):
    out = np.empty_like(r)
    small = np.abs(r) < 1e-8
    out[small] = 1 / np.log(2)
    out[~small] = np.log2(1 + r[~small]) / r[~small]
    return out

def r_fun(g):
    out = np.empty_like(g)
    small = np.abs(g) < 1e-8
    out[small] = np.log(2)
    out[~small] = (np.exp2(g[~small]) - 1) / g[~small]
    return out

if __name__ == "__main__":
    lo, hi = np.sqrt(0.5) - 1, np.sqrt(2.0) - 1
    for deg in (8, 9, 10):
        c = fit(q_fun, lo, hi, deg)
        x = np.linspace(lo, hi, 200001)
        err = np.max(np.abs(np.polyval(c[::-1], x) / q_fun(x) - 1))
        print("log2 Q deg", deg, "max rel err", err)
    for deg in (5, 6, 7):
        c = fit(r_fun, -0.5, 0.5, deg)
        x = np.linspace(-0.5, 0.5, 200001)
        err = np.max(np.abs((1 + x * np.polyval(c[::-1], x)) / np.exp2(x) - 1))
        print("exp2 R deg", deg, "max rel err", err)
    cq = fit(q_fun, lo, hi, 9).astype(np.float32)
    cr = fit(r_fun, -0.5, 0.5, 6).astype(np.float32)
    print("Q:", ", ".join(f"{float(v).hex()}" for v in cq))
    print("Q:", ", ".join(f"{v:.9g}f" for v in cq))
    print("R:", ", ".join(f"{float(v).hex()}" for v in cr))
    print("R:", ", ".join(f"{v:.9g}f" for v in cr))
