"""Fit used by csrc/k_mlp.hip: GELU(x) = max(x,0) - 0.5*u*2^(-q(u)),  u = min(|x|, 6.5),  q(u) = -log2(erfc(u/sqrt2)).
q is smooth (~ u^2 log2(e)/2 + log terms), so a degree-6 polynomial without constant term, fitted with weights
d(GELU)/dq = 0.5*u*erfc*ln2 (Lawson iterations towards minimax), reproduces GELU to ~3e-7 absolute in fp32 - the
level of Abramowitz-Stegun 7.1.26 - with one transcendental (exp2) instead of two (rcp + exp).  Prints the coefficients."""
import numpy as np
from scipy.special import erfc, erf

import sys
UMAX, DEG = 6.5, int(sys.argv[1]) if len(sys.argv) > 1 else 4   # number of coefficients of q: 6 -> 3.1e-7, 5 -> 7.1e-7, 4 -> 8.7e-6 (shipped)
u = np.linspace(0, UMAX, 20001)
E = erfc(u / np.sqrt(2))
q = -np.log2(E)
w = 0.5 * u * E * np.log(2) + 1e-9
V = np.vander(u / UMAX, DEG + 1, increasing=True)[:, 1:]
ww = w.copy()
for _ in range(200):
    c, *_ = np.linalg.lstsq(V * ww[:, None], q * ww, rcond=None)
    err = np.abs((V @ c - q) * w)
    ww = ww * (1 + 2 * err / err.max()); ww /= ww.max() / w.max()
coef = c / (UMAX ** np.arange(1, DEG + 1))
print(f"coefficients c1..c{DEG} of q(u) = u*(c1 + c2 u + ...):")
print(", ".join(f"{v:.9e}f" for v in coef))

x = np.linspace(-9, 9, 600001).astype(np.float32)
uu = np.minimum(np.abs(x), np.float32(UMAX))
acc = np.full_like(uu, np.float32(coef[-1]))
for k in range(DEG - 2, -1, -1):
    acc = (acc * uu + np.float32(coef[k])).astype(np.float32)
g = (np.maximum(x, 0) - np.float32(0.5) * uu * np.exp2(-(acc * uu).astype(np.float32))).astype(np.float32)
xd = x.astype(np.float64)
ref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
e = np.abs(g - ref)
print(f"max abs err {e.max():.3e} at x={x[e.argmax()]:.3f}; max rel err where |gelu|>1e-3: {(e / np.maximum(np.abs(ref), 1e-3)).max():.3e}")
