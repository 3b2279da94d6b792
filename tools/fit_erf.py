"""Fit of the branch-free erf used by csrc/common.h:  erf(t) = 1 - 2^(-t*Q(t)),  t = min(|x|, 4),  Q a polynomial.
Weighted least squares + Lawson re-weighting (minimax in absolute erf error) against scipy's fp64 erfc, then the
error of the fp32 evaluation (Horner in float32, exp2 rounded to float32).  usage: python tools/fit_erf.py [degree]"""
import sys
import numpy as np
from scipy.special import erfc, erf

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 9
t = np.concatenate([np.linspace(1e-6, 4.0, 40001), np.geomspace(1e-6, 1e-2, 2000)])
t.sort()
target = -np.log2(erfc(t)) / t                      # Q(t)
sens = erfc(t) * np.log(2.0) * t                    # d erf / d Q
V = np.vander(t, deg + 1, increasing=True)
w = np.ones_like(t)
for it in range(200):
    W = (sens * w)[:, None]
    coef, *_ = np.linalg.lstsq(V * W, target * sens * w, rcond=None)
    err = (V @ coef - target) * sens                # absolute erf error (linearised)
    w = w * (np.abs(err) / np.abs(err).max()) ** 0.5 + 1e-12
    w /= w.max()
exact_err = np.abs((1 - np.exp2(-t * (V @ coef))) - erf(t)).max()
print(f"degree {deg}: fp64 max abs err {exact_err:.3e}")

def eval32(x, c):
    x = x.astype(np.float32)
    tt = np.minimum(np.abs(x), np.float32(4.0))
    q = np.full_like(tt, np.float32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        q = (q.astype(np.float64) * tt + np.float64(np.float32(c[k]))).astype(np.float32)   # fma: one rounding
    e = np.float32(1.0) - np.exp2((-tt * q).astype(np.float32).astype(np.float64)).astype(np.float32)
    return np.copysign(e, x)

xs = np.concatenate([np.linspace(-6, 6, 2000001), np.random.default_rng(0).normal(size=1000000) * 1.5]).astype(np.float32)
d = eval32(xs, coef).astype(np.float64) - erf(xs.astype(np.float64))
print(f"fp32 evaluation: max abs err {np.abs(d).max():.3e}, mean err {d.mean():.3e}, rms {np.sqrt((d**2).mean()):.3e}")
# systematic part: mean signed error in bins of x (what survives averaging over many pixels)
bins = np.linspace(-4, 4, 33)
idx = np.digitize(xs, bins)
sysm = [abs(d[idx == i].mean()) for i in range(1, len(bins)) if (idx == i).sum() > 1000]
print(f"largest |bin-mean error| over 0.25-wide bins: {max(sysm):.3e}")
print("coefficients (highest first):")
for c in coef[::-1]:
    print(f"    {np.float32(c)!r}")
