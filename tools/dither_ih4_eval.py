"""Evaluation of a cheaper Gaussian for the default dither (round 5; NOT built): the sum of the four
bytes of a random word (Irwin-Hall-4, one v_sad_u8) Gaussianised by an odd polynomial.  Prints variance, kurtosis, the
sixth moment and the sup-CDF error of the map for degrees 3 / 5 / 7.     python tools/dither_ih4_eval.py"""
import numpy as np
from scipy.stats import norm
# pmf of the sum of 4 independent bytes (0..255): 0..1020
p1 = np.ones(256) / 256
p = np.convolve(np.convolve(p1, p1), np.convolve(p1, p1))
s = np.arange(p.size)
cdf = np.cumsum(p)
mid = cdf - 0.5 * p                      # mid-point rule: the quantile each level stands for
g = norm.ppf(mid)                        # exact Gaussianising map at every level
y = (s - 510.0) / 16.0                   # the packed-f16 variable: 64 + s/16 - 95.875 -> [-31.875, 31.875]
print('levels', p.size, 'g range', g[0], g[-1], 'p(extreme)', p[0])
for deg in (3, 5, 7):
    # odd polynomial least squares weighted by the pmf
    A = np.stack([y ** k for k in range(1, deg + 1, 2)], axis=1)
    w = np.sqrt(p)
    c, *_ = np.linalg.lstsq(A * w[:, None], g * w, rcond=None)
    z = A @ c
    m2 = (p * z ** 2).sum(); m4 = (p * z ** 4).sum(); m6 = (p * z ** 6).sum()
    err = np.abs(z - g)
    print('deg', deg, 'coef', c, 'var %.5f kurt %.4f m6/15 %.4f' % (m2, m4 / m2 ** 2, m6 / m2 ** 3 / 15),
          'max|z| %.3f' % np.abs(z).max(), 'max err within 4 sigma %.4f' % err[np.abs(g) < 4].max(),
          'err at 3 sigma %.4f' % err[np.argmin(np.abs(g - 3))])
    # CDF error: sup over levels of |Phi(z_k) - cdf_k|
    print('   sup |Phi(z) - F| = %.2e' % np.abs(norm.cdf(z) - mid).max())
# plain IH4 for comparison
z0 = (s - 510.0) / np.sqrt(4 * (256 ** 2 - 1) / 12)
print('plain IH4: kurt %.4f max %.3f' % ((p * z0 ** 4).sum(), np.abs(z0).max()))
