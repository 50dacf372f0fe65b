"""The frame comparison every parity test uses (tests/test_gpu_parity.py, tests/test_parity_util.py)."""
import numpy as np

TOL = 1e-4  # per RGB channel, BASELINE.json north_star


def maxdiff(a, b):
    """(max |a - b|, positions beyond TOL, positions that differ).  A NaN on both sides, or the same
    infinity on both sides, is agreement; a NaN on ONE side, or infinities that differ (or an infinity
    against a finite value), is an infinite difference — it counts as beyond TOL and every caller's
    `md <= TOL and nbad == 0` fails on it (np.nanmax / `nan > TOL` would let it through)."""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    both_nan = np.isnan(a64) & np.isnan(b64)
    same_inf = np.isinf(a64) & (a64 == b64)
    with np.errstate(invalid="ignore"):
        d = np.abs(a64 - b64)
    d = np.where(both_nan | same_inf, 0.0, d)
    d = np.where(np.isnan(d), np.inf, d)      # one-sided NaN, or inf - inf of opposite signs
    return float(d.max()) if d.size else 0.0, int((d > TOL).sum()), int((d != 0).sum())
