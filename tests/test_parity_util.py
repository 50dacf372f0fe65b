"""The checker itself (round-3 verdict, weak #2): only agreement passes.  Runs on CPU and on the GPU box."""
import numpy as np

from parity_util import TOL, maxdiff


def test_maxdiff_refuses_one_sided_nan_and_differing_infinities():
    """The checker itself (round-3 verdict, weak #2): only agreement passes."""
    z = np.zeros((2, 2, 3), np.float32)
    for bad in (np.nan, np.inf, -np.inf):
        g = z.copy()
        g[1, 0, 2] = bad
        for x, y in ((g, z), (z, g)):
            md, nbad, nne = maxdiff(x, y)
            assert not (md <= TOL and nbad == 0) and nbad == 1 and nne == 1
    p, m = z.copy(), z.copy()
    p[0, 0, 0], m[0, 0, 0] = np.inf, -np.inf
    assert maxdiff(p, m)[1] == 1
    n = z.copy()
    n[0, 1, 1] = np.nan
    assert maxdiff(n, n.copy()) == (0.0, 0, 0) and maxdiff(p, p.copy()) == (0.0, 0, 0)
