import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as o
    o.lib()
    return o


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.abs(b).max()) if b.size else 0.0, 1e-300)
    return float(np.abs(a - b).max() / den) if a.size else 0.0


# ---- per-variable-class metrics ---------------------------------------------------------------------------------
# rel_err above normalises by the largest entry of the WHOLE array.  With the demos' K / f0 (f0 = 600) the u0 / v0 columns
# of the point-frame blocks are ~600x smaller than the pose columns and the intrinsics x intrinsics entries of the frame
# blocks and of the reduced camera system ~1e5x smaller, so a global "rel 1e-10" pins those entries only to ~1e-5 of their
# own size.  The functions below compare every entry on the scale of ITS variable class: Gauss-Newton blocks are sums of
# outer products, so |B_ab| <= sqrt(B_aa B_bb) and 1 / sqrt(diag) of the oracle's block is the natural row / column scale.

def sym_scaled_err(a, b, d):
    """max |a - b| after scaling rows and columns by 1 / d (d: per-variable scales, shape [..., n]; a, b: [..., n, n]).
    Entries of a diagonally scaled positive semi-definite block are <= 1, its diagonal is exactly 1."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    if not a.size:
        return 0.0
    den = d[..., :, None] * d[..., None, :]
    den = np.where(den > 0, den, 1.0)
    return float((np.abs(a - b) / den).max())


def class_rel_err(a, b, class_axes):
    """max over variable classes of (max |a - b| over the class) / (max |b| over the class); a class = one index
    combination along `class_axes` (e.g. (point coordinate, frame variable) of the [O, 3, 10] point-frame blocks).
    A class the oracle holds only zeros for must be zero on the other side as well."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    other = tuple(ax for ax in range(b.ndim) if ax not in tuple(x % b.ndim for x in class_axes))
    num = np.abs(a - b).max(axis=other)
    den = np.abs(b).max(axis=other)
    assert np.all(num[den == 0] == 0), "a class the oracle holds zeros for is not zero"
    return float((num[den > 0] / den[den > 0]).max()) if np.any(den > 0) else 0.0
