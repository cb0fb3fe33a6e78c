"""The partitioned (parallel) factorisation of the explicit two-level PCG's block-banded coarse matrix (csrc/ps_k_bandpart.h)
against numpy's inverse and against the one-workgroup column walk it replaces (csrc/ps_k_band.h), through the C ABI
(ps_debug_band_inverse).  Stands in for the coarse level's share of scipy.sparse.linalg.spsolve (reference pyslam/problem.py:186)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def banded_spd(ncb, D, B, seed, cond_boost=0.6):
    rng = np.random.default_rng(seed)
    n = ncb * D
    M = rng.standard_normal((n, n))
    A = np.zeros((n, n))
    for i in range(ncb):
        for j in range(max(0, i - B), min(ncb, i + B + 1)):
            A[i * D:(i + 1) * D, j * D:(j + 1) * D] = M[i * D:(i + 1) * D, j * D:(j + 1) * D]
    A = A + A.T
    A += np.eye(n) * (np.abs(A).sum(1).max() * cond_boost)
    return A


CASES = [  # ncb, D, B, chunk nodes (0: automatic)
    (101, 6, 4, 0),      # C4's coarse level
    (501, 6, 3, 0),      # C2's
    (40, 6, 3, 9),
    (41, 6, 1, 5),
    (64, 6, 2, 3),       # chunks as small as the rule allows (m >= B), many separators
    (97, 3, 2, 0),       # SE(2)
    (23, 3, 4, 5),
    (30, 6, 4, 4),       # m == B: a chunk's first and last s rows are the same rows
    (200, 6, 3, 61),     # three big chunks, the last one short
]


@pytest.mark.parametrize('ncb,D,B,m', CASES)
def test_partitioned_inverse_equals_numpy_and_the_serial_walk(ncb, D, B, m):
    from pyslam_amd.device import band_inverse
    A = banded_spd(ncb, D, B, seed=ncb + D + B)
    ref = np.linalg.inv(A)
    scale = np.abs(ref).max()
    part, _ = band_inverse(A, ncb, D, B, m)
    serial, _ = band_inverse(A, ncb, D, B, -1)
    assert np.abs(part - ref).max() <= 2e-7 * scale          # fp32 storage of an fp64 computation
    assert np.abs(serial - ref).max() <= 2e-7 * scale
    assert np.array_equal(part, part.T)                      # exactly symmetric (mirrored stores): what a CG preconditioner must be
    # the same fp32 matrix up to the last bit or two of the fp64 results behind it
    assert np.abs(part.astype(np.float64) - serial.astype(np.float64)).max() <= 3e-7 * scale


def test_partitioned_inverse_of_an_ill_conditioned_chain():
    """A 1-D Laplacian-like chain (condition number ~ ncb^2, what a coarse level of smooth modes looks like): the separator
    system inherits the conditioning; fp64 throughout, so the inverse is still accurate to fp32 storage."""
    from pyslam_amd.device import band_inverse
    ncb, D, B = 120, 6, 3
    rng = np.random.default_rng(7)
    n = ncb * D
    # A = E^T E + 1e-4 I with E a banded "difference" operator: smooth vectors have tiny energy, cond(A) ~ 1e5-1e6
    E = np.zeros((n, n))
    for i in range(ncb):
        for j in range(max(0, i - 1), min(ncb, i + 2)):
            E[i * D:(i + 1) * D, j * D:(j + 1) * D] = (np.eye(D) if i == j else -0.5 * np.eye(D)) + 0.02 * rng.standard_normal((D, D))
    A = E.T @ E + 1e-4 * np.eye(n)                           # half-bandwidth 2 blocks <= B
    w = np.linalg.eigvalsh(A)
    assert w.min() > 0 and w.max() / w.min() > 1e4
    ref = np.linalg.inv(A)
    part, _ = band_inverse(A, ncb, D, B, 0)
    assert np.abs(part - ref).max() <= 1e-6 * np.abs(ref).max()


def test_partitioned_factorisation_reports_an_indefinite_matrix():
    from pyslam_amd import _native as nat
    from pyslam_amd.device import band_inverse
    A = banded_spd(60, 6, 3, seed=1)
    A[100, 100] = -5.0
    with pytest.raises(nat.NativeError):
        band_inverse(A, 60, 6, 3, 0)
