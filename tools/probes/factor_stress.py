"""The coarse level's factorisation kernels under a foreign load (include/pyslam_hip.h: ps_debug_factor_stress; DESIGN.md section 3
"side stream"): every kernel family on a lowest-priority / an ordinary stream, with / without streaming kernels on another stream,
outputs compared bit for bit with an idle run.  Measurement infrastructure.   python tools/probes/factor_stress.py [launches]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyslam_amd import _native as nat

launches = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 400
lib = nat.require_gpu()


def banded_spd(ncb, dof, bw, seed):
    rng = np.random.default_rng(seed)
    n = ncb * dof
    A = np.zeros((n, n))
    for i in range(ncb):
        for j in range(max(0, i - bw), i + 1):
            B = rng.standard_normal((dof, dof)) * (0.3 if i != j else 1.0)
            A[i * dof:(i + 1) * dof, j * dof:(j + 1) * dof] = B
            A[j * dof:(j + 1) * dof, i * dof:(i + 1) * dof] = B.T
    A = A @ A.T                      # SPD, band 2 bw -> keep bw: add a dominant diagonal to the banded part instead
    M = np.zeros_like(A)
    for i in range(ncb):
        for j in range(max(0, i - bw), min(ncb, i + bw + 1)):
            M[i * dof:(i + 1) * dof, j * dof:(j + 1) * dof] = A[i * dof:(i + 1) * dof, j * dof:(j + 1) * dof]
    M += np.eye(n) * (np.abs(M).sum(1).max() * 0.5)
    return np.ascontiguousarray(M)


CASES = [('serial band walk  k_band_chol + k_band_inverse_rl', 0, 45, 6, 3), ('partitioned band  BandPart', 1, 101, 6, 3),
         ('dense LDS         k_coarse_chol<6,true> + k_xcg_ainv', 2, 13, 6, 12), ('dense global      k_coarse_chol<6,false> + k_xcg_ainv', 3, 20, 6, 19),
         ('serial band, SE(2)', 0, 60, 3, 2), ('dense LDS, SE(2)', 2, 24, 3, 23)]
VARIANTS = [('', 0), (' [input produced by a kernel in front]', 8), (' [produced + event recorded in between]', 24)]
if '--produce' in sys.argv:
    CASES = [(n + v, m | bits, a, b, c) for n, m, a, b, c in CASES[:4] for v, bits in VARIANTS[1:]]
for name, mode, ncb, dof, bw in CASES:
    A = banded_spd(ncb, dof, bw, 11 + (mode & 7))
    for lowprio in (0, 1):
        for agg in ((2,) if '--lds' in sys.argv else (0, 1)):
            nd, npv = C.c_int32(), C.c_int32()
            rc = lib.ps_debug_factor_stress(nat.f64p(A), ncb, dof, bw, mode, launches, lowprio, agg, C.byref(nd), C.byref(npv))
            msg = '' if rc == 0 else '  ERROR ' + lib.ps_last_error().decode()
            print('%-98s %s stream, aggressor %-3s: %4d of %d outputs differ, %d non-positive pivots%s' % (
                name, 'LOWEST-priority' if lowprio else 'ordinary       ', {0: 'off', 1: 'ON', 2: 'LDS'}[agg], nd.value, launches, npv.value, msg), flush=True)
