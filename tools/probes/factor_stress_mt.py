"""ps_debug_factor_stress from several host threads at once: K lowest-priority (or ordinary) victim streams of ONE process, each with
its own buffers, each running [producer kernel -> (event) -> factorisation] jobs beside an aggressor -- the shape of
tools/hunt_explicit_flake.py (three handles, three side streams).  Measurement infrastructure.
    python tools/probes/factor_stress_mt.py [threads=3] [launches=300]"""
import ctypes as C
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyslam_amd import _native as nat
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = nat.require_gpu()


def banded_spd(ncb, dof, bw, seed):
    rng = np.random.default_rng(seed)
    n = ncb * dof
    M = np.zeros((n, n))
    for i in range(ncb):
        for j in range(max(0, i - bw), i + 1):
            B = rng.standard_normal((dof, dof)) * (0.3 if i != j else 1.0)
            M[i * dof:(i + 1) * dof, j * dof:(j + 1) * dof] = B
            M[j * dof:(j + 1) * dof, i * dof:(i + 1) * dof] = B.T
    M += np.eye(n) * (np.abs(M).sum(1).max() * 1.5)
    return np.ascontiguousarray(M)


for name, mode, ncb, dof, bw in (('serial band', 0, 45, 6, 3), ('partitioned band', 1, 49, 6, 3), ('dense LDS', 2, 13, 6, 12), ('dense global', 3, 45, 6, 44)):
    for bits, what in ((8, 'produced'), (24, 'produced + event')):
        for lowprio in (0, 1):
            for agg in (1, 2):
                res = [None] * K

                def work(k):
                    A = banded_spd(ncb, dof, bw, 100 + k)
                    nd, npv = C.c_int32(), C.c_int32()
                    rc = lib.ps_debug_factor_stress(nat.f64p(A), ncb, dof, bw, mode | bits, launches, lowprio, agg, C.byref(nd), C.byref(npv))
                    res[k] = (rc, nd.value, npv.value)
                th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
                for t in th: t.start()
                for t in th: t.join()
                print('%-18s [%-16s] %d x %s streams, aggressor %s: differ %s, pivots %s%s' % (
                    name, what, K, 'LOWEST-priority' if lowprio else 'ordinary       ', {1: 'copy', 2: 'LDS '}[agg],
                    [r[1] for r in res], [r[2] for r in res], '' if all(r[0] == 0 for r in res) else '  ERROR'), flush=True)
