"""GPU time of the coarse level's band factorisation + inverse: the one-workgroup column walk against the partitioned form
(csrc/ps_k_bandpart.h) at C2's and C4's coarse sizes, over chunk sizes.   python tools/bandpart_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyslam_amd.device import band_inverse
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_bandpart import banded_spd

for name, ncb, D, B, ms in (('C4', 101, 6, 4, (0, 8, 12, 16, 24)), ('C2', 501, 6, 3, (0, 20, 28, 36, 48, 64)),
                            ('C2 at 401 nodes', 401, 6, 3, (0,)), ('SE(2) 400 nodes', 400, 3, 2, (0,))):
    A = banded_spd(ncb, D, B, seed=3)
    ref = np.linalg.inv(A)
    inv, us = band_inverse(A, ncb, D, B, -1)
    print('%s: ncb %d D %d B %d: serial walk %.1f us (err %.1e)' % (name, ncb, D, B, us, np.abs(inv - ref).max() / np.abs(ref).max()))
    for m in ms:
        inv, us = band_inverse(A, ncb, D, B, m)
        print('    partitioned, chunk nodes %3d: %.1f us (err %.1e)' % (m, us, np.abs(inv - ref).max() / np.abs(ref).max()))
