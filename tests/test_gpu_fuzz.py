"""Randomised parity sweep (tools/fuzz_parity.py): seeded random BA / pose-graph / motion-only problems with random
sizes, losses and solver modes (direct, folded two-level CG, explicit PCG, no coarse level); one device iteration
against the oracle's step.  1 500 further seeds were run when this was written (0 failures)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.gpu
def test_random_problems_match_the_oracle_step():
    import fuzz_parity
    assert fuzz_parity.run(120, seed0=7000, verbose=False) == 0
