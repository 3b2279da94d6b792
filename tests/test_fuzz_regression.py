"""A fixed handful of cases of tools/fuzz_configs.py (random constructor arguments x input shapes against the CPU oracle) as a regression
guard: the sweep that found the band-kernel bug of round 5 keeps running on the seeds below.  Each case checks eval output, train output,
loss and every gradient (outputs 1e-4; gradients max(1e-4, 3 x the CPU fp32 path's distance from fp64))."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

# seeds that ran clean with the generator as committed, chosen for spread: instance / batch / group norms on either side, two encoder
# blocks, one to three decoder blocks, width 96, iso and separate heads, att_mean / mean, elu, and the shapes 33x47 (the one that exposed the
# bug: H = 1 mod 32), 40x100, 50x46, 72x60, 128x32, 64x64.  (Seed 17 -- width 96 at 33x47 -- is the documented limit of the tail
# corrections, DESIGN 3b: a per-frame product at 1.4e-4.)
CASES = [1, 4, 8, 9, 12, 15, 27]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_fuzz_case_stays_inside_the_contract(case):
    import fuzz_configs
    outside, refused = fuzz_configs.run_case(case)
    assert not refused, "a configuration that ran in round 5 is refused now"
    assert not outside
