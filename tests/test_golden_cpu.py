"""CPU: the oracle restatement reproduces the golden vectors that were generated
from the reference's own C (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from handbrake_amd import synth
import golden_cases as gc
import oracle_stream as os_

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_oracle_matches_golden(built, name):
    case = gc.CASES[name]
    want, _ = os_.load_golden(os.path.join(GOLD, name + ".npz"))
    frames = synth.stream(case["model"], case["w"], case["h"], case["n"])
    got = os_.run_chain(frames, case["orc"])
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[t][c], err_msg=f"{name} frame {t} plane {c}")
