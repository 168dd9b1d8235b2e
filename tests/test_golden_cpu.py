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
    frames = synth.stream(case["model"], case["w"], case["h"], case["n"], depth=case.get("depth", 8))
    got = os_.run_chain(frames, case["orc"], flags=synth.flags_for(case["model"]))
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[t][c], err_msg=f"{name} frame {t} plane {c}")
    meta = os_.run_chain.last_meta
    if meta is not None:
        _, gold_meta = os_.load_golden(os.path.join(GOLD, name + ".npz"))
        for t, m in enumerate(meta):
            assert m["combed"] == int(gold_meta[t][3]), f"{name} frame {t} combed"
            if "start" in m:
                assert (m["start"], m["stop"]) == (int(gold_meta[t][0]), int(gold_meta[t][1])), f"{name} frame {t} timestamps"
