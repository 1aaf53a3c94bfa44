"""GPU: a short randomised parity sweep (tests/tools/fuzz_parity.py runs the long one): random ragged shapes, missing values,
all ladder types, through whichever kernel the host picks, against the oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,env", [(101, {}), (102, {"AQ_CHAIN": "3"}), (103, {"AQ_MIS_C": "2", "AQ_CHAIN": "0"}),
                                      (104, {"AQ_TT": "2"}), (105, {"AQ_TT": "2", "AQ_CHAIN": "4", "AQ_STAGGER": "1"}),
                                      (106, {"AQ_LA_C": "2"}), (107, {"AQ_LA_C": "3", "AQ_LA_XHELPER": "1"})])
def test_random_shapes_match_oracle(seed, env, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_parity
    for k, v in env.items():
        monkeypatch.setenv(k, v)       # read by aq_vb_create
    worst = fuzz_parity.run(25, seed)
    assert worst["elbo"] < 1e-8 and worst["mu"] < 1e-6
