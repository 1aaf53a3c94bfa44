"""GPU: a short randomised parity sweep (tools/fuzz_parity.py runs the long one): random ragged shapes, missing values,
all ladder types, through whichever kernel the host picks, against the oracle."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,env", [(101, {}), (102, {"AQ_CHAIN": "3"}), (103, {"AQ_MIS_C": "2", "AQ_CHAIN": "0"})])
def test_random_shapes_match_oracle(seed, env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "25", str(seed)], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "25 cases ok" in out.stdout
