"""The CPU oracle (oracle/planner_oracle.py) against fixtures generated from the REAL
reference (oracle/make_golden.py).  fp32, tolerance 2e-5 abs (observed ~3e-6)."""
import pytest

from oracle import planner_oracle as po
from tests.golden_util import load_case, compare_outputs, compare_grads

CASES = ["c1_single_episode", "ragged_small", "c2_shape_b2", "c5_g64_b2", "c4_rxr_b1"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    z, cfg, batch = load_case(name)
    P = po.init_params(cfg, seed=0)
    outs, grads = po.step_with_grads(P, cfg, batch)
    compare_outputs(z, outs, atol=2e-5)
    compare_grads(z, grads, atol=2e-5, rel=1e-4)
