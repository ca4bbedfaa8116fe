"""Direct oracle-vs-reference check; only runs where /root/reference exists (build container)."""
import pytest
import torch

from oracle import planner_oracle as po
from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference not present")


def test_oracle_equals_reference_small():
    cfg = po.PlannerConfig.r2r()
    P = po.init_params(cfg, seed=3)
    model = rh.build_reference_model(cfg, P)
    batch = po.make_batch(cfg, B=2, L=10, V=13, G=6, seed=7, ragged=True)
    ro, rg = rh.reference_step(model, batch)
    oo, og = po.step_with_grads(P, cfg, batch)
    assert abs(float(ro["loss"] - oo["loss"])) < 1e-5
    fin = torch.isfinite(ro["global_logits"])
    assert float((ro["global_logits"][fin] - oo["global_logits"][fin]).abs().max()) < 2e-5
    for k in rg:
        assert float((rg[k] - og[k]).abs().max()) < 2e-5, k
