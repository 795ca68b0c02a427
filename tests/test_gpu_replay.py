"""BASELINE.json config 5 on the GPU: pairs rendered along the UZH-FPV indoor_forward_7 trajectory (cuahn_vio_amd/replay.py),
priors from the filter's mean propagation, through the prior-3 model the reference deploys (uzhfpv.launch:58) — HIP vs oracle."""
import numpy as np
import pytest

from conftest import TOL_COV_REL, tol_px_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [pytest.param(2, id="bf16x3"), pytest.param(3, id="f16x2"), pytest.param(0, id="fp32")])
def test_replayed_pairs_match_the_oracle(blob, oracle, precision):
    from cuahn_vio_amd import replay
    from cuahn_vio_amd.homography_net import HnetEngine
    fx = replay.load_fixture("indoor_forward_7")
    prev, curr, prior = replay.render_pairs(fx, first=120, count=10)        # 10 consecutive pairs of one stream (>= 8)
    assert np.abs(prior).max() > 5.0                                        # the drone moves: real priors, not zeros
    eng = HnetEngine(blob, variant="prior3", mc_samples=16, dropout_p=0.05, mc_seed=9, max_batch=10, precision=precision)
    mean, cov = eng.infer_batch(prev, curr, prior, pair_seq0=120)
    eng.close()
    for b in range(10):
        o = oracle.forward(prev[b], curr[b], prior[b], 3, 16, 0.05, 9, 120 + b)
        assert np.abs(mean[b] - o["mean"]).max() < tol_px_vs_oracle(precision), b
        assert np.abs(cov[b] - o["cov"]).max() / np.abs(o["cov"]).max() < TOL_COV_REL, b
