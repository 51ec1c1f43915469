"""Loop verification on the GPU: cv::solvePnPRansac (src/loopclosing.cpp:262-268) against the oracle's restatement.

The RANSAC stage (cv::RNG samples, EPnP per sample, float inlier test, best-count bookkeeping) contains no transcendental function
and is built without FMA contraction in the oracle's operation order: consensus set, count and winning hypothesis must be
IDENTICAL.  The refinement evaluates sin / cos in exp(): refined pose to 1e-9.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cmp(api, oracle, pw, uv, K, **kw):
    rc, rp, rin, rn = oracle.solve_pnp_ransac(pw, uv, K, **kw)
    if rc != 0:                                            # no model (OpenCV returns false): the device must say so too
        with pytest.raises(Exception):
            api.solve_pnp_ransac(pw, uv, K, **kw)
        return None, rin, 0
    gp, gin, gn = api.solve_pnp_ransac(pw, uv, K, **kw)
    assert gn == rn and np.array_equal(gin, rin)
    s = np.sign(np.dot(gp[:4], rp[:4]))
    assert np.abs(gp[:4] * s - rp[:4]).max() < 1e-9 and np.abs(gp[4:] - rp[4:]).max() < 1e-9
    return gp, gin, gn


@pytest.mark.parametrize("n,frac,noise,seed", [(150, 0.3, 0.5, 1), (60, 0.5, 0.8, 2), (400, 0.2, 0.3, 3), (25, 0.2, 0.5, 4), (1200, 0.6, 1.0, 5)])
def test_pnp_ransac_matches_oracle(api, oracle, synth, n, frac, noise, seed):
    pw, uv, K, pose, good = synth.pnp_problem(n, frac, noise, seed=seed)
    gp, gin, gn = _cmp(api, oracle, pw, uv, K)
    assert (gin & good).sum() >= 0.9 * good.sum() and (gin & ~good).sum() <= 0.05 * n + 1        # the consensus set is the true one
    s = np.sign(np.dot(gp[:4], pose[:4]))
    assert np.abs(gp[4:] - pose[4:]).max() < 0.05 and np.abs(gp[:4] * s - pose[:4]).max() < 2e-3  # and the pose is the true pose


def test_pnp_ransac_parameters_and_edge_cases(api, oracle, synth):
    pw, uv, K, pose, good = synth.pnp_problem(200, 0.4, 0.5, seed=9)
    _cmp(api, oracle, pw, uv, K, iterations=10)
    assert _cmp(api, oracle, pw, uv, K, iterations=40)[2] > 100
    _cmp(api, oracle, pw, uv, K, iterations=300, reproj_error=2.0, confidence=0.999)
    _cmp(api, oracle, pw[:5], uv[:5], K)                                  # exactly the model size
    # planar scene (all map points on a wall): EPnP's degenerate branch
    pw2 = pw.copy(); pw2[:, 2] = pw2[:, 2].mean()
    _cmp(api, oracle, pw2, uv, K)
    with pytest.raises(Exception):
        api.solve_pnp_ransac(pw[:4], uv[:4], K)                           # fewer points than the model needs
    # pure noise: whatever the oracle decides, the device decides the same
    rng = np.random.default_rng(0)
    pwr = rng.normal(0, 10, (80, 3)).astype(np.float32); uvr = rng.uniform(0, 1000, (80, 2)).astype(np.float32)
    _cmp(api, oracle, pwr, uvr, K)
