"""GPU parity: BA residual/Jacobian/block-Hessian build (f64).  LDS f64 atomics make the summation order
free, so agreement with the oracle is to rounding (rtol 1e-11 on O(1e6) entries), not bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _close(a, b, name):
    scale = max(1.0, np.abs(b).max())
    err = np.abs(a - b).max() / scale
    assert err < 1e-11, (name, err)


@pytest.mark.parametrize("n_kf,n_mp", [(10, 300), (7, 120), (2, 3), (20, 1000)])
def test_ba_build_matches_oracle(api, oracle, synth, n_kf, n_mp):
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem(seed=0xBA + n_kf, n_kf=n_kf, n_mp=n_mp)
    got = api.ba_build(poses, pts, ep, el, obs, fixed, K)
    ref = oracle.ba_build(poses, pts, ep, el, obs, fixed, K)
    for g, r, name in zip(got, ref, ["Hpp", "Hll", "Hpl", "bp", "bl", "chi2"]):
        _close(g, r, name)
    assert np.array_equal(got[0], np.transpose(got[0], (0, 2, 1)))        # exactly symmetric by construction
    assert len(ep) > n_kf


def test_ba_no_fixed_and_huber_off(api, oracle, synth):
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem(n_kf=5, n_mp=80)
    for fx, delta in ((None, 5.991), (fixed, 1e9), (np.ones_like(fixed), 5.991)):
        got = api.ba_build(poses, pts, ep, el, obs, fx, K, delta)
        ref = oracle.ba_build(poses, pts, ep, el, obs, fx if fx is not None else np.zeros_like(fixed), K, delta)
        for g, r, name in zip(got, ref, ["Hpp", "Hll", "Hpl", "bp", "bl", "chi2"]):
            _close(g, r, name)


def test_ba_build_batch(api, oracle, synth):
    import torch
    W, maxP, maxL, maxE = 6, 10, 300, 3000
    rng = np.random.default_rng(0)
    probs = [synth.ba_problem(seed=100 + w, n_kf=int(rng.integers(3, 11)), n_mp=int(rng.integers(20, 300))) for w in range(W)]
    poses = np.zeros((W, maxP, 7)); poses[..., 3] = 1
    pts = np.zeros((W, maxL, 3)); ep = np.zeros((W, maxE), np.int32); el = np.zeros((W, maxE), np.int32)
    obs = np.zeros((W, maxE, 2)); fixed = np.zeros((W, maxL), np.uint8); sizes = np.zeros((W, 3), np.int32)
    for w, (p, x, a, b, o, f, K) in enumerate(probs):
        assert len(a) <= maxE
        poses[w, :len(p)] = p; pts[w, :len(x)] = x; ep[w, :len(a)] = a; el[w, :len(a)] = b; obs[w, :len(a)] = o; fixed[w, :len(f)] = f
        sizes[w] = (len(p), len(x), len(a))
    dev = lambda a: torch.from_numpy(a).cuda()
    d = [dev(a) for a in (poses, pts, ep, el, obs, fixed, sizes)]
    Hpp = torch.zeros(W, maxP, 36, dtype=torch.float64, device="cuda"); Hll = torch.zeros(W, maxL, 9, dtype=torch.float64, device="cuda")
    Hpl = torch.zeros(W, maxE, 18, dtype=torch.float64, device="cuda"); bp = torch.zeros(W, maxP, 6, dtype=torch.float64, device="cuda")
    bl = torch.zeros(W, maxL, 3, dtype=torch.float64, device="cuda"); chi = torch.zeros(W, maxE, dtype=torch.float64, device="cuda")
    K = probs[0][6]
    api.ba_build_batch(*[t.data_ptr() for t in d], W, maxP, maxL, maxE, K, 5.991, Hpp.data_ptr(), Hll.data_ptr(), Hpl.data_ptr(),
                       bp.data_ptr(), bl.data_ptr(), chi.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for w, (p, x, a, b, o, f, K) in enumerate(probs):
        ref = oracle.ba_build(p, x, a, b, o, f, K)
        P, L, E = len(p), len(x), len(a)
        _close(Hpp[w, :P].cpu().numpy().reshape(P, 6, 6), ref[0], "Hpp")
        _close(Hll[w, :L].cpu().numpy().reshape(L, 3, 3), ref[1], "Hll")
        _close(Hpl[w, :E].cpu().numpy().reshape(E, 6, 3), ref[2], "Hpl")
        _close(bp[w, :P].cpu().numpy(), ref[3], "bp"); _close(bl[w, :L].cpu().numpy(), ref[4], "bl")
        _close(chi[w, :E].cpu().numpy(), ref[5], "chi2")
