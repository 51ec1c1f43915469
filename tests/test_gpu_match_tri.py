"""GPU parity: Hamming brute force (bit-exact) and stereo triangulation (f64, stated tolerance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nq,nt", [(1, 1), (7, 3), (300, 513), (2000, 2000), (2011, 1987), (5, 0)])
def test_hamming_match_bitexact(api, oracle, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nt > 20:                                     # plant exact duplicates and near matches -> exercises tie-breaking
        t[nt // 2] = t[3]; q[0] = t[3]
        q[1] = t[7]; q[1, 0] ^= 1
    idx, dist = api.hamming_match(q, t)
    ridx, rdist = oracle.hamming_match(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    if nt > 20:
        assert idx[0] == 3 and dist[0] == 0 and dist[1] == 1
    if nt == 0:
        assert (idx == -1).all() and (dist == -1).all()


def test_hamming_on_real_descriptors_and_filter(api, oracle, synth):
    L, R = synth.stereo_pair(1, 0)
    kl, dl = oracle.detect_and_compute(oracle.params(2000), L)
    kr, dr = oracle.detect_and_compute(oracle.params(2000), R)
    idx, dist = api.hamming_match(dl, dr)
    ridx, rdist = oracle.hamming_match(dl, dr)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    keep, mn = api.hamming_filter(dist); rkeep, rmn = oracle.hamming_filter(rdist)
    assert mn == rmn and np.array_equal(keep, rkeep)
    # size-independent properties: self-match is the identity with distance 0; distances are symmetric
    sidx, sdist = api.hamming_match(dl, dl)
    assert (sdist == 0).all()
    uniq = np.unique(dl, axis=0, return_index=True)[1]
    assert np.array_equal(sidx[uniq], uniq)


@pytest.mark.parametrize("rep", [1, 4])
def test_hamming_batch(api, oracle, rep):
    """5 pairs: fewer than 16 per call = one query tile per wave, four wave groups per block on every fourth train chunk (k_hamming_fp4<1, 4>);
    20 pairs: the batch form (k_hamming_fp4<4>).  Counts cover: fewer train chunks than wave groups, one row, none."""
    import torch
    rng = np.random.default_rng(3)
    B, cap = 5 * rep, 700
    nq = np.tile(np.array([700, 1, 350, 0, 512], np.int32), rep); nt = np.tile(np.array([650, 700, 2, 10, 0], np.int32), rep)
    if rep > 1: nt[5:10] = [33, 64, 65, 127, 129]                     # chunk counts 2, 2, 3, 4, 5
    q = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8); t = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    dnq, dnt = torch.from_numpy(nq).cuda(), torch.from_numpy(nt).cuda()
    di = torch.full((B, cap), -7, dtype=torch.int32, device="cuda"); dd = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    api.hamming_match_batch(dq.data_ptr(), dnq.data_ptr(), dt.data_ptr(), dnt.data_ptr(), B, cap, di.data_ptr(), dd.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    gi, gd = di.cpu().numpy(), dd.cpu().numpy()
    for b in range(B):
        ri, rd = oracle.hamming_match(q[b, :nq[b]], t[b, :nt[b]])
        assert np.array_equal(gi[b, :nq[b]], ri) and np.array_equal(gd[b, :nq[b]], rd)
        assert (gi[b, nq[b]:] == -7).all()          # rows past the count are never written


def test_triangulate_stereo(api, oracle, synth):
    K = synth.KITTI00
    b = K["bf"] / K["fx"]
    rng = np.random.default_rng(4)
    n = 3000
    Z = rng.uniform(3, 80, n); X = rng.uniform(-15, 15, n); Y = rng.uniform(-3, 3, n)
    xl = (K["fx"] * X / Z + K["cx"]).astype(np.float32); yl = (K["fy"] * Y / Z + K["cy"]).astype(np.float32)
    xr = (K["fx"] * (X - b) / Z + K["cx"] + rng.normal(0, 0.3, n)).astype(np.float32)
    yr = (yl + rng.normal(0, 0.3, n)).astype(np.float32)
    yr[:50] += 30                                    # gross epipolar violations -> rejected by the sigma ratio
    xr[50:80] = xl[50:80] + 5                        # negative disparity -> z < 0
    xyz, ok = api.triangulate_stereo(xl, yl, xr, yr, K["fx"], K["fy"], K["cx"], K["cy"], b)
    rxyz, rok = oracle.triangulate_stereo(xl, yl, xr, yr, K["fx"], K["fy"], K["cx"], K["cy"], b)
    assert np.array_equal(ok, rok)
    assert not ok[:80].any() and ok[80:].mean() > 0.95
    # tolerance: f64 one-sided Jacobi on both sides; agreement far below the measurement noise
    assert np.allclose(xyz[ok], rxyz[ok], rtol=1e-9, atol=1e-9)
    assert np.median(np.abs(xyz[ok][:, 2] - Z[ok]) / Z[ok]) < 0.05       # 0.3 px noise on 5..130 px disparities


@pytest.mark.parametrize("B", [2, 17])
def test_match_triangulate_in_one_call_equals_the_two_calls(api, B):
    """myslam_hamming_match_triangulate_batch = myslam_hamming_match_batch + myslam_triangulate_stereo_batch, every output bit for bit: for fewer
    than 16 pairs it is ONE launch (each matcher block triangulates its 128 queries, k_hamming_fp4<1, 4, true>), from 16 pairs on the two launches."""
    import torch
    rng = np.random.default_rng(11 + B)
    cap = 900
    nl = rng.integers(0, cap + 1, B).astype(np.int32); nr = rng.integers(0, cap + 1, B).astype(np.int32)
    nl[0] = cap; nr[0] = cap
    if B > 1: nr[1] = 0                                                # a pair without right key-points: no match, xyz = 0, ok = 0
    dl = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8); dr = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)
    kp = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    kl = np.zeros((B, cap), kp); kr = np.zeros((B, cap), kp)
    kl["x"] = rng.uniform(20, 1220, (B, cap)); kl["y"] = rng.uniform(20, 350, (B, cap))
    kr["x"] = rng.uniform(20, 1220, (B, cap)); kr["y"] = kl["y"] + rng.normal(0, 0.4, (B, cap))
    assert kp.itemsize == 28
    K = (718.856, 718.856, 607.1928, 185.2157); base = 0.537
    dev = lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype == kp else a).cuda()
    d = {k: dev(v) for k, v in dict(dl=dl, dr=dr, nl=nl, nr=nr, kl=kl, kr=kr).items()}
    s = torch.cuda.current_stream().cuda_stream

    def outs():
        return [torch.full((B, cap), -7, dtype=torch.int32, device="cuda"), torch.full((B, cap), -7, dtype=torch.int32, device="cuda"),
                torch.full((B, cap, 3), -7.0, dtype=torch.float64, device="cuda"), torch.full((B, cap), 9, dtype=torch.uint8, device="cuda")]
    a = outs()
    api.hamming_match_batch(d["dl"].data_ptr(), d["nl"].data_ptr(), d["dr"].data_ptr(), d["nr"].data_ptr(), B, cap, a[0].data_ptr(), a[1].data_ptr(), s)
    api.triangulate_stereo_batch(d["kl"].data_ptr(), d["kr"].data_ptr(), a[0].data_ptr(), d["nl"].data_ptr(), B, cap, K, base, a[2].data_ptr(), a[3].data_ptr(), s)
    b = outs()
    api.hamming_match_triangulate_batch(d["dl"].data_ptr(), d["nl"].data_ptr(), d["dr"].data_ptr(), d["nr"].data_ptr(), d["kl"].data_ptr(), d["kr"].data_ptr(),
                                        B, cap, K, base, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), s)
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), f"output {i}"
    assert int(a[3][0].sum()) >= 0 and (a[0][0] >= 0).all() and (a[3][:, :][a[0] == -7] == 9).all()      # slots past a pair's count are never written
    if B > 1: assert (b[0][1, :nl[1]] == -1).all() and not b[3][1, :nl[1]].any()
