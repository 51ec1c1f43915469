"""myslam_ba_flatten_window: the Map -> flat-array rules of Backend::OptimizeActiveMap (src/backend.cpp:139-206) — a host function of the
C-ABI library (no device needed), checked against a literal dict-walk of those lines."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def api_host(pkg):
    import os
    if not os.path.exists(pkg.api.LIB_PATH):
        pkg.build_library()
    return pkg.api


def reference_walk(active, mps, obs_of):
    """backend.cpp:139-206 on Python containers.  active: {kf_id: pose}; mps: {mp_id: (outlier, first_observer_kf)};
    obs_of: {mp_id: [(kf_id, (u, v), feat_outlier, row)]} in GetActiveObservations list order.
    Vertices sorted by id (g2o), edges grouped by landmark in list order, landmarks without an edge dropped."""
    kf_slot = {k: i for i, k in enumerate(sorted(active))}
    pts, fixed, ep, el, eo, es = [], [], [], [], [], []
    for m in sorted(mps):
        outlier, first = mps[m]
        if outlier:                                       # :163
            continue
        edges = []
        for kf, uv, fo, row in obs_of.get(m, []):
            assert kf in active                           # :187
            if fo:                                        # :189
                continue
            edges.append((kf_slot[kf], uv, row))
        if not edges:
            continue
        j = len(pts)
        pts.append(m); fixed.append(0 if first in active else 1)      # :175-177
        for slot, uv, row in edges:
            ep.append(slot); el.append(j); eo.append(uv); es.append(row)
    return sorted(active), pts, fixed, ep, el, eo, es


def random_map(seed, n_kf=6, n_mp=60):
    rng = np.random.default_rng(seed)
    kf_ids = rng.choice(np.arange(3, 200), n_kf, replace=False).astype(np.uint64)
    outside = np.array([1, 2, 250], np.uint64)                      # key-frames that left the window
    mp_ids = rng.choice(np.arange(10, 5000), n_mp, replace=False).astype(np.uint64)
    mp_out = (rng.random(n_mp) < 0.15).astype(np.uint8)
    first = np.where(rng.random(n_mp) < 0.3, rng.choice(outside, n_mp), rng.choice(kf_ids, n_mp)).astype(np.uint64)
    rows = []
    for i, m in enumerate(mp_ids):
        if rng.random() < 0.1:
            continue                                                 # an active map point without active observations
        for kf in rng.permutation(kf_ids)[: rng.integers(1, n_kf + 1)]:
            rows.append((m, kf, rng.uniform(0, 1241), rng.uniform(0, 376), rng.random() < 0.2))
    # interleave the map points' rows (the order of rows of ONE map point is its list order and must survive)
    order = np.argsort(rng.random(len(rows)), kind="stable")
    rows = [rows[i] for i in order]
    return kf_ids, mp_ids, mp_out, first, rows


@pytest.mark.parametrize("seed", range(8))
def test_flatten_matches_reference_walk(api_host, seed):
    kf_ids, mp_ids, mp_out, first, rows = random_map(seed)
    om = np.array([r[0] for r in rows], np.uint64); ok = np.array([r[1] for r in rows], np.uint64)
    uv = np.array([[r[2], r[3]] for r in rows], np.float32); fo = np.array([r[4] for r in rows], np.uint8)
    got = api_host.ba_flatten_window(kf_ids, mp_ids, mp_out, first, om, ok, uv, fo)
    active = {int(k): None for k in kf_ids}
    mps = {int(m): (bool(o), int(f)) for m, o, f in zip(mp_ids, mp_out, first)}
    obs_of = {}
    for row, (m, kf, u, v, f) in enumerate(rows):
        obs_of.setdefault(int(m), []).append((int(kf), (np.float32(u), np.float32(v)), bool(f), row))
    kfs, pts, fixed, ep, el, eo, es = reference_walk(active, mps, obs_of)
    assert [int(kf_ids[i]) for i in got["pose_src"]] == kfs
    assert [int(mp_ids[i]) for i in got["pt_src"]] == pts
    assert got["fixed"].tolist() == fixed and got["edge_pose"].tolist() == ep and got["edge_pt"].tolist() == el and got["edge_src"].tolist() == es
    assert np.array_equal(got["edge_obs"], np.array(eo, np.float64).reshape(-1, 2))          # toVec2: float -> double, exact
    # what the solve kernels need: edges grouped by landmark, every landmark has at least one edge
    assert np.all(np.diff(got["edge_pt"]) >= 0) and set(got["edge_pt"].tolist()) == set(range(len(pts)))
    assert len(pts) > 5 and 0 < sum(fixed) < len(pts) and len(es) < len(rows)


def test_flatten_errors_and_empty(api_host):
    kf = np.array([5, 9], np.uint64); mp = np.array([100, 101], np.uint64); mo = np.zeros(2, np.uint8); mf = np.array([5, 1], np.uint64)
    uv = np.zeros((1, 2), np.float32)
    with pytest.raises(api_host.MyslamError):      # observation from a key-frame outside the active set: the reference's assert (:187)
        api_host.ba_flatten_window(kf, mp, mo, mf, np.array([100], np.uint64), np.array([7], np.uint64), uv, np.zeros(1, np.uint8))
    with pytest.raises(api_host.MyslamError):      # observation of a map point that is not in the active set
        api_host.ba_flatten_window(kf, mp, mo, mf, np.array([555], np.uint64), np.array([5], np.uint64), uv, np.zeros(1, np.uint8))
    with pytest.raises(api_host.MyslamError):      # duplicate key-frame id
        api_host.ba_flatten_window(np.array([5, 5], np.uint64), mp, mo, mf, np.array([100], np.uint64), np.array([5], np.uint64), uv, np.zeros(1, np.uint8))
    got = api_host.ba_flatten_window(kf, mp, mo, mf, np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros((0, 2), np.float32), np.zeros(0, np.uint8))
    assert len(got["pt_src"]) == 0 and len(got["edge_pose"]) == 0 and got["pose_src"].tolist() == [0, 1]
    got = api_host.ba_flatten_window(kf, mp, mo, mf, np.array([101, 100, 101], np.uint64), np.array([9, 9, 5], np.uint64),
                                     np.array([[1, 2], [3, 4], [5, 6]], np.float32), np.array([0, 0, 0], np.uint8))
    assert got["pt_src"].tolist() == [0, 1] and got["fixed"].tolist() == [0, 1]           # 101's first observer (1) left the window
    assert got["edge_pt"].tolist() == [0, 1, 1] and got["edge_pose"].tolist() == [1, 1, 0] and got["edge_src"].tolist() == [1, 0, 2]
