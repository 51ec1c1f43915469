"""Known-answer and cross-implementation tests that pin the CPU oracle (CPU only, no GPU).

The reference has no tests and cannot be built here, so the oracle is pinned by
  * hand-derivable KATs from the reference's in-tree formulas (SURVEY.md §8 header numbers),
  * the reference's own in-tree FAST predicate (isFastCorner, ORBextractor.cpp:449-511) against the
    restated cv::FAST score,
  * independent second implementations (numpy / LAPACK / torch / a literal Python list walk of
    DistributeOctTree) of every stage.
"""
import math
import zlib

import numpy as np
import pytest


# ------------------------------------------------------------------------------------------- tables
def test_feature_budgets_and_umax(oracle):
    sc, isc, npl, umax = oracle.orb_tables(oracle.params(2000))
    assert npl.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert oracle.orb_tables(oracle.params(300))[2].tolist() == [65, 54, 45, 38, 31, 26, 22, 19]
    assert oracle.orb_tables(oracle.params(100))[2].tolist() == [22, 18, 15, 13, 10, 9, 7, 6]
    assert umax.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert int(sum(2 * u + 1 for u in umax[1:]) * 2 + 2 * umax[0] + 1) == 749         # circular patch size
    assert sc[0] == 1.0 and abs(sc[7] - 1.2 ** 7) < 1e-5


def test_pyramid_sizes(oracle):
    _, isc, _, _ = oracle.orb_tables(oracle.params())
    sizes = [oracle.level_size(1241, 376, float(s)) for s in isc]
    assert sizes == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    assert sum(w * h for w, h in sizes) == 1444097


def test_pattern_table(oracle):
    pat = oracle.pattern()
    assert pat.shape == (1024,) and pat.dtype == np.int8
    assert zlib.crc32(pat.tobytes()) == 0xD1A39030
    assert pat[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]            # ORBextractor.cpp:103-104
    assert pat[-4:].tolist() == [-1, -6, 0, -11]                      # :358
    assert np.abs(pat).max() <= 13


# ------------------------------------------------------------------------------------------- resize / blur
def test_resize_identity_and_constant(oracle, synth):
    img = synth.random_image(1, 60, 80)
    assert np.array_equal(oracle.resize(img, 80, 60), img)
    const = np.full((50, 70), 173, np.uint8)
    assert np.all(oracle.resize(const, 58, 42) == 173)


def test_resize_close_to_float_bilinear(oracle, synth):
    img = synth.random_image(2, 120, 150)
    dw, dh = 125, 100
    out = oracle.resize(img, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (150 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (120 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 149); y0 = np.clip(np.floor(sy).astype(int), 0, 119)
    x1 = np.clip(x0 + 1, 0, 149); y1 = np.clip(y0 + 1, 0, 119)
    fx = np.clip(sx - x0, 0, 1); fy = np.clip(sy - y0, 0, 1)
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy)[:, None] + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy[:, None]
    assert np.abs(out - ref).max() <= 1.0


def test_blur_coefficients_and_impulse(oracle):
    # kind 0 (sigma = 2): every normalised tap rounded to Q8 on its own, as OpenCV 3.4.8's getFixedpointGaussianKernel does -> sum 257;
    # kind 1 (sigma <= 0): OpenCV's fixed 7-tap table, sum 256
    g = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    assert np.rint(g / g.sum() * 256).astype(int).tolist() == [18, 34, 49, 55, 49, 34, 18]
    for kind, q in ((0, [18, 34, 49, 55, 49, 34, 18]), (1, [8, 28, 56, 72, 56, 28, 8])):
        assert sum(q) == (257 if kind == 0 else 256)
        img = np.zeros((21, 21), np.uint8); img[10, 10] = 255
        out = oracle.blur7(img, kind)
        qq = np.array(q, np.int64)
        expect = ((np.outer(qq, qq) * 255 + 32768) >> 16).astype(np.uint8)
        assert np.array_equal(out[7:14, 7:14], expect)
        for c in (0, 1, 100, 201, 254, 255):          # constant images: (c * sum^2 + 2^15) >> 16, saturated to 255 (ufixedpoint32 -> u8)
            const = np.full((16, 40), c, np.uint8)
            assert np.all(oracle.blur7(const, kind) == min(255, (c * sum(q) ** 2 + 32768) >> 16)), (kind, c)


def test_blur_reflect101(oracle):
    img = np.zeros((9, 30), np.uint8); img[:, 0] = 200            # column at the border
    out = oracle.blur7(img, 1)
    # with gfedcb|abcdefgh reflection the border pixel is counted once, its mirror positions hold zeros
    q = [8, 28, 56, 72, 56, 28, 8]
    assert out[4, 0] == (q[3] * 200 * 256 + 32768) >> 16
    assert out[4, 1] == ((q[2]) * 200 * 256 + 32768) >> 16        # x=1: taps at -2..4 -> reflect(-1)=1,(−2)=2: pixel 0 hit once


# ------------------------------------------------------------------------------------------- FAST
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_fast_score_constructed_ring(oracle):
    img = np.full((15, 15), 100, np.uint8)
    for k in range(9):                                              # 9 contiguous brighter pixels
        dx, dy = RING[(3 + k) % 16]
        img[7 + dy, 7 + dx] = 150
    s = oracle.fast_score_map(img, 7)
    assert s[7, 7] == 49                                            # corner for every t < 50
    img2 = img.copy(); dx, dy = RING[3]; img2[7 + dy, 7 + dx] = 100  # only 8 contiguous -> not a corner
    assert oracle.fast_score_map(img2, 7)[7, 7] == 0
    dark = np.full((15, 15), 100, np.uint8)
    for k in range(11):
        dx, dy = RING[k]; dark[7 + dy, 7 + dx] = 60 + k               # min diff over best 9-arc
    assert oracle.fast_score_map(dark, 7)[7, 7] == (100 - 68) - 1     # arcs of 9 among 60..70: best min d = 100-68


def test_fast_score_equals_reference_predicate(oracle, synth):
    """corner@t <=> score >= t, with the reference's own in-tree isFastCorner (ORBextractor.cpp:449-511)."""
    img = synth.random_image(3, 64, 96)
    rng = np.random.default_rng(0)
    for th in (7, 20, 35):
        smap = oracle.fast_score_map(img, 1)
        for _ in range(600):
            x = int(rng.integers(3, 93)); y = int(rng.integers(3, 61))
            assert oracle.is_fast_corner(img, x, y, th) == (smap[y, x] >= th), (x, y, th, smap[y, x])


def test_fast_detect_is_strict_local_max(oracle, synth):
    img = synth.random_image(4, 48, 70)
    smap = oracle.fast_score_map(img, 7).astype(int)
    xs, ys, sc = oracle.fast_detect(img, 7)
    assert len(xs) > 10
    assert list(zip(ys.tolist(), xs.tolist())) == sorted(zip(ys.tolist(), xs.tolist()))     # row-major order
    got = set(zip(xs.tolist(), ys.tolist()))
    for y in range(3, 45):
        for x in range(3, 67):
            s = smap[y, x]
            nb = smap[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
            is_max = s > 0 and s > nb.max()
            assert ((x, y) in got) == is_max
    # threshold 20 == threshold-7 result filtered by score (what the HIP path exploits)
    xs20, ys20, sc20 = oracle.fast_detect(img, 20)
    keep = sc >= 20
    assert np.array_equal(xs20, xs[keep]) and np.array_equal(ys20, ys[keep]) and np.array_equal(sc20, sc[keep])


def test_grid_fast_bounds_and_mask(oracle, synth):
    img = synth.random_image(5, 150, 220)
    xs, ys, sc = oracle.grid_fast(img)
    assert len(xs) > 50 and len(set(zip(xs.tolist(), ys.tolist()))) == len(xs)
    assert xs.min() >= 3 and ys.min() >= 3 and xs.max() < 220 - 32 - 3 and ys.max() < 150 - 32 - 3
    mask = np.full_like(img, 255); mask[:, :60] = 0                 # mask is indexed border-relative (reference quirk)
    xm, ym, _ = oracle.grid_fast(img, mask=mask)
    assert set(zip(xm.tolist(), ym.tolist())) == {(x, y) for x, y in zip(xs.tolist(), ys.tolist()) if x >= 60}


# ------------------------------------------------------------------------------------------- oct-tree
def py_octree(xs, ys, sc, minX, maxX, minY, maxY, N):
    """Literal Python walk of DistributeOctTree (ORBextractor.cpp:586-810), lists instead of std::list."""
    f32 = np.float32
    nIni = int(round(float(f32(maxX - minX) / f32(maxY - minY))))
    hX = f32(maxX - minX) / f32(nIni)
    serial = [0]

    def mk(ULx, ULy, BRx, BRy, keys):
        serial[0] += 1
        return dict(ULx=ULx, ULy=ULy, BRx=BRx, BRy=BRy, keys=keys, no=len(keys) == 1, id=serial[0])

    nodes = [mk(int(hX * f32(i)), 0, int(hX * f32(i + 1)), maxY - minY, []) for i in range(nIni)]
    for k in range(len(xs)):
        nodes[int(f32(xs[k]) / hX)]["keys"].append(k)
    L = [n for n in nodes if n["keys"]]
    for n in L:
        n["no"] = len(n["keys"]) == 1

    def divide(n):
        hx = math.ceil((n["BRx"] - n["ULx"]) / 2); hy = math.ceil((n["BRy"] - n["ULy"]) / 2)
        mx, my = n["ULx"] + hx, n["ULy"] + hy
        ks = [[], [], [], []]
        for k in n["keys"]:
            if xs[k] < mx:
                ks[0 if ys[k] < my else 2].append(k)
            else:
                ks[1 if ys[k] < my else 3].append(k)
        b = [(n["ULx"], n["ULy"], mx, my), (mx, n["ULy"], n["BRx"], my), (n["ULx"], my, mx, n["BRy"]), (mx, my, n["BRx"], n["BRy"])]
        return [mk(*b[q], ks[q]) for q in range(4)]

    finish = False
    while not finish:
        prev = len(L)
        newL, big, keep = [], [], []
        for n in L:
            if n["no"]:
                keep.append(n)
                continue
            for ch in divide(n):
                if ch["keys"]:
                    newL.insert(0, ch)
                    if len(ch["keys"]) > 1:
                        big.append(ch)
        L = newL + keep
        if len(L) >= N or len(L) == prev:
            finish = True
        elif len(L) + 3 * len(big) > N:
            while not finish:
                prev = len(L)
                order = sorted(big, key=lambda n: (len(n["keys"]), n["id"]))
                big = []
                for n in reversed(order):
                    for ch in divide(n):
                        if ch["keys"]:
                            L.insert(0, ch)
                            if len(ch["keys"]) > 1:
                                big.append(ch)
                    L.remove(n)
                    if len(L) >= N:
                        break
                if len(L) >= N or len(L) == prev:
                    finish = True
    out = []
    for n in L:
        best = n["keys"][0]
        for k in n["keys"][1:]:
            if sc[k] > sc[best]:
                best = k
        out.append(best)
    return out


@pytest.mark.parametrize("seed,n,N", [(0, 40, 10), (1, 300, 60), (2, 1000, 122), (3, 2500, 434), (4, 7, 50), (5, 2, 5),
                                      (6, 500, 500), (7, 64, 1)])
def test_octree_matches_literal_python_walk(oracle, seed, n, N):
    rng = np.random.default_rng(seed)
    W, H = 1209, 344                                               # KITTI level 0: maxBorder - minBorder
    pts = set()
    while len(pts) < n:
        pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
    pts = sorted(pts, key=lambda p: (p[1] // 32, p[0] // 31, p[1], p[0]))   # cell-major like the grid
    xs = np.array([p[0] for p in pts], np.int32); ys = np.array([p[1] for p in pts], np.int32)
    sc = rng.integers(7, 60, n).astype(np.int32)                   # many response ties
    got = oracle.octree(xs, ys, sc, 16, 16 + W, 16, 16 + H, N).tolist()
    assert got == py_octree(xs, ys, sc, 16, 16 + W, 16, 16 + H, N)
    assert len(set(got)) == len(got) and len(got) <= max(N + 3, 16)
    if n <= N:
        assert sorted(got) == list(range(n))                       # everything survives when under budget


def test_octree_clustered(oracle):
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.integers(100, 130, 300), rng.integers(3, 1200, 50)]).astype(np.int32)
    ys = np.concatenate([rng.integers(50, 80, 300), rng.integers(3, 340, 50)]).astype(np.int32)
    uniq = sorted(set(zip(xs.tolist(), ys.tolist())))
    xs = np.array([u[0] for u in uniq], np.int32); ys = np.array([u[1] for u in uniq], np.int32)
    sc = rng.integers(7, 255, len(xs)).astype(np.int32)
    for N in (5, 40, 200):
        assert oracle.octree(xs, ys, sc, 16, 1225, 16, 360, N).tolist() == py_octree(xs, ys, sc, 16, 1225, 16, 360, N)


# ------------------------------------------------------------------------------------------- orientation / BRIEF
def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        ref = math.degrees(math.atan2(y, x)) % 360
        got = oracle.fast_atan2(float(np.float32(y)), float(np.float32(x)))
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.02


def test_ic_angle_ramps(oracle):
    yy, xx = np.mgrid[0:64, 0:64]
    hr = (xx * 3).astype(np.uint8); vr = (yy * 3).astype(np.uint8)
    assert oracle.ic_angle(hr, 32, 32) == pytest.approx(0.0, abs=1e-3)       # pure horizontal ramp
    assert oracle.ic_angle(vr, 32, 32) == pytest.approx(90.0, abs=1e-3)      # pure vertical ramp
    assert oracle.ic_angle(255 - hr, 32, 32) == pytest.approx(180.0, abs=1e-3)
    assert oracle.ic_angle(255 - vr, 32, 32) == pytest.approx(270.0, abs=1e-3)


def test_sincos_is_correctly_rounded(oracle):
    rng = np.random.default_rng(2)
    ang = (rng.uniform(0, 360, 20000).astype(np.float32) * np.float32(np.pi / 180)).astype(np.float32)
    bad = 0
    for a in ang:
        s, c = oracle.sincos(float(a))
        bad += (np.float32(s) != np.float32(np.sin(np.float64(a)))) + (np.float32(c) != np.float32(np.cos(np.float64(a))))
    assert bad == 0


def test_brief_unrotated_matches_pattern(oracle, synth):
    img = synth.random_image(6, 80, 80)
    pat = oracle.pattern().reshape(256, 4).astype(int)
    d = oracle.brief(img, 40, 40, 0.0)
    bits = np.unpackbits(d, bitorder="little")
    expect = np.array([img[40 + p[1], 40 + p[0]] < img[40 + p[3], 40 + p[2]] for p in pat], np.uint8)
    assert np.array_equal(bits, expect)
    # 90 degrees: (x, y) -> (-y, x) exactly
    d90 = oracle.brief(img, 40, 40, 90.0)
    bits90 = np.unpackbits(d90, bitorder="little")
    expect90 = np.array([img[40 + p[0], 40 - p[1]] < img[40 + p[2], 40 - p[3]] for p in pat], np.uint8)
    assert np.array_equal(bits90, expect90)


def test_detect_and_compute_structure(oracle, synth):
    img = synth.random_image(7, 240, 320)
    p = oracle.params(500)
    kps, desc = oracle.detect_and_compute(p, img)
    _, _, npl, _ = oracle.orb_tables(p)
    assert len(kps) == len(desc) > 100
    assert np.all(np.diff(kps["octave"]) >= 0)                             # level-major output
    for l in range(8):
        assert (kps["octave"] == l).sum() <= npl[l] + 3
    sc, _, _, _ = oracle.orb_tables(p)
    assert np.all(kps["size"] == np.floor(31 * sc[kps["octave"]]))
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360))
    d0 = oracle.detect(oracle.params(100), img)
    assert 0 < len(d0) <= 103 and np.all(d0["angle"] == -1) and np.all(d0["size"] == 7) and np.all(d0["octave"] == 0)


def test_screen_and_calc_descriptors_roundtrip(oracle, synth):
    img = synth.random_image(8, 240, 320)
    p = oracle.params(300)
    kps, desc = oracle.detect_and_compute(p, img)
    # every DetectAndCompute keypoint is a FAST@7 corner on its level -> Screen keeps those away from the 19-px border
    out = oracle.screen(p, img, kps)
    assert len(out) > 0.8 * len(kps)
    d2 = oracle.calc_descriptors(p, img, out)
    assert d2.shape == (len(out), 32)
    lvl0 = out["octave"] == 0                                              # level 0: no scale round trip -> same descriptor
    idx = {(k["x"], k["y"], k["octave"]): i for i, k in enumerate(kps)}
    for j in np.nonzero(lvl0)[0]:
        i = idx[(out[j]["x"], out[j]["y"], 0)]
        assert out[j]["angle"] == kps[i]["angle"] and np.array_equal(d2[j], desc[i])


# ------------------------------------------------------------------------------------------- Hamming / triangulation
def test_hamming_known_patterns(oracle):
    q = np.zeros((3, 32), np.uint8); t = np.zeros((4, 32), np.uint8)
    q[1] = 0xFF; q[2, :4] = 0x0F
    t[1] = 0xFF; t[2, 0] = 0x01; t[3] = 0xFF                               # t[3] duplicates t[1]
    idx, dist = oracle.hamming_match(q, t)
    assert idx.tolist() == [0, 1, 2] and dist.tolist() == [0, 0, 15]        # q[2]: 16 bits, t[2] shares one
    idx, dist = oracle.hamming_match(q[1:2], t[[3, 1]])                     # ties -> lowest train index
    assert idx.tolist() == [0] and dist.tolist() == [0]
    keep, mn = oracle.hamming_filter(np.array([5, 30, 31, 10]))
    assert mn == 5 and keep.tolist() == [True, True, False, True]           # max(2*5, 30) = 30
    keep, mn = oracle.hamming_filter(np.array([40, 80, 81]))
    assert keep.tolist() == [True, True, False]


def test_triangulation_known_depth(oracle, synth):
    K = synth.KITTI00
    b = K["bf"] / K["fx"]
    for Z, X, Y in ((10.0, 1.0, -0.5), (40.0, -6.0, 1.0), (5.0, 0.0, 0.0)):
        ul = K["fx"] * X / Z + K["cx"]; vl = K["fy"] * Y / Z + K["cy"]
        ur = K["fx"] * (X - b) / Z + K["cx"]
        xyz, ok = oracle.triangulate_stereo([ul], [vl], [ur], [vl], K["fx"], K["fy"], K["cx"], K["cy"], b)
        assert ok[0] and np.allclose(xyz[0], [X, Y, Z], rtol=1e-4, atol=1e-4)   # pixel coords are f32
        assert abs(Z - K["bf"] / (ul - ur)) < 1e-6 * Z                       # Z = bf / d
    xyz, ok = oracle.triangulate_stereo([700.0], [100.0], [720.0], [100.0], K["fx"], K["fy"], K["cx"], K["cy"], b)
    assert not ok[0]                                                        # negative disparity -> behind the camera
    xyz, ok = oracle.triangulate_stereo([700.0], [100.0], [690.0], [140.0], K["fx"], K["fy"], K["cx"], K["cy"], b)
    assert not ok[0]                                                        # 40 px off the epipolar line -> sigma ratio


def test_triangulation_matches_lapack(oracle):
    rng = np.random.default_rng(5)
    for _ in range(50):
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))]); P1 = np.hstack([np.eye(3), [[-0.54], [0], [0]]])
        pts = np.array([[rng.uniform(-1, 1), rng.uniform(-.5, .5), 1], [rng.uniform(-1, 1), rng.uniform(-.5, .5), 1]])
        xyz, ratio = oracle.triangulate(np.stack([P0.ravel(), P1.ravel()]), pts)
        A = np.vstack([pts[0, 0] * P0[2] - P0[0], pts[0, 1] * P0[2] - P0[1], pts[1, 0] * P1[2] - P1[0], pts[1, 1] * P1[2] - P1[1]])
        _, s, Vt = np.linalg.svd(A)
        ref = Vt[3, :3] / Vt[3, 3]
        assert np.allclose(xyz, ref, rtol=1e-8, atol=1e-9) and ratio == pytest.approx(s[3] / s[2], rel=1e-7, abs=1e-12)


# ------------------------------------------------------------------------------------------- CALC
def test_calc_shape_chain_and_torch(oracle, synth):
    import torch
    import torch.nn.functional as F
    assert oracle.calc_nweights() == 137476 == 1664 + 131200 + 4612
    w = synth.calc_weights()
    x = synth._rng(9).uniform(0, 1, (120, 160)).astype(np.float32)
    got = oracle.calc_forward(w, x)
    o = 0
    def take(shape):
        nonlocal o
        n = int(np.prod(shape)); t = torch.from_numpy(w[o:o + n].reshape(shape).copy()); o += n
        return t
    w1, b1, w2, b2, w3, b3 = take((64, 1, 5, 5)), take((64,)), take((128, 64, 4, 4)), take((128,)), take((4, 128, 3, 3)), take((4,))
    t = torch.from_numpy(x)[None, None].double()
    t = F.relu(F.conv2d(t, w1.double(), b1.double(), stride=2, padding=4)); assert t.shape[2:] == (62, 82)
    t = F.max_pool2d(t, 3, 2, ceil_mode=True); assert t.shape[2:] == (31, 41)
    t = F.local_response_norm(t, 5, alpha=1e-4, beta=0.75, k=1.0)
    t = F.relu(F.conv2d(t, w2.double(), b2.double(), stride=1, padding=2)); assert t.shape[2:] == (32, 42)
    t = F.max_pool2d(t, 3, 2, ceil_mode=True); assert t.shape[2:] == (16, 21)
    t = F.local_response_norm(t, 5, alpha=1e-4, beta=0.75, k=1.0)
    t = F.relu(F.conv2d(t, w3.double(), b3.double())); assert t.shape[1:] == (4, 14, 19)
    ref = t.flatten().numpy(); ref = ref / np.linalg.norm(ref)
    assert got.shape == (1064,) and abs(np.linalg.norm(got) - 1) < 1e-5
    assert np.abs(got - ref).max() < 2e-5


def test_calc_preproc(oracle, synth):
    L, _ = synth.stereo_pair(0, 0)
    x, after = oracle.calc_preproc(L, blur_in_place=True)
    assert x.shape == (120, 160) and 0 <= x.min() and x.max() <= 1
    assert np.array_equal(after, oracle.blur7(L, 1)) and not np.array_equal(after, L)        # in-place side effect
    x2, after2 = oracle.calc_preproc(L, blur_in_place=False)
    assert np.array_equal(x, x2) and np.array_equal(after2, L)
    assert np.array_equal((x * 255).round().astype(np.uint8), oracle.resize(oracle.blur7(L, 1), 160, 120))


def test_lcddb_scan_rules(oracle, synth):
    db = synth.lcd_database(100)
    ids = np.arange(100, dtype=np.uint64) * 2                             # ids 0,2,...,198
    q = db[37].copy()
    best, mx, cnt = oracle.lcddb_query(db, ids, q, 300)
    assert best == 74 and mx == pytest.approx(1.0, abs=1e-5) and cnt >= 1
    # cut-off: stop at the first id with cur - id < 20
    best, mx, cnt = oracle.lcddb_query(db, ids, q, 90)                    # ids <= 70 are scanned (90-72=18 <20 stops)
    assert best != 74 and mx < 0.99
    brute = db[:36] @ q
    assert best == int(ids[np.argmax(brute)]) and cnt == int((brute > 0.92).sum())
    # ties: duplicate rows -> lowest id wins (strict '>')
    db2 = db.copy(); db2[50] = db2[10]
    best, _, _ = oracle.lcddb_query(db2, ids, db2[10], 300)
    assert best == 20
    assert oracle.lcddb_query(db, ids, q, 5) == (0, 0.0, 0)               # nothing older than 20 ids


# ------------------------------------------------------------------------------------------- BA
def _project(pose, pt, K):
    from scipy.spatial.transform import Rotation as R
    pc = R.from_quat(pose[:4]).as_matrix() @ pt + pose[4:]
    return np.array([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]])


def test_ba_jacobians_vs_finite_differences(oracle, synth):
    from scipy.spatial.transform import Rotation as R
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem(n_kf=3, n_mp=6, outlier_frac=0)
    fixed[:] = 0
    # single edge problems: H = w J^T J, b = -w J^T e with w = 1 (inlier) -> recover J from b via perturbation of e
    k = 4
    ip, il = ep[k], el[k]
    Hpp, Hll, Hpl, bp, bl, chi2 = oracle.ba_build(poses, pts, [ip], [il], [obs[k]], fixed, K, delta=1e9)
    e = obs[k] - _project(poses[ip], pts[il], K)
    assert chi2[0] == pytest.approx(e @ e, rel=1e-12)
    eps = 1e-6
    Jp = np.zeros((2, 3)); Jx = np.zeros((2, 6))
    for a in range(3):
        d = np.zeros(3); d[a] = eps
        Jp[:, a] = ((obs[k] - _project(poses[ip], pts[il] + d, K)) - e) / eps
    Rm = R.from_quat(poses[ip][:4]).as_matrix()
    for a in range(6):
        xi = np.zeros(6); xi[a] = eps
        T = oracle.se3_exp(xi)
        Rd = R.from_quat(T[:4]).as_matrix()
        newp = np.concatenate([R.from_matrix(Rd @ Rm).as_quat(), Rd @ poses[ip][4:] + T[4:]])
        Jx[:, a] = ((obs[k] - _project(newp, pts[il], K)) - e) / eps
    assert np.allclose(Hpp[ip], Jx.T @ Jx, rtol=2e-4, atol=1e-3)
    assert np.allclose(Hll[il], Jp.T @ Jp, rtol=2e-4, atol=1e-3)
    assert np.allclose(Hpl[0], Jx.T @ Jp, rtol=2e-4, atol=1e-3)
    assert np.allclose(bp[ip], -Jx.T @ e, rtol=2e-4, atol=1e-3) and np.allclose(bl[il], -Jp.T @ e, rtol=2e-4, atol=1e-3)


def test_ba_huber_and_fixed(oracle, synth):
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem(n_kf=4, n_mp=30)
    H = oracle.ba_build(poses, pts, ep, el, obs, fixed, K, delta=5.991)
    Hq = oracle.ba_build(poses, pts, ep, el, obs, fixed, K, delta=1e9)
    chi2 = H[5]
    out = chi2 > 5.991 ** 2
    assert out.any() and (~out).any()
    for k in np.nonzero(out)[0][:5]:                                        # weight = delta / sqrt(e2) on outliers
        if not fixed[el[k]]:
            assert np.allclose(H[2][k], Hq[2][k] * 5.991 / math.sqrt(chi2[k]), rtol=1e-12)
    for l in np.nonzero(fixed)[0]:
        assert not H[1][l].any() and not H[4][l].any()
        assert not H[2][el == l].any()
    assert np.allclose(H[0], np.transpose(H[0], (0, 2, 1)))


def test_ba_lm_converges(oracle, synth):
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem(n_kf=5, n_mp=60, outlier_frac=0.0)
    chi0 = oracle.ba_build(poses, pts, ep, el, obs, fixed, K)[5].sum()
    p2, x2, chi, it = oracle.ba_optimize(poses, pts, ep, el, obs, fixed, K, iters=10)
    assert it >= 1 and chi < 0.5 * chi0
    assert chi / len(ep) < 2.0                                              # ~ noise level (sigma 0.5 px -> E[e2] = 0.5)


def test_pyr_down_matches_scipy(oracle, synth):
    from scipy.ndimage import correlate1d
    for (h, w) in ((37, 52), (64, 64), (5, 9)):
        img = synth.random_image(40 + h, h, w)
        k = np.array([1, 4, 6, 4, 1])
        t = correlate1d(correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")     # mirror == REFLECT_101
        ref = ((t[::2, ::2] + 128) >> 8).astype(np.uint8)
        assert np.array_equal(oracle.pyr_down(img), ref)


def test_lk_recovers_subpixel_translation(oracle, synth):
    from scipy.ndimage import gaussian_filter, shift
    rng = np.random.default_rng(3)
    base = gaussian_filter(rng.uniform(0, 255, (200, 260)), 2.0)
    base = (base - base.min()) / (base.max() - base.min()) * 255
    a = np.clip(base, 0, 255).astype(np.uint8)
    dx, dy = 3.3, -1.7
    b = np.clip(shift(base, (dy, dx), order=3, mode="nearest"), 0, 255).astype(np.uint8)
    pts = rng.uniform([40, 40], [220, 160], size=(60, 2)).astype(np.float32)
    out, st, err = oracle.lk_track(a, b, pts, pts)
    assert st.all()
    d = out - pts
    assert np.abs(d[:, 0] - dx).max() < 0.15 and np.abs(d[:, 1] - dy).max() < 0.15
    # flat image: the minimum-eigenvalue test rejects every point
    flat = np.full((100, 120), 90, np.uint8)
    _, st2, _ = oracle.lk_track(flat, flat, pts[:10] / 2, pts[:10] / 2)
    assert not st2.any()


def test_pose_only_converges_and_flags_gross_outliers(oracle, synth):
    rng = np.random.default_rng(11)
    K = synth.KITTI00; Kt = (K["fx"], K["fy"], K["cx"], K["cy"])
    n = 150
    pts = np.stack([rng.uniform(-10, 10, n), rng.uniform(-3, 3, n), rng.uniform(5, 40, n)], 1)
    T = oracle.se3_exp(np.array([0.2, 0.05, -0.3, -0.01, 0.02, 0.01]))
    x, y, z, w = T[:4]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    pc = pts @ R.T + T[4:]
    obs = np.stack([Kt[0] * pc[:, 0] / pc[:, 2] + Kt[2], Kt[1] * pc[:, 1] / pc[:, 2] + Kt[3]], 1)
    bad = np.arange(0, n, 10)
    noisy = obs.copy(); noisy[bad] += 35.0
    p, out, ni = oracle.pose_only_optimize(np.array([0, 0, 0, 1, 0, 0, 0.0]), pts, noisy, Kt)
    assert np.abs(p - T).max() < 1e-6                       # exact observations for the inliers -> exact pose
    assert set(np.where(out)[0]) == set(bad.tolist()) and ni == n - len(bad)


# ---- loop correction (src/loopclosing.cpp:537-646) ----
def _pose_apply(p7, X):
    x, y, z, w = p7[:4] / np.linalg.norm(p7[:4])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return X @ R.T + p7[4:]


def test_se3_log_exp_and_compose(oracle):
    rng = np.random.default_rng(3)
    I7 = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for _ in range(50):
        xi = np.concatenate([rng.normal(0, 2, 3), rng.normal(0, 0.8, 3)])
        T = oracle.se3_exp(xi)
        assert np.abs(oracle.se3_log(T) - xi).max() < 1e-12
        assert np.abs(oracle.se3_log(oracle.se3_compose(T, T, invert_b=True))).max() < 1e-12          # T T^-1 = I
        X = rng.normal(0, 5, (4, 3))
        U = oracle.se3_exp(np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.5, 3)]))
        assert np.abs(_pose_apply(oracle.se3_compose(T, U), X) - _pose_apply(T, _pose_apply(U, X))).max() < 1e-12
    assert np.abs(oracle.se3_log(I7)).max() == 0
    # a pure translation has log = (t, 0); a pure rotation about z by 0.3 rad has log = (0, 0, 0, 0, 0, 0.3)
    assert np.allclose(oracle.se3_log(np.array([0, 0, 0, 1, 1.5, -2, 0.25])), [1.5, -2, 0.25, 0, 0, 0], atol=1e-15)
    assert np.allclose(oracle.se3_log(np.array([0, 0, np.sin(0.15), np.cos(0.15), 0, 0, 0])), [0, 0, 0, 0, 0, 0.3], atol=1e-15)


def test_pose_graph_recovers_a_consistent_graph(oracle, synth):
    """Measurements taken from the ground truth: the optimum is the ground truth itself (chi2 -> 0) wherever the anchors are true."""
    _, fixed, e0, e1, _, gt = synth.pose_graph(60, 2, seed=11)
    meas = np.stack([oracle.se3_compose(gt[a], gt[b], invert_b=True) for a, b in zip(e0, e1)])
    rng = np.random.default_rng(0)
    start = gt.copy()
    free = ~fixed.astype(bool)
    start[free, 4:] += rng.normal(0, 0.3, (free.sum(), 3))
    start[free, :4] += rng.normal(0, 0.02, (free.sum(), 4))
    chi0 = oracle.pose_graph_optimize(start, fixed, e0, e1, meas, iters=0)[1]
    p, chi, its = oracle.pose_graph_optimize(start, fixed, e0, e1, meas)
    assert chi0 > 1.0 and chi < 1e-12 * chi0
    s = np.sign(np.sum(p[:, :4] * gt[:, :4], axis=1))[:, None]
    assert np.abs(p[:, 4:] - gt[:, 4:]).max() < 1e-6 and np.abs(p[:, :4] * s - gt[:, :4]).max() < 1e-7
    assert np.array_equal(p[~free, 4:], gt[~free, 4:])


def test_pose_graph_error_definition(oracle):
    """error = log(M^-1 T0 T1^-1) (g2o_types.h:161-167): chi2 of a single edge, read back through iters = 0."""
    rng = np.random.default_rng(5)
    T0 = oracle.se3_exp(rng.normal(0, 0.5, 6)); T1 = oracle.se3_exp(rng.normal(0, 0.5, 6)); M = oracle.se3_exp(rng.normal(0, 0.5, 6))
    e = oracle.se3_log(oracle.se3_compose(oracle.se3_compose(np.array([0, 0, 0, 1, 0, 0, 0.0]), M, invert_b=True), oracle.se3_compose(T0, T1, invert_b=True)))
    chi = oracle.pose_graph_optimize(np.stack([T0, T1]), np.array([1, 1], np.uint8), [0], [1], M[None], iters=0)[1]
    assert abs(chi - e @ e) < 1e-13
    # satisfied edge: M = T0 T1^-1
    chi = oracle.pose_graph_optimize(np.stack([T0, T1]), np.array([1, 1], np.uint8), [0], [1], oracle.se3_compose(T0, T1, invert_b=True)[None], iters=0)[1]
    assert chi < 1e-28


def test_correct_map_points_keeps_camera_frame_position(oracle, synth):
    poses, fixed, e0, e1, meas, _ = synth.pose_graph(40, 1, seed=12)
    new = oracle.pose_graph_optimize(poses, fixed, e0, e1, meas)[0]
    rng = np.random.default_rng(2)
    kf = rng.integers(-1, 40, 300).astype(np.int32); pts = rng.normal(0, 20, (300, 3))
    out = oracle.correct_map_points(poses, new, kf, pts)
    assert np.array_equal(out[kf < 0], pts[kf < 0])
    for i in np.flatnonzero(kf >= 0)[:60]:
        assert np.abs(_pose_apply(new[kf[i]], out[i][None]) - _pose_apply(poses[kf[i]], pts[i][None])).max() < 1e-10     # :630-633
    assert np.array_equal(oracle.correct_map_points(poses, poses, kf, pts)[kf < 0], pts[kf < 0])
    assert np.abs(oracle.correct_map_points(poses, poses, kf, pts) - pts).max() < 1e-12


# ---- loop verification: PnP-RANSAC (src/loopclosing.cpp:262-268) ----
def test_cv_rng_is_the_multiply_with_carry_generator(oracle):
    """cv::RNG: state <- (uint32)state * 4164903690 + (state >> 32), output = (uint32)state, uniform(a, b) = a + output % (b - a)."""
    state = 2 ** 64 - 1
    want = []
    for _ in range(20):
        state = ((state & 0xffffffff) * 4164903690 + (state >> 32)) & (2 ** 64 - 1)
        want.append(7 + (state & 0xffffffff) % (1000 - 7))
    assert oracle.cv_rng_uniform(2 ** 64 - 1, 7, 1000, 20).tolist() == want
    assert oracle.cv_rng_uniform(0, 0, 10, 5).tolist() == oracle.cv_rng_uniform(0xffffffff, 0, 10, 5).tolist()      # RNG(0) -> 0xffffffff
    assert oracle.cv_rng_uniform(1, 4, 4, 3).tolist() == [4, 4, 4]


def test_epnp_is_exact_on_exact_data(oracle, synth):
    for seed, n in ((1, 5), (2, 5), (3, 6), (4, 12), (5, 40)):
        pw, uv, K, pose, _ = synth.pnp_problem(n, 0.0, 0.0, seed=seed)
        pw = pw.astype(np.float64)
        pc = _pose_apply(pose, pw)
        uv = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)     # exact pixels of the float32 points
        rc, R, t = oracle.epnp(pw, uv, K)
        assert rc == 0
        assert np.abs(pw @ R.T + t - pc).max() < 1e-6 and abs(np.linalg.det(R) - 1) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9


def test_pnp_ransac_consensus_and_least_squares_refinement(oracle, synth):
    from scipy.optimize import least_squares
    pw, uv, K, pose, good = synth.pnp_problem(160, 0.35, 0.5, seed=21)
    rc, p, inl, ni = oracle.solve_pnp_ransac(pw, uv, K)
    assert rc == 0 and ni == inl.sum() and (inl & good).sum() >= 0.95 * good.sum() and (inl & ~good).sum() <= 3
    # every flagged match reprojects within the threshold under the RANSAC model's refinement (nearly all: the mask belongs to the
    # unrefined model), nothing unflagged is close
    pc = _pose_apply(p, pw.astype(np.float64))
    e2 = (K[0] * pc[:, 0] / pc[:, 2] + K[2] - uv[:, 0]) ** 2 + (K[1] * pc[:, 1] / pc[:, 2] + K[3] - uv[:, 1]) ** 2
    assert (e2[inl] <= 5.991 ** 2).mean() > 0.97 and (e2[~inl] > 5.991 ** 2).mean() > 0.97
    # the refined pose is the least-squares optimum over the consensus set: an independent minimiser started there does not move
    P = pw[inl].astype(np.float64); U = uv[inl].astype(np.float64)

    def res(x):
        q = p.copy(); T = oracle.se3_compose(oracle.se3_exp(x), q)
        c = _pose_apply(T, P)
        return np.concatenate([K[0] * c[:, 0] / c[:, 2] + K[2] - U[:, 0], K[1] * c[:, 1] / c[:, 2] + K[3] - U[:, 1]])
    sol = least_squares(res, np.zeros(6), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert np.abs(sol.x).max() < 1e-6
    # all-inlier data: the first successful hypothesis ends the search (RANSACUpdateNumIters -> 0) and everything is flagged
    pw, uv, K, pose, good = synth.pnp_problem(50, 0.0, 0.1, seed=22)
    rc, p, inl, ni = oracle.solve_pnp_ransac(pw, uv, K)
    assert rc == 0 and ni == 50
    assert oracle.solve_pnp_ransac(pw[:4], uv[:4], K)[0] == -2


def test_fast_score_threshold_seed_is_irrelevant(oracle, synth):
    """cv::cornerScore<16> seeds its running maxima with the threshold; the oracle's score leaves the seed out.  For every
    pixel that IS a corner at the threshold (the only pixels whose score is ever used: cv::FAST scores detections) both give the same
    value, for both thresholds of the extractor; for non-corners the seeded form returns threshold - 1 and the oracle's map stores 0."""
    rng = np.random.default_rng(11)
    for img in (synth.random_image(3, 64, 96), synth.random_image(4, 64, 96, "noise"), synth.stereo_batch(1, n_rect=150, h=64, w=96)[0, 0]):
        n_corner = 0
        for _ in range(1500):
            x = int(rng.integers(3, img.shape[1] - 3)); y = int(rng.integers(3, img.shape[0] - 3))
            s = oracle.fast_score_px(img, x, y)
            for th in (7, 20):
                seeded = oracle.fast_score_seeded(img, x, y, th)
                if oracle.is_fast_corner(img, x, y, th):
                    assert s >= th and seeded == s, (x, y, th, s, seeded)
                    n_corner += 1
                else:
                    assert s < th and seeded == th - 1, (x, y, th, s, seeded)
        assert n_corner > 0


def test_gauss_taps_option(oracle, synth):
    """The sigma = 2 taps are a run-time table on the oracle side as on the device side (myslam_orb_set_gauss_taps): default
    [18,34,49,55,49,34,18] (sum 257, OpenCV 3.4.8's per-tap rounding); e.g. an error-diffusion rounding [18,34,48,56,48,34,18] or the
    residue-on-the-centre table [18,34,49,54,49,34,18] of rounds 1-2 changes the blurred image."""
    img = synth.random_image(9, 60, 80)
    a = oracle.blur7(img, 0)
    try:
        oracle.set_gauss_taps([18, 34, 48, 56, 48, 34, 18])
        b = oracle.blur7(img, 0)
    finally:
        oracle.set_gauss_taps(None)
    assert np.array_equal(oracle.blur7(img, 0), a) and not np.array_equal(a, b) and np.abs(a.astype(int) - b.astype(int)).max() <= 3
    try:
        oracle.set_gauss_taps([18, 34, 49, 54, 49, 34, 18])
        c = oracle.blur7(img, 0)
    finally:
        oracle.set_gauss_taps(None)
    assert not np.array_equal(a, c) and np.abs(a.astype(int) - c.astype(int)).max() <= 2
    with pytest.raises(AssertionError):
        oracle.set_gauss_taps([18, 34, 49, 56, 49, 34, 18])      # sum 258: 255 * 258 overflows the Q8.8 row sum
