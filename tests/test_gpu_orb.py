"""GPU parity tests of the ORB extractor: HIP path (through the C ABI) vs the CPU oracle, bit-exact.

Every stage is integer work or float work whose integer consequences must agree (BRIEF sample coordinates,
fastAtan2 angles), so the bar is exact equality of keypoint coordinates/order, angles (f32 bits) and
descriptor bytes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(240, 320), (376, 1241), (255, 333), (480, 640)]


def _kp_equal(a, b):
    if len(a) != len(b):
        return False
    return all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def _explain(a, b):
    msg = [f"n: {len(a)} vs {len(b)}"]
    n = min(len(a), len(b))
    for f in a.dtype.names:
        bad = np.nonzero(a[f][:n] != b[f][:n])[0]
        if len(bad):
            i = bad[0]
            msg.append(f"{f}: {len(bad)} differ, first at {i}: {a[i]} vs {b[i]}")
    return "; ".join(msg)


@pytest.mark.parametrize("h,w", SIZES)
def test_pyramid_and_blur_bitexact(api, oracle, synth, h, w):
    img = synth.random_image(100 + h, h, w)
    ext = api.ORBextractor(1000)
    ref = oracle.pyramid(oracle.params(1000), img)
    for l in range(8):
        got = ext.debug_pyramid(img, l)
        assert got.shape == ref[l].shape
        assert np.array_equal(got, ref[l]), f"level {l}: {np.count_nonzero(got != ref[l])} px differ"
        gb = ext.debug_pyramid(img, l, blurred=True)
        rb = oracle.blur7(ref[l], 0)
        assert np.array_equal(gb, rb), f"blur level {l}: {np.count_nonzero(gb != rb)} px differ"


@pytest.mark.parametrize("w", [252, 255, 256, 257, 258, 259, 260, 263, 264, 509, 512, 513, 516, 519, 771])
def test_blur_strip_borders_bitexact(api, oracle, synth, w):
    """The register-strip blur exchanges neighbour dwords between lanes and mirrors the right border with byte permutes whose
    selectors depend on w mod 4 and on where the border falls inside a 256-column wave strip: sweep those cases (odd heights too)."""
    h = 131 + (w % 3)
    img = synth.random_image(7000 + w, h, w)
    ext = api.ORBextractor(500, nlevels=2)
    gb = ext.debug_pyramid(img, 0, blurred=True)
    rb = oracle.blur7(img, 0)
    assert gb.shape == rb.shape
    assert np.array_equal(gb, rb), f"{np.count_nonzero(gb != rb)} px differ, first at {np.argwhere(gb != rb)[:3].tolist()}"


@pytest.mark.parametrize("h,w", SIZES[:3])
def test_fast_candidates_equal_as_sets(api, oracle, synth, h, w):
    img = synth.random_image(200 + w, h, w)
    ext = api.ORBextractor(1000)
    pyr = oracle.pyramid(oracle.params(1000), img)
    for l in range(8):
        xs, ys, sc = ext.debug_candidates(img, l)
        rx, ry, rs = oracle.grid_fast(pyr[l])
        got = set(zip(xs.tolist(), ys.tolist(), sc.tolist())); ref = set(zip(rx.tolist(), ry.tolist(), rs.tolist()))
        assert len(xs) == len(got)
        assert got == ref, f"level {l}: {len(got - ref)} extra, {len(ref - got)} missing; e.g. {sorted(got ^ ref)[:5]}"


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("nfeat", [500, 2000])
def test_detect_and_compute_bitexact(api, oracle, synth, h, w, nfeat):
    img = synth.random_image(300 + h + nfeat, h, w)
    ext = api.ORBextractor(nfeat)
    kps, desc = ext.DetectAndCompute(img)
    rk, rd = oracle.detect_and_compute(oracle.params(nfeat), img)
    assert _kp_equal(kps, rk), _explain(kps, rk)
    assert np.array_equal(desc, rd), f"{np.count_nonzero((desc != rd).any(axis=1))} descriptors differ"
    assert len(kps) > nfeat // 3


def test_kitti_resolution_stereo_pair(api, oracle, synth):
    L, R = synth.stereo_pair(0, 3)
    ext = api.ORBextractor(2000)
    for img in (L, R):
        kps, desc = ext.DetectAndCompute(img)
        rk, rd = oracle.detect_and_compute(oracle.params(2000), img)
        assert _kp_equal(kps, rk), _explain(kps, rk)
        assert np.array_equal(desc, rd)
        assert 1990 <= len(kps) <= 2024


@pytest.mark.parametrize("nfeat", [100, 300, 2000])
def test_detect_level0_bitexact(api, oracle, synth, nfeat):
    img = synth.random_image(400 + nfeat, 376, 1241)
    ext = api.ORBextractor(nfeat)
    kps = ext.Detect(img)
    rk = oracle.detect(oracle.params(nfeat), img)
    assert _kp_equal(kps, rk), _explain(kps, rk)


def test_mask_quirk_reproduced(api, oracle, synth):
    img = synth.random_image(500, 300, 420)
    mask = np.full_like(img, 255); mask[:, :150] = 0; mask[200:, :] = 0
    ext = api.ORBextractor(800)
    kps, desc = ext.DetectAndCompute(img, mask)
    rk, rd = oracle.detect_and_compute(oracle.params(800), img, mask)
    assert _kp_equal(kps, rk), _explain(kps, rk)
    assert np.array_equal(desc, rd)
    k0 = ext.Detect(img, mask); r0 = oracle.detect(oracle.params(800), img, mask)
    assert _kp_equal(k0, r0), _explain(k0, r0)
    assert len(kps) < len(oracle.detect_and_compute(oracle.params(800), img)[0])


def test_other_extractor_configs(api, oracle, synth):
    img = synth.random_image(600, 200, 260)
    for nfeat, sf, nl, ini, mn in [(300, 1.2, 3, 20, 7), (150, 1.5, 2, 30, 10), (64, 1.1, 1, 12, 5)]:
        ext = api.ORBextractor(nfeat, sf, nl, ini, mn)
        kps, desc = ext.DetectAndCompute(img)
        rk, rd = oracle.detect_and_compute(oracle.params(nfeat, sf, nl, ini, mn), img)
        assert _kp_equal(kps, rk), (nfeat, sf, nl, _explain(kps, rk))
        assert np.array_equal(desc, rd)


def test_flat_and_sparse_images(api, oracle):
    flat = np.full((240, 320), 90, np.uint8)
    ext = api.ORBextractor(500)
    kps, desc = ext.DetectAndCompute(flat)
    assert len(kps) == 0 and desc.shape == (0, 32)
    one = flat.copy(); one[100:140, 150:200] = 200                        # a single bright rectangle: 4 corners, th-7 fallback cells
    kps, desc = ext.DetectAndCompute(one)
    rk, rd = oracle.detect_and_compute(oracle.params(500), one)
    assert _kp_equal(kps, rk), _explain(kps, rk)
    assert np.array_equal(desc, rd) and len(kps) >= 4
    assert len(ext.Detect(flat)) == 0


def test_unsupported_and_empty_inputs(api, pkg):
    ext = api.ORBextractor(500)
    tiny = np.zeros((100, 120), np.uint8)                                 # 8 levels: level 7 is 28x33 -> no 30-px cell fits
    with pytest.raises(api.MyslamError) as e:
        ext.DetectAndCompute(tiny)
    assert e.value.code == api.ERR_UNSUPPORTED
    n = C.c_int(123)
    kp = np.zeros(8, api.KP_DTYPE); d = np.zeros((8, 32), np.uint8)
    rc = api.lib().myslam_orb_detect_and_compute(ext._h, None, 0, 0, 0, None, 0, kp.ctypes.data_as(C.c_void_p),
                                                 d.ctypes.data_as(C.c_void_p), 8, C.byref(n))
    assert rc == 0 and n.value == 0                                       # reference: silent return on empty image (:924)
    img = pkg.synth.random_image(1, 240, 320)
    with pytest.raises(api.MyslamError) as e:                            # caller's buffer too small -> CAPACITY, never truncation
        ext.DetectAndCompute(img, cap=10)
    assert e.value.code == api.ERR_CAPACITY


def test_row_pitch_is_honoured(api, oracle, synth):
    base = synth.random_image(700, 260, 400)
    view = base[:, 7:7 + 333]                                             # non-contiguous view -> api copies; build an explicit padded buffer too
    ext = api.ORBextractor(400)
    kps, desc = ext.DetectAndCompute(view)
    rk, rd = oracle.detect_and_compute(oracle.params(400), np.ascontiguousarray(view))
    assert _kp_equal(kps, rk) and np.array_equal(desc, rd)
    padded = np.zeros((260, 352), np.uint8); padded[:, :333] = view      # step 352 != cols 333
    n = C.c_int()
    cap = ext.max_keypoints()
    kp = np.zeros(cap, api.KP_DTYPE); d = np.zeros((cap, 32), np.uint8)
    rc = api.lib().myslam_orb_detect_and_compute(ext._h, padded.ctypes.data_as(C.c_void_p), 260, 333, 352, None, 0,
                                                 kp.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    assert rc == 0 and _kp_equal(kp[:n.value], rk) and np.array_equal(d[:n.value], rd)


def test_screen_and_calc_descriptors_bitexact(api, oracle, synth):
    """The loop-closing path: LoopClosing::ProcessNewKF (loopclosing.cpp:94-112)."""
    img = synth.random_image(800, 376, 1241)
    p = oracle.params(300)
    feats = oracle.detect(p, img)                                         # frontend features (level-0 FAST)
    pyr_kps = np.repeat(feats, 8)                                         # expand to 8 pyramid keypoints each (:94-105)
    pyr_kps["octave"] = np.tile(np.arange(8), len(feats)); pyr_kps["response"] = -1
    pyr_kps["class_id"] = np.repeat(np.arange(len(feats)), 8)
    ext = api.ORBextractor(300)
    out, kin_after = ext.ScreenAndComputeKPsParams(img, pyr_kps)
    rout = oracle.screen(p, img, pyr_kps)
    assert _kp_equal(out, rout), _explain(out, rout)
    assert len(out) >= len(feats)                                         # level 0 always survives away from the border
    d = ext.CalcDescriptors(img, out)
    rd = oracle.calc_descriptors(p, img, rout)
    assert np.array_equal(d, rd), f"{np.count_nonzero((d != rd).any(axis=1))} of {len(d)} descriptors differ"


def test_batch_equals_single_and_oracle(api, oracle, synth):
    import torch
    B = 6
    imgs = np.stack([synth.random_image(900 + i, 300, 420) for i in range(B)])
    ext = api.ORBextractor(700)
    cap = ext.max_keypoints()
    d_imgs = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
    assert d_imgs.is_contiguous()
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); d_st = torch.ones(B, dtype=torch.int32, device="cuda")
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):                                                  # twice: buffers are reused, counters must be reset
        ext.detect_and_compute_batch(d_imgs.data_ptr(), B, 300, 420, 420, 300 * 420, d_kps.data_ptr(), d_desc.data_ptr(),
                                     d_cnt.data_ptr(), d_st.data_ptr(), cap)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy(); assert (d_st.cpu().numpy() == 0).all()
        kps = d_kps.cpu().numpy().view(api.KP_DTYPE).reshape(B, cap)
        desc = d_desc.cpu().numpy().reshape(B, cap, 32)
        for b in range(B):
            rk, rd = oracle.detect_and_compute(oracle.params(700), imgs[b])
            assert _kp_equal(kps[b, :cnt[b]], rk), (rep, b, _explain(kps[b, :cnt[b]], rk))
            assert np.array_equal(desc[b, :cnt[b]], rd)


@pytest.mark.parametrize("fast_mode", [-1, 1, 0])
def test_large_batch_takes_the_batch_kernels(api, oracle, synth, fast_mode):
    """72 different images in one call: batches of 64 and more run the 256-thread oct-tree blocks (`k_octree<256>`; smaller ones the 512-thread
    form) and the tile-ordered descriptor pass — every image must still equal the oracle's single-image result.
    fast_mode: the dense path, the two-phase path or whichever the statistics choose — same candidates, same key-points."""
    import torch
    B, h, w, nf = 72, 240, 328, 300
    kinds = ("texture", "noise", "texture")
    imgs = np.stack([synth.random_image(7100 + i, h, w, kinds[i % 3]) for i in range(B)])
    imgs[5] = synth.stereo_batch(1, stream_id=3, n_rect=40, h=h, w=w)[0, 0]          # a sparse scene and a flat image among them
    imgs[9] = 128
    ext = api.ORBextractor(nf)
    ext.set_option(ext.OPT_FAST_MODE, fast_mode)
    cap = ext.max_keypoints(h, w)
    d_imgs = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); d_st = torch.ones(B, dtype=torch.int32, device="cuda")
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):
        ext.detect_and_compute_batch(d_imgs.data_ptr(), B, h, w, w, h * w, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), cap)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy(); assert (d_st.cpu().numpy() == 0).all()
        kps = d_kps.cpu().numpy().view(api.KP_DTYPE).reshape(B, cap); desc = d_desc.cpu().numpy().reshape(B, cap, 32)
        for b in range(B):
            rk, rd = oracle.detect_and_compute(oracle.params(nf), imgs[b])
            assert _kp_equal(kps[b, :cnt[b]], rk), (rep, b, _explain(kps[b, :cnt[b]], rk))
            assert np.array_equal(desc[b, :cnt[b]], rd), (rep, b)


def test_golden_fixture(api):
    """Committed oracle outputs (tests/golden/orb_small.npz): the HIP path must reproduce them byte for byte."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "orb_small.npz"))
    ext = api.ORBextractor(int(g["nfeatures"]))
    kps, desc = ext.DetectAndCompute(g["image"])
    assert kps.tobytes() == g["kps"].tobytes() and np.array_equal(desc, g["desc"])


def _clustered_image(seed, h=240, w=320):
    """Flat background + a few tiny, extremely corner-dense patches: forces the oct-tree far below the depth the
    LDS counting sort covers (the on-demand in-place partition path) and cells that need the th=7 fallback."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 100, np.uint8)
    for _ in range(3):
        y0 = int(rng.integers(30, h - 70)); x0 = int(rng.integers(30, w - 70))
        patch = rng.integers(0, 2, (20, 20)).astype(np.uint8) * int(rng.integers(60, 150))
        img[y0:y0 + 40, x0:x0 + 40] = 100 + np.kron(patch, np.ones((2, 2), np.uint8))
    ys = rng.integers(25, h - 25, 25); xs = rng.integers(25, w - 25, 25)       # a few isolated weak corners
    for y, x in zip(ys, xs):
        img[y:y + 3, x:x + 3] = 100 + int(rng.integers(9, 30))
    return img


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("nfeat", [60, 400, 1500])
def test_clustered_corners_deep_octree(api, oracle, seed, nfeat):
    img = _clustered_image(seed)
    ext = api.ORBextractor(nfeat)
    kps, desc = ext.DetectAndCompute(img)
    rk, rd = oracle.detect_and_compute(oracle.params(nfeat), img)
    assert _kp_equal(kps, rk), _explain(kps, rk)
    assert np.array_equal(desc, rd)
    k0 = ext.Detect(img); r0 = oracle.detect(oracle.params(nfeat), img)
    assert _kp_equal(k0, r0), _explain(k0, r0)


def test_many_random_images_small(api, oracle, synth):
    """Breadth: 40 seeded images of assorted sizes / budgets, all bit-exact."""
    rng = np.random.default_rng(123)
    for i in range(40):
        h = int(rng.integers(230, 420)); w = int(rng.integers(230, 700)); nfeat = int(rng.choice([50, 200, 777, 2500]))
        img = synth.random_image(5000 + i, h, w, kind="noise" if i % 7 == 0 else "texture")
        ext = api.ORBextractor(nfeat)
        kps, desc = ext.DetectAndCompute(img)
        rk, rd = oracle.detect_and_compute(oracle.params(nfeat), img)
        assert _kp_equal(kps, rk), (i, h, w, nfeat, _explain(kps, rk))
        assert np.array_equal(desc, rd), (i, h, w, nfeat)


def test_wide_image_small_budget_capacity(api, oracle, synth):
    """A very wide image has many root nodes (nIni = round(w/h)); the first unconditional split returns up to 4*nIni key-points per
    level even when the level's budget is smaller (ORBextractor.cpp:645-716).  The size-aware capacity query must cover it."""
    img = synth.random_image(31337, 118, 1010)
    ext = api.ORBextractor(60, 1.2, 3)
    p = oracle.params(60); p.nlevels = 3
    rk, rd = oracle.detect_and_compute(p, img)
    gk, gd = ext.DetectAndCompute(img)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    assert len(rk) > 60 + 3 * 3                               # more key-points than nfeatures: the reference behaves this way
    assert ext.max_keypoints(118, 1010) >= len(rk)
    with pytest.raises(api.MyslamError) as e:                 # an undersized caller buffer is refused, never truncated
        ext.DetectAndCompute(img, cap=len(rk) - 1)
    assert e.value.code == api.ERR_CAPACITY


def test_large_image_many_cells(api, oracle, synth):
    """1500 x 1200: level 0 has more than 1024 FAST grid cells, so the oct-tree's best-key phase cannot pack the candidate order
    into the sort entry and takes its two-pass form (k_octree, phase D)."""
    img = synth.random_image(77, 1200, 1500)
    gk, gd = api.ORBextractor(3000).DetectAndCompute(img)
    rk, rd = oracle.detect_and_compute(oracle.params(3000), img)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)


def test_two_handles_taking_turns_on_fast(api, oracle, synth):
    """myslam_orb_set_fast_event / _set_fast_gate: two handles on two streams that gate each other's FAST stage (bench.py's
    left / right pipeline) return exactly what a plain call returns, over several rounds with buffer reuse."""
    import torch
    B, H, W = 4, 260, 400
    imgs = [np.stack([synth.random_image(7000 + 10 * s + i, H, W) for i in range(B)]) for s in range(2)]
    ref = [[oracle.detect_and_compute(oracle.params(600), im) for im in side] for side in imgs]
    exts = [api.ORBextractor(600), api.ORBextractor(600)]
    cap = exts[0].max_keypoints()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    for e in evs:
        e.record(torch.cuda.current_stream())
    d_imgs = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in imgs]
    d_kps = [torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_desc = [torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_cnt = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    d_st = [torch.ones(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for s in range(2):
        exts[s].set_stream(streams[s].cuda_stream)
        exts[s].set_fast_event(evs[s].cuda_event)
        exts[s].set_fast_gate(evs[1 - s].cuda_event)
    for rep in range(3):
        for s in range(2):
            exts[s].detect_and_compute_batch(d_imgs[s].data_ptr(), B, H, W, W, H * W, d_kps[s].data_ptr(), d_desc[s].data_ptr(),
                                             d_cnt[s].data_ptr(), d_st[s].data_ptr(), cap)
        torch.cuda.synchronize()
        for s in range(2):
            assert evs[s].query()
            cnt = d_cnt[s].cpu().numpy(); assert (d_st[s].cpu().numpy() == 0).all()
            kps = d_kps[s].cpu().numpy().view(api.KP_DTYPE).reshape(B, cap)
            desc = d_desc[s].cpu().numpy().reshape(B, cap, 32)
            for b in range(B):
                rk, rd = ref[s][b]
                assert _kp_equal(kps[b, :cnt[b]], rk), (rep, s, b, _explain(kps[b, :cnt[b]], rk))
                assert np.array_equal(desc[b, :cnt[b]], rd)
            d_kps[s].zero_(); d_desc[s].zero_()
        torch.cuda.synchronize()
    for s in range(2):                     # gates off again: plain calls
        exts[s].set_fast_event(0); exts[s].set_fast_gate(0)
    exts[0].detect_and_compute_batch(d_imgs[0].data_ptr(), B, H, W, W, H * W, d_kps[0].data_ptr(), d_desc[0].data_ptr(),
                                     d_cnt[0].data_ptr(), d_st[0].data_ptr(), cap)
    torch.cuda.synchronize()
    assert (d_st[0].cpu().numpy() == 0).all() and d_cnt[0].cpu().numpy().tolist() == [len(r[0]) for r in ref[0]]


def test_gauss_taps_option_matches_oracle(api, oracle, synth):
    """myslam_orb_set_gauss_taps / orc_set_gauss_taps: the one-table change a maintainer makes when OpenCV's fixed-point Gaussian turns
    out to round differently (tools/dump_opencv_goldens.py): blurred levels and descriptors follow on both sides, key-points do not move."""
    img = synth.random_image(321, 240, 320)
    taps = [18, 34, 48, 56, 48, 34, 18]
    ext = api.ORBextractor(500)
    k0, d0 = ext.DetectAndCompute(img)
    ext.set_gauss_taps(taps)
    k1, d1 = ext.DetectAndCompute(img)
    try:
        oracle.set_gauss_taps(taps)
        rk, rd = oracle.detect_and_compute(oracle.params(500), img)
        for l in (0, 5):
            assert np.array_equal(ext.debug_pyramid(img, l, blurred=True), oracle.blur7(oracle.pyramid(oracle.params(500), img)[l], 0))
    finally:
        oracle.set_gauss_taps(None)
    assert k1.tobytes() == rk.tobytes() and np.array_equal(d1, rd)
    assert k1.tobytes() == k0.tobytes() and not np.array_equal(d1, d0)
    ext.set_gauss_taps(None)
    k2, d2 = ext.DetectAndCompute(img)
    assert np.array_equal(d2, d0)
    with pytest.raises(api.MyslamError):
        ext.set_gauss_taps([18, 34, 49, 56, 49, 34, 18])             # sum 258: the Q8.8 row sums would not fit 16 bits
    ext.set_gauss_taps([18, 34, 49, 54, 49, 34, 18])                 # the sum-256 table of rounds 1-2 stays selectable
    ext.set_gauss_taps(None)


def test_blur_saturates_like_ufixedpoint(api, oracle, synth):
    """The default sigma = 2 taps sum to 257 (OpenCV 3.4.8 rounds every tap on its own), so saturated image regions reach 257 before the
    u8 conversion: strip kernel (wide, aligned levels) and LDS kernel (narrow levels) both clamp like ufixedpoint32 -> uint8_t."""
    for h, w in ((376, 1241), (240, 320)):
        img = synth.random_image(5, h, w)
        img[20:90, 30:170] = 255; img[5:40, w - 60:] = 254; img[h - 30:, :80] = 255
        ext = api.ORBextractor(500)
        P = oracle.pyramid(oracle.params(500), img)
        for l in (0, 1, 4, 7):
            got = ext.debug_pyramid(img, l, blurred=True); ref = oracle.blur7(P[l], 0)
            assert np.array_equal(got, ref), (h, w, l)
            if l == 0:
                assert ref.max() == 255 and (ref[30:80, 40:160] == 255).all()
        k, d = ext.DetectAndCompute(img); rk, rd = oracle.detect_and_compute(oracle.params(500), img)
        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)


def test_get_tables(api, oracle):
    """a1: the constructor tables (ORBextractor.cpp:384-445) through myslam_orb_get_tables"""
    for nf, sf, nl in ((2000, 1.2, 8), (300, 1.2, 8), (100, 1.2, 8), (1000, 1.5, 4)):
        sc, isc, npl, um = api.ORBextractor(nf, sf, nl).tables()
        r = oracle.orb_tables(oracle.params(nf, scale=sf, nlevels=nl))
        assert np.array_equal(sc.view(np.uint32), np.asarray(r[0], np.float32).view(np.uint32)) and np.array_equal(isc.view(np.uint32), np.asarray(r[1], np.float32).view(np.uint32))
        assert np.array_equal(npl, r[2]) and np.array_equal(um, r[3])
    assert api.ORBextractor(2000).tables()[2].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert api.ORBextractor(2000).tables()[3].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


@pytest.mark.parametrize("copy_input", [0, 1])
@pytest.mark.parametrize("cols,step,gap", [(423, 423, 0), (423, 431, 17), (640, 640, 0), (257, 300, 5)])
def test_level0_read_in_place(api, oracle, synth, copy_input, cols, step, gap):
    """A *_batch call reads the full-resolution level of every image but the last in place (MYSLAM_ORB_OPT_COPY_INPUT 0, the default):
    odd widths, row pitches that are not multiples of 4 and gaps between the images must give the bytes the copying path gives — the
    oracle's — for every image of the batch, the in-place ones and the copied last one, with and without a mask, and for Detect."""
    import torch
    B, rows = 5, 240
    imgs = [synth.random_image(4200 + i, rows, cols, "texture" if i % 2 else "noise") for i in range(B)]
    stride = rows * step + gap
    buf = np.full(B * stride + 64, 0xA5, np.uint8)                      # pitch padding and gaps hold garbage, not zeros
    for i, im in enumerate(imgs):
        v = buf[i * stride:i * stride + rows * step].reshape(rows, step)
        v[:, :cols] = im
    d = torch.from_numpy(buf).cuda()
    mask = np.full((rows, cols), 255, np.uint8); mask[60:140, 100:220] = 0
    mbuf = np.zeros(B * stride + 64, np.uint8)
    for i in range(B):
        mbuf[i * stride:i * stride + rows * step].reshape(rows, step)[:, :cols] = mask
    dm = torch.from_numpy(mbuf).cuda()
    ext = api.ORBextractor(600)
    ext.set_option(ext.OPT_COPY_INPUT, copy_input)
    cap = ext.max_keypoints(rows, cols)
    kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    for use_mask in (False, True):
        for _ in range(2):                                               # second call: the FAST path chosen from statistics
            ext.detect_and_compute_batch(d.data_ptr(), B, rows, cols, step, stride, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), cap,
                                         d_masks=dm.data_ptr() if use_mask else 0)
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        k = kps.cpu().numpy().view(api.KP_DTYPE).reshape(B, cap); dd = desc.cpu().numpy().reshape(B, cap, 32)
        for i in range(B):
            rk, rd = oracle.detect_and_compute(oracle.params(600), imgs[i], mask if use_mask else None)
            n = int(cnt[i])
            assert n == len(rk) and k[i, :n].tobytes() == rk.tobytes() and np.array_equal(dd[i, :n], rd), (copy_input, use_mask, i)
    # single-image entry points on the handle that has just read a batch in place: they stage and copy their own image
    other = synth.random_image(977, rows, cols, "texture")
    pyr = oracle.pyramid(oracle.params(600), other)
    for lvl in (0, 1, 3):
        assert np.array_equal(ext.debug_pyramid(other, lvl), pyr[lvl]), (copy_input, "debug_pyramid", lvl)
    xs, ys, sc = ext.debug_candidates(other, 0)
    rx, ry, rs = oracle.grid_fast(pyr[0])
    assert sorted(zip(ys.tolist(), xs.tolist(), sc.tolist())) == sorted(zip(ry.tolist(), rx.tolist(), rs.tolist()))
    gk, gd = ext.DetectAndCompute(other)
    rk, rd = oracle.detect_and_compute(oracle.params(600), other)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    det = api.ORBextractor(300)
    det.set_option(det.OPT_COPY_INPUT, copy_input)
    dcap = det.max_keypoints(rows, cols)
    dk = torch.zeros(B * dcap * 28, dtype=torch.uint8, device="cuda")
    det.detect_batch(d.data_ptr(), B, rows, cols, step, stride, dk.data_ptr(), cnt.data_ptr(), st.data_ptr(), dcap)
    torch.cuda.synchronize()
    k = dk.cpu().numpy().view(api.KP_DTYPE).reshape(B, dcap)
    for i in range(B):
        rk = oracle.detect(oracle.params(300), imgs[i])
        n = int(cnt[i])
        assert n == len(rk) and k[i, :n].tobytes() == rk.tobytes(), (copy_input, "detect", i)


def test_batch_of_overlapping_images_is_copied(api, oracle, synth):
    """img_stride < rows * step (here 0: the SAME image five times, and rows * step / 2: every image shares half its rows with the next
    one): the in-place path would rely on over-reads landing in the next image — such layouts take the copying path and give the
    oracle's bytes."""
    import torch
    rows, cols, step = 240, 333, 336
    img = synth.random_image(4300, rows + rows // 2 * 4, cols)
    buf = np.full(img.shape[0] * step + 64, 0xA5, np.uint8)
    buf[:img.shape[0] * step].reshape(-1, step)[:, :cols] = img
    d = torch.from_numpy(buf).cuda()
    ext = api.ORBextractor(400)
    cap = ext.max_keypoints(rows, cols)
    for B, stride in ((5, 0), (5, rows // 2 * step)):
        kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
        ext.detect_and_compute_batch(d.data_ptr(), B, rows, cols, step, stride, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), cap)
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        k = kps.cpu().numpy().view(api.KP_DTYPE).reshape(B, cap); dd = desc.cpu().numpy().reshape(B, cap, 32)
        for i in range(B):
            r0 = i * stride // step
            rk, rd = oracle.detect_and_compute(oracle.params(400), img[r0:r0 + rows])
            n = int(cnt[i])
            assert n == len(rk) and k[i, :n].tobytes() == rk.tobytes() and np.array_equal(dd[i, :n], rd), (stride, i)
