"""Paths a run does not necessarily take must stay bit-compatible with the oracle:
  * the two paths of the grid-FAST kernel (compass pre-test + compaction / dense scoring) and its own choice between them,
  * the extractor with and without its internal Gaussian-pyramid stream,
  * the capability fallbacks (generic resize for scale factors above 1.25, the LDS-tiled Gaussian for unaligned / tiny images).
(The generic CALC layer kernels against the fused ones: tests/test_gpu_lcd.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1, -1])
@pytest.mark.parametrize("kind,nrect", [("texture", 0), ("noise", 0), ("scene", 300), ("scene", 6000)])
def test_fast_paths_bitexact(api, oracle, synth, mode, kind, nrect):
    """FAST_MODE 0 (two-phase), 1 (dense) and -1 (chosen from the previous launch's statistics; called three times so that the choice
    is exercised) on corner-dense texture, pure noise, a sparse scene (a few % of corners) and the BASELINE scene."""
    if kind == "scene":
        img = synth.stereo_batch(1, stream_id=3, n_rect=nrect, h=240, w=420)[0, 0]
    else:
        img = synth.random_image(77, 240, 420, kind)
    rk, rd = oracle.detect_and_compute(oracle.params(800), img)
    ext = api.ORBextractor(800)
    ext.set_option(ext.OPT_FAST_MODE, mode)
    for rep in range(3 if mode < 0 else 1):
        gk, gd = ext.DetectAndCompute(img)
        assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd), (mode, kind, rep)
    for lvl in (0, 3, 7):                                           # candidate sets before the oct-tree
        xs, ys, sc = ext.debug_candidates(img, lvl)
        rx, ry, rs = oracle.grid_fast(oracle.pyramid(oracle.params(800), img)[lvl])
        assert sorted(zip(ys.tolist(), xs.tolist(), sc.tolist())) == sorted(zip(ry.tolist(), rx.tolist(), rs.tolist())), (mode, kind, lvl)


def test_fast_statistics_readback(api, synth):
    """myslam_orb_debug_readback(what = 6): the numbers the path choice is made from — forced modes report their own path, the
    surviving-pair fraction of noise is far above that of a sparse scene, and the automatic mode follows them"""
    noise = synth.random_image(11, 240, 420, "noise")
    sparse = synth.stereo_batch(1, stream_id=4, n_rect=150, h=240, w=420)[0, 0]
    frac = {}
    for name, img in (("noise", noise), ("sparse", sparse)):
        ext = api.ORBextractor(500)
        ext.set_option(ext.OPT_FAST_MODE, 0)
        ext.DetectAndCompute(img)
        stat, pairs, path = ext.fast_statistics(0)
        assert path == 0 and 0 < pairs and stat <= pairs
        frac[name] = stat / pairs
        ext.set_option(ext.OPT_FAST_MODE, 1)
        ext.DetectAndCompute(img)
        stat1, pairs1, path1 = ext.fast_statistics(0)
        assert path1 == 1 and pairs1 == pairs and stat1 <= pairs1
    assert frac["noise"] > 0.6 and frac["sparse"] < 0.3
    ext = api.ORBextractor(500)                               # automatic: first launch two-phase (no history), then what the history says
    ext.DetectAndCompute(noise)
    assert ext.fast_statistics(0)[2] == 0
    ext.DetectAndCompute(noise)
    assert ext.fast_statistics(0)[2] == 1
    ext.DetectAndCompute(sparse)                              # still decided by the noise image's statistics
    ext.DetectAndCompute(sparse)
    assert ext.fast_statistics(0)[2] == 0


def test_fast_mode_switches_between_batches(api, oracle, synth):
    """One handle, alternating noise and sparse images: whatever path the statistics select, every result equals the oracle's."""
    ext = api.ORBextractor(500)
    imgs = [synth.random_image(5, 240, 400, "noise"), synth.stereo_batch(1, stream_id=2, n_rect=200, h=240, w=400)[0, 0]]
    refs = [oracle.detect_and_compute(oracle.params(500), im) for im in imgs]
    for i in (0, 0, 1, 1, 1, 0, 1, 0, 0):
        gk, gd = ext.DetectAndCompute(imgs[i])
        assert gk.tobytes() == refs[i][0].tobytes() and np.array_equal(gd, refs[i][1])


@pytest.mark.parametrize("aux", [0, 1, 2])
def test_internal_stream_modes(api, oracle, synth, aux):
    import torch
    imgs = np.stack([synth.random_image(900 + i, 240, 320) for i in range(4)])
    ext = api.ORBextractor(600)
    ext.set_option(ext.OPT_INTERNAL_STREAM, aux)
    cap = ext.max_keypoints(240, 320)
    d = torch.from_numpy(imgs).cuda()
    kps = torch.zeros(4 * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(4 * cap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda"); st = torch.zeros(4, dtype=torch.int32, device="cuda")
    for _ in range(2):
        ext.detect_and_compute_batch(d.data_ptr(), 4, 240, 320, 320, 240 * 320, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), cap)
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    k = kps.cpu().numpy().view(api.KP_DTYPE).reshape(4, cap); dd = desc.cpu().numpy().reshape(4, cap, 32)
    for i in range(4):
        rk, rd = oracle.detect_and_compute(oracle.params(600), imgs[i])
        n = int(cnt[i])
        assert k[i, :n].tobytes() == rk.tobytes() and np.array_equal(dd[i, :n], rd)


def test_capability_fallbacks(api, oracle, synth):
    """scale factor 1.5 -> the generic resize kernel (the register strips cover factors up to 1.25); a 7-pixel-high LCD input ->
    the LDS-tiled Gaussian."""
    img = synth.random_image(31, 300, 500)
    p = oracle.params(700, scale=1.5, nlevels=4)
    gk, gd = api.ORBextractor(700, scaleFactor=1.5, nlevels=4).DetectAndCompute(img)
    rk, rd = oracle.detect_and_compute(p, img)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    small = synth.random_image(32, 7, 301)
    lcd = api.DeepLCD(synth.calc_weights())
    _, blurred = lcd.calcDescrOriginalImg(small, blur_in_place=True)
    assert np.array_equal(blurred, oracle.calc_preproc(small, blur_in_place=True)[1])


@pytest.mark.parametrize("shape", [(376, 1241), (240, 420), (97, 333), (96, 96), (100, 90)])
def test_matrix_core_gaussian_bitexact(api, oracle, synth, shape):
    """MYSLAM_ORB_OPT_BLUR_MFMA: the Gaussian pyramid as banded int8 matrix products (k_blur7_mfma) — every blurred level and the whole
    extraction must equal the oracle bit for bit: saturated flat-white areas (the default tap table sums to 257), borders, partial tiles,
    levels too narrow for the matrix-core form (they keep the strip kernel inside the same call), single frames and a batch whose level 0
    is read in place; another tap table through the same path."""
    import torch
    h, w = shape
    img = synth.random_image(41 + h, h, w, "texture")
    img[: h // 3, : w // 2] = 255; img[h // 2:, w // 3:] = 0                      # saturation: 255 * 257 * 257 needs the clamp
    nlev = 8 if min(h, w) >= 200 else 3 if w == 90 else 2      # (100, 90): level 2 is 62 columns wide — below the matrix-core form's 64: strip kernel
    par = oracle.params(600, nlevels=nlev)
    ext = api.ORBextractor(600, nlevels=nlev); ext.set_option(ext.OPT_BLUR_MFMA, 1)
    rk, rd = oracle.detect_and_compute(par, img)
    for rep in range(2):                                                            # second call = the replayed graph of the host-pointer path
        gk, gd = ext.DetectAndCompute(img)
        assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd), (shape, rep)
    pyr = oracle.pyramid(par, img)
    for lvl in range(nlev):
        assert np.array_equal(ext.debug_pyramid(img, lvl, blurred=True), oracle.blur7(pyr[lvl], 0)), (shape, lvl)
    # batch: level 0 of every image but the last is read where the caller put it (pitch = cols: rows of any alignment)
    B = 5
    imgs = np.stack([np.roll(img, 7 * i, axis=1) for i in range(B)])
    d = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
    cap = ext.max_keypoints(h, w)
    kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    ext.detect_and_compute_batch(d.data_ptr(), B, h, w, w, h * w, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), cap)
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    k = kps.cpu().numpy().view(api.KP_DTYPE).reshape(B, cap); dd = desc.cpu().numpy().reshape(B, cap, 32)
    for i in range(B):
        rki, rdi = oracle.detect_and_compute(par, imgs[i])
        n = int(cnt[i])
        assert k[i, :n].tobytes() == rki.tobytes() and np.array_equal(dd[i, :n], rdi), (shape, i)
    # another tap table (the sum-256 one of rounds 1-2) through the same kernels
    q = [18, 34, 49, 54, 49, 34, 18]
    ext.set_gauss_taps(q); oracle.set_gauss_taps(q)
    try:
        rk2, rd2 = oracle.detect_and_compute(par, img)
        gk2, gd2 = ext.DetectAndCompute(img)
        assert gk2.tobytes() == rk2.tobytes() and np.array_equal(gd2, rd2)
    finally:
        oracle.set_gauss_taps(None)


@pytest.mark.gpu
def test_option_change_between_host_calls_takes_effect(api, oracle, synth):
    """An option that changes the launched kernels, set between the first (eager) and the second (capturing) host-pointer call: the handle's
    generation moves, the cached graph of the old setting is dropped, the matrix-core Gaussian's tables are built before the capture begins
    (advisor, round 4) — every call equals the oracle and later calls still take the graph path (same results, no error)."""
    img = synth.random_image(77, 240, 420, "texture")
    par = oracle.params(500)
    rk, rd = oracle.detect_and_compute(par, img)
    ext = api.ORBextractor(500)
    gk, gd = ext.DetectAndCompute(img)                                   # eager call of this key
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    ext.set_option(ext.OPT_BLUR_MFMA, 1)                                 # now the second call captures: tables must exist before it does
    for rep in range(3):
        gk, gd = ext.DetectAndCompute(img)
        assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd), rep
    ext.set_option(ext.OPT_BLUR_MFMA, 0)
    for rep in range(3):
        gk, gd = ext.DetectAndCompute(img)
        assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd), rep
