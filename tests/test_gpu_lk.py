"""Pyramidal LK tracker on the GPU vs the oracle restatement (oracle/lk_oracle.cpp): bit-exact positions, status and error.
Reference call sites: Frontend::TrackLastFrame / FindFeaturesInRight, src/frontend.cpp:150-153, 358-361."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _points(oracle, img, n=300):
    k = oracle.detect(oracle.params(n), img)
    return np.stack([k["x"], k["y"]], 1).astype(np.float32)


@pytest.mark.parametrize("kind", ["temporal", "stereo"])
def test_lk_matches_oracle_bitexact(api, oracle, synth, kind):
    L, R = synth.stereo_pair(0, 3)
    L2, _ = synth.stereo_pair(0, 4)
    nxt = L2 if kind == "temporal" else R
    pts = _points(oracle, L)
    init = pts.copy()
    if kind == "stereo":
        init[:, 0] -= 12.0                                    # a projected-landmark style initial guess
    lk = api.LKTracker()
    g_pts, g_st, g_err = lk.track(L, nxt, pts, init)
    r_pts, r_st, r_err = oracle.lk_track(L, nxt, pts, init)
    assert np.array_equal(g_st, r_st) and r_st.mean() > 0.9
    assert np.array_equal(g_pts.view(np.uint32), r_pts.view(np.uint32)), np.abs(g_pts - r_pts).max()
    assert np.array_equal(g_err.view(np.uint32), r_err.view(np.uint32))
    d = r_pts[r_st] - pts[r_st]
    if kind == "temporal":
        assert abs(np.median(d[:, 0]) + 2.0) < 0.05 and abs(np.median(d[:, 1])) < 0.05          # the synthetic scene shifts by 2 px per frame
    else:
        assert np.median(d[:, 0]) < -5 and np.abs(d[:, 1]).mean() < 0.5                         # disparity along x only


def test_lk_borders_small_images_and_lost_points(api, oracle, synth):
    """points next to / outside the border, windows hanging over the edge (REFLECT_101 image, zero derivative border), an image so
    small that the pyramid stops early, flat regions (minEig test), far-off initial guesses"""
    rng = np.random.default_rng(5)
    for (h, w) in ((97, 131), (40, 45), (260, 333)):
        a = synth.random_image(900 + h, h, w)
        b = np.roll(a, (1, -2), axis=(0, 1)).copy()
        b[:, -2:] = a[:, -2:]
        pts = np.concatenate([rng.uniform([-8, -8], [w + 8, h + 8], size=(200, 2)),
                              np.array([[0, 0], [w - 1, h - 1], [0.5, h - 0.5], [w - 0.25, 0.25], [-20, 5], [w + 30, h + 30]])]).astype(np.float32)
        init = (pts + rng.normal(0, 1.5, size=pts.shape)).astype(np.float32)
        init[:5] += 40
        flat = a.copy(); flat[10:30, 10:40] = 77
        for prev, nxt in ((a, b), (flat, flat)):
            lk = api.LKTracker()
            g_pts, g_st, g_err = lk.track(prev, nxt, pts, init)
            r_pts, r_st, r_err = oracle.lk_track(prev, nxt, pts, init)
            assert np.array_equal(g_st, r_st)
            assert np.array_equal(g_pts.view(np.uint32), r_pts.view(np.uint32)), (h, w, np.abs(g_pts - r_pts).max())
            assert np.array_equal(g_err.view(np.uint32), r_err.view(np.uint32))
        assert not r_st.all()


def test_lk_other_parameters_and_empty(api, oracle, synth):
    a = synth.random_image(77, 120, 160); b = np.roll(a, 1, axis=1).copy()
    pts = np.random.default_rng(1).uniform([8, 8], [150, 110], size=(64, 2)).astype(np.float32)
    for win, lv, it in ((7, 1, 5), (15, 4, 30), (9, 0, 10)):
        g = api.LKTracker(win=win, max_level=lv, max_iters=it).track(a, b, pts, pts)
        r = oracle.lk_track(a, b, pts, pts, win=win, max_level=lv, max_iters=it)
        assert np.array_equal(g[1], r[1]) and np.array_equal(g[0].view(np.uint32), r[0].view(np.uint32))
    e = api.LKTracker().track(a, b, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert len(e[0]) == 0


def test_lk_batch(api, oracle, synth):
    import torch
    B, cap = 3, 256
    frames = [synth.stereo_pair(0, t) for t in range(B)]
    prev = np.stack([f[0] for f in frames]); nxt = np.stack([f[1] for f in frames])
    H, W = prev.shape[1:]
    pts = np.zeros((B, cap, 2), np.float32); cnt = np.zeros(B, np.int32)
    for b in range(B):
        p = _points(oracle, prev[b], 200 - 30 * b)[:cap]
        pts[b, :len(p)] = p; cnt[b] = len(p)
    init = pts.copy(); init[..., 0] -= 10
    d = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (prev, nxt, pts, init, cnt)]
    st = torch.zeros(B, cap, dtype=torch.uint8, device="cuda"); err = torch.zeros(B, cap, device="cuda")
    lk = api.LKTracker(stream=torch.cuda.current_stream().cuda_stream)
    lk.track_batch(d[0].data_ptr(), d[1].data_ptr(), B, H, W, W, H * W, d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), cap, st.data_ptr(), err.data_ptr())
    torch.cuda.synchronize()
    for b in range(B):
        n = cnt[b]
        r_pts, r_st, r_err = oracle.lk_track(prev[b], nxt[b], pts[b, :n], init[b, :n])
        assert np.array_equal(st[b, :n].cpu().numpy().astype(bool), r_st)
        assert np.array_equal(d[3][b, :n].cpu().numpy().view(np.uint32), r_pts.view(np.uint32))


def test_lk_image_cache_and_prefetch(api, oracle, synth):
    """myslam_lk_track_cached / myslam_lk_prefetch: the handle keeps the device copy and the pyramid of the two images it saw last under caller
    tokens (the `next` image of frame t is the `prev` image of frame t + 1; the image of frame t + 1 can be uploaded ahead).  Every call must
    return what the uncached call returns, bit for bit: a walk through five frames with hits, misses, a prefetch, an image modified in place
    (new token: must be uploaded again; the OLD token would still name the old bytes), an uncached stereo call in between, token 0, and a change
    of geometry."""
    imgs = [synth.stereo_pair(0, t)[0] for t in range(5)]
    R2 = synth.stereo_pair(0, 2)[1]
    pts = _points(oracle, imgs[0])
    plain, lk = api.LKTracker(), api.LKTracker()

    def same(a, b):
        return np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    tok = {i: 100 + i for i in range(5)}
    assert same(lk.track_cached(imgs[0], tok[0], imgs[1], tok[1], pts, pts), plain.track(imgs[0], imgs[1], pts, pts))          # two misses
    lk.prefetch(imgs[2], tok[2])                                                                                               # frame 2 ahead of its call
    assert same(lk.track_cached(imgs[1], tok[1], imgs[2], tok[2], pts, pts), plain.track(imgs[1], imgs[2], pts, pts))          # two hits
    # an uncached call on the same handle (FindFeaturesInRight: left -> right of the key-frame) leaves the cache alone
    assert same(lk.track(imgs[2], R2, pts, pts - [12.0, 0.0]), plain.track(imgs[2], R2, pts, pts - [12.0, 0.0]))
    # DeepLCD blurs the key-frame's image in place: other bytes, a NEW token -> uploaded again
    blurred = imgs[2].copy(); blurred[1:-1, 1:-1] = ((blurred[:-2, 1:-1].astype(np.int32) + blurred[2:, 1:-1] + blurred[1:-1, :-2] + blurred[1:-1, 2:]) // 4).astype(np.uint8)
    want = plain.track(blurred, imgs[3], pts, pts)
    assert same(lk.track_cached(blurred, 999, imgs[3], tok[3], pts, pts), want)
    assert not same(want, plain.track(imgs[2], imgs[3], pts, pts))                                                             # (the blur does change the tracks)
    assert same(lk.track_cached(imgs[3], tok[3], imgs[4], 0, pts, pts), plain.track(imgs[3], imgs[4], pts, pts))               # token 0: never looked up or kept
    assert same(lk.track_cached(imgs[3], tok[3], imgs[4], 0, pts, pts), plain.track(imgs[3], imgs[4], pts, pts))
    assert same(lk.track_cached(imgs[4], 7, imgs[4], 7, pts, pts), plain.track(imgs[4], imgs[4], pts, pts))                    # one token for both images
    # another geometry under a token seen before: nothing cached is usable
    small = [np.ascontiguousarray(im[40:300, 100:900]) for im in imgs[:2]]
    ps = _points(oracle, small[0])
    assert same(lk.track_cached(small[0], tok[0], small[1], tok[1], ps, ps), plain.track(small[0], small[1], ps, ps))
    assert same(lk.track_cached(imgs[0], tok[0], imgs[1], tok[1], pts, pts), plain.track(imgs[0], imgs[1], pts, pts))
    r = oracle.lk_track(imgs[0], imgs[1], pts, pts)
    g = lk.track_cached(imgs[0], tok[0], imgs[1], tok[1], pts, pts)
    assert np.array_equal(g[0].view(np.uint32), r[0].view(np.uint32)) and np.array_equal(g[1], r[1])
