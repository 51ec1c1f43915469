"""GPU fuzz: DetectAndCompute / pyramid / blur on random image sizes and parameters vs the oracle (bit-exact)."""
import sys, numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401  (loads the HIP runtime torch ships before ours)
import __graft_entry__ as g
pkg = g.load_package(); api, synth = pkg.api, pkg.synth
sys.path.insert(0, "oracle")
from pyoracle import Oracle
o = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    h = int(rng.integers(110, 520)); w = int(rng.integers(140, 1300))
    nf = int(rng.choice([60, 200, 500, 1000, 2000])); nl = int(rng.choice([1, 3, 5, 8])); sf = float(rng.choice([1.1, 1.2, 1.2, 1.3]))
    u = rng.random()
    kind = "noise" if u < 0.2 else ("texture" if u < 0.55 else "sparse")
    if kind == "sparse":      # a scene with few rectangles: a few % of FAST corners, the regime of the two-phase FAST path
        img = np.ascontiguousarray(synth.stereo_batch(1, stream_id=int(rng.integers(1000)), n_rect=int(rng.integers(20, 1500)), h=h, w=w)[0, 0])
    else:
        img = np.ascontiguousarray(synth.random_image(int(rng.integers(1 << 30)), h, w, kind))
    try:
        ext = api.ORBextractor(nf, sf, nl)
    except Exception as e:
        print("create failed", h, w, nf, nl, sf, e); continue
    p = o.params(nf); p.scale_factor = sf; p.nlevels = nl
    mode = int(rng.choice([-1, -1, 0, 1]))
    ext.set_option(ext.OPT_FAST_MODE, mode)
    # round-4 options, both must leave every byte unchanged: the descriptor kernel as a limited (persistent) grid, the Gaussian on the matrix cores
    side = int(rng.choice([0, 0, 1, 2, 3])); mf = int(rng.random() < 0.3)
    ext.set_option(ext.OPT_SIDE_BLOCKS_PER_CU, side); ext.set_option(ext.OPT_BLUR_MFMA, mf)
    if mode < 0 and rng.random() < 0.5:
        try:
            ext.DetectAndCompute(np.ascontiguousarray(synth.random_image(int(rng.integers(1 << 30)), h, w, "noise")))     # primes the path statistics
        except api.MyslamError:
            pass
    try:
        gk, gd = ext.DetectAndCompute(img)
    except api.MyslamError as e:
        # image too small for the FAST grid at some level: the oracle must refuse as well
        try:
            o.detect_and_compute(p, img, cap=20000); print("GPU refused but oracle ran", h, w, nf, nl, sf, e); bad += 1
        except AssertionError:
            pass
        continue
    rk, rd = o.detect_and_compute(p, img, cap=ext.max_keypoints(h, w) + 8)
    ok = gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    if ok:
        for l in range(nl):
            ok = ok and np.array_equal(ext.debug_pyramid(img, l, blurred=True), o.blur7(o.pyramid(p, img)[l], 0))
    if ok and rng.random() < 0.5:
        # the same handle on more frames of the same shape: the second call captures a HIP graph, the later ones replay it (both FAST
        # statistics parities); masks on and off alternate the graph key
        for rep in range(4):
            img2 = np.ascontiguousarray(synth.random_image(int(rng.integers(1 << 30)), h, w, "texture" if rep % 2 else "noise"))
            mask = None
            if rng.random() < 0.3:
                mask = np.full((h, w), 255, np.uint8); y0, x0 = int(rng.integers(0, h - 20)), int(rng.integers(0, w - 20)); mask[y0:y0 + h // 3, x0:x0 + w // 3] = 0
            g2, d2 = ext.DetectAndCompute(img2, mask)
            r2, e2 = o.detect_and_compute(p, img2, mask, cap=ext.max_keypoints(h, w) + 8)
            ok = ok and g2.tobytes() == r2.tobytes() and np.array_equal(d2, e2)
    if ok and h * w < 160 * 1024 and rng.random() < 0.15:
        # the image as entry 17 of a batch of 64 + different frames: the batch forms of the kernels (256-thread oct-tree blocks, tile-ordered
        # descriptor pass, level 0 read in place) must give the single-image bytes
        import torch
        B = 64 + int(rng.integers(0, 9))
        batch = np.stack([img if i == 17 else synth.random_image(int(rng.integers(1 << 30)), h, w, "texture") for i in range(B)])
        cap = ext.max_keypoints(h, w)
        d_imgs = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_c = torch.zeros(B, dtype=torch.int32, device="cuda"); d_s = torch.ones(B, dtype=torch.int32, device="cuda")
        ext.set_stream(torch.cuda.current_stream().cuda_stream)
        ext.detect_and_compute_batch(d_imgs.data_ptr(), B, h, w, w, h * w, d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), cap)
        torch.cuda.synchronize()
        c = d_c.cpu().numpy(); kk = d_k.cpu().numpy().view(api.KP_DTYPE).reshape(B, cap); dd = d_d.cpu().numpy().reshape(B, cap, 32)
        ok = ok and int(d_s.abs().sum()) == 0 and kk[17, :c[17]].tobytes() == rk.tobytes() and np.array_equal(dd[17, :c[17]], rd)
        for b2 in (0, B - 1):
            r2, e2 = o.detect_and_compute(p, batch[b2], cap=cap + 8)
            ok = ok and kk[b2, :c[b2]].tobytes() == r2.tobytes() and np.array_equal(dd[b2, :c[b2]], e2)
        ext.set_stream(0)
    if not ok:
        bad += 1
        print("MISMATCH", dict(h=h, w=w, nf=nf, nl=nl, sf=sf, kind=kind, mode=mode, side=side, mfma=mf, n_gpu=len(gk), n_ref=len(rk)))
print(f"fuzz done: {N} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
