"""One handle per thread (include/myslam_hip.h: "a handle serves one thread at a time; create one handle per thread" — the reference shares
one extractor between its frontend and loop-closing threads, src/system.cpp:31,54,66): four host threads drive their own handles and
the handle-free entry points at the same time — graph capture and replay of the one-frame extractor calls included — and every
result must equal what the oracle computed beforehand."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_four_threads_with_their_own_handles(api, oracle, synth):
    nthreads, rounds = 4, 6
    work = []
    for t in range(nthreads):
        h, w = 240 + 24 * t, 360 + 40 * t
        imgs = [synth.random_image(9100 + 10 * t + i, h, w, "texture" if i % 2 else "noise") for i in range(3)]
        p = oracle.params(300 + 100 * t)
        ref = [oracle.detect_and_compute(p, im) for im in imgs]
        det = [oracle.detect(oracle.params(100 + 20 * t), im) for im in imgs]
        ham = oracle.hamming_match(ref[0][1], ref[1][1])
        pts0 = np.stack([ref[0][0]["x"], ref[0][0]["y"]], 1).astype(np.float32)[:120]
        lk = oracle.lk_track(imgs[0], np.roll(imgs[0], -3, axis=1), pts0, pts0)
        poses, pts, ep, el, obs, fixed, Kt = synth.ba_problem(seed=0xBA + t, n_kf=5 + t % 2, n_mp=60 + 10 * t)
        ba = oracle.ba_optimize_active_map(poses, pts, ep, el, obs, fixed, Kt)
        # a synchronous host-pointer call without a stream of its own (it runs on a non-blocking stream per calling thread: never the legacy
        # stream, which HIP refuses to touch while ANOTHER thread records a graph on a blocking stream — the one-frame extractor calls here do)
        pw, uv, Kp, _, _ = synth.pnp_problem(120 + 30 * t, 0.4, 0.5, seed=70 + t)
        pnp = oracle.solve_pnp_ransac(pw, uv, Kp)
        work.append(dict(imgs=imgs, nf=300 + 100 * t, nd=100 + 20 * t, ref=ref, det=det, ham=ham, pts0=pts0, lk=lk,
                         ba_in=(poses, pts, ep, el, obs, fixed, Kt), ba=ba, pnp_in=(pw, uv, Kp), pnp=pnp))
    errors = []
    start = threading.Barrier(nthreads)

    def run(t):
        try:
            wk = work[t]
            ext, dex, lkt = api.ORBextractor(wk["nf"]), api.ORBextractor(wk["nd"]), api.LKTracker()
            start.wait()
            for r in range(rounds):
                for i, im in enumerate(wk["imgs"]):           # the same shapes again and again: eager, capture, replay
                    k, d = ext.DetectAndCompute(im)
                    assert k.tobytes() == wk["ref"][i][0].tobytes() and np.array_equal(d, wk["ref"][i][1]), ("DetectAndCompute", t, r, i)
                    kd = dex.Detect(im)
                    assert kd.tobytes() == wk["det"][i].tobytes(), ("Detect", t, r, i)
                    if wk["pnp"][0] == 0:
                        gp, gin, gn = api.solve_pnp_ransac(*wk["pnp_in"])
                        assert gn == wk["pnp"][3] and np.array_equal(gin, wk["pnp"][2]), ("pnp", t, r, i)
                idx, dist = api.hamming_match(wk["ref"][0][1], wk["ref"][1][1])
                assert np.array_equal(idx, wk["ham"][0]) and np.array_equal(dist, wk["ham"][1]), ("hamming", t, r)
                o, s, _ = lkt.track(wk["imgs"][0], np.roll(wk["imgs"][0], -3, axis=1), wk["pts0"], wk["pts0"])
                assert np.array_equal(s, wk["lk"][1]) and np.array_equal(o, wk["lk"][0]), ("lk", t, r)
                p2, x2, chi, out, rd, no = api.ba_optimize_active_map(*wk["ba_in"])
                rp, rx, rchi, rout, rrd, rno = wk["ba"]
                assert rd == rrd and np.allclose(p2, rp, rtol=1e-7, atol=1e-9) and np.allclose(x2, rx, rtol=1e-7, atol=1e-8), ("ba", t, r)
                assert np.array_equal(out[np.abs(rchi - 5.991) > 1e-6], rout[np.abs(rchi - 5.991) > 1e-6])
        except Exception as e:      # noqa: BLE001  (reported below, with the thread that raised it)
            errors.append((t, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors
    assert not any(th.is_alive() for th in threads)
