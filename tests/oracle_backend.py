"""The CPU oracle behind the back-end interface of the product's chain (<pkg>/chain.py: Frontend / Backend / LoopClosing / Map of the
reference as one sequential schedule).  Test infrastructure: the same Chain runs once through chain.HipBackend (the C ABI) and once
through this class, and tests/test_gpu_sequence.py compares the two logs entry by entry."""
import numpy as np


class OracleBackend:
    name = "oracle"

    def __init__(self, o, weights, cfg, chain_mod):
        c = dict(chain_mod.DEFAULT_CONFIG, **(cfg or {}))
        mk = lambda n: o.params(int(c[n]), float(c["ORBextractor.scaleFactor"]), int(c["ORBextractor.nLevels"]),
                                int(c["ORBextractor.iniThFAST"]), int(c["ORBextractor.minThFAST"]))
        self.o, self.w = o, weights
        self.p_init, self.p_orb = mk("ORBextractor.nInitFeatures"), mk("ORBextractor.nNewFeatures")
        self.db_rows, self.db_ids = [], []

    def detect(self, img, mask, init):
        return self.o.detect(self.p_init if init else self.p_orb, img, mask)

    def lk_track(self, a, b, p0, p1):
        return self.o.lk_track(a, b, p0, p1)

    def triangulate(self, xl, yl, xr, yr, K):
        return self.o.triangulate_stereo(xl, yl, xr, yr, K["fx"], K["fy"], K["cx"], K["cy"], K["baseline"])

    def pose_only(self, pose, p3, obs, Kt, pre=0):
        return self.o.pose_only_optimize(pose, p3, obs, Kt, pre_optimize=pre)

    def ba(self, poses, pts, ep, el, obs, fixed, Kt):
        return self.o.ba_optimize_active_map(poses, pts, ep, el, obs, fixed, Kt)

    def lcd_descr(self, img):
        x, blurred = self.o.calc_preproc(img, blur_in_place=True)
        return self.o.calc_forward(self.w, x), blurred

    def screen(self, img, kps):
        return self.o.screen(self.p_orb, img, kps)

    def calc_desc(self, img, kps):
        return self.o.calc_descriptors(self.p_orb, img, kps)

    def db_add(self, kf_id, d):
        self.db_rows.append(np.asarray(d, np.float32)); self.db_ids.append(kf_id)

    def db_size(self):
        return len(self.db_ids)

    def db_query(self, d, cur, thr_low):
        if not self.db_rows:
            return (0, 0.0, 0)
        return self.o.lcddb_query(np.stack(self.db_rows), np.array(self.db_ids, np.uint64), d, cur, thr_low)

    def hamming(self, q, t):
        return self.o.hamming_match(q, t)

    def pnp(self, p3, p2, Kt):
        rc, pose, inl, n = self.o.solve_pnp_ransac(p3, p2, Kt)
        if rc != 0:
            raise RuntimeError(f"solvePnPRansac: no model ({rc})")
        return pose, inl, n

    def pgo(self, poses, fixed, e0, e1, meas):
        return self.o.pose_graph_optimize(poses, fixed, e0, e1, meas)

    def correct_points(self, old, new, first, pts):
        return self.o.correct_map_points(old, new, first, pts)

    def local_fusion(self, poses, cur, corrected, first, pts):
        return self.o.loop_local_fusion(poses, cur, corrected, first, pts)


class CheckedBackend:
    """LOCK-STEP comparison: every operator call of a chain goes to the HIP back end AND to the oracle back end with the SAME inputs, the two
    results are compared at the bars of that operator's own parity test, and the HIP result is what the chain continues with.  This is the
    entry-by-entry check of a whole sequence: two free-running chains cannot give it, because LK's initial flow is a re-projection with a
    float pose (frontend.cpp:136-147) — one ulp in the f32 start point moves a converged track by up to ~5e-3 px (measured, round 4), and
    everything behind it follows."""
    name = "hip (checked against the oracle call by call)"

    def __init__(self, hip, orc):
        self.h, self.o = hip, orc
        self.calls = {}
        self.dev = {}

    def _soft(self, op, budget):
        """One more call of `op` was accepted on the oracle's one-ulp self-spread instead of the fixed bar: counted (soft_bar_uses) and budgeted."""
        self.soft = getattr(self, "soft", {})
        self.soft[op] = self.soft.get(op, 0) + 1
        self.dev["soft_bar_uses_" + op] = float(self.soft[op])
        assert self.soft[op] <= budget, f"{op}: the soft bar (oracle self-spread) was needed {self.soft[op]} times, budget {budget}: an ordering / accept-reject regression, not a weak problem"

    def _note(self, op, key=None, val=0.0):
        self.calls[op] = self.calls.get(op, 0) + 1
        if key:
            self.dev[key] = max(self.dev.get(key, 0.0), float(val))

    def detect(self, img, mask, init):
        a, b = self.h.detect(img, mask, init), self.o.detect(img, mask, init)
        assert a.tobytes() == b.tobytes(), f"Detect #{self.calls.get('detect', 0)}"
        self._note("detect")
        return a

    def lk_track(self, a, b, p0, p1):
        gn, gs, ge = self.h.lk_track(a, b, p0, p1); rn, rs, _ = self.o.lk_track(a, b, p0, p1)
        k = self.calls.get("lk_track", 0)
        assert np.array_equal(gs, rs), f"LK status, call {k}"
        assert gn[gs.astype(bool)].tobytes() == rn[rs.astype(bool)].tobytes(), f"LK tracks, call {k}: not bit-identical on identical inputs"
        self._note("lk_track")
        return gn, gs, ge

    def triangulate(self, xl, yl, xr, yr, K):
        (gx, gok), (rx, rok) = self.h.triangulate(xl, yl, xr, yr, K), self.o.triangulate(xl, yl, xr, yr, K)
        assert np.array_equal(gok, rok) and np.allclose(gx[gok], rx[rok], rtol=1e-9, atol=1e-9)
        self._note("triangulate", "triangulate_abs", np.abs(gx[gok] - rx[rok]).max() if gok.any() else 0.0)
        return gx, gok

    def pose_only(self, pose, p3, obs, Kt, pre=0):
        (gp, go, gi), (rp, ro, ri) = self.h.pose_only(pose, p3, obs, Kt, pre), self.o.pose_only(pose, p3, obs, Kt, pre)
        self._note("pose_only", "pose_only_abs", np.abs(gp - rp).max())
        # (the operator test holds 1e-8 on well-conditioned synthetic problems; a real frame's last Levenberg steps sit at the noise floor)
        ok = np.allclose(gp, rp, rtol=1e-6, atol=1e-6) and np.array_equal(go, ro) and gi == ri
        if not ok:
            # A frame that keeps ~35 of its matches (the tracker is about to ask for a key-frame) can leave a Levenberg accept / reject decision on
            # the last bits of a sum: the ORACLE's own result then moves by more than the bar when one observation changes by one ulp
            # (tests/golden/pose_only_weak_frame.npz, tests/test_gpu_pose_only.py::test_weak_frame).  The rule of the chaotic local-BA window and of
            # the flat-valley pose graph: the bar is three times the oracle's own one-ulp spread, the flags those of one of the oracle's runs.
            import os
            d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(d, exist_ok=True)
            np.savez(os.path.join(d, f"pose_only_mismatch_{self.calls['pose_only'] - 1}.npz"), pose=pose, p3=p3, obs=obs, Kt=np.array(Kt), pre=pre, hip_pose=gp, hip_outlier=go,
                     oracle_pose=rp, oracle_outlier=ro)
            rng = np.random.default_rng(0)
            runs = [self.o.pose_only(pose, p3, obs * (1 + rng.choice([-1.0, 1.0], size=obs.shape) * 2.2e-16), Kt, pre) for _ in range(4)]
            spread = max(float(np.abs(q[0] - rp).max()) for q in runs)
            self.dev["pose_only_oracle_one_ulp_spread"] = max(self.dev.get("pose_only_oracle_one_ulp_spread", 0.0), spread)
            # (advisor, round 5) the soft bar has a BUDGET: it is for the rare weak frame, not a way round an ordering regression — the lock-step runs
            # of rounds 4 / 5 took it on 0 .. 2 of ~300 frames per sequence; more than max(2, 1 %) of the calls fails the run whatever the spreads say
            self._soft("pose_only", max(2, self.calls["pose_only"] // 100))
            ok = float(np.abs(gp - rp).max()) <= 3.0 * spread and any(np.array_equal(go, q[1]) and gi == q[2] for q in runs + [(rp, ro, ri)])
            assert ok, (f"pose-only call {self.calls['pose_only'] - 1}", np.abs(gp - rp).max(), gi, ri, "oracle one-ulp spread", spread)
        return gp, go, gi

    def ba(self, poses, pts, ep, el, obs, fixed, Kt):
        g = self.h.ba(poses, pts, ep, el, obs, fixed, Kt); r = self.o.ba(poses, pts, ep, el, obs, fixed, Kt)
        k = self.calls.get("ba", 0)
        assert np.allclose(g[0], r[0], rtol=1e-6, atol=1e-6) and np.allclose(g[1], r[1], rtol=1e-6, atol=1e-6), (f"local BA {k}: iterates", np.abs(g[0] - r[0]).max(), np.abs(g[1] - r[1]).max())
        assert np.allclose(g[2], r[2], rtol=1e-5, atol=1e-7), (f"local BA {k}: chi2", np.abs(g[2] - r[2]).max())
        near = np.abs(r[2] - 5.991) < 1e-4                        # flags may only differ where chi2 sits on the threshold
        assert np.array_equal(g[3][~near], r[3][~near]) and g[4] == r[4] and abs(g[5] - r[5]) <= int(near.sum()), f"local BA {k}: flags / rounds"
        self._note("ba", "ba_pose_abs", np.abs(g[0] - r[0]).max())
        return g

    def lcd_descr(self, img):
        (gd, gb), (rd, rb) = self.h.lcd_descr(img), self.o.lcd_descr(img)
        assert np.abs(gd - rd).max() < 2e-5 and np.array_equal(gb, rb)
        self._note("lcd_descr", "lcd_abs", np.abs(gd - rd).max())
        return gd, gb

    def screen(self, img, kps):
        a, b = self.h.screen(img, kps), self.o.screen(img, kps)
        assert a.tobytes() == b.tobytes(), "ScreenAndComputeKPsParams"
        self._note("screen")
        return a

    def calc_desc(self, img, kps):
        a, b = self.h.calc_desc(img, kps), self.o.calc_desc(img, kps)
        assert np.array_equal(a, b), "CalcDescriptors"
        self._note("calc_desc")
        return a

    def db_add(self, kf_id, d):
        self.h.db_add(kf_id, d); self.o.db_add(kf_id, d)

    def db_size(self):
        assert self.h.db_size() == self.o.db_size()
        return self.h.db_size()

    def db_query(self, d, cur, thr_low):
        g, r = self.h.db_query(d, cur, thr_low), self.o.db_query(d, cur, thr_low)
        rows = np.stack(self.o.db_rows)
        near = int((np.abs(rows @ np.asarray(d, np.float32) - thr_low) < 1e-5).sum())
        assert g[0] == r[0] and abs(g[1] - r[1]) < 2e-5 and abs(g[2] - r[2]) <= near, (g, r)
        self._note("db_query", "db_score_abs", abs(g[1] - r[1]))
        return g

    def hamming(self, q, t):
        (gi, gd), (ri, rd) = self.h.hamming(q, t), self.o.hamming(q, t)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
        self._note("hamming")
        return gi, gd

    def pnp(self, p3, p2, Kt):
        (gp, gi, gn), (rp, ri, rn) = self.h.pnp(p3, p2, Kt), self.o.pnp(p3, p2, Kt)
        s = 1.0 if np.dot(gp[:4], rp[:4]) >= 0 else -1.0
        assert gn == rn and np.array_equal(gi, ri), ("PnP consensus", gn, rn, int((gi != ri).sum()))
        dq, dt = float(np.abs(gp[:4] * s - rp[:4]).max()), float(np.abs(gp[4:] - rp[4:]).max())
        # the refinement stops at |step| < 1e-12 in its own parameters on both sides; on the sequences' ~60-match problems the two poses
        # then agree to ~1e-9 (seen: 1.2e-9 in t): the bar is 1e-8 (tests/test_gpu_pnp.py holds 1e-9 on its well-conditioned problems)
        assert dq < 1e-8 and dt < 1e-8, ("PnP refined pose", dq, dt)
        self.dev["pnp_pose_abs"] = max(self.dev.get("pnp_pose_abs", 0.0), dq, dt)
        self._note("pnp")
        return gp, gi, gn

    def pgo(self, poses, fixed, e0, e1, meas):
        g, r = self.h.pgo(poses, fixed, e0, e1, meas), self.o.pgo(poses, fixed, e0, e1, meas)
        dev = float(np.abs(g[0] - r[0]).max())
        ok = abs(g[1] - r[1]) <= 1e-3 * abs(r[1]) + 1e-12 and dev < 5e-4
        spread = None
        if not ok and abs(g[1] - r[1]) <= 1e-3 * abs(r[1]) + 1e-12:
            # A long chain closed by ONE loop edge (52 key-frames, 52 edges: tests/golden/pgo_flat_valley.npz) leaves the 20 Levenberg iterations of
            # LoopClosing::PoseGraphOptimization in a flat valley: chi2 agrees to 1e-9 while the poses still move by 1e-3 per further iteration, and
            # the ORACLE's own result moves by 2e-3 .. 6e-3 when one measurement changes by one ulp.  There the bar is the oracle's own spread
            # (the rule of the chaotic local-BA window, tests/golden/ba_chaotic_window.npz; tests/test_gpu_pgo.py allows 10 x): chi2 to 1e-3, poses
            # within three times that spread.  The second loop of the same drive (58 key-frames, 2 loop edges) is softer still: the oracle moves by
            # 0.015 .. 0.026 under one ulp and by 0.78 between 20 and 25 iterations, chi2 by 5e-6; the HIP result sits 0.037 away.
            rng = np.random.default_rng(0)
            spread = max(float(np.abs(self.o.pgo(poses, fixed, e0, e1, meas * (1 + rng.choice([-1.0, 1.0], size=meas.shape) * 2.2e-16))[0] - r[0]).max())
                         for _ in range(4))
            ok = dev <= 3.0 * spread
            self.dev["pgo_oracle_one_ulp_spread"] = max(self.dev.get("pgo_oracle_one_ulp_spread", 0.0), spread)
            self._soft("pgo", 2)                                  # budget: the two flat-valley graphs of the two-lap drive, nothing more
        if not ok:          # keep the failing problem for an offline look (gpurun_out/ travels back from the GPU box)
            import os
            d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(d, exist_ok=True)
            np.savez(os.path.join(d, "pgo_mismatch.npz"), poses=poses, fixed=fixed, e0=e0, e1=e1, meas=meas, hip_poses=g[0], hip_chi2=g[1], oracle_poses=r[0], oracle_chi2=r[1])
        assert ok, ("pose graph: chi2 (HIP, oracle)", float(g[1]), float(r[1]), "largest pose deviation", dev, "oracle one-ulp self-spread", spread,
                    "key-frames", len(poses), "edges", len(e0))
        self._note("pgo", "pgo_pose_abs", dev)
        return g

    def correct_points(self, old, new, first, pts):
        g, r = self.h.correct_points(old, new, first, pts), self.o.correct_points(old, new, first, pts)
        assert np.abs(g - r).max() < 1e-10
        self._note("correct_points")
        return g

    def local_fusion(self, poses, cur, corrected, first, pts):
        g, r = self.h.local_fusion(poses, cur, corrected, first, pts), self.o.local_fusion(poses, cur, corrected, first, pts)
        assert np.allclose(g[0], r[0], rtol=0, atol=1e-12) and np.allclose(g[1], r[1], rtol=1e-12, atol=1e-11)
        self._note("local_fusion")
        return g
