"""A whole tracking / mapping / loop-closing CHAIN over the operators of the path, written once and run through two back ends:
the HIP library (ctypes over the C ABI, api.py) and the CPU oracle (pyoracle).  It follows the reference's control flow
  Frontend::StereoInit / TrackLastFrame / EstimateCurrentPose / InsertKeyFrame (DetectFeatures, FindFeaturesInRight,
  TriangulateNewPoints)                                              src/frontend.cpp:126-276, 300-488
  Backend::OptimizeActiveMap                                         src/backend.cpp:126-266
  LoopClosing::ProcessNewKF / DetectLoop / MatchFeatures / ComputeCorrectPose / OptimizeCurrentPose / PoseGraphOptimization
                                                                     src/loopclosing.cpp:83-203, 208-335, 339-433, 537-646
with plain Python lists in place of Map / KeyFrame / Feature / MapPoint (no threads, no viewer).  One deliberate simplification keeps
the 2-D side of the two runs bit-identical: LK always starts from the feature's last position (the reference's branch for features
without a map point, frontend.cpp:141-145) instead of from the re-projection with the predicted pose, so no floating-point pose ever
feeds the integer tracker.  tests/test_gpu_sequence.py compares the two logs entry by entry."""
import numpy as np

CHI2_TH = 5.991


# ---- SE3 as (qx qy qz qw tx ty tz), Tcw ------------------------------------------------------------------------------------------
def q_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_q(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2; q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2; q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2; q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2; q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q)
    return q if q[3] >= 0 else -q


def T_of(p7):
    T = np.eye(4); T[:3, :3] = q_to_R(np.asarray(p7[:4], float)); T[:3, 3] = p7[4:]
    return T


def p7_of(T):
    return np.concatenate([R_to_q(T[:3, :3]), T[:3, 3]])


IDENT = np.array([0, 0, 0, 1, 0, 0, 0], float)


# ---- the two back ends behind one interface ----------------------------------------------------------------------------------------
class HipBackend:
    name = "hip"

    def __init__(self, api, weights):
        self.api = api
        self.det_init, self.det, self.orb = api.ORBextractor(300), api.ORBextractor(120), api.ORBextractor(300)
        self.lk, self.lcd = api.LKTracker(), api.DeepLCD(weights)
        self.db = api.LoopDatabase(256)

    def detect(self, img, mask, init):
        return (self.det_init if init else self.det).Detect(img, mask)

    def lk_track(self, a, b, p0, p1):
        return self.lk.track(a, b, p0, p1)

    def triangulate(self, xl, yl, xr, yr, K):
        return self.api.triangulate_stereo(xl, yl, xr, yr, K["fx"], K["fy"], K["cx"], K["cy"], K["bf"] / K["fx"])

    def pose_only(self, pose, p3, obs, Kt, pre=0):
        return self.api.pose_only_optimize(pose, p3, obs, Kt, pre_optimize=pre)

    def ba(self, poses, pts, ep, el, obs, fixed, Kt):
        return self.api.ba_optimize_active_map(poses, pts, ep, el, obs, fixed, Kt)

    def lcd_descr(self, img):
        return self.lcd.calcDescrOriginalImg(img, blur_in_place=True)

    def screen(self, img, kps):
        return self.orb.ScreenAndComputeKPsParams(img, kps)[0]

    def calc_desc(self, img, kps):
        return self.orb.CalcDescriptors(img, kps)

    def db_add(self, kf_id, d):
        self.db.AddToDatabase(kf_id, d)

    def db_query(self, d, cur):
        return self.db.query(d, cur)

    def hamming(self, q, t):
        return self.api.hamming_match(q, t)

    def pnp(self, p3, p2, Kt):
        return self.api.solve_pnp_ransac(p3, p2, Kt)

    def pgo(self, poses, fixed, e0, e1, meas):
        return self.api.pose_graph_optimize(poses, fixed, e0, e1, meas)

    def correct_points(self, old, new, first, pts):
        return self.api.correct_map_points(old, new, first, pts)

    def local_fusion(self, poses, cur, corrected, first, pts):
        return self.api.loop_local_fusion(poses, cur, corrected, first, pts)


class OracleBackend:
    name = "oracle"

    def __init__(self, o, weights):
        self.o, self.w = o, weights
        self.db_rows, self.db_ids = [], []

    def detect(self, img, mask, init):
        return self.o.detect(self.o.params(300 if init else 120), img, mask)

    def lk_track(self, a, b, p0, p1):
        return self.o.lk_track(a, b, p0, p1)

    def triangulate(self, xl, yl, xr, yr, K):
        return self.o.triangulate_stereo(xl, yl, xr, yr, K["fx"], K["fy"], K["cx"], K["cy"], K["bf"] / K["fx"])

    def pose_only(self, pose, p3, obs, Kt, pre=0):
        return self.o.pose_only_optimize(pose, p3, obs, Kt, pre_optimize=pre)

    def ba(self, poses, pts, ep, el, obs, fixed, Kt):
        return self.o.ba_optimize_active_map(poses, pts, ep, el, obs, fixed, Kt)

    def lcd_descr(self, img):
        x, blurred = self.o.calc_preproc(img, blur_in_place=True)
        return self.o.calc_forward(self.w, x), blurred

    def screen(self, img, kps):
        return self.o.screen(self.o.params(300), img, kps)

    def calc_desc(self, img, kps):
        return self.o.calc_descriptors(self.o.params(300), img, kps)

    def db_add(self, kf_id, d):
        self.db_rows.append(np.asarray(d, np.float32)); self.db_ids.append(kf_id)

    def db_query(self, d, cur):
        if not self.db_rows:
            return (0, 0.0, 0)
        return self.o.lcddb_query(np.stack(self.db_rows), np.array(self.db_ids, np.uint64), d, cur)

    def hamming(self, q, t):
        return self.o.hamming_match(q, t)

    def pnp(self, p3, p2, Kt):
        rc, pose, inl, n = self.o.solve_pnp_ransac(p3, p2, Kt)
        assert rc == 0
        return pose, inl, n

    def pgo(self, poses, fixed, e0, e1, meas):
        return self.o.pose_graph_optimize(poses, fixed, e0, e1, meas)

    def correct_points(self, old, new, first, pts):
        return self.o.correct_map_points(old, new, first, pts)

    def local_fusion(self, poses, cur, corrected, first, pts):
        return self.o.loop_local_fusion(poses, cur, corrected, first, pts)


# ---- the chain -----------------------------------------------------------------------------------------------------------------------
class Chain:
    def __init__(self, be, api, K, frames, kf_every=6, window=7, anchor_gauge=False, lcd_min_db=25, lcd_thr_high=0.94, lcd_thr_low=0.92):
        """frames: list of (left, right) uint8 images; api: the product's host helpers (pyramid expansion, match -> feature pairs).
        anchor_gauge: DIAGNOSTIC, not the reference's behaviour — after every local BA the window (poses + the landmarks it moved) is
        put back rigidly so that its oldest key-frame keeps the pose it had (the reference's graph fixes no pose, backend.cpp:139-150)"""
        self.be, self.api, self.K, self.frames = be, api, K, frames
        self.anchor_gauge = anchor_gauge
        # the loop detector's configuration values (config/*.yaml: LCD.nDatabaseMinSize, LCD.similarityScoreThreshold.high / .low; the KITTI
        # files say 50 / 0.94 / 0.92 — a 200-frame sequence makes 34 key-frames, so the gate is lowered, the thresholds are KITTI's)
        self.lcd_min_db, self.lcd_thr_high, self.lcd_thr_low = lcd_min_db, lcd_thr_high, lcd_thr_low
        self.detected = []          # (current key-frame index, loop key-frame index) pairs DetectLoop accepted
        self.Kt = (K["fx"], K["fy"], K["cx"], K["cy"])
        self.kf_every, self.window = kf_every, window
        self.log = []
        self.points = {}            # map point id -> xyz (world = camera 0)
        self.first_kf = {}          # map point id -> index of the key-frame that first observed it
        self.obs = {}               # map point id -> {kf index: pixel}
        self.kfs = []               # dicts: frame, pose, px [n,2] f32, mp [n] ids (-1 = none), rel (relative pose to the previous KF), descr, pyr, desc
        self.next_mp = 0
        self.poses = []             # Tcw per frame

    def rec(self, tag, *arrays):
        self.log.append((tag, [np.array(a) for a in arrays]))

    # Frontend::FindFeaturesInRight + triangulation of the features that have no map point yet (frontend.cpp:335-379, 451-488 / 385-417)
    def stereo_points(self, L, R, px, mp, pose, kf_index):
        nxt, st, _ = self.be.lk_track(L, R, px, px)
        self.rec("lk_right", nxt, st)
        need = np.nonzero((mp < 0) & st)[0]
        if len(need):
            xyz, ok = self.be.triangulate(px[need, 0], px[need, 1], nxt[need, 0], nxt[need, 1], self.K)
            self.rec("triangulate", ok, xyz[ok])
            Twc = np.linalg.inv(T_of(pose))
            for j, i in enumerate(need):
                if ok[j]:
                    self.points[self.next_mp] = Twc[:3, :3] @ xyz[j] + Twc[:3, 3]
                    self.first_kf[self.next_mp] = kf_index
                    self.obs[self.next_mp] = {}
                    mp[i] = self.next_mp
                    self.next_mp += 1
        return mp

    def insert_keyframe(self, t, pose, px, mp, init):
        L, R = self.frames[t]
        kf_index = len(self.kfs)
        # Frontend::DetectFeatures: mask out a 41 x 41 square around every tracked feature (frontend.cpp:300-315)
        mask = np.full(L.shape, 255, np.uint8)
        for (x, y) in px:
            x0, y0 = int(round(float(x))), int(round(float(y)))
            mask[max(y0 - 20, 0):y0 + 21, max(x0 - 20, 0):x0 + 21] = 0
        new = self.be.detect(L, mask, init)
        self.rec("detect", new)
        npx = np.stack([new["x"], new["y"]], 1).astype(np.float32).reshape(-1, 2)
        px = np.concatenate([px, npx]).astype(np.float32); mp = np.concatenate([mp, np.full(len(npx), -1, np.int64)])
        mp = self.stereo_points(L, R, px, mp, pose, kf_index)
        for i, m in enumerate(mp):
            if m >= 0:
                self.obs[m][kf_index] = px[i].astype(np.float64)
        rel = IDENT.copy() if not self.kfs else p7_of(T_of(pose) @ np.linalg.inv(T_of(self.kfs[-1]["pose"])))      # mRelativePoseToLastKF
        kf = {"frame": t, "pose": np.array(pose, float), "px": px.copy(), "mp": mp.copy(), "rel": rel, "img": L.copy()}
        self.kfs.append(kf)
        if kf_index > 0:
            self.local_ba()
            pose = self.kfs[-1]["pose"].copy()
        self.process_new_kf(kf)
        return pose, px, mp

    # Backend::OptimizeActiveMap (backend.cpp:126-266): the last `window` key-frames, every map point they observe
    def local_ba(self):
        win = list(range(max(0, len(self.kfs) - self.window), len(self.kfs)))            # Map::GetActiveKeyFrames
        idx = set(win)
        act = [m for m, o in self.obs.items() if any(k in idx for k in o)]               # Map::GetActiveMapPoints (container order is arbitrary)
        rows = [(m, k) for m in act for k in sorted(self.obs[m]) if k in idx]            # MapPoint::GetActiveObservations, list order = insertion order
        # the graph-build rules of backend.cpp:139-206 live behind the C ABI (myslam_ba_flatten_window): skip outliers, fix landmarks whose
        # first observer left the window, vertices by id, edges grouped by landmark
        fl = self.api.ba_flatten_window(win, act, np.zeros(len(act), np.uint8), [self.first_kf[m] for m in act],
                                        [m for m, _ in rows], [k for _, k in rows], np.array([self.obs[m][k] for m, k in rows], np.float32).reshape(-1, 2),
                                        np.zeros(len(rows), np.uint8))
        win = [win[i] for i in fl["pose_src"]]; mps = [act[i] for i in fl["pt_src"]]
        edge_ref = [rows[i] for i in fl["edge_src"]]; fixed = fl["fixed"]
        poses = np.stack([self.kfs[k]["pose"] for k in win])
        pts = np.stack([self.points[m] for m in mps])
        p2, x2, chi, out, rounds, nout = self.be.ba(poses, pts, fl["edge_pose"], fl["edge_pt"], fl["edge_obs"], fixed, self.Kt)
        self.rec("ba", p2, x2, out, np.array([rounds, nout]), chi)
        if self.anchor_gauge:
            G = np.linalg.inv(T_of(poses[0])) @ T_of(p2[0])                   # world motion of the oldest key-frame: Twc_old * Tcw_new
            Gi = np.linalg.inv(G)
            p2 = np.stack([p7_of(T_of(q) @ Gi) for q in p2])
            x2 = np.where(fixed[:, None] != 0, x2, x2 @ G[:3, :3].T + G[:3, 3])
        for i, k in enumerate(win):
            self.kfs[k]["pose"] = p2[i].copy()
        for j, m in enumerate(mps):
            if not fixed[j]:
                self.points[m] = x2[j].copy()
        for e, (m, k) in enumerate(edge_ref):                        # backend.cpp:234-250: outlier observations are detached
            if out[e]:
                del self.obs[m][k]
                kf = self.kfs[k]
                kf["mp"][kf["mp"] == m] = -1
        for m in [m for m, o in self.obs.items() if not o]:          # map points without observations leave the map
            del self.obs[m], self.points[m], self.first_kf[m]

    # LoopClosing::ProcessNewKF + DetectLoop + AddToDatabase (loopclosing.cpp:83-161, 651-659)
    def process_new_kf(self, kf):
        d, blurred = self.be.lcd_descr(kf["img"])                    # blurs the key-frame's image in place (reference quirk 7)
        kf["img"] = blurred
        feats = np.zeros(len(kf["px"]), self.api.KP_DTYPE)
        feats["x"], feats["y"], feats["size"], feats["angle"], feats["octave"], feats["class_id"] = kf["px"][:, 0], kf["px"][:, 1], 7, -1, 0, -1
        pyr = self.api.expand_pyramid_keypoints(feats, 8)
        kf["pyr"] = self.be.screen(kf["img"], pyr)
        kf["desc"] = self.be.calc_desc(kf["img"], kf["pyr"])
        kf["descr"] = d
        kf_id = len(self.kfs) - 1
        q = self.be.db_query(d, 3 * kf_id + 5)                       # key-frame ids spaced by 3: the `cur - id < 20` cut-off hides the last five
        self.rec("lcd", d, kf["pyr"], kf["desc"], np.array([q[0], q[2]]), np.array([q[1]]))
        # LoopClosingRun (loopclosing.cpp:62-75): the detector runs once the database holds more than nDatabaseMinSize key-frames (the current
        # key-frame is added afterwards, :78); DetectLoop's decision (:140-148): maxScore >= high and at most 3 scores above low
        if kf_id > self.lcd_min_db and q[1] >= self.lcd_thr_high and q[2] <= 3:
            self.detected.append((kf_id, int(q[0]) // 3))
        self.be.db_add(3 * kf_id, d)

    def run(self):
        L0, R0 = self.frames[0]
        pose = IDENT.copy()
        pose, px, mp = self.insert_keyframe(0, pose, np.zeros((0, 2), np.float32), np.zeros(0, np.int64), True)     # Frontend::StereoInit
        self.poses.append(pose.copy())
        last_pose, prev_pose = pose.copy(), pose.copy()
        for t in range(1, len(self.frames)):
            Lp, L = self.frames[t - 1][0], self.frames[t][0]
            nxt, st, _ = self.be.lk_track(Lp, L, px, px)                                       # Frontend::TrackLastFrame
            self.rec("lk_track", nxt, st)
            keep = st & (mp >= 0)
            for i in np.nonzero(keep)[0]:
                if mp[i] not in self.points:                                                    # culled by the back end meanwhile
                    keep[i] = False
            px, mp = nxt[keep].astype(np.float32), mp[keep]
            pred = p7_of(T_of(last_pose) @ np.linalg.inv(T_of(prev_pose)) @ T_of(last_pose))  # constant-velocity prediction (frontend.cpp:90,111)
            p3 = np.stack([self.points[m] for m in mp]) if len(mp) else np.zeros((0, 3))
            pose, outl, ninl = self.be.pose_only(pred, p3, px.astype(np.float64), self.Kt)     # Frontend::EstimateCurrentPose
            self.rec("pose_only", pose, outl, np.array([ninl]))
            px, mp = px[~outl], mp[~outl]                                                       # outlier features lose their map point (:231-241)
            prev_pose, last_pose = last_pose, pose.copy()
            if t % self.kf_every == 0:
                pose, px, mp = self.insert_keyframe(t, pose, px, mp, False)
                last_pose = pose.copy()
            self.poses.append(pose.copy())
        # the loop-closing thread works asynchronously in the reference; here the loops DetectLoop accepted are closed after the last frame
        for cur_i, loop_i in self.detected:
            self.close_loop(cur_i, loop_i)
        return self

    # the loop closer on a candidate DetectLoop accepted: MatchFeatures, ComputeCorrectPose, OptimizeCurrentPose, PoseGraphOptimization
    def close_loop(self, cur_i, loop_i):
        cur, loop = self.kfs[cur_i], self.kfs[loop_i]
        ti, dist = self.be.hamming(loop["desc"], cur["desc"])                                   # query = loop KF, train = current KF (:172)
        pairs = self.api.match_feature_pairs(ti, dist, loop["pyr"], cur["pyr"])
        self.rec("loop_match", ti, dist, pairs)
        p3, p2, = [], []
        for (cf, lf) in pairs:                                                                  # :215-238
            m = loop["mp"][lf]
            if m >= 0 and m in self.points:
                p3.append(self.points[m]); p2.append(cur["px"][cf])
        p3, p2 = np.array(p3, np.float32).reshape(-1, 3), np.array(p2, np.float32).reshape(-1, 2)
        self.n_loop_matches = len(p3)
        pose, inl, n = self.be.pnp(p3, p2, self.Kt)                                             # :262-272
        self.rec("pnp", inl, np.array([n]), pose)
        pose2, outl, ninl = self.be.pose_only(pose, p3.astype(np.float64), p2.astype(np.float64), self.Kt, pre=1)     # OptimizeCurrentPose :339-433
        self.rec("loop_pose", pose2, outl, np.array([ninl]))
        # LoopLocalFusion (:466-507): the active window moves rigidly with the corrected current key-frame, active map points follow the
        # active key-frame that first observes them
        n_kf = len(self.kfs)
        act = list(range(max(0, n_kf - self.window), n_kf))
        aidx = {k: i for i, k in enumerate(act)}
        ids = sorted(self.points)
        first_act = np.array([min((aidx[k] for k in self.obs[m] if k in aidx), default=-1) for m in ids], np.int32)
        pts = np.stack([self.points[m] for m in ids])
        loop_edge = p7_of(T_of(pose2) @ np.linalg.inv(T_of(loop["pose"])))                        # mRelativePoseToLoopKF (:441-446)
        aposes, pts = self.be.local_fusion(np.stack([self.kfs[k]["pose"] for k in act]), aidx[cur_i], pose2, first_act, pts)
        self.rec("local_fusion", aposes, pts)
        for i, k in enumerate(act):
            self.kfs[k]["pose"] = aposes[i].copy()
        for j, m in enumerate(ids):
            self.points[m] = pts[j].copy()
        # PoseGraphOptimization (:537-610): chain edges mRelativePoseToLastKF, one loop edge; active key-frames, the loop key-frame and
        # key-frame 0 are fixed (:557-562)
        poses = np.stack([k["pose"] for k in self.kfs])
        fixed = np.zeros(n_kf, np.uint8); fixed[act] = 1; fixed[loop_i] = 1; fixed[0] = 1
        e0 = list(range(1, n_kf)) + [cur_i]; e1 = list(range(0, n_kf - 1)) + [loop_i]
        meas = [self.kfs[i]["rel"] for i in range(1, n_kf)] + [loop_edge]
        new_poses, chi2, iters = self.be.pgo(poses, fixed, np.array(e0, np.int32), np.array(e1, np.int32), np.stack(meas))
        self.rec("pgo", new_poses, np.array([chi2]), np.array([iters]))
        # map points outside the active map follow the key-frame that first observed them (:612-633)
        first = np.array([-1 if first_act[j] >= 0 else self.first_kf[m] for j, m in enumerate(ids)], np.int32)
        pts2 = self.be.correct_points(poses, new_poses, first, pts)                             # :612-640
        self.rec("correct_points", pts2)
        self.final_poses = new_poses
