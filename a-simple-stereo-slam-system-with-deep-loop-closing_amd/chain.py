"""Frontend / Backend / LoopClosing / Map of the reference as ONE sequential schedule over the operators of the hot path — the host
orchestration of BASELINE configs[0] (`run_kitti_stereo`), written once and run through a back end object: `HipBackend` below (ctypes
over the C ABI of libmyslam_hip.so) is the product; tests/ holds an oracle back end with the same methods as the checker.

What follows the reference line by line (plain Python objects in place of the shared_ptr / weak_ptr graph):
  Frontend::GrabStereoImage / Track / TrackLastFrame / EstimateCurrentPose / StereoInit / DetectFeatures / FindFeaturesInRight /
  BuildInitMap / InsertKeyFrame / TriangulateNewPoints                                   src/frontend.cpp:41-488
     - a key-frame is inserted when the pose-only inlier count falls to numFeatures.trackingGood or below (TRACKING_BAD, :97-120)
     - LK starts from the re-projection with the constant-velocity pose for features that have a live map point (:136-147, :345-352)
  KeyFrame::CreateKF (features shared with the frame)                                     src/keyframe.cpp:6-44
  Map::InsertKeyFrame / RemoveOldActiveKeyframe / RemoveOldActiveMapPoints / RemoveAllOutlierMapPoints / RemoveMapPoint
                                                                                          src/map.cpp:15-167
  MapPoint::Add/Remove(Active)Observation                                                 src/mappoint.cpp:21-57
  Backend::ProcessNewKeyFrame / OptimizeActiveMap                                         src/backend.cpp:105-266
  LoopClosing::InsertNewKeyFrame / LoopClosingRun / ProcessNewKF / DetectLoop / MatchFeatures / ComputeCorrectPose /
  OptimizeCurrentPose / LoopCorrect / LoopLocalFusion / PoseGraphOptimization / AddToDatabase
                                                                                          src/loopclosing.cpp:51-687
  System::GetCamera (both cameras from the Camera.right.* keys, f32 values)               src/system.cpp:101-146

What a sequential program has to DECIDE (the reference runs three threads whose interleaving is a race):
  * a new key-frame is handed to the back end at once: Map::InsertKeyFrame, LoopClosing::InsertNewKeyFrame, then OptimizeActiveMap
    (the back-end thread's own order, backend.cpp:82-100), then the loop closer's turn for that key-frame (ProcessNewKF … AddToDatabase)
    — all before the next frame is tracked;
  * DeepLCD blurs the key-frame's image IN PLACE (deeplcd.cpp:46) and cv::Mat copies share pixels: KeyFrame::mImageLeft IS the frame's
    mLeftImg, which the next TrackLastFrame reads as the previous image.  With the loop closer's turn taken before the next frame (above)
    the tracker sees the blurred image (`lcd_blur_reaches_tracker=True`, the default; False = the frontend wins the race);
  * std::unordered_map iteration order (Map::RemoveOldActiveKeyframe's max / min search, the edge order of the pose graph) is taken as
    ascending id.
Two deliberate deviations from the reference's text (host/myslam_system.hpp: the same): LoopLocalFusion skips a matched pair whose two
features already share one map point (loopclosing.cpp:516-527 would re-add its observations to itself and RemoveMapPoint() it); a live
active map point without any observation is an error (backend.cpp:175 would take front() of an empty list), an outlier one is skipped
first as backend.cpp:163 does.
`kf_every > 0` replaces the inlier-count rule by "every n-th frame" (the schedule of rounds 2-3, kept as an option)."""
import numpy as np

CHI2_TH = 5.991
INITING, TRACKING_GOOD, TRACKING_BAD, LOST = range(4)

# the reference's config/stereo/gray/KITTI00-02.yaml values (the YAML given on the command line overrides them)
DEFAULT_CONFIG = {"numFeatures.initGood": 100, "numFeatures.trackingGood": 50, "numFeatures.trackingBad": 10,
                  "ORBextractor.nInitFeatures": 300, "ORBextractor.nNewFeatures": 100, "ORBextractor.scaleFactor": 1.2,
                  "ORBextractor.nLevels": 8, "ORBextractor.iniThFAST": 20, "ORBextractor.minThFAST": 7, "Map.activeMap.size": 7,
                  "LCD.similarityScoreThreshold.high": 0.94, "LCD.similarityScoreThreshold.low": 0.92, "LCD.nDatabaseMinSize": 50}


# ---- SE3 as (qx qy qz qw tx ty tz), Tcw --------------------------------------------------------------------------------------------
# Every sum below is written out in the order host/myslam_system.hpp (the compiled twin of this file) uses — no BLAS call, no reduction
# whose order is numpy's choice: the pose-only optimiser stops on an iteration budget, so a last-bit difference in its start is a 1e-9
# difference in its result and a different LK start a few frames later.  With the order pinned the two hosts agree bit for bit
# (tests/test_gpu_runner.py).
def mm(A, B):
    """A @ B for the small matrices of this file: sum over k in ascending order, plain multiplies and adds"""
    C = A[:, 0:1] * B[0:1, :]
    for k in range(1, A.shape[1]):
        C = C + A[:, k:k + 1] * B[k:k + 1, :]
    return C


def mv(R, v):
    """R @ v, columns in ascending order"""
    r = R[:, 0] * v[0]
    for k in range(1, R.shape[1]):
        r = r + R[:, k] * v[k]
    return r


def q_to_R(q):
    n = np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    x, y, z, w = q[0] / n, q[1] / n, q[2] / n, q[3] / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_q(R):
    t = R[0, 0] + R[1, 1] + R[2, 2]
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


def T_inv(T):
    Ti = np.eye(4); Ti[:3, :3] = T[:3, :3].T; Ti[:3, 3] = mv(-T[:3, :3].T, T[:3, 3])
    return Ti


def se3_log_norm(T):
    """|Sophus::SE3d::log()| (the 6-vector (upsilon, omega)): Map::RemoveOldActiveKeyframe's distance (map.cpp:88), the `error > 1` test of
    ComputeCorrectPose (loopclosing.cpp:283)"""
    R, t = T[:3, :3], T[:3, 3]
    c = min(1.0, max(-1.0, (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) / 2.0))
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        om = 0.5 * w
    elif np.pi - th < 1e-6:                      # near pi: axis from the diagonal
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / max(ax[k], 1e-300)
        ax = ax / np.sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2])
        if ax[0] * w[0] + ax[1] * w[1] + ax[2] * w[2] < 0:
            ax = -ax
        om = th * ax
    else:
        om = th / (2.0 * np.sin(th)) * w
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        coef = 1.0 / 12.0
    else:
        h = 0.5 * th
        coef = (1.0 - th * np.cos(h) / (2.0 * np.sin(h))) / (th * th)
    Vinv = np.eye(3) - 0.5 * Om + coef * mm(Om, Om)
    u = mv(Vinv, t)
    return float(np.sqrt((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + (om[0] * om[0] + om[1] * om[1] + om[2] * om[2])))


IDENT = np.array([0, 0, 0, 1, 0, 0, 0], float)


def camera_from_config(cfg):
    """System::GetCamera (system.cpp:101-146): BOTH cameras take the Camera.right.* keys (reference quirk 8), every value passes through a
    float, baseline = bf / fx in float"""
    f = lambda k: np.float32(float(cfg[k]))
    fx, fy, cx, cy, bf = f("Camera.right.fx"), f("Camera.right.fy"), f("Camera.right.cx"), f("Camera.right.cy"), f("Camera.bf")
    return {"fx": float(fx), "fy": float(fy), "cx": float(cx), "cy": float(cy), "bf": float(bf), "baseline": float(np.float32(bf / fx))}


# ---- the map's objects ------------------------------------------------------------------------------------------------------------
class Feature:                              # include/myslam/feature.h:14-35
    __slots__ = ("x", "y", "mp", "kf", "outlier")

    def __init__(self, x, y):
        self.x, self.y = np.float32(x), np.float32(y)     # mkpPosition.pt (cv::Point2f)
        self.mp = None                      # mpMapPoint (weak_ptr: `live()` is lock() != nullptr)
        self.kf = None                      # mpKF
        self.outlier = False                # mbIsOutlier

    def live(self):
        return self.mp if (self.mp is not None and self.mp.alive) else None


class MapPoint:                             # include/myslam/mappoint.h:13-61
    __slots__ = ("id", "pos", "obs", "active_obs", "outlier", "alive")

    def __init__(self, mp_id, pos):
        self.id, self.pos = mp_id, np.array(pos, float)
        self.obs, self.active_obs = [], []
        self.outlier = False
        self.alive = True                   # False once the Map dropped its shared_ptr (the only owner): every weak_ptr expires


class KeyFrame:                             # include/myslam/keyframe.h:14-60
    def __init__(self, kf_id, frame):
        self.id, self.frame_id, self.ts = kf_id, frame.id, frame.ts
        self.img = frame.L                  # cv::Mat copy = the SAME pixels as the frame's mLeftImg
        self.feats = frame.feats            # the shared Feature objects
        self.pose = IDENT.copy()
        self.last_kf, self.rel_to_last = None, None
        self.loop_kf, self.rel_to_loop = None, None
        self.descr = self.pyr = self.desc = None


class Frame:                                # include/myslam/frame.h:12-49
    def __init__(self, frame_id, ts, L, R):
        self.id, self.ts, self.L, self.R = frame_id, ts, L, R
        self.feats, self.right = [], []     # mvpFeaturesLeft; mvpFeaturesRight as (x, y) or None
        self.rel = np.eye(4)                # RelativePose(): pose relative to the reference key-frame


# ---- the product's back end: every operator through the C ABI ----------------------------------------------------------------------
class HipBackend:
    name = "hip"

    def __init__(self, api, weights=None, cfg=None, lcd=None):
        c = dict(DEFAULT_CONFIG, **(cfg or {}))
        mk = lambda n: api.ORBextractor(int(c[n]), float(c["ORBextractor.scaleFactor"]), int(c["ORBextractor.nLevels"]),
                                        int(c["ORBextractor.iniThFAST"]), int(c["ORBextractor.minThFAST"]))
        self.api = api
        self.det_init = mk("ORBextractor.nInitFeatures")              # Frontend::_mpORBextractorInit (frontend.cpp:34)
        self.orb = mk("ORBextractor.nNewFeatures")                    # System::_mpORBextractor, shared by Frontend and LoopClosing (system.cpp:31,54,66)
        self.lk = api.LKTracker()
        self.lcd = lcd if lcd is not None else api.DeepLCD(weights)
        self.db = api.LoopDatabase(64)                                # grows like the std::map

    def detect(self, img, mask, init):
        return (self.det_init if init else self.orb).Detect(img, mask)

    def lk_track(self, a, b, p0, p1):
        return self.lk.track(a, b, p0, p1)

    def triangulate(self, xl, yl, xr, yr, K):
        return self.api.triangulate_stereo(xl, yl, xr, yr, K["fx"], K["fy"], K["cx"], K["cy"], K["baseline"])

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

    def db_size(self):
        return len(self.db)

    def db_query(self, d, cur, thr_low):
        return self.db.query(d, cur, thr_low)

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


# ---- the system ----------------------------------------------------------------------------------------------------------------------
class Chain:
    def __init__(self, be, api, K, frames, cfg=None, kf_every=0, lcd_blur_reaches_tracker=True, timestamps=None, log=True, correct_threshold=1.0):
        """be: back end (HipBackend or the tests' oracle back end); api: the product's host helpers (pyramid expansion, feature pairing,
        window flattening: plain host functions of the library); K: camera_from_config() or a dict with fx fy cx cy bf [baseline];
        frames: a sequence of (left, right) uint8 images or a callable t -> (left, right).  correct_threshold: the reference corrects a
        confirmed loop only when |log(T_cur * T_corrected^-1)| > 1 (loopclosing.cpp:283-288); tests lower it to reach LoopLocalFusion and
        the pose graph on a short track."""
        self.be, self.api = be, api
        self.K = dict(K)
        self.K.setdefault("baseline", self.K["bf"] / self.K["fx"])
        self.Kt = (self.K["fx"], self.K["fy"], self.K["cx"], self.K["cy"])
        self.frames = frames
        self.ts = timestamps
        c = dict(DEFAULT_CONFIG, **(cfg or {}))
        self.n_init_good, self.n_good, self.n_bad = int(c["numFeatures.initGood"]), int(c["numFeatures.trackingGood"]), int(c["numFeatures.trackingBad"])
        self.window = int(c["Map.activeMap.size"])
        self.nlevels = int(c["ORBextractor.nLevels"])
        self.thr_high, self.thr_low = float(c["LCD.similarityScoreThreshold.high"]), float(c["LCD.similarityScoreThreshold.low"])
        self.lcd_min_db = int(c["LCD.nDatabaseMinSize"])
        self.kf_every = kf_every
        self.correct_threshold = correct_threshold
        self.blur_shared = lcd_blur_reaches_tracker
        self.keep_log = log
        self.log = []
        # Frontend
        self.status = INITING
        self.cur = self.last = self.ref_kf = None
        self.rel_motion = np.eye(4)                 # _mseRelativeMotion
        self.next_frame_id = self.next_kf_id = self.next_mp_id = 0       # the static id factories (frame.cpp:10, keyframe.cpp:14, mappoint.cpp:8)
        # Map
        self.all_kfs, self.active_kfs = {}, {}      # id -> KeyFrame
        self.all_mps, self.active_mps = {}, {}      # id -> MapPoint
        self.outlier_mps = []                       # _mlistOutlierMapPoints
        self.map_cur_kf = None
        # LoopClosing
        self.db = {}                                # _mvDatabase: id -> KeyFrame (the descriptors live in the back end's device matrix too)
        self.last_closed_kf = None
        self.loops = []                             # (current KF, loop KF) pairs that passed every verification
        self.need_correct = False
        self.poses = []                             # Tcw of every tracked frame (as estimated when the frame was processed)
        self.kf_frames = []
        self.stats = {"lk_init_from_projection": 0, "lk_init_from_last": 0}

    def rec(self, tag, *arrays):
        if self.keep_log:
            self.log.append((tag, [np.array(a) for a in arrays]))

    def frame_images(self, t):
        return self.frames(t) if callable(self.frames) else self.frames[t]

    # ---------------------------------------------------------------- cameras (camera.cpp:7-45; left extrinsics = identity, right = (-baseline, 0, 0))
    def world2pixel(self, pw, Tcw, right=False):
        pc = mv(Tcw[:3, :3], pw) + Tcw[:3, 3]
        if right:
            pc = pc + np.array([-self.K["baseline"], 0.0, 0.0])
        return np.array([self.K["fx"] * pc[0] / pc[2] + self.K["cx"], self.K["fy"] * pc[1] / pc[2] + self.K["cy"]])

    # ---------------------------------------------------------------- Frontend
    def grab(self, t, ts=None):
        """Frontend::GrabStereoImage (frontend.cpp:41-80); False = the tracker is LOST (the reference quits)"""
        L, R = self.frame_images(t)
        self.cur = Frame(self.next_frame_id, float(ts if ts is not None else (self.ts[t] if self.ts is not None else t)), L, R)
        self.next_frame_id += 1
        if self.status == INITING:
            self.stereo_init()
        elif self.status in (TRACKING_GOOD, TRACKING_BAD):
            self.track()
        else:
            return False
        if self.ref_kf is not None:
            self.poses.append(p7_of(mm(self.cur.rel, T_of(self.ref_kf.pose))))
        else:
            self.poses.append(IDENT.copy())
        self.last = self.cur
        return True

    def run(self, n=None):
        n = n if n is not None else len(self.frames)
        for t in range(n):
            if not self.grab(t):
                break
        return self

    def stereo_init(self):                          # frontend.cpp:281-295
        self.detect_features()
        if self.find_features_in_right() < self.n_init_good:
            return False
        self.build_init_map()
        self.status = TRACKING_GOOD
        return True

    def track(self):                                # frontend.cpp:85-124
        cur, last = self.cur, self.last
        cur.rel = mm(self.rel_motion, last.rel)
        self.track_last_frame()
        n_inl = self.estimate_current_pose()
        if n_inl > self.n_good:
            self.status = TRACKING_GOOD
        elif n_inl > self.n_bad:
            self.status = TRACKING_BAD
        else:
            self.status = LOST
        self.rel_motion = mm(cur.rel, T_inv(last.rel))
        insert = self.status == TRACKING_BAD if self.kf_every <= 0 else (self.status != LOST and cur.id % self.kf_every == 0)
        if insert:
            self.detect_features()
            self.find_features_in_right()
            self.triangulate_new_points()
            self.insert_keyframe()

    def track_last_frame(self):                     # frontend.cpp:129-172
        cur, last = self.cur, self.last
        Tcw = mm(cur.rel, T_of(self.ref_kf.pose))
        p0 = np.zeros((len(last.feats), 2), np.float32); p1 = np.zeros((len(last.feats), 2), np.float32)
        for i, f in enumerate(last.feats):
            p0[i] = (f.x, f.y)
            mp = f.live()
            if mp is not None and not mp.outlier:   # initial flow = the re-projection with the predicted pose
                p1[i] = self.world2pixel(mp.pos, Tcw).astype(np.float32)
                self.stats["lk_init_from_projection"] += 1
            else:
                p1[i] = (f.x, f.y)
                self.stats["lk_init_from_last"] += 1
        nxt, st, _ = self.be.lk_track(last.L, cur.L, p0, p1)
        self.rec("lk_track", nxt, st, p1)
        for i, f in enumerate(last.feats):
            if st[i] and f.live() is not None:      # status && !mpMapPoint.expired()
                g = Feature(nxt[i, 0], nxt[i, 1])
                g.mp = f.mp
                cur.feats.append(g)

    def estimate_current_pose(self):                # frontend.cpp:176-276
        cur = self.cur
        feats = [f for f in cur.feats if f.live() is not None and not f.mp.outlier]
        p3 = np.array([f.mp.pos for f in feats], float).reshape(-1, 3)
        obs = np.array([[f.x, f.y] for f in feats], np.float64).reshape(-1, 2)
        pose0 = p7_of(mm(cur.rel, T_of(self.ref_kf.pose)))
        pose, outl, n_inl = self.be.pose_only(pose0, p3, obs, self.Kt)
        self.rec("pose_only", pose, outl, np.array([n_inl]))
        cur.rel = mm(T_of(pose), T_inv(T_of(self.ref_kf.pose)))
        for f, o in zip(feats, outl):
            if o:
                mp = f.live()
                if mp is not None and cur.id - self.ref_kf.frame_id <= 2:      # a map point that fails right after its creation leaves the map
                    mp.outlier = True
                    self.outlier_mps.append(mp.id)
                f.mp = None
                f.outlier = False
        return n_inl

    def detect_features(self):                      # frontend.cpp:300-330
        cur = self.cur
        h, w = cur.L.shape
        mask = np.full((h, w), 255, np.uint8)
        for f in cur.feats:                         # cv::rectangle(pt - (20, 20), pt + (20, 20), 0, CV_FILLED): Point2f -> Point rounds (cvRound)
            x0, x1 = int(np.rint(np.float32(f.x - np.float32(20)))), int(np.rint(np.float32(f.x + np.float32(20))))
            y0, y1 = int(np.rint(np.float32(f.y - np.float32(20)))), int(np.rint(np.float32(f.y + np.float32(20))))
            mask[max(y0, 0):max(y1 + 1, 0), max(x0, 0):max(x1 + 1, 0)] = 0
        new = self.be.detect(cur.L, mask, self.status == INITING)
        self.rec("detect", new)
        for x, y in zip(new["x"], new["y"]):
            cur.feats.append(Feature(x, y))
        return len(new)

    def find_features_in_right(self):               # frontend.cpp:335-379
        cur = self.cur
        Tcw = mm(cur.rel, T_of(self.ref_kf.pose)) if self.ref_kf is not None else np.eye(4)
        n = len(cur.feats)
        p0 = np.zeros((n, 2), np.float32); p1 = np.zeros((n, 2), np.float32)
        for i, f in enumerate(cur.feats):
            p0[i] = (f.x, f.y)
            mp = f.live()
            if mp is not None and not mp.outlier:
                p1[i] = self.world2pixel(mp.pos, Tcw, right=True).astype(np.float32)
            else:
                p1[i] = (f.x, f.y)
        nxt, st, _ = self.be.lk_track(cur.L, cur.R, p0, p1)
        self.rec("lk_right", nxt, st, p1)
        cur.right = [(nxt[i, 0], nxt[i, 1]) if st[i] else None for i in range(n)]
        return int(np.count_nonzero(st))

    def _triangulate(self, idx):
        cur = self.cur
        xl = np.array([cur.feats[i].x for i in idx], np.float32); yl = np.array([cur.feats[i].y for i in idx], np.float32)
        xr = np.array([cur.right[i][0] for i in idx], np.float32); yr = np.array([cur.right[i][1] for i in idx], np.float32)
        xyz, ok = self.be.triangulate(xl, yl, xr, yr, self.K)           # triangulation() && z > 0 (algorithm.h:16-33, frontend.cpp:401,470)
        self.rec("triangulate", ok, xyz[ok])
        return xyz, ok

    def _new_map_point(self, pos, feat):
        mp = MapPoint(self.next_mp_id, pos)
        self.next_mp_id += 1
        feat.mp = mp
        self.all_mps[mp.id] = mp                    # Map::InsertMapPoint
        return mp

    def build_init_map(self):                       # frontend.cpp:385-417
        cur = self.cur
        idx = [i for i in range(len(cur.feats)) if cur.right[i] is not None]
        if idx:
            xyz, ok = self._triangulate(idx)
            for j, i in enumerate(idx):
                if ok[j]:
                    self._new_map_point(xyz[j], cur.feats[i])
        self.insert_keyframe()

    def triangulate_new_points(self):               # frontend.cpp:451-488
        cur = self.cur
        Twc = T_inv(mm(cur.rel, T_of(self.ref_kf.pose)))
        idx = [i for i, f in enumerate(cur.feats) if f.live() is None and cur.right[i] is not None]      # !expired() -> skip
        if idx:
            xyz, ok = self._triangulate(idx)
            for j, i in enumerate(idx):
                if ok[j]:
                    self._new_map_point(mv(Twc[:3, :3], xyz[j]) + Twc[:3, 3], cur.feats[i])

    def insert_keyframe(self):                      # frontend.cpp:424-447 + KeyFrame::CreateKF (keyframe.cpp:29-44)
        cur = self.cur
        kf = KeyFrame(self.next_kf_id, cur)
        self.next_kf_id += 1
        for f in kf.feats:
            f.kf = kf
            mp = f.live()
            if mp is not None:
                mp.obs.append(f)                    # MapPoint::AddObservation
        if self.status == INITING:
            kf.pose = IDENT.copy()
        else:
            kf.pose = p7_of(mm(cur.rel, T_of(self.ref_kf.pose)))
            kf.last_kf = self.ref_kf
            kf.rel_to_last = p7_of(cur.rel)
        self.ref_kf = kf
        cur.rel = np.eye(4)
        self.kf_frames.append(cur.id)
        self.backend_new_keyframe(kf)

    # ---------------------------------------------------------------- Backend + Map
    def backend_new_keyframe(self, kf):             # Backend::ProcessNewKeyFrame + the optimisation it triggers (backend.cpp:82-121)
        self.map_insert_keyframe(kf)
        queued = self.lc_insert_new_keyframe(kf)
        self.optimize_active_map()
        if queued:
            self.loop_closing_turn(kf)

    def map_insert_keyframe(self, kf):              # map.cpp:15-45
        self.map_cur_kf = kf
        self.all_kfs[kf.id] = kf; self.active_kfs[kf.id] = kf
        for f in kf.feats:
            mp = f.live()
            if mp is not None:
                mp.active_obs.append(f)             # AddActiveObservation
                self.active_mps[mp.id] = mp
        if len(self.active_kfs) > self.window:
            self.remove_old_active_keyframe()
            self.remove_old_active_map_points()

    def remove_old_active_keyframe(self):           # map.cpp:75-120
        cur = self.map_cur_kf
        Twc = T_inv(T_of(cur.pose))
        max_dis, min_dis, max_id, min_id = 0.0, 9999.0, 0, 0
        for kid in sorted(self.active_kfs):
            kf = self.active_kfs[kid]
            if kf is cur:
                continue
            dis = se3_log_norm(mm(T_of(kf.pose), Twc))
            if dis > max_dis:
                max_dis, max_id = dis, kid
            elif dis < min_dis:
                min_dis, min_id = dis, kid
        gone = self.active_kfs[min_id] if min_dis < 0.2 else self.active_kfs[max_id]
        del self.active_kfs[gone.id]
        for f in gone.feats:
            mp = f.live()
            if mp is not None:
                self._remove_active_obs(mp, f)

    @staticmethod
    def _remove_active_obs(mp, f):                  # mappoint.cpp:36-45
        for i, g in enumerate(mp.active_obs):
            if g is f:
                del mp.active_obs[i]
                break

    @staticmethod
    def _remove_obs(mp, f):                         # mappoint.cpp:48-58
        for i, g in enumerate(mp.obs):
            if g is f:
                del mp.obs[i]
                f.mp = None
                break

    def remove_old_active_map_points(self):         # map.cpp:124-137
        for mid in [m for m, mp in self.active_mps.items() if not mp.active_obs]:
            del self.active_mps[mid]

    def remove_all_outlier_map_points(self):        # map.cpp:163-171
        for mid in self.outlier_mps:
            mp = self.all_mps.pop(mid, None)
            self.active_mps.pop(mid, None)
            if mp is not None:
                mp.alive = False
        self.outlier_mps = []

    def remove_map_point(self, mp):                 # map.cpp:141-149
        self.all_mps.pop(mp.id, None); self.active_mps.pop(mp.id, None)
        mp.alive = False

    def optimize_active_map(self):                  # backend.cpp:126-266
        kfs = [self.active_kfs[k] for k in sorted(self.active_kfs)]
        mps = [self.active_mps[m] for m in sorted(self.active_mps)]
        if not kfs or not mps:
            return
        rows = [(mp, f) for mp in mps for f in mp.active_obs]
        # GetObservations().front()->mpKF (:175); an outlier point is dropped by the flatten rules (:163) before :175 could touch it
        first = [(mp.obs[0].kf.id if mp.obs else mp.active_obs[0].kf.id if mp.active_obs else 0) for mp in mps]
        if any(not m.outlier and not m.obs and not m.active_obs for m in mps):
            raise RuntimeError("active map point without observations")
        # the graph-build rules of :139-206 live behind the C ABI (myslam_ba_flatten_window): skip outlier map points / features, fix the
        # landmarks whose first observer left the window, vertices by id, edges grouped by landmark
        fl = self.api.ba_flatten_window([k.id for k in kfs], [m.id for m in mps], [1 if m.outlier else 0 for m in mps], first,
                                        [mp.id for mp, _ in rows], [f.kf.id for _, f in rows],
                                        np.array([[f.x, f.y] for _, f in rows], np.float32).reshape(-1, 2), [1 if f.outlier else 0 for _, f in rows])
        if len(fl["edge_src"]) == 0:
            return
        win = [kfs[i] for i in fl["pose_src"]]; pts_mp = [mps[i] for i in fl["pt_src"]]
        poses = np.stack([k.pose for k in win]); pts = np.stack([m.pos for m in pts_mp])
        p2, x2, chi, out, rounds, nout = self.be.ba(poses, pts, fl["edge_pose"], fl["edge_pt"], fl["edge_obs"], fl["fixed"], self.Kt)
        self.rec("ba", p2, x2, out, np.array([rounds, nout]), chi)
        for e, r in enumerate(fl["edge_src"]):      # :234-250
            mp, f = rows[r]
            if out[e]:
                f.outlier = True
                self._remove_active_obs(mp, f)
                self._remove_obs(mp, f)
                if not mp.obs:
                    mp.outlier = True
                    self.outlier_mps.append(mp.id)
                f.mp = None
            else:
                f.outlier = False
        for i, k in enumerate(win):                 # :252-266 (under the map mutex)
            k.pose = p2[i].copy()
        for j, m in enumerate(pts_mp):
            m.pos = x2[j].copy()
        self.remove_all_outlier_map_points()
        self.remove_old_active_map_points()

    # ---------------------------------------------------------------- LoopClosing
    def lc_insert_new_keyframe(self, kf):           # loopclosing.cpp:671-681: the 5 key-frames after a closed loop are skipped
        if self.last_closed_kf is None or kf.id - self.last_closed_kf.id > 5:
            return True
        kf.img = None
        return False

    def loop_closing_turn(self, kf):                # one pass of LoopClosingRun's body (loopclosing.cpp:51-77)
        self.process_new_kf(kf)
        confirmed = False
        if len(self.db) > self.lcd_min_db:
            loop = self.detect_loop(kf)
            if loop is not None:
                pairs = self.match_features(kf, loop)
                if pairs is not None:
                    confirmed = self.compute_correct_pose(kf, loop, pairs)
                    if confirmed:
                        self.loop_correct(kf, loop)
        if not confirmed:
            self.db[kf.id] = kf                     # AddToDatabase (:651-659)
            self.be.db_add(kf.id, kf.descr)

    def process_new_kf(self, kf):                   # loopclosing.cpp:83-121
        d, blurred = self.be.lcd_descr(kf.img)      # blurs the key-frame's image in place (reference quirk 7) ...
        if self.blur_shared and self.cur is not None and self.cur.L is kf.img:
            self.cur.L = blurred                    # ... and those pixels are the frame's mLeftImg (see the module header)
        kf.img = blurred
        feats = np.zeros(len(kf.feats), self.api.KP_DTYPE)
        feats["x"] = [f.x for f in kf.feats]; feats["y"] = [f.y for f in kf.feats]
        feats["size"], feats["angle"], feats["octave"], feats["class_id"] = 7, -1, 0, -1
        pyr = self.api.expand_pyramid_keypoints(feats, self.nlevels)
        kf.pyr = self.be.screen(kf.img, pyr)
        kf.desc = self.be.calc_desc(kf.img, kf.pyr)
        kf.descr = d
        kf.img = None                               # LoopClosing.bShowResult 0: mImageLeft.release() (:117-120)
        self.rec("lcd", d, kf.pyr, kf.desc)

    def detect_loop(self, kf):                      # loopclosing.cpp:124-161
        best, mx, cnt = self.be.db_query(kf.descr, kf.id, self.thr_low)
        self.rec("detect_loop", np.array([best, cnt]), np.array([mx]))
        if mx < self.thr_high or cnt > 3:
            return None
        return self.db[int(best)]

    def match_features(self, kf, loop):             # loopclosing.cpp:166-203
        ti, dist = self.be.hamming(loop.desc, kf.desc)                 # query = loop key-frame, train = current key-frame (:172)
        pairs = self.api.match_feature_pairs(ti, dist, loop.pyr, kf.pyr)
        self.rec("loop_match", ti, dist, pairs)
        return pairs if len(pairs) >= 10 else None

    def compute_correct_pose(self, kf, loop, pairs):            # loopclosing.cpp:208-335
        valid = [(int(cf), int(lf)) for cf, lf in pairs if loop.feats[lf].live() is not None]     # matches without a map point leave the set
        if len(valid) < 10:
            return False
        p3 = np.array([loop.feats[lf].mp.pos for _, lf in valid], np.float32).reshape(-1, 3)     # cv::Point3f
        p2 = np.array([[kf.feats[cf].x, kf.feats[cf].y] for cf, _ in valid], np.float32).reshape(-1, 2)
        try:
            pose, inl, n = self.be.pnp(p3, p2, self.Kt)
        except RuntimeError:                        # the reference's try / catch around solvePnPRansac (:262-270): "no model" is how both back
            return False                            # ends report OpenCV's `false`; anything else (a checker's assertion) is not swallowed
        self.rec("pnp", inl, np.array([n]), pose)
        # OptimizeCurrentPose (:339-433): the map points in double, the pixels through toVec2
        P3 = np.array([loop.feats[lf].mp.pos for _, lf in valid], float).reshape(-1, 3)
        pose2, outl, n_inl = self.be.pose_only(pose, P3, p2.astype(np.float64), self.Kt, pre=1)
        self.rec("loop_pose", pose2, outl, np.array([n_inl]))
        valid = [v for v, o in zip(valid, outl) if not o]
        if len(valid) < 10:
            return False
        self.need_correct = se3_log_norm(mm(T_of(kf.pose), T_inv(T_of(pose2)))) > self.correct_threshold
        kf.loop_kf = loop
        kf.rel_to_loop = p7_of(mm(T_of(pose2), T_inv(T_of(loop.pose))))
        self.last_closed_kf = kf
        self.loops.append((kf, loop))
        self._corrected, self._valid = pose2, valid
        return True

    def loop_correct(self, kf, loop):               # loopclosing.cpp:438-462
        if not self.need_correct:
            return
        self.loop_local_fusion(kf, loop)
        self.pose_graph_optimization(loop)

    def loop_local_fusion(self, kf, loop):          # loopclosing.cpp:466-533
        act = [self.active_kfs[k] for k in sorted(self.active_kfs)]
        slot = {k.id: i for i, k in enumerate(act)}
        assert kf.id in slot, "the current key-frame left the active window before its loop was closed"
        mps = [self.active_mps[m] for m in sorted(self.active_mps)]
        first = np.array([slot.get(m.active_obs[0].kf.id, -1) if m.active_obs else -1 for m in mps], np.int32)
        pts = np.stack([m.pos for m in mps]) if mps else np.zeros((0, 3))
        aposes, pts = self.be.local_fusion(np.stack([k.pose for k in act]), slot[kf.id], self._corrected, first, pts)
        self.rec("local_fusion", aposes, pts)
        for i, k in enumerate(act):
            k.pose = aposes[i].copy()
        for j, m in enumerate(mps):
            m.pos = pts[j].copy()
        for cf, lf in self._valid:                  # the current key-frame's map points are replaced by the loop key-frame's (:510-532)
            loop_mp = loop.feats[lf].live()
            if loop_mp is None:
                continue
            cur_mp = kf.feats[cf].live()
            if cur_mp is not None:
                if cur_mp is loop_mp:
                    continue
                for g in list(cur_mp.obs):
                    loop_mp.obs.append(g)
                    g.mp = loop_mp
                self.remove_map_point(cur_mp)
            else:
                kf.feats[cf].mp = loop_mp

    def pose_graph_optimization(self, loop):        # loopclosing.cpp:537-646
        kfs = [self.all_kfs[k] for k in sorted(self.all_kfs)]
        idx = {k.id: i for i, k in enumerate(kfs)}
        poses = np.stack([k.pose for k in kfs])
        fixed = np.array([1 if (k.id in self.active_kfs or k.id == loop.id or k.id == 0) else 0 for k in kfs], np.uint8)
        e0, e1, meas = [], [], []
        for k in kfs:
            if k.last_kf is not None:
                e0.append(idx[k.id]); e1.append(idx[k.last_kf.id]); meas.append(k.rel_to_last)
            if k.loop_kf is not None:
                e0.append(idx[k.id]); e1.append(idx[k.loop_kf.id]); meas.append(k.rel_to_loop)
        new_poses, chi2, iters = self.be.pgo(poses, fixed, np.array(e0, np.int32), np.array(e1, np.int32), np.stack(meas))
        self.rec("pgo", new_poses, np.array([chi2]), np.array([iters]))
        mps = [m for mid, m in sorted(self.all_mps.items()) if mid not in self.active_mps and m.obs]
        if mps:                                     # map points outside the active map follow the key-frame that first observed them (:612-633)
            first = np.array([idx.get(m.obs[0].kf.id, -1) for m in mps], np.int32)
            pts2 = self.be.correct_points(poses, new_poses, first, np.stack([m.pos for m in mps]))
            self.rec("correct_points", pts2)
            for j, m in enumerate(mps):
                m.pos = pts2[j].copy()
        for i, k in enumerate(kfs):
            k.pose = new_poses[i].copy()

    # ---------------------------------------------------------------- System::SaveTrajectory / SaveLoopEdges (system.cpp:153-224)
    def save(self, out_dir):
        import os
        os.makedirs(out_dir, exist_ok=True)
        kfs = [self.all_kfs[k] for k in sorted(self.all_kfs)]
        self.api.save_trajectory(os.path.join(out_dir, "trajectory.txt"), np.array([k.id for k in kfs], np.uint64),
                                 np.array([k.ts for k in kfs]), np.stack([k.pose for k in kfs]))
        lp = [k for k in kfs if k.loop_kf is not None]
        z7 = np.zeros((0, 7))
        self.api.save_loop_edges(os.path.join(out_dir, "loopEdges.txt"), np.array([k.id for k in lp], np.uint64), np.array([k.ts for k in lp]),
                                 np.stack([k.pose for k in lp]) if lp else z7, np.array([k.loop_kf.id for k in lp], np.uint64),
                                 np.array([k.loop_kf.ts for k in lp]), np.stack([k.loop_kf.pose for k in lp]) if lp else z7)
