"""Deterministic synthetic inputs for the hot path.

numpy only; used by tests/ and bench.py on both boxes (no dataset, no network).  Seeds are fixed so
the CPU oracle and the HIP path always see identical bytes.

  * stereo stream  S(stream_id): 1241x376 u8 left/right pairs, scene shifted by (2t, 0) px per frame
  * CALC weights   N(0, 1/fan_in), seed 0xCA1C (the real caffemodel is not available)
  * loop database  |N(0,1)|^1064 rows, L2-normalised, seed 0xDB
  * BA window      10 key-frames x 300 landmarks, K = KITTI00, seed 0xBA
"""
import os

import numpy as np

KITTI00 = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448)   # config/stereo/gray/KITTI00-02.yaml
IMG_H, IMG_W = 376, 1241


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-1.0, 1.0, size=(gh, gw))
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(int); x0 = xs.astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_scene(stream_id=0, h=IMG_H, w=IMG_W, n_rect=6000):
    """Base scene (float64, un-clamped) for a stream; wraps horizontally."""
    rng = _rng(0x5EED0000 + stream_id)
    img = np.full((h, w), 128.0)
    img += 48.0 * (_value_noise(rng, h, w, 64) + 0.5 * _value_noise(rng, h, w, 16) + 0.25 * _value_noise(rng, h, w, 4)) / 1.75
    # 6000 axis-aligned rectangles via a 2-D difference array
    x0 = rng.integers(0, w, n_rect); y0 = rng.integers(0, h, n_rect)
    sw = rng.integers(3, 25, n_rect); sh = rng.integers(3, 25, n_rect)
    val = rng.uniform(-90.0, 90.0, n_rect)
    x1 = np.minimum(x0 + sw, w); y1 = np.minimum(y0 + sh, h)
    diff = np.zeros((h + 1, w + 1))
    np.add.at(diff, (y0, x0), val); np.add.at(diff, (y0, x1), -val)
    np.add.at(diff, (y1, x0), -val); np.add.at(diff, (y1, x1), val)
    img += np.cumsum(np.cumsum(diff, axis=0), axis=1)[:h, :w]
    return img


def _depth_map(stream_id, h, w):
    rng = _rng(0xDE97 + stream_id)
    nplanes = 8
    z = rng.uniform(5.0, 50.0, nplanes + 1)
    xs = np.linspace(0, nplanes, w)
    i = np.minimum(xs.astype(int), nplanes - 1)
    f = xs - i
    zrow = z[i] * (1 - f) + z[i + 1] * f
    tilt = np.linspace(1.15, 0.85, h)[:, None]          # nearer towards the bottom of the image
    return zrow[None, :] * tilt


def stereo_pair(stream_id=0, t=0, h=IMG_H, w=IMG_W, scene=None, bf=KITTI00["bf"]):
    """(left, right) uint8 images of frame t of stream `stream_id`."""
    if scene is None:
        scene = make_scene(stream_id, h, w)
    rng = _rng((0x5EED0000 + stream_id) * 1000003 + t)
    base = np.roll(scene, -2 * t, axis=1)
    left = base + rng.uniform(-3.0, 3.0, size=base.shape)
    d = bf / _depth_map(stream_id, h, w)                 # right(x) = left(x + d)
    xs = np.arange(w)[None, :] + d
    x0 = np.floor(xs).astype(int); fx = xs - x0
    x0c = np.clip(x0, 0, w - 1); x1c = np.clip(x0 + 1, 0, w - 1)
    rows = np.arange(h)[:, None]
    right = base[rows, x0c] * (1 - fx) + base[rows, x1c] * fx + rng.uniform(-3.0, 3.0, size=base.shape)
    to_u8 = lambda a: np.ascontiguousarray(np.clip(np.rint(a), 0, 255).astype(np.uint8))
    return to_u8(left), to_u8(right)


def stereo_batch(n_pairs, stream_id=0, t0=0, h=IMG_H, w=IMG_W, n_rect=6000):
    """[n_pairs, 2, h, w] uint8: frames t0.. of one stream (left=[:,0], right=[:,1]).  n_rect = 6000 is the BASELINE stream
    (SURVEY.md §8(d)); fewer rectangles give a sparser scene (a few % FAST corners, closer to real imagery)."""
    scene = make_scene(stream_id, h, w, n_rect)
    out = np.empty((n_pairs, 2, h, w), np.uint8)

    def one(i):
        out[i, 0], out[i, 1] = stereo_pair(stream_id, t0 + i, h, w, scene)
    workers = min(32, n_pairs, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    if workers <= 1:
        for i in range(n_pairs):
            one(i)
    else:           # frames are independent (own PRNG each): rendered on a thread pool, same bytes as the serial loop
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, range(n_pairs)))
    return out


def random_image(seed, h, w, kind="texture"):
    """Small seeded test images for parity tests."""
    rng = _rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    img = 128 + 40 * _value_noise(rng, h, w, 16) + 25 * _value_noise(rng, h, w, 4)
    n_rect = max(8, (h * w) // 80)
    x0 = rng.integers(0, w, n_rect); y0 = rng.integers(0, h, n_rect)
    sw = rng.integers(3, 25, n_rect); sh = rng.integers(3, 25, n_rect)
    val = rng.uniform(-90.0, 90.0, n_rect)
    x1 = np.minimum(x0 + sw, w); y1 = np.minimum(y0 + sh, h)
    diff = np.zeros((h + 1, w + 1))
    np.add.at(diff, (y0, x0), val); np.add.at(diff, (y0, x1), -val)
    np.add.at(diff, (y1, x0), -val); np.add.at(diff, (y1, x1), val)
    img += np.cumsum(np.cumsum(diff, axis=0), axis=1)[:h, :w]
    img += rng.uniform(-3, 3, size=(h, w))
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))      # C order: row pitch = width


# ----------------------------------------------------------------------------------------------
CALC_SHAPES = [("conv1.w", (64, 1, 5, 5)), ("conv1.b", (64,)), ("conv2.w", (128, 64, 4, 4)), ("conv2.b", (128,)),
               ("conv3.w", (4, 128, 3, 3)), ("conv3.b", (4,))]


def calc_weights(seed=0xCA1C):
    """Flat f32 blob in the layout both oracle and HIP path use (137 476 floats)."""
    rng = _rng(seed)
    parts = []
    for name, shp in CALC_SHAPES:
        if name.endswith(".w"):
            fan_in = shp[1] * shp[2] * shp[3]
            parts.append(rng.normal(0.0, np.sqrt(1.0 / fan_in), size=shp).astype(np.float32).ravel())
        else:
            parts.append(rng.uniform(0.0, 0.1, size=shp).astype(np.float32).ravel())
    return np.concatenate(parts)


def _gauss2(n, sigma):
    c = (n - 1) / 2
    y, x = np.mgrid[0:n, 0:n]
    g = np.exp(-((x - c) ** 2 + (y - c) ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def calc_weights_handcrafted(c3_gain=6.0, c3_bias=0.05):
    """A deterministic, NON-degenerate weight set of the CALC architecture, built by hand (the trained calc.caffemodel is a download the
    build environment cannot make; N(0, 1/fan_in) weights give cosine scores of 0.99 between ANY two frames, so the loop rule of
    src/loopclosing.cpp:124-161 can never be exercised with them).  CALC is trained to reproduce HOG-like appearance; this bank is the
    classical hand-made analogue:
      conv1 (64 @ 5x5 / 2): smoothed intensity and its complement at two scales (ON / OFF pathways), 16 signed gradient directions x 2
                            scales, 7 ridge / valley orientations x 2 polarities x 2 scales;
      conv2 (128 @ 4x4):    Gaussian pooling of every conv1 channel, and the same sharpened against its neighbours in the bank;
      conv3 (4 @ 3x3):      centre-minus-surround of the pooled ON / OFF maps at cell scale (an 8 x 8 input-pixel cell per output), biased
                            down so that the ReLU leaves a sparse blob map, plus a little oriented-energy so the whole bank matters.
    The descriptor is a 14 x 19 x 4 map of bright / dark blob dominance: the same place seen again scores >= 0.97 against itself, a
    key-frame 0.4 m to the side <= 0.85, unrelated places ~0.6 (tests/test_gpu_sequence.py measures it on the rendered sequence).
    Same flat layout as calc_weights()."""
    w1 = np.zeros((64, 1, 5, 5), np.float32); b1 = np.zeros(64, np.float32)
    w1[0, 0] = _gauss2(5, 1.2); w1[1, 0] = -_gauss2(5, 1.2); b1[1] = 1.0          # input in [0, 1]: both pathways stay in [0, 1]
    w1[2, 0] = _gauss2(5, 2.0); w1[3, 0] = -_gauss2(5, 2.0); b1[3] = 1.0
    y, x = np.mgrid[-2:3, -2:3].astype(float)
    k = 4
    for sig in (0.9, 1.5):
        g = np.exp(-(x * x + y * y) / (2 * sig * sig)); gx = -x * g; gy = -y * g; nrm = np.abs(gx).sum()
        for d in range(16):
            th = 2 * np.pi * d / 16
            w1[k, 0] = 2 * (np.cos(th) * gx + np.sin(th) * gy) / nrm; k += 1
    for sig in (1.0, 1.6):
        for d in range(7):
            th = np.pi * d / 7
            u = x * np.cos(th) + y * np.sin(th); v = -x * np.sin(th) + y * np.cos(th)
            f = (1 - u * u / (sig * sig)) * np.exp(-(u * u) / (2 * sig * sig) - (v * v) / (2 * (1.5 * sig) ** 2)); f -= f.mean(); f /= np.abs(f).sum()
            w1[k, 0] = 2 * f; w1[k + 1, 0] = -2 * f; k += 2
    assert k == 64
    w2 = np.zeros((128, 64, 4, 4), np.float32); b2 = np.zeros(128, np.float32)
    p4 = _gauss2(4, 1.2)
    for c in range(64):
        w2[c, c] = p4
        w2[64 + c, c] = 1.5 * p4
        for o in (c - 1, c + 1):
            if 0 <= o < 64:
                w2[64 + c, o] -= 0.5 * p4
    w3 = np.zeros((4, 128, 3, 3), np.float32); b3 = np.full(4, -c3_bias, np.float32)
    cs = -np.ones((3, 3)) / 8; cs[1, 1] = 1.0
    for j in range(4):
        w3[j, j] = c3_gain * cs
    for d in range(16):
        th = 2 * np.pi * d / 16
        w3[0, 4 + d] += 0.05 * abs(np.cos(th)) * _gauss2(3, 1.0); w3[1, 4 + d] += 0.05 * abs(np.sin(th)) * _gauss2(3, 1.0)
    return np.concatenate([w1.ravel(), b1, w2.ravel(), b2, w3.ravel(), b3]).astype(np.float32)


def lcd_database(n, seed=0xDB, dim=1064):
    rng = _rng(seed)
    db = np.abs(rng.standard_normal(size=(n, dim))).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(db, np.float32)


# ----------------------------------------------------------------------------------------------
def _quat_from_yaw(yaw):
    return np.array([0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)])      # rotation about camera y


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def ba_problem(seed=0xBA, n_kf=10, n_mp=300, noise_px=0.5, outlier_frac=0.03, perturb=True, K=KITTI00,
               h=IMG_H, w=IMG_W):
    """A sliding-window BA instance: poses (n_kf,7: qx qy qz qw tx ty tz, Tcw), points (n_mp,3),
    edges (pose_idx, pt_idx, obs uv), fixed flags.  ~n_kf*n_mp edges."""
    rng = _rng(seed)
    poses = np.zeros((n_kf, 7))
    for i in range(n_kf):
        yaw = np.deg2rad(5.0) * (i / max(n_kf - 1, 1) - 0.5) * 2
        twc = np.array([0.15 * np.sin(i * 0.7), 0.0, 1.0 * i])          # 1 m spacing along z
        q = _quat_from_yaw(yaw)                                         # Rwc
        Rwc = _quat_to_R(q)
        Rcw = Rwc.T
        tcw = -Rcw @ twc
        poses[i, :4] = [-q[0], -q[1], -q[2], q[3]]
        poses[i, 4:] = tcw
    pts = np.stack([rng.uniform(-10, 10, n_mp), rng.uniform(-3, 3, n_mp), rng.uniform(5, 40, n_mp) + n_kf], axis=1)
    # every landmark is observed by every key-frame it projects into; edges grouped by landmark (as backend.cpp:161-205 builds them)
    Rs = np.stack([_quat_to_R(poses[i, :4]) for i in range(n_kf)])
    pc = np.einsum("iab,jb->jia", Rs, pts) + poses[None, :, 4:]                 # [n_mp, n_kf, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K["fx"] * pc[..., 0] / pc[..., 2] + K["cx"]; v = K["fy"] * pc[..., 1] / pc[..., 2] + K["cy"]
    vis = (pc[..., 2] >= 0.5) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
    el, ep = np.nonzero(vis)
    ep = ep.astype(np.int32); el = el.astype(np.int32)
    obs = np.stack([u[vis], v[vis]], axis=1).astype(np.float64).reshape(-1, 2)
    obs += rng.normal(0, noise_px, size=obs.shape)
    nout = int(outlier_frac * len(ep))
    if nout:
        idx = rng.choice(len(ep), nout, replace=False)
        obs[idx] += rng.uniform(-40, 40, size=(nout, 2))
    fixed = (rng.uniform(size=n_mp) < 0.1).astype(np.uint8)            # landmarks first seen outside the window
    if perturb:
        pts = pts + rng.normal(0, 0.05, size=pts.shape) * (1 - fixed)[:, None]
        poses = poses.copy()
        poses[:, 4:] += rng.normal(0, 0.02, size=(n_kf, 3))
    Kt = (K["fx"], K["fy"], K["cx"], K["cy"])
    return poses, pts, ep, el, obs, fixed, Kt


def ba_windows(n, seed0=0xBA, **kw):
    """n DISTINCT sliding-window BA instances (seeds seed0, seed0 + 1, ...) padded to common capacities, in the layout of the
    *_batch entry points: (poses [n,maxP,7], points [n,maxL,3], edge_pose [n,maxE], edge_pt [n,maxE], obs [n,maxE,2],
    fixed [n,maxL], sizes [n,3] = (nposes, npts, nedges)), plus the camera tuple."""
    probs = [ba_problem(seed=seed0 + i, **kw) for i in range(n)]
    maxP = max(len(p[0]) for p in probs); maxL = max(len(p[1]) for p in probs); maxE = max(len(p[2]) for p in probs)
    poses = np.zeros((n, maxP, 7)); poses[:, :, 3] = 1.0
    pts = np.zeros((n, maxL, 3)); ep = np.zeros((n, maxE), np.int32); el = np.zeros((n, maxE), np.int32)
    obs = np.zeros((n, maxE, 2)); fixed = np.zeros((n, maxL), np.uint8); sizes = np.zeros((n, 3), np.int32)
    for i, (po, pt, e0, e1, ob, fx, _) in enumerate(probs):
        poses[i, :len(po)] = po; pts[i, :len(pt)] = pt; ep[i, :len(e0)] = e0; el[i, :len(e1)] = e1; obs[i, :len(ob)] = ob
        fixed[i, :len(fx)] = fx; sizes[i] = (len(po), len(pt), len(e0))
    return (poses, pts, ep, el, obs, fixed, sizes), probs[0][6]


# ---- loop correction: synthetic pose graphs ------------------------------------------------------------------------

def _so3_exp(w):
    th = float(np.linalg.norm(w))
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)


def _R_to_quat(R):
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2; q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0, 0, 0, (R[k, j] - R[j, k]) / s]
        q[i] = 0.25 * s; q[j] = (R[j, i] + R[i, j]) / s; q[k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def _T_to_pose7(T):
    return np.concatenate([_R_to_quat(T[:3, :3]), T[:3, 3]])


def pose_graph(n_kf=200, n_loops=2, seed=0x9A, odo_noise=(0.002, 0.02), loop_noise=(0.0005, 0.005), n_active=10, laps=2.2):
    """A key-frame pose graph as LoopClosing::PoseGraphOptimization builds it (src/loopclosing.cpp:546-601):
    a car driving `laps` times round a circuit, key-frame poses Tcw integrated from noisy odometry (so they drift),
    one edge per (KF, previous KF) with the odometry as measurement, `n_loops` loop edges (KF, much older KF at the same
    place) with a near-true relative pose, and the reference's fixed set {KF 0, the last loop's loop-KF, the last n_active}.
    Returns poses (n,7), fixed (n,), e0, e1 (E,), meas (E,7) with meas = T[e0] * T[e1]^-1, and the ground truth (n,7)."""
    rng = _rng(seed)
    per_lap = n_kf / laps
    Tgt = []
    for i in range(n_kf):
        a = 2 * np.pi * i / per_lap
        r = 40.0 + 6.0 * np.sin(3 * a)
        twc = np.array([r * np.cos(a), 0.3 * np.sin(2 * a), r * np.sin(a)])
        Rwc = _so3_exp(np.array([0.0, -(a + np.pi / 2), 0.0])) @ _so3_exp(np.array([0.02 * np.sin(5 * a), 0, 0.03 * np.cos(4 * a)]))
        T = np.eye(4); T[:3, :3] = Rwc.T; T[:3, 3] = -Rwc.T @ twc
        Tgt.append(T)

    def noisy(T, s_rot, s_tr):
        N = np.eye(4); N[:3, :3] = _so3_exp(rng.normal(0, s_rot, 3)); N[:3, 3] = rng.normal(0, s_tr, 3)
        return N @ T

    e0, e1, meas = [], [], []
    Test = [Tgt[0]]
    for i in range(1, n_kf):
        M = noisy(Tgt[i] @ np.linalg.inv(Tgt[i - 1]), *odo_noise)
        e0.append(i); e1.append(i - 1); meas.append(_T_to_pose7(M))
        Test.append(M @ Test[-1])
    fixed = np.zeros(n_kf, np.uint8)
    fixed[0] = 1
    fixed[max(0, n_kf - n_active):] = 1
    lap = int(round(per_lap))
    cand = [i for i in range(lap + 5, n_kf - n_active - 1)]
    n_old = max(n_loops - 1, 0)
    picks = sorted(rng.choice(cand, size=min(n_old, len(cand)), replace=False).tolist()) if n_old and cand else []
    if n_loops:
        picks.append(n_kf - 1)                                          # the loop being closed starts at the current key-frame
    picks = [i for i in picks if i - 21 >= 1]                           # a loop needs a key-frame at least 21 back (cf. loopclosing.cpp:131)
    for k, i in enumerate(picks):
        j = i - lap + int(rng.integers(-2, 3))
        j = min(max(j, 1), i - 21)
        M = noisy(Tgt[i] @ np.linalg.inv(Tgt[j]), *loop_noise)
        e0.append(i); e1.append(j); meas.append(_T_to_pose7(M))
        if k == len(picks) - 1:
            # LoopLocalFusion (:466-533) has already moved the active window onto the corrected current pose
            fixed[j] = 1
            Tcorr = M @ Test[j]
            rel = np.linalg.inv(Test[i]) @ Tcorr
            for a in range(max(0, n_kf - n_active), n_kf):
                Test[a] = Test[a] @ rel
    poses = np.stack([_T_to_pose7(T) for T in Test])
    gt = np.stack([_T_to_pose7(T) for T in Tgt])
    return poses, fixed, np.array(e0, np.int32), np.array(e1, np.int32), np.array(meas).reshape(-1, 7), gt


# ---- loop verification: 3D-2D matches between a loop key-frame's map points and the current key-frame --------------------------

def pnp_problem(n=120, outlier_frac=0.3, noise_px=0.5, seed=0x919, K=KITTI00, w=IMG_W, h=IMG_H):
    """Map points in front of a camera at a known pose (the loop-corrected pose LoopClosing::ComputeCorrectPose is after,
    src/loopclosing.cpp:208-268), their pixels with Gaussian noise, and a fraction of gross mismatches.
    Returns pts3d (n,3) f32, pts2d (n,2) f32, Kt, true pose7 (Tcw), inlier ground truth (n,) bool."""
    rng = _rng(seed)
    yaw = rng.uniform(-0.6, 0.6)
    Rcw = _so3_exp(np.array([rng.uniform(-0.05, 0.05), yaw, rng.uniform(-0.05, 0.05)]))
    tcw = np.array([rng.uniform(-3, 3), rng.uniform(-0.5, 0.5), rng.uniform(-3, 3)])
    Kt = (K["fx"], K["fy"], K["cx"], K["cy"])
    pc = np.stack([rng.uniform(-12, 12, n), rng.uniform(-3, 2, n), rng.uniform(4, 45, n)], 1)
    u = Kt[0] * pc[:, 0] / pc[:, 2] + Kt[2]; v = Kt[1] * pc[:, 1] / pc[:, 2] + Kt[3]
    u = np.clip(u, 0, w - 1); v = np.clip(v, 0, h - 1)                 # keep the pixels inside the image, move the points with them
    pc[:, 0] = (u - Kt[2]) * pc[:, 2] / Kt[0]; pc[:, 1] = (v - Kt[3]) * pc[:, 2] / Kt[1]
    pw = (pc - tcw) @ Rcw                                              # Rcw^T (pc - t)
    uv = np.stack([u, v], 1) + rng.normal(0, noise_px, (n, 2))
    good = np.ones(n, bool)
    nout = int(round(outlier_frac * n))
    if nout:
        idx = rng.choice(n, nout, replace=False)
        uv[idx] = np.stack([rng.uniform(0, w, nout), rng.uniform(0, h, nout)], 1)
        good[idx] = False
    T = np.eye(4); T[:3, :3] = Rcw; T[:3, 3] = tcw
    return pw.astype(np.float32), uv.astype(np.float32), Kt, _T_to_pose7(T), good


# ---- a geometrically consistent stereo SEQUENCE with ground truth (the stand-in for BASELINE configs[0], KITTI-00) ----------------------
# World: ONE continuous textured wall whose depth zig-zags along x (a piecewise-linear profile Z(X): no occlusion edges, every pixel
# has a well-defined depth).  The rig (left camera + right camera `baseline` to its right) moves out along the wall and back to its
# start, with a small yaw.  Every pixel is the bilinear sample of the wall point its viewing ray hits, so LK tracks, stereo
# disparities, triangulated landmarks and camera poses are mutually consistent and the true poses are known.

SEQ_K = {"fx": 420.0, "fy": 420.0, "cx": 359.5, "cy": 119.5, "bf": 420.0 * 0.54}       # a 720 x 240 camera, 0.54 m baseline


def sequence_scene(seed=0x5E0, tex_w=4096, tex_h=1024, texels_per_m=40.0, x_min=-45.0, x_max=55.0, z_lo=8.0, z_hi=24.0):
    rng = _rng(seed)
    xs = [x_min]
    while xs[-1] < x_max:
        xs.append(xs[-1] + rng.uniform(3.0, 7.0))
    xs = np.array(xs)
    zs = np.empty(len(xs)); zs[0] = min(max(14.0, z_lo), z_hi)
    for i in range(1, len(xs)):                       # depth between z_lo and z_hi (8 and 24 m), slope at most 1.0
        lo = max(z_lo, zs[i - 1] - (xs[i] - xs[i - 1])); hi = min(z_hi, zs[i - 1] + (xs[i] - xs[i - 1]))
        zs[i] = rng.uniform(lo, hi)
    seg = np.sqrt(np.diff(xs) ** 2 + np.diff(zs) ** 2)
    arc = np.concatenate([[0.0], np.cumsum(seg)])     # arc length of the wall at every knot: the texture is not stretched by the slant
    t = 128.0 + 50.0 * (_value_noise(rng, tex_h, tex_w, 64) + 0.5 * _value_noise(rng, tex_h, tex_w, 16) + 0.25 * _value_noise(rng, tex_h, tex_w, 4)) / 1.75
    nr = 9000
    rx = rng.integers(0, tex_w, nr); ry = rng.integers(0, tex_h, nr); sw = rng.integers(4, 36, nr); sh = rng.integers(4, 36, nr)
    val = rng.uniform(-80.0, 80.0, nr)
    diff = np.zeros((tex_h + 1, tex_w + 1))
    np.add.at(diff, (ry, rx), val); np.add.at(diff, (ry, np.minimum(rx + sw, tex_w)), -val)
    np.add.at(diff, (np.minimum(ry + sh, tex_h), rx), -val); np.add.at(diff, (np.minimum(ry + sh, tex_h), np.minimum(rx + sw, tex_w)), val)
    t += np.cumsum(np.cumsum(diff, axis=0), axis=1)[:tex_h, :tex_w]
    return {"x": xs, "z": zs, "arc": arc, "tex": t, "texels_per_m": texels_per_m}


def sequence_poses(n=200, kind="figure", reach=25.0, laps=1):
    """camera position [x, y, z] and yaw of frame t.  "figure": a flat figure along the wall that starts in full sideways motion (a
    vehicle that is already driving: the first key-frames see parallax), swings 3 m to either side and ends where it started.
    "outback": a drive of `reach` metres along the wall and back to the start (up to reach * pi / n metres per frame: tens of pixels of
    flow per frame at KITTI resolution, features leave the view for good, the way back revisits every place)"""
    t = np.arange(n) / (n - 1)
    if kind == "ramp":          # `laps` times out to `reach` and back, starting from rest: x = reach sin^2(pi laps t); up to reach pi laps / n metres per frame mid-leg
        x = reach * np.sin(np.pi * laps * t) ** 2
        zc = 1.0 * np.sin(2 * np.pi * laps * t) ** 2
        yaw = np.deg2rad(3.0) * np.sin(6 * np.pi * laps * t)
        return np.stack([x, np.zeros(n), zc], 1), yaw
    if kind == "legs":          # `laps` times out and back at a CONSTANT `reach` metres per frame (cosine-eased starts, turns and stops of 12 frames):
        # the reference's key-frame rule (inliers <= trackingGood) then fires at a steady rate instead of in bursts at a sine's peak speed
        half = n // (2 * laps); ease = 12
        prof = np.ones(half)
        prof[:ease] = 0.5 - 0.5 * np.cos(np.pi * (np.arange(ease) + 0.5) / ease); prof[half - ease:] = prof[:ease][::-1]
        vel = np.concatenate([np.concatenate([prof, -prof]) for _ in range(laps)])
        vel = np.concatenate([vel, np.zeros(n - len(vel))])
        x = reach * np.concatenate([[0.0], np.cumsum(vel)[:-1]])
        tt = np.arange(n) / max(half, 1)
        zc = 1.0 * np.sin(np.pi * tt) ** 2
        yaw = np.deg2rad(3.0) * np.sin(3 * np.pi * tt)
        return np.stack([x, np.zeros(n), zc], 1), yaw
    if kind == "oneway":        # a drive in ONE direction at `reach` metres per frame (eased start): no place is seen twice
        x = reach * np.concatenate([[0.0], np.cumsum(np.minimum(1.0, (np.arange(n - 1) + 0.5) / 12.0))])
        tt = np.arange(n) / 100.0
        return np.stack([x, np.zeros(n), np.sin(np.pi * tt) ** 2], 1), np.deg2rad(3.0) * np.sin(3 * np.pi * tt)
    if kind == "outback":
        x = reach * np.sin(np.pi * t)
        zc = 1.0 * np.sin(2 * np.pi * t) ** 2
        yaw = np.deg2rad(3.0) * np.sin(6 * np.pi * t)
        return np.stack([x, np.zeros(n), zc], 1), yaw
    x = 3.0 * np.sin(2 * np.pi * t)                                                    # 0 -> 3 m -> -3 m -> 0
    zc = 0.8 * (1.0 - np.cos(2 * np.pi * t))
    yaw = np.deg2rad(2.0) * np.sin(4 * np.pi * t)
    return np.stack([x, np.zeros(n), zc], 1), yaw


# ---- a corridor driven FORWARD (round 5): the motion of a car on a road — features stream outwards from the vanishing point, grow, change
# pyramid level and leave through the image border, depth runs from 3 m to infinity.  Two textured side walls (x = -a, x = +a, up to `wall_h`
# above the camera), a textured ground plane (y = +cam_h, y down) and a featureless sky; every surface is sampled through a small mip chain
# chosen by its texel rate per pixel, so that distant parts do not alias into frame-to-frame noise.
def _mips(t, n=5):
    out = [t]
    for _ in range(n - 1):
        a = out[-1]
        h, w = (a.shape[0] // 2) * 2, (a.shape[1] // 2) * 2
        out.append(0.25 * (a[0:h:2, 0:w:2] + a[1:h:2, 0:w:2] + a[0:h:2, 1:w:2] + a[1:h:2, 1:w:2]))
    return out


def _texture(rng, th, tw, nrect):
    t = 128.0 + 50.0 * (_value_noise(rng, th, tw, 64) + 0.5 * _value_noise(rng, th, tw, 16) + 0.25 * _value_noise(rng, th, tw, 4)) / 1.75
    rx = rng.integers(0, tw, nrect); ry = rng.integers(0, th, nrect); sw = rng.integers(4, 36, nrect); sh = rng.integers(4, 36, nrect)
    val = rng.uniform(-80.0, 80.0, nrect)
    diff = np.zeros((th + 1, tw + 1))
    np.add.at(diff, (ry, rx), val); np.add.at(diff, (ry, np.minimum(rx + sw, tw)), -val)
    np.add.at(diff, (np.minimum(ry + sh, th), rx), -val); np.add.at(diff, (np.minimum(ry + sh, th), np.minimum(rx + sw, tw)), val)
    return t + np.cumsum(np.cumsum(diff, axis=0), axis=1)[:th, :tw]


def corridor_scene(seed=0xC0881D0, half_width=7.0, cam_h=1.65, wall_h=9.0, texels_per_m=32.0, tex_len=8192):
    rng = _rng(seed)
    wall_rows = int((wall_h + cam_h) * texels_per_m)
    ground_rows = int(2 * half_width * texels_per_m)
    walls = [_mips(_texture(rng, wall_rows, tex_len, 7000)) for _ in range(2)]
    ground = _mips(128.0 + 0.6 * (_texture(rng, ground_rows, tex_len, 5000) - 128.0))       # a road: lower contrast than the walls
    return {"a": half_width, "cam_h": cam_h, "wall_h": wall_h, "s": texels_per_m, "walls": walls, "ground": ground, "tex_len": tex_len}


def _sample_mips(mips, tu, tv, rate):
    """trilinear sample: (tu, tv) texel coordinates at level 0 (tu wraps), rate = texels per pixel"""
    lvl = np.clip(np.log2(np.maximum(rate, 1.0)), 0.0, len(mips) - 1.001)
    l0 = np.floor(lvl).astype(int); fl = lvl - l0
    out = np.zeros(tu.shape)
    for k in range(len(mips)):
        for which, wgt in ((0, 1.0 - fl), (1, fl)):
            m = (l0 + which) == k
            if not m.any():
                continue
            t = mips[k]; th, tw = t.shape
            u = np.mod(tu[m] / (1 << k) - 0.5 * (1 - 1.0 / (1 << k)), tw - 1); v = np.clip(tv[m] / (1 << k) - 0.5 * (1 - 1.0 / (1 << k)), 0, th - 1.001)
            u0 = np.floor(u).astype(int); v0 = np.floor(v).astype(int); fu = u - u0; fv = v - v0
            val = (t[v0, u0] * (1 - fu) + t[v0, u0 + 1] * fu) * (1 - fv) + (t[v0 + 1, u0] * (1 - fu) + t[v0 + 1, u0 + 1] * fu) * fv
            out[m] += wgt[m] * val
    return out


def render_corridor_camera(scene, c, yaw, h=IMG_H, w=IMG_W, K=KITTI00, noise_seed=None):
    a, ch, wh, s = scene["a"], scene["cam_h"], scene["wall_h"], scene["s"]
    xn = (np.arange(w) - K["cx"]) / K["fx"]; yn = (np.arange(h) - K["cy"]) / K["fy"]
    dx = (np.cos(yaw) * xn + np.sin(yaw))[None, :].repeat(h, 0); dz = (-np.sin(yaw) * xn + np.cos(yaw))[None, :].repeat(h, 0)
    dy = yn[:, None].repeat(w, 1)
    big = 1e9
    with np.errstate(divide="ignore", invalid="ignore"):
        lam_l = np.where(dx < -1e-9, (-a - c[0]) / dx, big); lam_r = np.where(dx > 1e-9, (a - c[0]) / dx, big)
        lam_g = np.where(dy > 1e-9, (ch - c[1]) / dy, big)
    lam_w = np.minimum(lam_l, lam_r)
    yw = c[1] + lam_w * dy                                         # height on the wall (y down): sky above -wall_h
    wall_ok = (lam_w < big) & (yw > -wh) & (yw <= ch + 1e-6) & (lam_w * dz > 0.3)
    use_wall = wall_ok & (lam_w <= lam_g)
    use_ground = ~use_wall & (lam_g < big) & (np.abs(c[0] + lam_g * dx) <= a + 1e-6) & (lam_g * dz > 0.3)
    img = np.full((h, w), 150.0) - 25.0 * np.clip((yn[:, None] + 0.6), 0, 1)           # sky: a smooth vertical gradient, no corners
    px = 1.0 / K["fx"]
    for side, sel in ((0, use_wall & (lam_l <= lam_r)), (1, use_wall & (lam_r < lam_l))):
        if sel.any():
            lam = lam_w[sel]
            z = c[2] + lam * dz[sel]; y = c[1] + lam * dy[sel]
            rate = lam * px * s / np.maximum(np.abs(dx[sel]) / np.sqrt(dx[sel] ** 2 + dz[sel] ** 2), 0.05)       # grazing angle stretches the footprint along z
            img[sel] = _sample_mips(scene["walls"][side], (z + 1000.0) * s, (y + wh) * s, rate)
    if use_ground.any():
        lam = lam_g[use_ground]
        z = c[2] + lam * dz[use_ground]; x = c[0] + lam * dx[use_ground]
        rate = lam * px * s / np.maximum(dy[use_ground] / np.sqrt(dy[use_ground] ** 2 + dz[use_ground] ** 2), 0.05)
        img[use_ground] = _sample_mips(scene["ground"], (z + 1000.0) * s, (x + a) * s, rate)
    if noise_seed is not None:
        img = img + _rng(noise_seed).uniform(-1.5, 1.5, size=img.shape)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def render_corridor_stereo(scene, c, yaw, t=0, **kw):
    K = kw.get("K", KITTI00)
    bl = K["bf"] / K["fx"]
    cr = np.asarray(c, float) + bl * np.array([np.cos(yaw), 0.0, -np.sin(yaw)])
    return render_corridor_camera(scene, c, yaw, noise_seed=3000 + 2 * t, **kw), render_corridor_camera(scene, cr, yaw, noise_seed=3001 + 2 * t, **kw)


def corridor_poses(n=200, speed=0.9):
    """a drive along +z at `speed` metres per frame (eased start), a slow lateral sway of +-1.2 m and +-2.5 degrees of yaw"""
    z = speed * np.concatenate([[0.0], np.cumsum(np.minimum(1.0, (np.arange(n - 1) + 0.5) / 10.0))])
    tt = np.arange(n) / 60.0
    return np.stack([1.2 * np.sin(2 * np.pi * tt / 3.0), np.zeros(n), z], 1), np.deg2rad(2.5) * np.sin(2 * np.pi * tt / 2.0)


def pose7_from_twc(c, yaw):
    """(qx qy qz qw tx ty tz) of Tcw for a camera at position c with yaw (rotation about the camera's y axis)"""
    Rwc = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
    Rcw = Rwc.T
    return np.concatenate([_R_to_quat(Rcw), -Rcw @ np.asarray(c, float)])


def sequence_depth(scene, c, yaw, u, K=SEQ_K):
    """(lambda, arc-length coordinate) of the wall point seen at image column(s) u: the ray c + lambda * (dx, yn, dz) meets Z(X)"""
    xn = (np.asarray(u, float) - K["cx"]) / K["fx"]
    dx = np.cos(yaw) * xn + np.sin(yaw); dz = -np.sin(yaw) * xn + np.cos(yaw)
    xs, zs, arc = scene["x"], scene["z"], scene["arc"]
    best = np.full(xn.shape, np.inf); sarc = np.zeros(xn.shape)
    for k in range(len(xs) - 1):
        m = (zs[k + 1] - zs[k]) / (xs[k + 1] - xs[k])
        den = dz - m * dx
        lam = (zs[k] + m * (c[0] - xs[k]) - c[2]) / np.where(np.abs(den) < 1e-9, 1e-9, den)
        X = c[0] + lam * dx
        hit = (lam > 0.5) & (X >= xs[k]) & (X <= xs[k + 1]) & (lam < best)
        best = np.where(hit, lam, best)
        sarc = np.where(hit, arc[k] + (X - xs[k]) * np.sqrt(1 + m * m), sarc)
    return best, sarc


def render_camera(scene, c, yaw, h=240, w=720, K=SEQ_K, noise_seed=None):
    lam, sarc = sequence_depth(scene, c, yaw, np.arange(w), K)
    yn = (np.arange(h) - K["cy"]) / K["fy"]
    tex = scene["tex"]; TH, TW = tex.shape
    s = scene["texels_per_m"]
    tu = np.mod(sarc * s, TW - 1)[None, :].repeat(h, 0)
    tv = np.mod((c[1] + lam[None, :] * yn[:, None]) * s + TH / 2, TH - 1)
    u0 = np.floor(tu).astype(int); v0 = np.floor(tv).astype(int); fu = tu - u0; fv = tv - v0
    img = (tex[v0, u0] * (1 - fu) + tex[v0, u0 + 1] * fu) * (1 - fv) + (tex[v0 + 1, u0] * (1 - fu) + tex[v0 + 1, u0 + 1] * fu) * fv
    if noise_seed is not None:
        img = img + _rng(noise_seed).uniform(-1.5, 1.5, size=img.shape)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def render_stereo(scene, c, yaw, t=0, **kw):
    """(left, right) of the rig at position c / yaw; the right camera sits `baseline` along the rig's x axis"""
    K = kw.get("K", SEQ_K)
    bl = K["bf"] / K["fx"]
    cr = np.asarray(c, float) + bl * np.array([np.cos(yaw), 0.0, -np.sin(yaw)])
    return render_camera(scene, c, yaw, noise_seed=1000 + 2 * t, **kw), render_camera(scene, cr, yaw, noise_seed=1001 + 2 * t, **kw)
