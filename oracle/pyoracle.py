"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: may be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h", ".inc"))]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28 == C.sizeof(KeyPoint)


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int)]


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


CALC_LAYER_DTYPE = np.dtype([("type", "<i4"), ("num_output", "<i4"), ("kernel", "<i4"), ("stride", "<i4"), ("pad", "<i4"),
                             ("local_size", "<i4"), ("alpha", "<f4"), ("beta", "<f4"), ("k", "<f4")])


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.orc_orb_pattern.restype = C.POINTER(C.c_int8)
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_lcd_score.restype = C.c_float
        L.orc_calc_nweights.restype = C.c_size_t
        L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]

    # ---- tables ----
    @staticmethod
    def params(nfeatures=2000, scale=1.2, nlevels=8, ini=20, mn=7):
        return OrbParams(nfeatures, scale, nlevels, ini, mn)

    def orb_tables(self, p):
        n = p.nlevels
        sc = np.zeros(n, np.float32); isc = np.zeros(n, np.float32)
        npl = np.zeros(n, np.int32); umax = np.zeros(16, np.int32)
        rc = self.lib.orc_orb_tables(C.byref(p), _p(sc), _p(isc), _p(npl), _p(umax))
        assert rc == 0
        return sc, isc, npl, umax

    def level_size(self, cols, rows, inv_scale):
        w = C.c_int(); h = C.c_int()
        self.lib.orc_level_size(cols, rows, C.c_float(inv_scale), C.byref(w), C.byref(h))
        return w.value, h.value

    def pattern(self):
        return np.ctypeslib.as_array(self.lib.orc_orb_pattern(), shape=(1024,)).copy()

    # ---- image primitives ----
    def resize(self, src, dw, dh):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        rc = self.lib.orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0],
                                           _p(dst), dw, dh, dw)
        assert rc == 0
        return dst

    def blur7(self, src, kind=0):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros_like(src)
        rc = self.lib.orc_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0],
                                            _p(dst), dst.strides[0], kind)
        assert rc == 0
        return dst

    def pyramid(self, p, img):
        img = np.ascontiguousarray(img, np.uint8)
        _, isc, _, _ = self.orb_tables(p)
        lv = []
        for l in range(p.nlevels):
            w, h = self.level_size(img.shape[1], img.shape[0], float(isc[l]))
            lv.append(np.zeros((h, w), np.uint8))
        ptrs = (C.c_void_p * p.nlevels)(*[a.ctypes.data for a in lv])
        rc = self.lib.orc_build_pyramid(C.byref(p), _p(img), img.shape[0], img.shape[1], img.strides[0], ptrs)
        assert rc == 0, rc
        return lv

    # ---- FAST ----
    def set_gauss_taps(self, q7=None):
        """sigma = 2 taps of the ORB blur (process global; None restores the default)"""
        q = None if q7 is None else np.ascontiguousarray(q7, np.int32)
        assert self.lib.orc_set_gauss_taps(_p(q) if q is not None else None) == 0

    def fast_score_px(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return self.lib.orc_fast_score_px(_p(img), img.strides[0], int(x), int(y))

    def fast_score_seeded(self, img, x, y, th):
        img = np.ascontiguousarray(img, np.uint8)
        return self.lib.orc_fast_score_seeded(_p(img), img.strides[0], int(x), int(y), int(th))

    def fast_score_map(self, img, th):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros_like(img)
        self.lib.orc_fast_score_map(_p(img), img.shape[1], img.shape[0], img.strides[0], th, _p(out))
        return out

    def fast_detect(self, img, th):
        img = np.ascontiguousarray(img, np.uint8)
        cap = img.size
        xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
        n = C.c_int()
        rc = self.lib.orc_fast_detect(_p(img), img.shape[1], img.shape[0], img.strides[0], th,
                                      _p(xs), _p(ys), _p(sc), cap, C.byref(n))
        assert rc == 0
        return xs[:n.value], ys[:n.value], sc[:n.value]

    def is_fast_corner(self, img, x, y, th):
        img = np.ascontiguousarray(img, np.uint8)
        return bool(self.lib.orc_is_fast_corner(_p(img), img.strides[0], x, y, th))

    def grid_fast(self, img, ini=20, mn=7, mask=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = img.size // 2 + 16
        xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
        n = C.c_int()
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        rc = self.lib.orc_grid_fast(_p(img), img.shape[1], img.shape[0], img.strides[0],
                                    _p(mask) if mask is not None else None,
                                    mask.strides[0] if mask is not None else 0,
                                    ini, mn, _p(xs), _p(ys), _p(sc), cap, C.byref(n))
        assert rc == 0, rc
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def octree(self, xs, ys, sc, minX, maxX, minY, maxY, N):
        xs = np.ascontiguousarray(xs, np.int32); ys = np.ascontiguousarray(ys, np.int32)
        sc = np.ascontiguousarray(sc, np.int32)
        cap = len(xs) + 8
        out = np.zeros(cap, np.int32)
        n = C.c_int()
        rc = self.lib.orc_distribute_octree(_p(xs), _p(ys), _p(sc), len(xs), minX, maxX, minY, maxY, N,
                                            _p(out), cap, C.byref(n))
        assert rc == 0, rc
        return out[:n.value].copy()

    # ---- orientation / descriptor ----
    def fast_atan2(self, y, x):
        return self.lib.orc_fast_atan2(C.c_float(y), C.c_float(x))

    def ic_angle(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return self.lib.orc_ic_angle(_p(img), img.strides[0], x, y)

    def sincos(self, rad):
        s = C.c_float(); c = C.c_float()
        self.lib.orc_sincos(C.c_float(rad), C.byref(s), C.byref(c))
        return s.value, c.value

    def brief(self, blurred, x, y, angle_deg):
        blurred = np.ascontiguousarray(blurred, np.uint8)
        d = np.zeros(32, np.uint8)
        self.lib.orc_brief(_p(blurred), blurred.strides[0], x, y, C.c_float(angle_deg), _p(d))
        return d

    # ---- operators ----
    def detect_and_compute(self, p, img, mask=None, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = cap or (p.nfeatures * 2 + 64)
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        rc = self.lib.orc_detect_and_compute(C.byref(p), _p(img), img.shape[0], img.shape[1], img.strides[0],
                                             _p(mask) if mask is not None else None,
                                             mask.strides[0] if mask is not None else 0,
                                             _p(kps), _p(desc), cap, C.byref(n))
        assert rc == 0, rc
        return kps[:n.value].copy(), desc[:n.value].copy()

    def detect(self, p, img, mask=None, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = cap or (p.nfeatures * 2 + 64)
        kps = np.zeros(cap, KP_DTYPE)
        n = C.c_int()
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        rc = self.lib.orc_detect(C.byref(p), _p(img), img.shape[0], img.shape[1], img.strides[0],
                                 _p(mask) if mask is not None else None,
                                 mask.strides[0] if mask is not None else 0, _p(kps), cap, C.byref(n))
        assert rc == 0, rc
        return kps[:n.value].copy()

    def screen(self, p, img, kps_in):
        img = np.ascontiguousarray(img, np.uint8)
        kin = np.ascontiguousarray(kps_in, KP_DTYPE).copy()
        kout = np.zeros(len(kin) + 1, KP_DTYPE)
        n = C.c_int()
        rc = self.lib.orc_screen(C.byref(p), _p(img), img.shape[0], img.shape[1], img.strides[0],
                                 _p(kin), len(kin), _p(kout), len(kout), C.byref(n))
        assert rc == 0, rc
        return kout[:n.value].copy()

    def calc_descriptors(self, p, img, kps):
        img = np.ascontiguousarray(img, np.uint8)
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.zeros((len(kps), 32), np.uint8)
        rc = self.lib.orc_calc_descriptors(C.byref(p), _p(img), img.shape[0], img.shape[1], img.strides[0],
                                           _p(kps), len(kps), _p(desc))
        assert rc == 0, rc
        return desc

    # ---- hamming / triangulation ----
    def hamming_match(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        idx = np.zeros(len(q), np.int32); dist = np.zeros(len(q), np.int32)
        rc = self.lib.orc_hamming_match(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
        assert rc == 0
        return idx, dist

    def hamming_filter(self, dist):
        dist = np.ascontiguousarray(dist, np.int32)
        keep = np.zeros(len(dist), np.uint8)
        mn = C.c_int()
        self.lib.orc_hamming_filter(_p(dist), len(dist), _p(keep), C.byref(mn))
        return keep.astype(bool), mn.value

    def triangulate(self, poses, pts):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        xyz = np.zeros(3); r = C.c_double()
        rc = self.lib.orc_triangulate(_p(poses), _p(pts), len(poses), _p(xyz), C.byref(r))
        assert rc == 0
        return xyz, r.value

    def triangulate_stereo(self, xl, yl, xr, yr, fx, fy, cx, cy, baseline):
        xl, yl, xr, yr = [np.ascontiguousarray(a, np.float32) for a in (xl, yl, xr, yr)]
        n = len(xl)
        xyz = np.zeros((n, 3)); ok = np.zeros(n, np.uint8)
        rc = self.lib.orc_triangulate_stereo(_p(xl), _p(yl), _p(xr), _p(yr), n, C.c_double(fx), C.c_double(fy),
                                             C.c_double(cx), C.c_double(cy), C.c_double(baseline), _p(xyz), _p(ok))
        assert rc == 0
        return xyz, ok.astype(bool)

    # ---- CALC ----
    def calc_nweights(self):
        return self.lib.orc_calc_nweights()

    def calc_preproc(self, img, blur_in_place=False):
        """returns (120x160 f32 input, image after the call)"""
        img = np.ascontiguousarray(img, np.uint8).copy()
        out = np.zeros((120, 160), np.float32)
        rc = self.lib.orc_calc_preproc(_p(img), img.shape[0], img.shape[1], img.strides[0],
                                       1 if blur_in_place else 0, _p(out))
        assert rc == 0
        return out, img

    def calc_forward(self, weights, x):
        weights = np.ascontiguousarray(weights, np.float32); x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(1064, np.float32)
        rc = self.lib.orc_calc_forward(_p(weights), C.c_size_t(weights.size), _p(x), _p(out))
        assert rc == 0, rc
        return out

    def calc_forward_net(self, layers, weights, x):
        """layers: structured array / list of (type, num_output, kernel, stride, pad, local_size, alpha, beta, k) records"""
        L = np.ascontiguousarray(layers, CALC_LAYER_DTYPE)
        w = np.ascontiguousarray(weights, np.float32).ravel(); x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(1064, np.float32)
        rc = self.lib.orc_calc_forward_net(_p(L), len(L), _p(w), C.c_size_t(w.size), _p(x), _p(out))
        assert rc == 0, rc
        return out

    def lcd_score(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        self.lib.orc_lcd_score.argtypes = [C.c_void_p, C.c_void_p]
        return float(self.lib.orc_lcd_score(_p(a), _p(b)))

    def lcddb_query(self, db, ids, q, cur_id, thr_low=0.92):
        db = np.ascontiguousarray(db, np.float32); ids = np.ascontiguousarray(ids, np.uint64)
        q = np.ascontiguousarray(q, np.float32)
        best = C.c_uint64(); mx = C.c_float(); cnt = C.c_int()
        rc = self.lib.orc_lcddb_query(_p(db), _p(ids), len(ids), _p(q), C.c_uint64(cur_id), C.c_float(thr_low),
                                      C.byref(best), C.byref(mx), C.byref(cnt))
        assert rc == 0
        return best.value, mx.value, cnt.value

    # ---- BA ----
    def ba_build(self, poses, points, ep, el, obs, fixed, K, delta=5.991):
        poses = np.ascontiguousarray(poses, np.float64); points = np.ascontiguousarray(points, np.float64)
        ep = np.ascontiguousarray(ep, np.int32); el = np.ascontiguousarray(el, np.int32)
        obs = np.ascontiguousarray(obs, np.float64); fixed = np.ascontiguousarray(fixed, np.uint8)
        P, L, E = len(poses), len(points), len(ep)
        Hpp = np.zeros((P, 6, 6)); Hll = np.zeros((L, 3, 3)); Hpl = np.zeros((E, 6, 3))
        bp = np.zeros((P, 6)); bl = np.zeros((L, 3)); chi2 = np.zeros(E)
        rc = self.lib.orc_ba_build(_p(poses), P, _p(points), L, _p(ep), _p(el), _p(obs), E, _p(fixed),
                                   C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                                   C.c_double(delta), _p(Hpp), _p(Hll), _p(Hpl), _p(bp), _p(bl), _p(chi2))
        assert rc == 0, rc
        return Hpp, Hll, Hpl, bp, bl, chi2

    def ba_optimize(self, poses, points, ep, el, obs, fixed, K, delta=5.991, iters=10):
        poses = np.ascontiguousarray(poses, np.float64).copy(); points = np.ascontiguousarray(points, np.float64).copy()
        ep = np.ascontiguousarray(ep, np.int32); el = np.ascontiguousarray(el, np.int32)
        obs = np.ascontiguousarray(obs, np.float64); fixed = np.ascontiguousarray(fixed, np.uint8)
        chi = C.c_double(); it = C.c_int()
        rc = self.lib.orc_ba_optimize(_p(poses), len(poses), _p(points), len(points), _p(ep), _p(el), _p(obs), len(ep),
                                      _p(fixed), C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                                      C.c_double(delta), iters, C.byref(chi), C.byref(it))
        assert rc == 0, rc
        return poses, points, chi.value, it.value

    def ba_optimize_active_map(self, poses, points, ep, el, obs, fixed, K, delta=5.991, chi2_th=5.991, rounds=5, iters=10):
        poses = np.ascontiguousarray(poses, np.float64).copy(); points = np.ascontiguousarray(points, np.float64).copy()
        ep = np.ascontiguousarray(ep, np.int32); el = np.ascontiguousarray(el, np.int32)
        obs = np.ascontiguousarray(obs, np.float64); fixed = np.ascontiguousarray(fixed, np.uint8)
        chi = np.zeros(len(ep)); out = np.zeros(len(ep), np.uint8); r = C.c_int(); no = C.c_int()
        rc = self.lib.orc_ba_optimize_active_map(_p(poses), len(poses), _p(points), len(points), _p(ep), _p(el), _p(obs), len(ep),
                                                 _p(fixed), C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                                                 C.c_double(delta), C.c_double(chi2_th), rounds, iters, _p(chi), _p(out),
                                                 C.byref(r), C.byref(no))
        assert rc == 0, rc
        return poses, points, chi, out, r.value, no.value

    def ba_optimize_active_map_traced(self, *a, **kw):
        """ba_optimize_active_map + the per-iteration trace: rows of (robust chi2 of the last Levenberg trial, chi2 of the accepted state,
        lambda after the iteration, trials) over all rounds — what g2o's post-iteration hook sees (tools/dump_reference_goldens.cpp)."""
        buf = np.zeros((256, 4))
        self.lib.orc_ba_set_trace(_p(buf), len(buf))
        try:
            out = self.ba_optimize_active_map(*a, **kw)
            n = self.lib.orc_ba_trace_rows()
        finally:
            self.lib.orc_ba_set_trace(None, 0)
        return out, buf[:n].copy()

    def bench_frames(self, frames, K, weights, db, ids, ba_windows, stages=3, threads=1, nfeatures=2000, n_warmup=0, n_tasks=None):
        """CPU-baseline driver (bench_oracle.cpp): frames [n,2,H,W] u8 through the whole per-frame pipeline on `threads` threads.
        ba_windows = (poses [nw,maxP,7], points [nw,maxL,3], ep [nw,maxE], el [nw,maxE], obs [nw,maxE,2], fixed [nw,maxL],
        sizes [nw,3]); task i uses frame i % n and window i % nw; n_tasks (default n) tasks run, the first n_warmup of them untimed.
        Returns (wall seconds over the tasks after the warm-up, per-task stage seconds [n_tasks,5]: orb, match+tri, lcd+db, ba build, ba solve)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, _, H, W = frames.shape
        weights = np.ascontiguousarray(weights, np.float32); db = np.ascontiguousarray(db, np.float32)
        ids = np.ascontiguousarray(ids, np.uint64)
        poses, pts, ep, el, obs, fixed, sizes = ba_windows
        poses = np.ascontiguousarray(poses, np.float64); pts = np.ascontiguousarray(pts, np.float64)
        ep = np.ascontiguousarray(ep, np.int32); el = np.ascontiguousarray(el, np.int32)
        obs = np.ascontiguousarray(obs, np.float64); fixed = np.ascontiguousarray(fixed, np.uint8)
        sizes = np.ascontiguousarray(sizes, np.int32).reshape(-1, 3)
        nw, maxP, maxL, maxE = len(sizes), poses.shape[1], pts.shape[1], ep.shape[1]
        assert poses.shape == (nw, maxP, 7) and pts.shape == (nw, maxL, 3) and el.shape == (nw, maxE) and obs.shape == (nw, maxE, 2)
        n_tasks = n if n_tasks is None else int(n_tasks)
        sec = C.c_double(); st = np.zeros((n_tasks, 5))
        rc = self.lib.orc_bench_frames(_p(frames), n, n_tasks, H, W, nfeatures, C.c_double(K["fx"]), C.c_double(K["fy"]), C.c_double(K["cx"]),
                                       C.c_double(K["cy"]), C.c_double(K["bf"] / K["fx"]), _p(weights), C.c_size_t(weights.size),
                                       _p(db), _p(ids), len(ids), _p(poses), _p(pts), _p(ep), _p(el), _p(obs), _p(fixed), _p(sizes),
                                       nw, maxP, maxL, maxE, int(stages), int(threads), int(n_warmup), C.byref(sec), _p(st))
        assert rc == 0, rc
        return sec.value, st

    def cpu_capacity(self, threads, work_ms=60):
        """threads the host really runs concurrently (affinity masks overstate it under a CPU quota): bench_oracle.cpp"""
        self.lib.orc_cpu_capacity.restype = C.c_double
        return float(self.lib.orc_cpu_capacity(int(threads), int(work_ms)))

    def pose_only_optimize(self, pose, pts3d, obs, K, chi2_th=5.991, rounds=4, iters=10, pre_optimize=0):
        pose = np.ascontiguousarray(pose, np.float64).copy(); pts3d = np.ascontiguousarray(pts3d, np.float64); obs = np.ascontiguousarray(obs, np.float64)
        n = len(pts3d); out = np.zeros(max(n, 1), np.uint8); ni = C.c_int()
        rc = self.lib.orc_pose_only_optimize(_p(pose), _p(pts3d), _p(obs), n, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                                             C.c_double(chi2_th), rounds, iters, pre_optimize, _p(out), C.byref(ni))
        assert rc == 0, rc
        return pose, out[:n].astype(bool), ni.value

    # ---- LK tracker ----
    def pyr_down(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        assert self.lib.orc_pyr_down(_p(img), w, h, img.strides[0], _p(out), out.strides[0]) == 0
        return out

    def lk_track(self, prev, nxt, prev_pts, next_pts, win=11, max_level=3, max_iters=30, eps=0.01, min_eig=1e-4):
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2); npts = np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
        n = len(pp); st = np.zeros(n, np.uint8); err = np.zeros(n, np.float32)
        rc = self.lib.orc_lk_track(_p(prev), _p(nxt), prev.shape[0], prev.shape[1], prev.strides[0], nxt.strides[0], _p(pp), _p(npts), n,
                                   win, max_level, max_iters, C.c_float(eps), C.c_float(min_eig), _p(st), _p(err))
        assert rc >= 0, rc
        return npts, st.astype(bool), err

    def se3_exp(self, xi):
        xi = np.ascontiguousarray(xi, np.float64); out = np.zeros(7)
        self.lib.orc_se3_exp(_p(xi), _p(out))
        return out

    # ---- loop correction: pose graph ----
    def se3_log(self, pose7):
        p = np.ascontiguousarray(pose7, np.float64); out = np.zeros(6)
        self.lib.orc_se3_log(_p(p), _p(out))
        return out

    def se3_compose(self, a7, b7, invert_b=False):
        a = np.ascontiguousarray(a7, np.float64); b = np.ascontiguousarray(b7, np.float64); out = np.zeros(7)
        self.lib.orc_se3_compose(_p(a), _p(b), int(invert_b), _p(out))
        return out

    def pose_graph_optimize(self, poses, fixed, e0, e1, meas, iters=20):
        poses = np.ascontiguousarray(poses, np.float64).copy(); fixed = np.ascontiguousarray(fixed, np.uint8)
        e0 = np.ascontiguousarray(e0, np.int32); e1 = np.ascontiguousarray(e1, np.int32); meas = np.ascontiguousarray(meas, np.float64)
        chi = C.c_double(); it = C.c_int()
        rc = self.lib.orc_pose_graph_optimize(_p(poses), len(poses), _p(fixed), _p(e0), _p(e1), _p(meas), len(e0), iters, C.byref(chi), C.byref(it))
        assert rc == 0, rc
        return poses, chi.value, it.value

    def loop_local_fusion(self, active_poses, cur, corrected_cur, first_active_kf, pts):
        poses = np.ascontiguousarray(active_poses, np.float64).reshape(-1, 7).copy(); cc = np.ascontiguousarray(corrected_cur, np.float64)
        kf = np.ascontiguousarray(first_active_kf, np.int32); pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3).copy()
        rc = self.lib.orc_loop_local_fusion(_p(poses), len(poses), int(cur), _p(cc), _p(kf), _p(pts), len(pts))
        assert rc == 0, rc
        return poses, pts

    def correct_map_points(self, old_poses, new_poses, kf, pts):
        old_poses = np.ascontiguousarray(old_poses, np.float64); new_poses = np.ascontiguousarray(new_poses, np.float64)
        kf = np.ascontiguousarray(kf, np.int32); pts = np.ascontiguousarray(pts, np.float64).copy()
        rc = self.lib.orc_correct_map_points(_p(old_poses), _p(new_poses), len(old_poses), _p(kf), _p(pts), len(pts))
        assert rc == 0, rc
        return pts

    # ---- loop verification: PnP-RANSAC ----
    def solve_pnp_ransac(self, pts3d, pts2d, K, iterations=100, reproj_error=5.991, confidence=0.99):
        p3 = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); p2 = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
        n = len(p3); pose = np.zeros(7); inl = np.zeros(max(n, 1), np.uint8); ni = C.c_int()
        rc = self.lib.orc_solve_pnp_ransac(_p(p3), _p(p2), n, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), iterations,
                                           C.c_double(reproj_error), C.c_double(confidence), _p(pose), _p(inl), C.byref(ni))
        return rc, pose, inl[:n].astype(bool), ni.value

    def epnp(self, pw, uv, K):
        pw = np.ascontiguousarray(pw, np.float64).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        R = np.zeros(9); t = np.zeros(3)
        rc = self.lib.orc_epnp(_p(pw), _p(uv), len(pw), C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), _p(R), _p(t))
        return rc, R.reshape(3, 3), t

    def cv_rng_uniform(self, seed, a, b, count):
        out = np.zeros(count, np.int32)
        self.lib.orc_cv_rng_uniform(C.c_uint64(seed), a, b, count, _p(out))
        return out
