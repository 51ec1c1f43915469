"""ctypes mirror of the reference's operator interface over libmyslam_hip.so (include/myslam_hip.h).

Names follow the reference: ORBextractor (include/myslam/ORBextractor.h), DeepLCD
(include/myslam/deeplcd.h), the BFMatcher-style `hamming_match`, `triangulation` (algorithm.h), the
loop database of LoopClosing::DetectLoop and the BA block build of Backend::OptimizeActiveMap.
Host-buffer calls take numpy arrays; *_batch calls take raw device pointers (e.g. torch.Tensor.data_ptr()).
There is no CPU fallback: a missing library or GPU raises.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmyslam_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED = 0, -1, -2, -3, -4
LCD_DIM = 1064

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class MyslamError(RuntimeError):
    def __init__(self, where, code):
        names = {-1: "INVALID", -2: "HIP", -3: "CAPACITY", -4: "UNSUPPORTED"}
        super().__init__(f"{where} failed: {code} ({names.get(code, '?')})")
        self.code = code


HEADER_PATH = os.path.join(_HERE, "..", "include", "myslam_hip.h")
CALC_LAYER_DTYPE = np.dtype([("type", "<i4"), ("num_output", "<i4"), ("kernel", "<i4"), ("stride", "<i4"), ("pad", "<i4"),
                             ("local_size", "<i4"), ("alpha", "<f4"), ("beta", "<f4"), ("k", "<f4")])     # myslam_calc_layer
assert CALC_LAYER_DTYPE.itemsize == 36
CALC_CONV, CALC_RELU, CALC_POOL_MAX, CALC_LRN = 1, 2, 3, 4
CAND_DTYPE = np.dtype([("best_id", "<u8"), ("max_score", "<f4"), ("cnt", "<i4")])      # myslam_lcd_candidate
assert CAND_DTYPE.itemsize == 16
OWNED_DTYPE = np.dtype([("pre_best_id", "<u8"), ("pre_max_score", "<f4"), ("pre_cnt", "<i4"),
                        ("suf_best_id", "<u8"), ("suf_max_score", "<f4"), ("suf_cnt", "<i4")])      # myslam_lcd_owned_candidate
assert OWNED_DTYPE.itemsize == 32

_lib = None
_SCALARS = {"int": C.c_int, "float": C.c_float, "double": C.c_double, "size_t": C.c_size_t, "uint64_t": C.c_uint64, "long": C.c_long,
            "int32_t": C.c_int32, "uint8_t": C.c_uint8}


def header_prototypes(path=HEADER_PATH):
    """{function name: (return type, [parameter types])} parsed from include/myslam_hip.h — every pointer is 'ptr'."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    out = {}
    for m in re.finditer(r"\b(int|float|size_t|const char\s*\*)\s+(myslam_\w+)\s*\(([^()]*)\)\s*;", txt):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if params and params != "void":
            for prm in params.split(","):
                prm = prm.strip()
                if "*" in prm:
                    types.append("ptr")
                else:
                    toks = [t for t in prm.split() if t != "const"]
                    types.append(toks[0])
        out[name] = ("char*" if "char" in ret else ret, types)
    return out


def lib():
    """Load libmyslam_hip.so (fails loudly when it has not been built) and declare every prototype of the header, so that 64-bit
    pointers, size_t and floating-point arguments never depend on how a call site wraps them."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() / build.py first (no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (ret, types) in header_prototypes().items():
            fn = getattr(L, name)              # AttributeError = the library does not export what the header declares
            fn.restype = {"int": C.c_int, "float": C.c_float, "size_t": C.c_size_t, "char*": C.c_char_p}[ret]
            fn.argtypes = [C.c_void_p if t == "ptr" else _SCALARS[t] for t in types]
        _lib = L
    return _lib


def _check(code, where):
    if code != OK:
        raise MyslamError(where, code)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    return lib().myslam_hip_device_count()


def version():
    """'myslam_hip <version> (gfx950) build <digest of the sources>'"""
    return lib().myslam_hip_version().decode()


def build_id():
    return version().rsplit(" ", 1)[-1]


def shader_clock_mhz(stream=0, spin_us=200.0):
    """the shader clock right now, measured on the device (one wave spins for spin_us); synchronises `stream`"""
    v = C.c_float()
    _check(lib().myslam_prof_shader_clock_mhz(stream, float(spin_us), C.byref(v)), "myslam_prof_shader_clock_mhz")
    return float(v.value)


# ---------------------------------------------------------------------------------- profiling
def prof_enable(on=True):
    lib().myslam_prof_enable(1 if on else 0)


def prof_reset():
    lib().myslam_prof_reset()


def prof_read():
    """{kernel name: (total ms, launches)} since the last reset (synchronises)."""
    L = lib()
    out = {}
    for i in range(L.myslam_prof_count()):
        name = C.c_char_p(); ms = C.c_double(); n = C.c_long()
        L.myslam_prof_get(i, C.byref(name), C.byref(ms), C.byref(n))
        out[name.value.decode()] = (ms.value, n.value)
    return out


# ---------------------------------------------------------------------------------- ORB
class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) — ORBextractor.h:55-56."""

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, stream=None):
        self._h = C.c_void_p()
        _check(lib().myslam_orb_create(C.byref(self._h), int(nfeatures), C.c_float(scaleFactor), int(nlevels),
                                       int(iniThFAST), int(minThFAST)), "myslam_orb_create")
        self.nfeatures, self.nlevels = nfeatures, nlevels
        if stream is not None:
            self.set_stream(stream)

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.myslam_orb_destroy(self._h)
            self._h = C.c_void_p()

    def set_stream(self, stream_ptr):
        _check(lib().myslam_orb_set_stream(self._h, C.c_void_p(stream_ptr)), "myslam_orb_set_stream")

    def set_fast_event(self, event_ptr):
        """Record the hipEvent_t `event_ptr` (0 = off) on the handle's stream after the FAST stage of every following batched call."""
        _check(lib().myslam_orb_set_fast_event(self._h, C.c_void_p(event_ptr or None)), "myslam_orb_set_fast_event")

    def set_fast_gate(self, event_ptr):
        """Make the handle's stream wait for the hipEvent_t `event_ptr` (0 = off) before the FAST stage of every following batched call."""
        _check(lib().myslam_orb_set_fast_gate(self._h, C.c_void_p(event_ptr or None)), "myslam_orb_set_fast_gate")

    OPT_FAST_MODE, OPT_INTERNAL_STREAM, OPT_STOP_AFTER, OPT_COPY_INPUT, OPT_BLUR_MFMA, OPT_SIDE_BLOCKS_PER_CU = 1, 2, 3, 4, 5, 6

    def set_option(self, option, value):
        """scheduling / debugging knobs (myslam_orb_set_option): none of them changes a result"""
        _check(lib().myslam_orb_set_option(self._h, int(option), int(value)), "myslam_orb_set_option")

    def set_gauss_taps(self, q7=None):
        q = None if q7 is None else np.ascontiguousarray(q7, np.int32)
        _check(lib().myslam_orb_set_gauss_taps(self._h, _p(q) if q is not None else None), "myslam_orb_set_gauss_taps")

    def tables(self):
        n = self.nlevels
        sc = np.zeros(n, np.float32); isc = np.zeros(n, np.float32); npl = np.zeros(n, np.int32); um = np.zeros(16, np.int32)
        _check(lib().myslam_orb_get_tables(self._h, _p(sc), _p(isc), _p(npl), _p(um)), "myslam_orb_get_tables")
        return sc, isc, npl, um

    def max_keypoints(self, rows=None, cols=None):
        if rows is not None:
            return lib().myslam_orb_max_keypoints_for(self._h, int(rows), int(cols))
        return lib().myslam_orb_max_keypoints(self._h)

    @staticmethod
    def _img(img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.ndim == 2
        return img

    def DetectAndCompute(self, image, mask=None, cap=None):
        img = self._img(image)
        cap = cap or self.max_keypoints(*img.shape)
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
        m = self._img(mask) if mask is not None else None
        _check(lib().myslam_orb_detect_and_compute(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                                   _p(m), m.strides[0] if m is not None else 0,
                                                   _p(kps), _p(desc), cap, C.byref(n)), "myslam_orb_detect_and_compute")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def Detect(self, image, mask=None, cap=None):
        img = self._img(image)
        cap = cap or self.max_keypoints(*img.shape)
        kps = np.zeros(cap, KP_DTYPE); n = C.c_int()
        m = self._img(mask) if mask is not None else None
        _check(lib().myslam_orb_detect(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                       _p(m), m.strides[0] if m is not None else 0, _p(kps), cap, C.byref(n)),
               "myslam_orb_detect")
        return kps[:n.value].copy()

    def ScreenAndComputeKPsParams(self, image, keypoints):
        """returns (out_keypoints, keypoints as modified in place by the call)"""
        img = self._img(image)
        kin = np.ascontiguousarray(keypoints, KP_DTYPE).copy()
        kout = np.zeros(max(len(kin), 1), KP_DTYPE); n = C.c_int()
        _check(lib().myslam_orb_screen_and_compute_params(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                                          _p(kin), len(kin), _p(kout), len(kout), C.byref(n)),
               "myslam_orb_screen_and_compute_params")
        return kout[:n.value].copy(), kin

    def CalcDescriptors(self, image, keypoints):
        img = self._img(image)
        k = np.ascontiguousarray(keypoints, KP_DTYPE)
        desc = np.zeros((len(k), 32), np.uint8)
        _check(lib().myslam_orb_calc_descriptors(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                                 _p(k), len(k), _p(desc)), "myslam_orb_calc_descriptors")
        return desc

    # device-resident batch (pointers are ints)
    def detect_and_compute_batch(self, d_imgs, batch, rows, cols, step, img_stride, d_kps, d_desc, d_counts, d_status, cap,
                                 d_masks=0):
        _check(lib().myslam_orb_detect_and_compute_batch(self._h, C.c_void_p(d_imgs), batch, rows, cols, step,
                                                         C.c_size_t(img_stride), C.c_void_p(d_masks or None),
                                                         C.c_void_p(d_kps), C.c_void_p(d_desc), C.c_void_p(d_counts),
                                                         C.c_void_p(d_status or None), cap),
               "myslam_orb_detect_and_compute_batch")

    def detect_batch(self, d_imgs, batch, rows, cols, step, img_stride, d_kps, d_counts, d_status, cap, d_masks=0):
        _check(lib().myslam_orb_detect_batch(self._h, C.c_void_p(d_imgs), batch, rows, cols, step, C.c_size_t(img_stride),
                                             C.c_void_p(d_masks or None), C.c_void_p(d_kps), C.c_void_p(d_counts),
                                             C.c_void_p(d_status or None), cap), "myslam_orb_detect_batch")

    # stage taps
    def debug_pyramid(self, image, level, blurred=False):
        img = self._img(image)
        w = C.c_int(); h = C.c_int()
        _check(lib().myslam_orb_debug_pyramid(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0], level,
                                              1 if blurred else 0, None, 0, C.byref(w), C.byref(h)), "debug_pyramid(size)")
        out = np.zeros((h.value, w.value), np.uint8)
        _check(lib().myslam_orb_debug_pyramid(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0], level,
                                              1 if blurred else 0, _p(out), out.strides[0], C.byref(w), C.byref(h)),
               "debug_pyramid")
        return out

    def debug_candidates(self, image, level, mask=None):
        img = self._img(image)
        cap = 65536
        xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32); n = C.c_int()
        m = self._img(mask) if mask is not None else None
        _check(lib().myslam_orb_debug_candidates(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                                 _p(m), m.strides[0] if m is not None else 0, level,
                                                 _p(xs), _p(ys), _p(sc), cap, C.byref(n)), "debug_candidates")
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def fast_statistics(self, level):
        """what the last grid-FAST launch measured on `level`: (statistic, pixel pairs, path) — myslam_orb_debug_readback(what = 6)"""
        out = np.zeros(4, np.uint32)
        _check(lib().myslam_orb_debug_readback(self._h, 6, 0, level, _p(out), out.nbytes, 0), "debug_readback")
        return int(out[0]), int(out[1]), int(out[2])


# ---------------------------------------------------------------------------------- Hamming / triangulation
def hamming_match(query, train):
    """cv::BFMatcher(NORM_HAMMING).match(query, train) -> (trainIdx, distance) per query row."""
    q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
    idx = np.zeros(len(q), np.int32); dist = np.zeros(len(q), np.int32)
    _check(lib().myslam_hamming_match(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist)), "myslam_hamming_match")
    return idx, dist


def hamming_match_batch(d_q, d_nq, d_t, d_nt, batch, cap, d_idx, d_dist, stream=0):
    _check(lib().myslam_hamming_match_batch(C.c_void_p(d_q), C.c_void_p(d_nq), C.c_void_p(d_t), C.c_void_p(d_nt), batch, cap,
                                            C.c_void_p(d_idx), C.c_void_p(d_dist), C.c_void_p(stream or None)),
           "myslam_hamming_match_batch")


def hamming_filter(dist):
    d = np.ascontiguousarray(dist, np.int32)
    keep = np.zeros(len(d), np.uint8); mn = C.c_int()
    _check(lib().myslam_hamming_filter(_p(d), len(d), _p(keep), C.byref(mn)), "myslam_hamming_filter")
    return keep.astype(bool), mn.value


def expand_pyramid_keypoints(feats, nlevels=8):
    """LoopClosing::ProcessNewKF (loopclosing.cpp:94-105): nlevels pyramid key-points per feature, class_id = feature index"""
    f = np.ascontiguousarray(feats, KP_DTYPE)
    out = np.zeros(len(f) * nlevels, KP_DTYPE)
    _check(lib().myslam_expand_pyramid_keypoints(_p(f), len(f), nlevels, _p(out)), "myslam_expand_pyramid_keypoints")
    return out


def match_feature_pairs(train_idx, dist, loop_pyr_kps, cur_pyr_kps):
    """LoopClosing::MatchFeatures (loopclosing.cpp:175-194): (current feature id, loop feature id) pairs, std::set order"""
    ti = np.ascontiguousarray(train_idx, np.int32); d = np.ascontiguousarray(dist, np.int32)
    lk = np.ascontiguousarray(loop_pyr_kps, KP_DTYPE); ck = np.ascontiguousarray(cur_pyr_kps, KP_DTYPE)
    assert len(ti) == len(d) == len(lk)
    pairs = np.zeros((max(len(ti), 1), 2), np.int32); n = C.c_int()
    _check(lib().myslam_match_feature_pairs(_p(ti), _p(d), len(ti), _p(lk), _p(ck), len(ck), _p(pairs), C.byref(n)), "myslam_match_feature_pairs")
    return pairs[:n.value].copy()


def triangulate_stereo(xl, yl, xr, yr, fx, fy, cx, cy, baseline):
    xl, yl, xr, yr = [np.ascontiguousarray(a, np.float32) for a in (xl, yl, xr, yr)]
    n = len(xl)
    xyz = np.zeros((n, 3)); ok = np.zeros(n, np.uint8)
    _check(lib().myslam_triangulate_stereo(_p(xl), _p(yl), _p(xr), _p(yr), n, C.c_double(fx), C.c_double(fy), C.c_double(cx),
                                           C.c_double(cy), C.c_double(baseline), _p(xyz), _p(ok)), "myslam_triangulate_stereo")
    return xyz, ok.astype(bool)


def triangulate_stereo_batch(d_kl, d_kr, d_match, d_nl, batch, cap, K, baseline, d_xyz, d_ok, stream=0):
    _check(lib().myslam_triangulate_stereo_batch(C.c_void_p(d_kl), C.c_void_p(d_kr), C.c_void_p(d_match), C.c_void_p(d_nl),
                                                 batch, cap, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]),
                                                 C.c_double(K[3]), C.c_double(baseline), C.c_void_p(d_xyz), C.c_void_p(d_ok),
                                                 C.c_void_p(stream or None)), "myslam_triangulate_stereo_batch")


def hamming_match_triangulate_batch(d_q, d_nq, d_t, d_nt, d_kl, d_kr, batch, cap, K, baseline, d_idx, d_dist, d_xyz, d_ok, stream=0):
    """myslam_hamming_match_triangulate_batch: match + triangulation of every match, one launch for fewer than 16 pairs."""
    _check(lib().myslam_hamming_match_triangulate_batch(C.c_void_p(d_q), C.c_void_p(d_nq), C.c_void_p(d_t), C.c_void_p(d_nt), C.c_void_p(d_kl),
                                                        C.c_void_p(d_kr), batch, cap, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]),
                                                        C.c_double(K[3]), C.c_double(baseline), C.c_void_p(d_idx), C.c_void_p(d_dist),
                                                        C.c_void_p(d_xyz), C.c_void_p(d_ok), C.c_void_p(stream or None)),
           "myslam_hamming_match_triangulate_batch")


# ---------------------------------------------------------------------------------- DeepLCD
def calc_default_layers():
    """the SURVEY A.6 layer list as CALC_LAYER_DTYPE records"""
    n = lib().myslam_lcd_default_layers(None, 0)
    L = np.zeros(n, CALC_LAYER_DTYPE)
    assert lib().myslam_lcd_default_layers(_p(L), n) == n
    return L


def calc_parse_caffe(prototxt_path, caffemodel_path):
    """(layer records, flat weights) of a deploy.prototxt + .caffemodel pair — host only, no device needed"""
    nl = C.c_int(); nw = C.c_size_t()
    _check(lib().myslam_calc_parse_caffe(prototxt_path.encode(), caffemodel_path.encode(), None, 0, C.byref(nl), None, 0, C.byref(nw)),
           "myslam_calc_parse_caffe")
    L = np.zeros(nl.value, CALC_LAYER_DTYPE); w = np.zeros(nw.value, np.float32)
    _check(lib().myslam_calc_parse_caffe(prototxt_path.encode(), caffemodel_path.encode(), _p(L), len(L), C.byref(nl), _p(w), w.size, C.byref(nw)),
           "myslam_calc_parse_caffe")
    return L, w


class DeepLCD:
    """DeepLCD(weights) — include/myslam/deeplcd.h:33.  `weights` = flat f32 blob of the SURVEY A.6 layer list; `layers` = explicit
    CALC_LAYER_DTYPE records; DeepLCD.from_caffe(prototxt, caffemodel) = the reference's constructor arguments."""
    OPT_GENERIC_KERNELS = 1
    OPT_CONV2_BF16X6 = 2
    OPT_SKIP_KERNELS = 3

    def __init__(self, weights=None, stream=None, layers=None, caffe=None, path=None):
        self._h = C.c_void_p()
        if caffe is not None:
            _check(lib().myslam_lcd_create_from_caffe(C.byref(self._h), caffe[0].encode(), caffe[1].encode()), "myslam_lcd_create_from_caffe")
        elif path is not None:
            _check(lib().myslam_lcd_create_from_file(C.byref(self._h), path.encode()), "myslam_lcd_create_from_file")
        else:
            w = np.ascontiguousarray(weights, np.float32).ravel()
            if layers is not None:
                L = np.ascontiguousarray(layers, CALC_LAYER_DTYPE)
                _check(lib().myslam_lcd_create_from_layers(C.byref(self._h), _p(L), len(L), _p(w), w.size), "myslam_lcd_create_from_layers")
            else:
                _check(lib().myslam_lcd_create(C.byref(self._h), _p(w), w.size), "myslam_lcd_create")
        if stream is not None:
            _check(lib().myslam_lcd_set_stream(self._h, C.c_void_p(stream)), "myslam_lcd_set_stream")

    @classmethod
    def from_caffe(cls, prototxt_path="calc_model/deploy.prototxt", caffemodel_path="calc_model/calc.caffemodel", stream=None):
        return cls(caffe=(prototxt_path, caffemodel_path), stream=stream)

    def set_option(self, option, value):
        _check(lib().myslam_lcd_set_option(self._h, int(option), int(value)), "myslam_lcd_set_option")

    def uses_fused_kernels(self):
        return lib().myslam_lcd_uses_fused_kernels(self._h) == 1

    def conv2_products(self):
        """3: conv2 runs as f16 x 3 on the matrix cores, 6: as bf16 x 6, 0: generic kernels"""
        return lib().myslam_lcd_conv2_products(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.myslam_lcd_destroy(self._h)
            self._h = C.c_void_p()

    def calcDescrOriginalImg(self, image, blur_in_place=True):
        """returns (descriptor[1064], image after the call) — the reference blurs the caller's image in place."""
        img = np.ascontiguousarray(image, np.uint8).copy()
        d = np.zeros(LCD_DIM, np.float32)
        _check(lib().myslam_lcd_calc_descr_original_img(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                                        1 if blur_in_place else 0, _p(d)), "myslam_lcd_calc_descr_original_img")
        return d, img

    def calcDescr(self, im160x120):
        img = np.ascontiguousarray(im160x120, np.uint8)
        assert img.shape == (120, 160)
        d = np.zeros(LCD_DIM, np.float32)
        _check(lib().myslam_lcd_calc_descr(self._h, _p(img), img.strides[0], _p(d)), "myslam_lcd_calc_descr")
        return d

    @staticmethod
    def score(d1, d2):
        a = np.ascontiguousarray(d1, np.float32); b = np.ascontiguousarray(d2, np.float32)
        return float(lib().myslam_lcd_score(_p(a), _p(b)))

    def describe_batch(self, d_imgs, batch, rows, cols, step, img_stride, d_out, blur_in_place=False):
        _check(lib().myslam_lcd_describe_batch(self._h, C.c_void_p(d_imgs), batch, rows, cols, step, C.c_size_t(img_stride),
                                               1 if blur_in_place else 0, C.c_void_p(d_out)), "myslam_lcd_describe_batch")

    def debug_forward(self, x120x160, stage):
        x = np.ascontiguousarray(x120x160, np.float32)
        sizes = [62 * 82 * 64, 31 * 41 * 64, 32 * 42 * 128, 16 * 21 * 128, LCD_DIM]
        out = np.zeros(sizes[stage], np.float32)
        _check(lib().myslam_lcd_debug_forward(self._h, _p(x), _p(out), stage, C.c_size_t(out.size)), "myslam_lcd_debug_forward")
        return out


class LoopDatabase:
    """LoopClosing::_mvDatabase + DetectLoop()/AddToDatabase() — loopclosing.cpp:124-161, 651-659."""

    def __init__(self, capacity, stream=None):
        self._h = C.c_void_p()
        _check(lib().myslam_lcddb_create(C.byref(self._h), int(capacity)), "myslam_lcddb_create")
        if stream is not None:
            _check(lib().myslam_lcddb_set_stream(self._h, C.c_void_p(stream)), "myslam_lcddb_set_stream")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.myslam_lcddb_destroy(self._h)
            self._h = C.c_void_p()

    def __len__(self):
        return lib().myslam_lcddb_size(self._h)

    def capacity(self):
        return lib().myslam_lcddb_capacity(self._h)

    def reserve(self, rows):
        _check(lib().myslam_lcddb_reserve(self._h, int(rows)), "myslam_lcddb_reserve")

    def AddToDatabase(self, kf_id, descr):
        d = np.ascontiguousarray(descr, np.float32)
        _check(lib().myslam_lcddb_append(self._h, C.c_uint64(kf_id), _p(d)), "myslam_lcddb_append")

    def append_batch(self, ids, d_descr, n):
        ids = np.ascontiguousarray(ids, np.uint64)
        _check(lib().myslam_lcddb_append_batch(self._h, _p(ids), C.c_void_p(d_descr), n), "myslam_lcddb_append_batch")

    def append_batch_async(self, ids, d_descr, n, stream):
        """AddToDatabase inside a pipelined step: copies enqueued on `stream`, no wait (MyslamError(CAPACITY) when the rows do not fit: reserve ahead)"""
        ids = np.ascontiguousarray(ids, np.uint64)
        _check(lib().myslam_lcddb_append_batch_async(self._h, _p(ids), C.c_void_p(d_descr), n, C.c_void_p(stream or None)), "myslam_lcddb_append_batch_async")

    def query(self, descr, cur_id, thr_low=0.92):
        d = np.ascontiguousarray(descr, np.float32)
        best = C.c_uint64(); mx = C.c_float(); cnt = C.c_int()
        _check(lib().myslam_lcddb_query(self._h, _p(d), C.c_uint64(cur_id), C.c_float(thr_low), C.byref(best), C.byref(mx),
                                        C.byref(cnt)), "myslam_lcddb_query")
        return best.value, mx.value, cnt.value

    def DetectLoop(self, descr, cur_id, thr_high=0.94, thr_low=0.92):
        """bool + candidate id, the decision rule of loopclosing.cpp:147."""
        best, mx, cnt = self.query(descr, cur_id, thr_low)
        if mx < thr_high or cnt > 3:
            return False, None
        return True, best

    def query_batch(self, d_q, cur_ids, nq, d_best, d_max, d_cnt, thr_low=0.92):
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_query_batch(self._h, C.c_void_p(d_q), _p(cur), nq, C.c_float(thr_low), C.c_void_p(d_best),
                                              C.c_void_p(d_max), C.c_void_p(d_cnt)), "myslam_lcddb_query_batch")


    def update_query_limits(self, cur_ids):
        """new cur_ids (or appended rows) for a query that was recorded into a HIP graph (StepGraph); MyslamError(CAPACITY) = record again"""
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_update_query_limits(self._h, _p(cur), len(cur)), "myslam_lcddb_update_query_limits")

    def query_batch_sharded(self, d_q, cur_ids, nq, d_cand, thr_low=0.92):
        """per-shard records (myslam_lcd_candidate, 16 bytes each, device memory) for the multi-GPU exchange"""
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_query_batch_sharded(self._h, d_q, _p(cur), nq, thr_low, d_cand), "myslam_lcddb_query_batch_sharded")

    def query_batch_owned(self, d_q, cur_ids, nq, d_cand, thr_low=0.92):
        """records of a shard whose ids interleave with the other shards' (myslam_lcd_owned_candidate, 32 bytes each, device memory)"""
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_query_batch_owned(self._h, d_q, _p(cur), nq, thr_low, d_cand), "myslam_lcddb_query_batch_owned")

    def generation(self):
        """number of times the descriptor matrix has moved (growth): recorded steps are valid for the generation they were recorded in"""
        return lib().myslam_lcddb_generation(self._h)

    def context(self, stream):
        """A query context on `stream` (myslam_lcddb_query_ctx): several streams scan this ONE database concurrently."""
        return LoopQueryContext(self, stream)


class LoopQueryContext:
    """myslam_lcddb_query_ctx: the per-stream half of a loop database (row-limit staging, partial results, recorded-step state).
    LoopClosing::_mvDatabase is one std::map per process (loopclosing.h:120): L streams of one GPU scan it through L contexts."""

    def __init__(self, db, stream):
        self._db = db                       # keeps the database alive: contexts are destroyed before it
        self._h = C.c_void_p()
        _check(lib().myslam_lcddb_query_ctx_create(C.byref(self._h), db._h, C.c_void_p(stream)), "myslam_lcddb_query_ctx_create")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None and getattr(self._db, "_h", None) and self._db._h.value:
            _lib.myslam_lcddb_query_ctx_destroy(self._h)
        self._h = C.c_void_p()

    def query_batch(self, d_q, cur_ids, nq, d_best, d_max, d_cnt, thr_low=0.92):
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_ctx_query_batch(self._h, C.c_void_p(d_q), _p(cur), nq, C.c_float(thr_low), C.c_void_p(d_best),
                                                  C.c_void_p(d_max), C.c_void_p(d_cnt)), "myslam_lcddb_ctx_query_batch")

    def update_query_limits(self, cur_ids):
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_ctx_update_query_limits(self._h, _p(cur), len(cur)), "myslam_lcddb_ctx_update_query_limits")

    def query_batch_sharded(self, d_q, cur_ids, nq, d_cand, thr_low=0.92):
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_ctx_query_batch_sharded(self._h, d_q, _p(cur), nq, thr_low, d_cand), "myslam_lcddb_ctx_query_batch_sharded")


    def query_batch_owned(self, d_q, cur_ids, nq, d_cand, thr_low=0.92):
        cur = np.ascontiguousarray(cur_ids, np.uint64)
        _check(lib().myslam_lcddb_ctx_query_batch_owned(self._h, d_q, _p(cur), nq, thr_low, d_cand), "myslam_lcddb_ctx_query_batch_owned")


class StepGraph:
    """One batched step recorded into a HIP graph (myslam_graph_begin / _end) and replayed with one launch.
        g = StepGraph.record(origin_stream, [side streams...], body)      # body() issues the step's *_batch calls on those streams
        g.launch(origin_stream)"""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def record(cls, origin, sides, body):
        arr = (C.c_void_p * max(1, len(sides)))(*[C.c_void_p(s) for s in sides])
        _check(lib().myslam_graph_begin(C.c_void_p(origin), arr, len(sides)), "myslam_graph_begin")
        err = None
        try:
            body()
        except BaseException as e:          # the capture must be closed whatever the body did
            err = e
        h = C.c_void_p()
        rc = lib().myslam_graph_end(C.c_void_p(origin), arr, len(sides), C.byref(h))
        if err is not None:
            if rc == OK:
                lib().myslam_graph_destroy(h)
            raise err
        _check(rc, "myslam_graph_end")
        return cls(h)

    def launch(self, stream):
        _check(lib().myslam_graph_launch(self._h, C.c_void_p(stream)), "myslam_graph_launch")

    def node_count(self):
        return lib().myslam_graph_node_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.myslam_graph_destroy(self._h)
            self._h = C.c_void_p()


def lcd_merge_candidates(gathered):
    """gathered: [nshards, nq] CAND_DTYPE records in ascending id-range order (host) -> (best_id u64, max_score f32, cnt i32)"""
    g = np.ascontiguousarray(gathered, CAND_DTYPE)
    assert g.ndim == 2
    ns, nq = g.shape
    best = np.zeros(nq, np.uint64); mx = np.zeros(nq, np.float32); cnt = np.zeros(nq, np.int32)
    _check(lib().myslam_lcd_merge_candidates(_p(g), ns, nq, _p(best), _p(mx), _p(cnt)), "myslam_lcd_merge_candidates")
    return best, mx, cnt


def lcd_merge_candidates_device(d_gathered, nshards, nq, d_best, d_max, d_cnt, stream=0):
    _check(lib().myslam_lcd_merge_candidates_device(d_gathered, nshards, nq, d_best, d_max, d_cnt, stream or None),
           "myslam_lcd_merge_candidates_device")


def lcd_merge_owned_candidates(gathered):
    """gathered: [nshards, nq] OWNED_DTYPE records, any shard order (host) -> (best_id u64, max_score f32, cnt i32) of ONE scan of the whole map"""
    g = np.ascontiguousarray(gathered, OWNED_DTYPE)
    assert g.ndim == 2
    ns, nq = g.shape
    best = np.zeros(nq, np.uint64); mx = np.zeros(nq, np.float32); cnt = np.zeros(nq, np.int32)
    _check(lib().myslam_lcd_merge_owned_candidates(_p(g), ns, nq, _p(best), _p(mx), _p(cnt)), "myslam_lcd_merge_owned_candidates")
    return best, mx, cnt


def lcd_merge_owned_candidates_device(d_gathered, nshards, nq, d_best, d_max, d_cnt, stream=0):
    _check(lib().myslam_lcd_merge_owned_candidates_device(d_gathered, nshards, nq, d_best, d_max, d_cnt, stream or None),
           "myslam_lcd_merge_owned_candidates_device")


# ---------------------------------------------------------------------------------- BA
def ba_build(poses, points, edge_pose, edge_pt, obs, fixed, K, delta=5.991):
    poses = np.ascontiguousarray(poses, np.float64); points = np.ascontiguousarray(points, np.float64)
    ep = np.ascontiguousarray(edge_pose, np.int32); el = np.ascontiguousarray(edge_pt, np.int32)
    obs = np.ascontiguousarray(obs, np.float64)
    fixed = np.ascontiguousarray(fixed, np.uint8) if fixed is not None else None
    P, L, E = len(poses), len(points), len(ep)
    Hpp = np.zeros((P, 6, 6)); Hll = np.zeros((L, 3, 3)); Hpl = np.zeros((E, 6, 3)); bp = np.zeros((P, 6)); bl = np.zeros((L, 3))
    chi2 = np.zeros(E)
    _check(lib().myslam_ba_build(_p(poses), P, _p(points), L, _p(ep), _p(el), _p(obs), E, _p(fixed), C.c_double(K[0]),
                                 C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), C.c_double(delta), _p(Hpp), _p(Hll),
                                 _p(Hpl), _p(bp), _p(bl), _p(chi2)), "myslam_ba_build")
    return Hpp, Hll, Hpl, bp, bl, chi2


def ba_build_batch(d_poses, d_points, d_ep, d_el, d_obs, d_fixed, d_sizes, nwin, maxP, maxL, maxE, K, delta,
                   d_Hpp, d_Hll, d_Hpl, d_bp, d_bl, d_chi2, stream=0):
    _check(lib().myslam_ba_build_batch(C.c_void_p(d_poses), C.c_void_p(d_points), C.c_void_p(d_ep), C.c_void_p(d_el),
                                       C.c_void_p(d_obs), C.c_void_p(d_fixed or None), C.c_void_p(d_sizes), nwin, maxP, maxL, maxE,
                                       C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), C.c_double(delta),
                                       C.c_void_p(d_Hpp), C.c_void_p(d_Hll), C.c_void_p(d_Hpl), C.c_void_p(d_bp), C.c_void_p(d_bl),
                                       C.c_void_p(d_chi2), C.c_void_p(stream or None)), "myslam_ba_build_batch")


def ba_optimize(poses, points, edge_pose, edge_pt, obs, fixed, K, delta=5.991, iters=10):
    """optimizer.optimize(iters) of Backend::OptimizeActiveMap (backend.cpp:212-214) on the device: returns
    (poses, points, robust chi2, iterations)."""
    poses = np.ascontiguousarray(poses, np.float64).copy(); points = np.ascontiguousarray(points, np.float64).copy()
    ep = np.ascontiguousarray(edge_pose, np.int32); el = np.ascontiguousarray(edge_pt, np.int32)
    obs = np.ascontiguousarray(obs, np.float64)
    fixed = np.ascontiguousarray(fixed, np.uint8) if fixed is not None else None
    chi = C.c_double(); it = C.c_int()
    _check(lib().myslam_ba_optimize(_p(poses), len(poses), _p(points), len(points), _p(ep), _p(el), _p(obs), len(ep), _p(fixed),
                                    C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), C.c_double(delta),
                                    int(iters), C.byref(chi), C.byref(it)), "myslam_ba_optimize")
    return poses, points, chi.value, it.value


def ba_optimize_batch(d_poses, d_points, d_ep, d_el, d_obs, d_fixed, d_sizes, nwin, maxP, maxL, maxE, K, delta, iters,
                      d_scratch, d_chi2, d_iters, d_status, stream=0):
    _check(lib().myslam_ba_optimize_batch(C.c_void_p(d_poses), C.c_void_p(d_points), C.c_void_p(d_ep), C.c_void_p(d_el),
                                          C.c_void_p(d_obs), C.c_void_p(d_fixed or None), C.c_void_p(d_sizes), nwin, maxP, maxL, maxE,
                                          C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), C.c_double(delta),
                                          int(iters), C.c_void_p(d_scratch), C.c_void_p(d_chi2), C.c_void_p(d_iters),
                                          C.c_void_p(d_status), C.c_void_p(stream or None)), "myslam_ba_optimize_batch")


def ba_flatten_window(active_kf_ids, mp_ids, mp_outlier, mp_first_observer_kf, obs_mp_id, obs_kf_id, obs_uv, obs_feat_outlier):
    """The graph-build rules of Backend::OptimizeActiveMap (src/backend.cpp:139-206) on the Map's tables (host only).
    Returns dict(pose_src, pt_src, edge_pose, edge_pt, edge_obs, edge_src, fixed): slots -> input rows, edges grouped by landmark."""
    kf = np.ascontiguousarray(active_kf_ids, np.uint64); mp = np.ascontiguousarray(mp_ids, np.uint64)
    mo = np.ascontiguousarray(mp_outlier, np.uint8); mf = np.ascontiguousarray(mp_first_observer_kf, np.uint64)
    om = np.ascontiguousarray(obs_mp_id, np.uint64); ok = np.ascontiguousarray(obs_kf_id, np.uint64)
    uv = np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 2); fo = np.ascontiguousarray(obs_feat_outlier, np.uint8)
    assert len(mp) == len(mo) == len(mf) and len(om) == len(ok) == len(uv) == len(fo)
    pose_src = np.zeros(max(len(kf), 1), np.int32); pt_src = np.zeros(max(len(mp), 1), np.int32); fixed = np.zeros(max(len(mp), 1), np.uint8)
    n = max(len(om), 1)
    ep = np.zeros(n, np.int32); el = np.zeros(n, np.int32); eo = np.zeros((n, 2), np.float64); es = np.zeros(n, np.int32)
    npt = C.c_int32(); ne = C.c_int32()
    _check(lib().myslam_ba_flatten_window(_p(kf), len(kf), _p(mp), _p(mo), _p(mf), len(mp), _p(om), _p(ok), _p(uv), _p(fo), len(om),
                                          _p(pose_src), _p(pt_src), C.byref(npt), _p(ep), _p(el), _p(eo), _p(es), C.byref(ne), _p(fixed)),
           "myslam_ba_flatten_window")
    L, E = npt.value, ne.value
    return dict(pose_src=pose_src[:len(kf)], pt_src=pt_src[:L], fixed=fixed[:L], edge_pose=ep[:E], edge_pt=el[:E], edge_obs=eo[:E], edge_src=es[:E])


def ba_optimize_active_map(poses, points, edge_pose, edge_pt, obs, fixed, K, delta=5.991, chi2_th=5.991, rounds=5, iters=10):
    """The solve stage of Backend::OptimizeActiveMap (src/backend.cpp:208-243).
    Returns (poses, points, edge_chi2, outlier flags, failed rounds, outlier count)."""
    poses = np.ascontiguousarray(poses, np.float64).copy(); points = np.ascontiguousarray(points, np.float64).copy()
    ep = np.ascontiguousarray(edge_pose, np.int32); el = np.ascontiguousarray(edge_pt, np.int32)
    obs = np.ascontiguousarray(obs, np.float64)
    fixed = np.ascontiguousarray(fixed, np.uint8) if fixed is not None else None
    chi = np.zeros(len(ep)); out = np.zeros(len(ep), np.uint8); r = C.c_int(); no = C.c_int()
    _check(lib().myslam_ba_optimize_active_map(_p(poses), len(poses), _p(points), len(points), _p(ep), _p(el), _p(obs), len(ep), _p(fixed),
                                               C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), C.c_double(delta),
                                               C.c_double(chi2_th), int(rounds), int(iters), _p(chi), _p(out), C.byref(r), C.byref(no)),
           "myslam_ba_optimize_active_map")
    return poses, points, chi, out, r.value, no.value


BA_OPT_LANDMARKS_IN_HBM = 1
BA_OPT_BUILD_POSE_ATOMICS = 2


def ba_set_option(option, value):
    """myslam_ba_set_option: process-wide scheduling knob of the solve kernel (include/myslam_hip.h)."""
    _check(lib().myslam_ba_set_option(int(option), int(value)), "myslam_ba_set_option")


def ba_optimize_active_map_batch(d_poses, d_points, d_ep, d_el, d_obs, d_fixed, d_sizes, nwin, maxP, maxL, maxE, K, delta, chi2_th,
                                 rounds, iters, d_scratch, d_edge_chi2, d_outlier, d_rounds, d_nout, d_status, stream=0):
    _check(lib().myslam_ba_optimize_active_map_batch(
        C.c_void_p(d_poses), C.c_void_p(d_points), C.c_void_p(d_ep), C.c_void_p(d_el), C.c_void_p(d_obs), C.c_void_p(d_fixed or None),
        C.c_void_p(d_sizes), nwin, maxP, maxL, maxE, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
        C.c_double(delta), C.c_double(chi2_th), int(rounds), int(iters), C.c_void_p(d_scratch), C.c_void_p(d_edge_chi2),
        C.c_void_p(d_outlier), C.c_void_p(d_rounds), C.c_void_p(d_nout), C.c_void_p(d_status), C.c_void_p(stream or None)),
        "myslam_ba_optimize_active_map_batch")


# ---------------------------------------------------------------------------------- LK tracker
class LKTracker:
    """cv::calcOpticalFlowPyrLK(..., Size(11,11), 3, TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW) as the reference
    calls it in Frontend::TrackLastFrame / FindFeaturesInRight (src/frontend.cpp:150-153, 358-361)."""

    def __init__(self, win=11, max_level=3, max_iters=30, eps=0.01, min_eig=1e-4, stream=None):
        self._h = C.c_void_p()
        _check(lib().myslam_lk_create(C.byref(self._h), int(win), int(max_level), int(max_iters), C.c_float(eps), C.c_float(min_eig)),
               "myslam_lk_create")
        if stream is not None:
            _check(lib().myslam_lk_set_stream(self._h, C.c_void_p(stream)), "myslam_lk_set_stream")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            lib().myslam_lk_destroy(self._h)
            self._h = C.c_void_p()

    def track(self, prev, nxt, prev_pts, next_pts):
        """host arrays; returns (next_pts, status bool, err)"""
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        npts = np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
        n = len(pp); st = np.zeros(max(n, 1), np.uint8); err = np.zeros(max(n, 1), np.float32)
        _check(lib().myslam_lk_track(self._h, _p(prev), _p(nxt), prev.shape[0], prev.shape[1], prev.strides[0], nxt.strides[0],
                                     _p(pp), _p(npts), n, _p(st), _p(err)), "myslam_lk_track")
        return npts, st[:n].astype(bool), err[:n]

    def track_cached(self, prev, prev_token, nxt, next_token, prev_pts, next_pts):
        """track() with the handle's two-image cache: a non-zero token names an image whose bytes never change (myslam_lk_track_cached)"""
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        npts = np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
        n = len(pp); st = np.zeros(max(n, 1), np.uint8); err = np.zeros(max(n, 1), np.float32)
        _check(lib().myslam_lk_track_cached(self._h, _p(prev), C.c_uint64(prev_token), _p(nxt), C.c_uint64(next_token), prev.shape[0], prev.shape[1],
                                            prev.strides[0], nxt.strides[0], _p(pp), _p(npts), n, _p(st), _p(err)), "myslam_lk_track_cached")
        return npts, st[:n].astype(bool), err[:n]

    def prefetch(self, img, token):
        """asynchronous upload + pyramid of an image the next track_cached() call will name by `token`; the array must stay alive until then"""
        img = np.ascontiguousarray(img, np.uint8)
        self._keep = img
        _check(lib().myslam_lk_prefetch(self._h, _p(img), C.c_uint64(token), img.shape[0], img.shape[1], img.strides[0]), "myslam_lk_prefetch")

    def track_batch(self, d_prev, d_next, batch, rows, cols, step, stride, d_prev_pts, d_next_pts, d_counts, cap, d_status, d_err=0):
        _check(lib().myslam_lk_track_batch(self._h, C.c_void_p(d_prev), C.c_void_p(d_next), batch, rows, cols, step, C.c_size_t(stride),
                                           C.c_void_p(d_prev_pts), C.c_void_p(d_next_pts), C.c_void_p(d_counts), cap,
                                           C.c_void_p(d_status), C.c_void_p(d_err or None)), "myslam_lk_track_batch")


# ---------------------------------------------------------------------------------- pose-only optimisation
def pose_only_optimize(pose, pts3d, obs, K, chi2_th=5.991, rounds=4, iters=10, pre_optimize=0):
    """The g2o part of Frontend::EstimateCurrentPose (src/frontend.cpp:176-276).  Returns (pose7, outlier flags, inlier count)."""
    pose = np.ascontiguousarray(pose, np.float64).copy()
    pts3d = np.ascontiguousarray(pts3d, np.float64).reshape(-1, 3); obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
    n = len(pts3d); out = np.zeros(max(n, 1), np.uint8); ni = C.c_int()
    _check(lib().myslam_pose_only_optimize(_p(pose), _p(pts3d), _p(obs), n, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]),
                                           C.c_double(K[3]), C.c_double(chi2_th), int(rounds), int(iters), int(pre_optimize), _p(out), C.byref(ni)),
           "myslam_pose_only_optimize")
    return pose, out[:n].astype(bool), ni.value


def pose_only_optimize_batch(d_poses, d_pts3d, d_obs, d_counts, batch, cap, K, chi2_th, rounds, iters, d_outlier, d_ninl, d_status, stream=0,
                             pre_optimize=0):
    _check(lib().myslam_pose_only_optimize_batch(C.c_void_p(d_poses), C.c_void_p(d_pts3d), C.c_void_p(d_obs), C.c_void_p(d_counts), batch, cap,
                                                 C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), C.c_double(chi2_th),
                                                 int(rounds), int(iters), int(pre_optimize), C.c_void_p(d_outlier), C.c_void_p(d_ninl), C.c_void_p(d_status),
                                                 C.c_void_p(stream or None)), "myslam_pose_only_optimize_batch")


# ---------------------------------------------------------------------------------- loop correction
def pose_graph_optimize(poses, fixed, edge_v0, edge_v1, meas, iters=20):
    """The g2o part of LoopClosing::PoseGraphOptimization (src/loopclosing.cpp:537-610).  poses (n,7) Tcw; meas[k] = the
    measured T[v0] * T[v1]^-1 of edge k.  Returns (poses, final chi2, iterations done)."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7).copy(); fixed = np.ascontiguousarray(fixed, np.uint8)
    e0 = np.ascontiguousarray(edge_v0, np.int32); e1 = np.ascontiguousarray(edge_v1, np.int32)
    meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 7)
    assert len(fixed) == len(poses) and len(e0) == len(e1) == len(meas)
    chi = C.c_double(); it = C.c_int()
    _check(lib().myslam_pose_graph_optimize(_p(poses), len(poses), _p(fixed), _p(e0), _p(e1), _p(meas), len(e0), int(iters), C.byref(chi), C.byref(it)),
           "myslam_pose_graph_optimize")
    return poses, chi.value, it.value


def correct_map_points(old_poses, new_poses, first_kf, points):
    """src/loopclosing.cpp:621-633: every map point keeps its camera-frame position in the key-frame that first observed it."""
    old_poses = np.ascontiguousarray(old_poses, np.float64).reshape(-1, 7); new_poses = np.ascontiguousarray(new_poses, np.float64).reshape(-1, 7)
    first_kf = np.ascontiguousarray(first_kf, np.int32); points = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
    assert old_poses.shape == new_poses.shape and len(first_kf) == len(points)
    _check(lib().myslam_correct_map_points(_p(old_poses), _p(new_poses), len(old_poses), _p(first_kf), _p(points), len(points)), "myslam_correct_map_points")
    return points


def loop_local_fusion(active_poses, cur, corrected_cur, first_active_kf, points):
    """LoopClosing::LoopLocalFusion (src/loopclosing.cpp:466-507), arithmetic part: returns (corrected active poses, corrected points)"""
    poses = np.ascontiguousarray(active_poses, np.float64).reshape(-1, 7).copy(); cc = np.ascontiguousarray(corrected_cur, np.float64)
    kf = np.ascontiguousarray(first_active_kf, np.int32); pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
    assert len(kf) == len(pts)
    _check(lib().myslam_loop_local_fusion(_p(poses), len(poses), int(cur), _p(cc), _p(kf), _p(pts), len(pts)), "myslam_loop_local_fusion")
    return poses, pts


def solve_pnp_ransac(pts3d, pts2d, K, iterations=100, reproj_error=5.991, confidence=0.99):
    """cv::solvePnPRansac as LoopClosing::ComputeCorrectPose calls it (src/loopclosing.cpp:262-268).
    Returns (pose7 Tcw, inlier flags, inlier count); raises MyslamError(UNSUPPORTED) when no model exists."""
    p3 = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); p2 = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
    assert len(p3) == len(p2)
    n = len(p3); pose = np.zeros(7); inl = np.zeros(max(n, 1), np.uint8); ni = C.c_int()
    _check(lib().myslam_solve_pnp_ransac(_p(p3), _p(p2), n, C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), int(iterations),
                                         C.c_double(reproj_error), C.c_double(confidence), _p(pose), _p(inl), C.byref(ni)), "myslam_solve_pnp_ransac")
    return pose, inl[:n].astype(bool), ni.value


# ---------------------------------------------------------------------------------- host-side formats (no device needed)
def read_png_gray(path):
    """cv::imread(path, IMREAD_GRAYSCALE) for the KITTI grey PNGs (app/run_kitti_stereo.cpp:66-67) -> uint8 [rows, cols]"""
    r = C.c_int(); c = C.c_int(); b = path.encode()
    _check(lib().myslam_io_read_png_gray(b, None, 0, C.byref(r), C.byref(c)), "myslam_io_read_png_gray")
    out = np.zeros((r.value, c.value), np.uint8)
    _check(lib().myslam_io_read_png_gray(b, _p(out), out.size, C.byref(r), C.byref(c)), "myslam_io_read_png_gray")
    return out


def load_images(sequence_path):
    """LoadImages (app/run_kitti_stereo.cpp:114-144) -> (left paths, right paths, timestamps)"""
    n = C.c_int(); b = sequence_path.encode()
    _check(lib().myslam_io_load_images(b, None, 0, C.byref(n)), "myslam_io_load_images")
    ts = np.zeros(max(n.value, 1), np.float64)
    _check(lib().myslam_io_load_images(b, _p(ts), len(ts), C.byref(n)), "myslam_io_load_images")
    buf = C.create_string_buffer(len(b) + 64)

    def path(i, right):
        _check(lib().myslam_io_image_path(b, i, right, buf, len(buf)), "myslam_io_image_path")
        return buf.value.decode()
    return [path(i, 0) for i in range(n.value)], [path(i, 1) for i in range(n.value)], ts[:n.value]


def save_trajectory(path, ids, timestamps, poses7_cw):
    """System::SaveTrajectory (src/system.cpp:153-180); poses are Tcw (KeyFrame::Pose()) as qx qy qz qw tx ty tz"""
    ids = np.ascontiguousarray(ids, np.uint64); ts = np.ascontiguousarray(timestamps, np.float64); ps = np.ascontiguousarray(poses7_cw, np.float64).reshape(-1, 7)
    assert len(ids) == len(ts) == len(ps)
    _check(lib().myslam_io_save_trajectory(path.encode(), _p(ids), _p(ts), _p(ps), len(ids)), "myslam_io_save_trajectory")


def save_loop_edges(path, cur_ids, cur_ts, cur_poses, loop_ids, loop_ts, loop_poses):
    """System::SaveLoopEdges (src/system.cpp:188-224)"""
    a = [np.ascontiguousarray(cur_ids, np.uint64), np.ascontiguousarray(cur_ts, np.float64), np.ascontiguousarray(cur_poses, np.float64).reshape(-1, 7),
         np.ascontiguousarray(loop_ids, np.uint64), np.ascontiguousarray(loop_ts, np.float64), np.ascontiguousarray(loop_poses, np.float64).reshape(-1, 7)]
    _check(lib().myslam_io_save_loop_edges(path.encode(), *[_p(x) for x in a], len(a[0])), "myslam_io_save_loop_edges")
