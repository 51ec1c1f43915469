#!/usr/bin/env python3
"""Pins the CPU oracle to the libraries the reference links — for a maintainer whose machine HAS them (this build environment has
neither OpenCV nor Caffe, so every OpenCV-derived definition of the oracle is "parity unpinned").

    python tools/dump_opencv_goldens.py            # needs `import cv2` (ideally OpenCV 3.4.8, the reference's version, README.md:28)
    python -m pytest tests/test_opencv_pin.py -q   # compares oracle/ with the fixtures written under tests/golden/opencv_*.npz

Every fixture stores its INPUTS next to OpenCV's outputs (and cv2.__version__), so the comparison needs nothing but the file.
Sections: resize (pyramid steps + the 160x120 CALC input), GaussianBlur (7x7 sigma 2 and sigma 0), FAST (score / NMS on cell-sized
ROIs, thresholds 20 and 7), fastAtan2, BFMatcher(NORM_HAMMING), calcOpticalFlowPyrLK (the reference's parameters), solvePnPRansac
(the reference's parameters) and, when `caffe` and calc_model/ are present, the CALC forward pass.
Where a section disagrees, the definitions that were chosen without a reference can be changed in one place on each side:
Gaussian taps — orc_set_gauss_taps / myslam_orb_set_gauss_taps; everything else — oracle/*.cpp and the kernel named in DESIGN.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PYR_SIZES = [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("OpenCV (cv2) is not importable here: run this on a machine that has it (the reference uses OpenCV 3.4.8)")
    synth = load_package().synth
    ver = np.array(cv2.__version__)
    L, R = synth.stereo_pair(0, 0)
    L1, _ = synth.stereo_pair(0, 1)
    tex = synth.random_image(1234, 240, 320)

    # 1. cv::resize INTER_LINEAR: the cascaded pyramid (ORBextractor.cpp:1243) and the CALC input (deeplcd.cpp:50)
    pyr = [L]
    for (w, h) in PYR_SIZES[1:]:
        pyr.append(cv2.resize(pyr[-1], (w, h), interpolation=cv2.INTER_LINEAR))
    small = cv2.resize(L, (160, 120))
    np.savez_compressed(os.path.join(OUT, "opencv_resize.npz"), version=ver, src=L, small=small, **{f"level{i}": p for i, p in enumerate(pyr)})

    # 2. cv::GaussianBlur 7x7: sigma 2 (ORBextractor.cpp:966) and sigma 0 (deeplcd.cpp:46), BORDER_REFLECT_101 / default
    g2 = [cv2.GaussianBlur(p, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101) for p in (pyr[0], pyr[3], pyr[7], tex)]
    g0 = cv2.GaussianBlur(L, (7, 7), 0)
    k2 = cv2.getGaussianKernel(7, 2).ravel(); k0 = cv2.getGaussianKernel(7, 0).ravel()
    np.savez_compressed(os.path.join(OUT, "opencv_blur.npz"), version=ver, src0=pyr[0], src1=pyr[3], src2=pyr[7], src3=tex,
                        out0=g2[0], out1=g2[1], out2=g2[2], out3=g2[3], lcd_src=L, lcd_out=g0, kernel_sigma2=k2, kernel_sigma0=k0)

    # 3. cv::FAST (9_16, nonmax on) on cell-sized ROIs as ComputeKeyPointsOctTree cuts them (:858-864) and on a whole level
    rois, outs = [], {}
    rng = np.random.default_rng(7)
    for i in range(12):
        y0 = int(rng.integers(16, 376 - 60)); x0 = int(rng.integers(16, 1241 - 60))
        roi = np.ascontiguousarray(L[y0:y0 + 38, x0:x0 + 37])
        rois.append(roi)
        for th in (20, 7):
            kps = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True).detect(roi, None)
            outs[f"roi{i}_th{th}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    for th in (20, 7):
        kps = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True).detect(pyr[4], None)
        outs[f"level4_th{th}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
        kps = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=False).detect(pyr[4], None)
        outs[f"level4_nonms_th{th}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    np.savez_compressed(os.path.join(OUT, "opencv_fast.npz"), version=ver, level4=pyr[4], **{f"roi{i}": r for i, r in enumerate(rois)}, **outs)

    # 4. cv::fastAtan2 (ORBextractor.cpp:54) on moment-like arguments
    yy = rng.integers(-200000, 200000, 20000).astype(np.float32); xx = rng.integers(-200000, 200000, 20000).astype(np.float32)
    yy[:8] = [0, 0, 1, -1, 5, -5, 0, 7]; xx[:8] = [0, 1, 0, 0, 5, 5, -3, -7]
    ang = np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(yy, xx)], np.float32)
    np.savez_compressed(os.path.join(OUT, "opencv_atan2.npz"), version=ver, y=yy, x=xx, angle=ang)

    # 5. BFMatcher(NORM_HAMMING).match (loopclosing.cpp:33,172)
    q = rng.integers(0, 256, (700, 32), dtype=np.uint8); t = rng.integers(0, 256, (1033, 32), dtype=np.uint8)
    t[500] = t[3]; q[0] = t[3]
    m = cv2.BFMatcher(cv2.NORM_HAMMING).match(q, t)
    np.savez_compressed(os.path.join(OUT, "opencv_hamming.npz"), version=ver, query=q, train=t,
                        train_idx=np.array([x.trainIdx for x in m], np.int32), dist=np.array([x.distance for x in m], np.float32),
                        query_idx=np.array([x.queryIdx for x in m], np.int32))

    # 6. calcOpticalFlowPyrLK as Frontend::TrackLastFrame / FindFeaturesInRight call it (frontend.cpp:150-153, 358-361)
    pts = cv2.goodFeaturesToTrack(L, 150, 0.01, 20).reshape(-1, 2).astype(np.float32)
    crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)
    res = {}
    for name, (a, b) in {"next": (L, L1), "right": (L, R)}.items():
        nxt, st, err = cv2.calcOpticalFlowPyrLK(a, b, pts, pts.copy(), winSize=(11, 11), maxLevel=3, criteria=crit, flags=cv2.OPTFLOW_USE_INITIAL_FLOW)
        res[name + "_pts"] = nxt.reshape(-1, 2); res[name + "_status"] = st.ravel(); res[name + "_err"] = err.ravel()
    np.savez_compressed(os.path.join(OUT, "opencv_lk.npz"), version=ver, prev=L, next=L1, right=R, pts=pts, **res)

    # 7. solvePnPRansac as LoopClosing::ComputeCorrectPose calls it (loopclosing.cpp:262-268)
    pw, uv, K, _, _ = synth.pnp_problem(200, 0.3, 0.5, seed=11)
    Kmat = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float64)
    ok, rvec, tvec, inl = cv2.solvePnPRansac(pw.astype(np.float32), uv.astype(np.float32), Kmat, None, None, None, False, 100, 5.991, 0.99)
    Rm, _ = cv2.Rodrigues(rvec)
    np.savez_compressed(os.path.join(OUT, "opencv_pnp.npz"), version=ver, pts3d=pw.astype(np.float32), pts2d=uv.astype(np.float32), K=np.array(K),
                        ok=np.array(bool(ok)), R=Rm, t=tvec.ravel(), inliers=(inl.ravel() if inl is not None else np.zeros(0, np.int32)))

    # 8. the CALC net itself (deeplcd.cpp:55-91), when Caffe and the model files are at hand
    proto, model = "calc_model/deploy.prototxt", "calc_model/calc.caffemodel"
    try:
        import caffe
        if os.path.exists(proto) and os.path.exists(model):
            net = caffe.Net(proto, model, caffe.TEST)
            x = (cv2.resize(cv2.GaussianBlur(L, (7, 7), 0), (160, 120)).astype(np.float32) * (1.0 / 255.0))
            net.blobs[net.inputs[0]].data[...] = x[None, None]
            out = net.forward()[net.outputs[0]].ravel().copy()
            np.savez_compressed(os.path.join(OUT, "caffe_calc.npz"), input=x, output=out, prototxt=np.array(open(proto).read()), image=L)
            print("caffe_calc.npz written (copy calc_model/ next to it to run tests/test_opencv_pin.py::test_calc_forward)")
    except ImportError:
        print("caffe not importable: CALC forward fixture skipped")
    print("fixtures written to", OUT, "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
