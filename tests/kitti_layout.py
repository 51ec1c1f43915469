"""A rendered stereo sequence written in KITTI layout (times.txt, image_0 / image_1/%06d.png, a YAML with the reference's keys): the
stand-in for BASELINE configs[0]'s data set, which is not available in the build environment.  Used by tests/test_gpu_runner.py, by
tests/test_chain_host.py and by tests/golden/make_kitti_layout_trajectory.py (the committed trajectory fixture)."""
import numpy as np

# the reference's config/stereo/gray/KITTI00-02.yaml, key for key (values typed in here: nothing reads /root/reference at run time)
KITTI00_02_YAML = """%YAML:1.0
Camera.left.fx: 718.856
Camera.left.fy: 718.856
Camera.left.cx: 607.1928
Camera.left.cy: 185.2157
Camera.right.fx: 718.856
Camera.right.fy: 718.856
Camera.right.cx: 607.1928
Camera.right.cy: 185.2157
Camera.left.k1: 0.0
Camera.left.k2: 0.0
Camera.left.p1: 0.0
Camera.left.p2: 0.0
Camera.right.k1: 0.0
Camera.right.k2: 0.0
Camera.right.p1: 0.0
Camera.right.p2: 0.0
Camera.bNeedUndistortion: 0
Camera.bf: 386.1448
Camera.fps: -1
numFeatures.initGood: 100
numFeatures.trackingGood: 50
numFeatures.trackingBad: 10
ORBextractor.nInitFeatures: 300
ORBextractor.nNewFeatures: 100
ORBextractor.scaleFactor: 1.2
ORBextractor.nLevels: 8
ORBextractor.iniThFAST: 20
ORBextractor.minThFAST: 7
Map.activeMap.size: 7
LCD.similarityScoreThreshold.high: 0.94
LCD.similarityScoreThreshold.low: 0.92
LoopClosing.bShowResult: 0
LCD.nDatabaseMinSize: 50
Viewer.bShow: 0
"""

N_FRAMES, H, W, REACH = 200, 376, 1241, 60.0


def parse_yaml(text):
    kv = {}
    for line in text.splitlines():
        line = line.split("#", 1)[0].strip()
        if ":" in line and not line.startswith("%"):
            k, v = line.split(":", 1)
            kv[k.strip()] = v.strip()
    return kv


def camera(synth):
    kv = parse_yaml(KITTI00_02_YAML)
    return {"fx": float(kv["Camera.right.fx"]), "fy": float(kv["Camera.right.fy"]), "cx": float(kv["Camera.right.cx"]),
            "cy": float(kv["Camera.right.cy"]), "bf": float(kv["Camera.bf"])}


def render(synth, n=N_FRAMES):
    """n stereo pairs at 1241 x 376 with the KITTI00-02 intrinsics: a drive of 60 m along a textured wall (8-24 m away) and back, up to
    0.94 m per frame — tens of pixels of flow per frame, features leave the view for good (the reference's key-frame rule acts)"""
    scene = synth.sequence_scene(x_max=REACH + 35.0, tex_w=8192)
    C, yaw = synth.sequence_poses(n, kind="outback", reach=REACH)
    K = camera(synth)
    return [synth.render_stereo(scene, C[t], yaw[t], t, h=H, w=W, K=K) for t in range(n)], C, yaw


def write(seq_dir, frames, png_files, times=None):
    import os
    os.makedirs(os.path.join(seq_dir, "image_0"), exist_ok=True); os.makedirs(os.path.join(seq_dir, "image_1"), exist_ok=True)
    for t, (L, R) in enumerate(frames):
        png_files.write_png_gray(os.path.join(seq_dir, "image_0", f"{t:06d}.png"), L, filters=True)
        png_files.write_png_gray(os.path.join(seq_dir, "image_1", f"{t:06d}.png"), R, filters=True)
    ts = times if times is not None else [0.1 * t for t in range(len(frames))]
    with open(os.path.join(seq_dir, "times.txt"), "w") as f:
        f.write("".join(f"{x:.6e}\n" for x in ts))
    return ts


def ate(chain_mod, synth, poses7, C, yaw):
    """RMSE / worst distance of the camera centres against the rendered path, both expressed in the frame of camera 0"""
    T0 = chain_mod.T_of(synth.pose7_from_twc(C[0], yaw[0]))
    est = np.array([chain_mod.T_inv(chain_mod.T_of(p))[:3, 3] for p in poses7])
    gt = np.array([chain_mod.T_inv(chain_mod.T_of(synth.pose7_from_twc(C[t], yaw[t])) @ chain_mod.T_inv(T0))[:3, 3] for t in range(len(poses7))])
    return float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1)))), float(np.abs(est - gt).max())
