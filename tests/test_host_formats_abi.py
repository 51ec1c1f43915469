"""The host-side formats behind the C ABI (csrc/io.hip: myslam_io_*) — what tools/run_kitti_stereo.py reads and writes.  No device needed."""
import os

import numpy as np
import pytest

import png_files


@pytest.fixture(scope="module")
def api_host(pkg):
    if not os.path.exists(pkg.api.LIB_PATH):
        pkg.build_library()
    return pkg.api


def test_png_and_sequence_listing(api_host, synth, tmp_path):
    seq = tmp_path / "00"; (seq / "image_0").mkdir(parents=True); (seq / "image_1").mkdir()
    imgs = [synth.random_image(70 + i, 94, 311, "texture" if i % 2 else "noise") for i in range(3)]
    for i, im in enumerate(imgs):
        png_files.write_png_gray(str(seq / "image_0" / f"{i:06d}.png"), im, filters=bool(i % 2))
        png_files.write_png_gray(str(seq / "image_1" / f"{i:06d}.png"), im[:, ::-1], filters=True)
    (seq / "times.txt").write_text("0.000000e+00\n1.037875e-01\n\n2.074438e-01\n")
    L, R, ts = api_host.load_images(str(seq))
    assert len(L) == 3 and L[2].endswith("/image_0/000002.png") and R[1].endswith("/image_1/000001.png")
    assert np.allclose(ts, [0.0, 0.1037875, 0.2074438])
    for i in range(3):
        assert np.array_equal(api_host.read_png_gray(L[i]), imgs[i]) and np.array_equal(api_host.read_png_gray(R[i]), imgs[i][:, ::-1])
    with pytest.raises(api_host.MyslamError):
        api_host.read_png_gray(str(seq / "image_0" / "000009.png"))


def test_trajectory_and_loop_edge_files(api_host, tmp_path):
    """System::SaveTrajectory / SaveLoopEdges (src/system.cpp:153-224): `id ts tx ty tz qx qy qz qw`, fixed, 6 decimals, pose = Tcw^-1,
    ascending id order whatever the input order"""
    yaw = 0.3
    q = np.array([0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)])                # Rcw = rotation about y by +0.3
    Rcw = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
    twc = np.array([1.5, -0.25, 7.0])
    poses = np.array([np.concatenate([q, -Rcw @ twc]), [0, 0, 0, 1, 0, 0, 0]])
    p = str(tmp_path / "trajectory.txt")
    api_host.save_trajectory(p, [5, 2], [0.5, 0.2], poses)
    lines = open(p).read().strip().split("\n")
    assert lines[0].replace("-0.000000", "0.000000") == "2 0.200000 0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 1.000000"      # (-R^T t of a zero translation prints as -0.000000, as Sophus' inverse does)
    f = [float(x) for x in lines[1].split()]
    assert f[0] == 5 and f[1] == 0.5 and np.allclose(f[2:5], twc, atol=1e-6)
    assert np.allclose(f[5:9], [0, -np.sin(yaw / 2), 0, np.cos(yaw / 2)], atol=1e-6)       # Rwc = Rcw^T, quaternion with w >= 0
    assert all(len(x.split(".")[1]) == 6 for x in lines[1].split()[1:])
    e = str(tmp_path / "loopEdges.txt")
    api_host.save_loop_edges(e, [9, 4], [0.9, 0.4], np.stack([poses[0], poses[1]]), [1, 0], [0.1, 0.0], np.stack([poses[1], poses[0]]))
    el = open(e).read().strip().split("\n")
    assert len(el) == 4 and el[0].startswith("4 0.400000") and el[1].startswith("0 0.000000") and el[2].startswith("9 0.900000") and el[3].startswith("1 0.100000")
