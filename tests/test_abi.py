"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/myslam_hip.h declares.  No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "myslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(myslam_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared()
    for must in ["myslam_orb_detect_and_compute", "myslam_orb_detect", "myslam_orb_screen_and_compute_params",
                 "myslam_orb_calc_descriptors", "myslam_hamming_match", "myslam_triangulate_stereo",
                 "myslam_lcd_calc_descr_original_img", "myslam_lcd_calc_descr", "myslam_lcd_score",
                 "myslam_lcddb_append", "myslam_lcddb_query", "myslam_ba_build"]:
        assert must in names


def test_library_builds_loads_and_exports_every_symbol(pkg):
    path = pkg.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"declared in myslam_hip.h but not exported: {missing}"


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "myslam_hip.h")).read()
    assert "torch" not in text.lower() and "at::" not in text


def test_product_does_not_touch_the_oracle():
    """The product path may not include, link or call anything under oracle/."""
    pkg_dir = os.path.join(ROOT, "a-simple-stereo-slam-system-with-deep-loop-closing_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("# oracle/", ""), f


def test_keypoint_layout(pkg):
    assert pkg.api.KP_DTYPE.itemsize == 28     # cv::KeyPoint
