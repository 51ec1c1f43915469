"""The alternative kernels kept behind developer switches (one cell per block FAST, LDS-tile blur, one-row resize, one wave per
key-point describe, unfused conv1, f32-input MFMA conv2 tilings, two-pass DeepLCD input, xor / popcount and int8 matrix-core Hamming; and the experimental int8 matrix-core Gaussian) must stay bit-compatible with the default path: run the extractor + CALC against the oracle in
child processes with the switches set (they are read once per process)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
import torch
import __graft_entry__ as g
pkg = g.load_package(); api, synth = pkg.api, pkg.synth
sys.path.insert(0, %(root)r + "/oracle")
from pyoracle import Oracle
o = Oracle()
for (h, w, nf) in ((240, 320, 500), (301, 517, 1200)):
    img = synth.random_image(1000 + h, h, w)
    gk, gd = api.ORBextractor(nf).DetectAndCompute(img)
    rk, rd = o.detect_and_compute(o.params(nf), img)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd), "ORB differs"
L, _ = synth.stereo_pair(0, 1)
wts = synth.calc_weights()
d, _ = api.DeepLCD(wts).calcDescrOriginalImg(L, blur_in_place=False)
x, _ = o.calc_preproc(L)
assert np.abs(d - o.calc_forward(wts, x)).max() < 2e-5, "CALC differs"
db = synth.lcd_database(300); ids = np.arange(300, dtype=np.uint64)
D = api.LoopDatabase(300)
for i in range(300):
    D.AddToDatabase(i, db[i])
import torch as _t
qs = _t.from_numpy(db[:64].copy()).cuda(); cur = np.full(64, 400, np.uint64)
d_best = _t.zeros(64, dtype=_t.int64, device="cuda"); d_max = _t.zeros(64, device="cuda"); d_cnt = _t.zeros(64, dtype=_t.int32, device="cuda")
D.query_batch(qs.data_ptr(), cur, 64, d_best.data_ptr(), d_max.data_ptr(), d_cnt.data_ptr()); _t.cuda.synchronize()
for k in range(64):
    ref = o.lcddb_query(db, ids, db[k], 400)
    assert int(d_best[k]) == ref[0] and abs(float(d_max[k]) - ref[1]) < 2e-5 and int(d_cnt[k]) == ref[2], "DB scan differs"
rng = np.random.default_rng(5)
for nq, nt in ((700, 1033), (33, 2), (2000, 1999)):
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    t[nt // 2] = t[0]; q[0] = t[0]
    gi, gd = api.hamming_match(q, t); ri, rd = o.hamming_match(q, t)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd), "Hamming differs"
print("FALLBACK OK")
'''


@pytest.mark.parametrize("env", [
    {"MYSLAM_FAST_V": "3", "MYSLAM_BLUR_V": "2", "MYSLAM_RESIZE_V": "1", "MYSLAM_DESC_V": "1", "MYSLAM_CONV1_V": "1", "MYSLAM_HAMMING_V": "1", "MYSLAM_ORB_AUX": "0", "MYSLAM_CONV2_V": "0", "MYSLAM_LCD_PRE_V": "1", "MYSLAM_DBSCAN_V": "1"},
    {"MYSLAM_FAST_V": "2", "MYSLAM_BLUR_V": "1", "MYSLAM_ORB_AUX": "1", "MYSLAM_HAMMING_V": "2", "MYSLAM_CONV1_V": "2", "MYSLAM_POOL2_V": "1"},
    {"MYSLAM_FAST_V": "2", "MYSLAM_FAST_T": "64", "MYSLAM_CONV2_V": "1"},
    {"MYSLAM_BLUR_V": "4"},
])
def test_alternative_kernels_match_oracle(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK OK" in r.stdout, (env, r.stdout[-2000:], r.stderr[-2000:])
