#!/usr/bin/env python3
"""Regenerates the committed golden fixtures from the CPU oracle (inputs + expected outputs only — data).
The reference has no golden vectors and cannot be built, so these pin the oracle itself
against accidental change and give the GPU tests a second, file-based comparison point.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
from pyoracle import Oracle  # noqa: E402

synth = load_package().synth
o = Oracle()

img = synth.random_image(4242, 240, 320)
p = o.params(400)
kps, desc = o.detect_and_compute(p, img)
np.savez_compressed(os.path.join(HERE, "orb_small.npz"), image=img, nfeatures=400, kps=kps, desc=desc,
                    detect100=o.detect(o.params(100), img))

rng = np.random.default_rng(7)
q = rng.integers(0, 256, (64, 32), dtype=np.uint8); t = rng.integers(0, 256, (80, 32), dtype=np.uint8)
t[40] = t[2]; q[5] = t[2]
idx, dist = o.hamming_match(q, t)
np.savez_compressed(os.path.join(HERE, "hamming_small.npz"), q=q, t=t, idx=idx, dist=dist)

w = synth.calc_weights()
x = synth._rng(99).uniform(0, 1, (120, 160)).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "calc_small.npz"), x=x, descr=o.calc_forward(w, x), weights_seed=0xCA1C)

poses, pts, ep, el, obs, fixed, K = synth.ba_problem(n_kf=4, n_mp=25)
Hpp, Hll, Hpl, bp, bl, chi2 = o.ba_build(poses, pts, ep, el, obs, fixed, K)
np.savez_compressed(os.path.join(HERE, "ba_small.npz"), poses=poses, pts=pts, ep=ep, el=el, obs=obs, fixed=fixed, K=np.array(K),
                    Hpp=Hpp, Hll=Hll, Hpl=Hpl, bp=bp, bl=bl, chi2=chi2)
print("golden fixtures written to", HERE)
