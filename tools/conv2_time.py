#!/usr/bin/env python3
"""conv2 stand-alone time (ms per 512 frames) through the library's profiling hooks:  python tools/conv2_time.py [frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import torch
pkg = load_package(); api, synth = pkg.api, pkg.synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H, W = 376, 1241
fr = synth.stereo_batch(8)
imgs = torch.from_numpy(np.tile(fr[:, 0], (P // 8, 1, 1))).cuda()
lcd = api.DeepLCD(synth.calc_weights(), stream=torch.cuda.current_stream().cuda_stream)
out = torch.zeros(P, 1064, device="cuda")
for _ in range(3):
    lcd.describe_batch(imgs.data_ptr(), P, H, W, W, H * W, out.data_ptr(), blur_in_place=False)
torch.cuda.synchronize()
api.prof_reset(); api.prof_enable(True)
for _ in range(20):
    lcd.describe_batch(imgs.data_ptr(), P, H, W, W, H * W, out.data_ptr(), blur_in_place=False)
torch.cuda.synchronize(); api.prof_enable(False)
pr = api.prof_read()
print({k: round(v[0] / v[1], 4) for k, v in pr.items() if v[1] > 0}, "checksum", float(out.abs().sum()))
