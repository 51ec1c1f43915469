#!/usr/bin/env python3
"""The two statistics the grid-FAST kernel switches its path on, for the bench scenes of different density:
fraction of pixel pairs that survive the compass pre-test (two-phase path) and fraction of 4-pixel rows that hold a corner (dense
path).  Together with `bench.py --workload orb_match --streams 1 --orb-internal-stream 0 --scene-rects R --fast-mode 0|1` (the
stand-alone time of both paths) this gives the break-even the thresholds in orb_kernels.hip (FastCtl) are set from.

    python tools/fast_path_table.py [rects ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    api, synth = pkg.api, pkg.synth
    rects = [int(a) for a in sys.argv[1:]] or [300, 1000, 2000, 2750, 3500, 6000]
    B = 8
    for r in rects:
        fr = synth.stereo_batch(B, stream_id=0, n_rect=r)
        imgs = np.ascontiguousarray(fr[:, 0])
        d = torch.from_numpy(imgs).cuda()
        row = [r]
        for mode in (0, 1):
            ext = api.ORBextractor(2000)
            ext.set_option(ext.OPT_FAST_MODE, mode)
            cap = ext.max_keypoints(imgs.shape[1], imgs.shape[2])
            kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
            cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
            ext.detect_and_compute_batch(d.data_ptr(), B, imgs.shape[1], imgs.shape[2], imgs.shape[2], imgs.shape[1] * imgs.shape[2],
                                         kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), cap)
            torch.cuda.synchronize()
            s = p = 0
            per = []
            for lvl in range(8):
                a, b, path = ext.fast_statistics(lvl)
                assert path == mode
                s += a; p += b; per.append(round(a / max(b, 1), 3))
            row += [round(s / max(p, 1), 4), per]
        print("rects %5d   surviving pairs / pairs %.4f %s   corner rows / pairs %.4f %s" % (row[0], row[1], row[2], row[3], row[4]))


if __name__ == "__main__":
    main()
