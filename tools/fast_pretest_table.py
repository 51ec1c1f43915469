#!/usr/bin/env python3
"""What a compass pre-test at +-iniThFAST (20) instead of +-minThFAST (7) would let the two-phase FAST path skip — measured on the CPU
(numpy + the oracle's pyramid and score map), per pyramid level of the bench scenes:
  s7 / s20   fraction of the pixel PAIRS inside the FAST grid that survive the compass pre-test at 7 / at 20 (either pixel of a pair has
             two cyclically adjacent compass points beyond +-th on the same side: the necessary condition k_fast_strip tests)
  redo       fraction of grid cells without any pixel of score >= 20: the reference re-runs those cells at threshold 7
             (src/ORBextractor.cpp:858-864), a +-20 pre-test would have to as well
  strips     fraction of the kernel's 4-cell strips that contain such a cell
The two-phase path costs 0.78 + 1.63 s ms per 512 images alone, the dense path 1.48 (orb_kernels.hip FastCtl): a +-20 first pass pays when
0.78 + 1.63 s20 + redo_strips x (cost of the strip at +-7) < min(1.48, 0.78 + 1.63 s7).
    python tools/fast_pretest_table.py [rects ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_package  # noqa: E402
from pyoracle import Oracle  # noqa: E402


def compass(img, th):
    """per pixel (interior, 3-px border excluded): necessary condition of a FAST-9 corner at threshold th"""
    v = img[3:-3, 3:-3].astype(np.int32)
    p0 = img[6:, 3:-3].astype(np.int32); p8 = img[:-6, 3:-3].astype(np.int32)         # (0, +3), (0, -3)
    p4 = img[3:-3, 6:].astype(np.int32); p12 = img[3:-3, :-6].astype(np.int32)        # (+3, 0), (-3, 0)
    dmin = np.minimum(np.minimum(np.maximum(p0, p4), np.maximum(p4, p8)), np.minimum(np.maximum(p8, p12), np.maximum(p12, p0)))
    bmax = np.maximum(np.maximum(np.minimum(p0, p4), np.minimum(p4, p8)), np.maximum(np.minimum(p8, p12), np.minimum(p12, p0)))
    return (v - th > dmin) | (bmax > v + th)


def main():
    pkg = load_package(); synth = pkg.synth
    o = Oracle()
    rects = [int(a) for a in sys.argv[1:]] or [6000, 1000, 300]
    par = o.params(2000)
    for r in rects:
        img = synth.stereo_batch(1, stream_id=0, n_rect=r)[0, 0]
        pyr = o.pyramid(par, img)
        tot = {"pairs": 0, "s7": 0, "s20": 0, "cells": 0, "redo": 0, "strips": 0, "rstrips": 0}
        rows = []
        for lvl, im in enumerate(pyr):
            h, w = im.shape
            minB, maxBX, maxBY = 16, w - 16, h - 16
            nC, nR = (maxBX - minB) // 30, (maxBY - minB) // 30
            wC, hC = -(-(maxBX - minB) // nC), -(-(maxBY - minB) // nR)
            sc = o.fast_score_map(im, 7)                                                   # full-size map, 0 where no corner at 7
            c7, c20 = compass(im, 7), compass(im, 20)                                      # index [y - 3, x - 3]
            reg = (slice(minB + 3 - 3, maxBY - 3 - 3), slice(minB + 3 - 3, maxBX - 3 - 3))   # pixels FAST can report: [minB + 3, maxB - 3)
            a7, a20 = c7[reg], c20[reg]
            wp = a7.shape[1] // 2 * 2
            p7 = a7[:, 0:wp:2] | a7[:, 1:wp:2]; p20 = a20[:, 0:wp:2] | a20[:, 1:wp:2]
            ncell = redo = 0; strips = rstrips = 0
            for i in range(nR):
                flags = []
                for j in range(nC):
                    y0, x0 = minB + i * hC, minB + j * wC
                    y1, x1 = min(y0 + hC + 6, maxBY), min(x0 + wC + 6, maxBX)
                    if y0 >= maxBY - 3 or x0 >= maxBX - 6:
                        continue
                    cell = sc[y0 + 3:y1 - 3, x0 + 3:x1 - 3]
                    ncell += 1
                    f = not (cell.size and cell.max() >= 20)
                    redo += f; flags.append(f)
                for k in range(0, len(flags), 4):
                    strips += 1; rstrips += any(flags[k:k + 4])
            rows.append((lvl, p7.mean(), p20.mean(), redo / max(ncell, 1), rstrips / max(strips, 1)))
            tot["pairs"] += p7.size; tot["s7"] += p7.sum(); tot["s20"] += p20.sum(); tot["cells"] += ncell; tot["redo"] += redo
            tot["strips"] += strips; tot["rstrips"] += rstrips
        print(f"rects {r:5d}: pairs surviving +-7 {tot['s7'] / tot['pairs']:.3f}, +-20 {tot['s20'] / tot['pairs']:.3f}; cells to redo at 7: "
              f"{tot['redo'] / tot['cells']:.3f}, strips holding one: {tot['rstrips'] / tot['strips']:.3f}")
        print("   per level (s7, s20, redo cells, redo strips): " + "  ".join(f"L{l}: {a:.2f} {b:.2f} {c:.2f} {d:.2f}" for l, a, b, c, d in rows))


if __name__ == "__main__":
    main()
