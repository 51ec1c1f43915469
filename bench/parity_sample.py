"""bench/parity_sample.py — bench.py's `parity_sample` leg: K frames of one more step of the timed workload against the CPU oracle at the bars of the parity
tests (untimed, after every timed pass).  See bench/cpu_baseline.py for the rule about importing oracle/."""
import os
import sys

import numpy as np

from .config import ROOT


def parity_sample(api, synth, frames, sample, cap, Kt, K, bufs, db_np, db_ids, cur_id, ba_w, calc_w):
    """K frames of ONE 512-pair step of the timed workload (the step function of the timed region, run once more after it) against the
    oracle at the bars of the parity tests: key-point structs and descriptor bytes of both images, match indices and distances,
    triangulation flags (identical) and coordinates (1e-9), DeepLCD descriptor (2e-5), the database scan's (best id, max score, count),
    one BA window's blocks (1e-11 of the largest entry).  `bufs` holds host copies of the step's output buffers.  Returns the
    `parity_sample` object; "ok": False on any mismatch (the caller exits non-zero)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle
    o = Oracle()
    P = len(frames)
    res = {"frames": len(sample), "frame_indices": [int(i) for i in sample], "pairs_per_step": P, "ok": True, "mismatches": []}
    bad = lambda what: (res["mismatches"].append(what), res.__setitem__("ok", False))
    par = o.params(2000)
    kps = bufs["kps"].view(api.KP_DTYPE).reshape(2 * P, cap); desc = bufs["desc"].reshape(2 * P, cap, 32); cnt = bufs["cnt"]
    n_kp = 0; lcd_dev = 0.0; xyz_dev = 0.0; score_dev = 0.0; ba_dev = 0.0; n_match = 0; n_ok = 0
    for i in sample:
        ref = []
        for side in (0, 1):
            rk, rd = o.detect_and_compute(par, frames[i, side])
            b = i + side * P
            n = int(cnt[b]); n_kp += n
            if n != len(rk) or kps[b, :n].tobytes() != rk.tobytes():
                bad(f"frame {i} {'LR'[side]}: key-points")
            elif not np.array_equal(desc[b, :n], rd):
                bad(f"frame {i} {'LR'[side]}: descriptors")
            ref.append((rk, rd))
        (kl, dl), (kr, dr) = ref
        ridx, rdist = o.hamming_match(dl, dr)
        nl = len(kl); n_match += nl
        if not (np.array_equal(bufs["midx"][i * cap:i * cap + nl], ridx) and np.array_equal(bufs["mdist"][i * cap:i * cap + nl], rdist)):
            bad(f"frame {i}: Hamming match")
        rxyz, rok = o.triangulate_stereo(kl["x"], kl["y"], kr["x"][ridx], kr["y"][ridx], K["fx"], K["fy"], K["cx"], K["cy"], K["bf"] / K["fx"])
        ok = bufs["ok"][i * cap:i * cap + nl].astype(bool); xyz = bufs["xyz"].reshape(-1, 3)[i * cap:i * cap + nl]
        n_ok += int(rok.sum())
        # a match of exactly zero disparity is a point at infinity: the DLT's homogeneous coordinate is rounding noise and so is the sign of z
        # (1e17 m here, 1e17 m behind the camera in the oracle; Eigen's bdcSvd in the reference is no different) — such points are not compared
        finite = (np.abs(rxyz[:, 2]) < 1e9) & (np.abs(xyz[:, 2]) < 1e9)
        rok = rok & finite
        if not np.array_equal(ok[finite], rok[finite]):
            bad(f"frame {i}: triangulation flags")
        elif rok.any():
            d = float(np.max(np.abs(xyz[rok] - rxyz[rok]) / np.maximum(1.0, np.abs(rxyz[rok]))))
            xyz_dev = max(xyz_dev, d)
            if d > 1e-9:
                bad(f"frame {i}: triangulated coordinates ({d:.2e})")
        if "descr" in bufs:
            x, _ = o.calc_preproc(frames[i, 0])
            rd_ = o.calc_forward(calc_w, x)
            d = float(np.abs(bufs["descr"][i] - rd_).max()); lcd_dev = max(lcd_dev, d)
            if d >= 2e-5:
                bad(f"frame {i}: DeepLCD descriptor ({d:.2e})")
            rb, rm, rc = o.lcddb_query(db_np, db_ids, bufs["descr"][i], cur_id)         # the scan of the descriptor the device scanned with
            near = int((np.abs(db_np @ bufs["descr"][i] - 0.92) < 1e-5).sum())          # counts may differ only for scores within float noise of the threshold
            d = abs(float(bufs["max"][i]) - rm); score_dev = max(score_dev, d)
            if int(bufs["best"][i]) != rb or d >= 2e-5 or abs(int(bufs["dbcnt"][i]) - rc) > near:
                bad(f"frame {i}: database scan ({int(bufs['best'][i])}, {float(bufs['max'][i])}, {int(bufs['dbcnt'][i])}) vs ({rb}, {rm}, {rc})")
    if "ba" in bufs:
        for i in sample[:1] + sample[-1:]:
            po, pt, ep, el, ob, fx, sz = [a[i] for a in ba_w]
            np_, nl_, ne_ = [int(v) for v in sz]
            ref = o.ba_build(po[:np_], pt[:nl_], ep[:ne_], el[:ne_], ob[:ne_], fx[:nl_], Kt)
            got = [bufs["ba"][0][i].reshape(-1, 6, 6)[:np_], bufs["ba"][1][i].reshape(-1, 3, 3)[:nl_], bufs["ba"][2][i].reshape(-1, 6, 3)[:ne_],
                   bufs["ba"][3][i].reshape(-1, 6)[:np_], bufs["ba"][4][i].reshape(-1, 3)[:nl_], bufs["ba"][5][i][:ne_]]
            for nm, g, r in zip(("Hpp", "Hll", "Hpl", "bp", "bl", "chi2"), got, ref):
                d = float(np.abs(g - r).max() / max(1.0, np.abs(r).max())); ba_dev = max(ba_dev, d)
                if d > 1e-11:
                    bad(f"window {i}: BA block {nm} ({d:.2e})")
    res.update({"orb": "bit-exact" if not any("key-points" in m or "descriptors" in m for m in res["mismatches"]) else "MISMATCH",
                "keypoints_compared": n_kp, "matches_compared": n_match, "triangulated_compared": n_ok,
                "match": "bit-exact" if not any("Hamming" in m for m in res["mismatches"]) else "MISMATCH",
                "triangulation_max_rel": xyz_dev, "lcd_max_abs": lcd_dev if "descr" in bufs else None,
                "db_score_max_abs": score_dev if "descr" in bufs else None, "ba_max_rel": ba_dev if "ba" in bufs else None,
                "bars": "ORB / match / flags identical; xyz 1e-9 rel; DeepLCD 2e-5 abs; DB best id identical, score 2e-5, count up to scores within 1e-5 "
                        "of the threshold; BA blocks 1e-11 of the largest entry (tests/test_gpu_*.py)",
                "note": "one more step of the timed workload (same step function, same batch, same buffers) run after the timed region; K frames of "
                        "it pulled to the host and compared with the CPU oracle (oracle/, parity unpinned — DESIGN.md section 5)"})
    return res
