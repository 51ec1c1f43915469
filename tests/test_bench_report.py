"""CPU checks of the parts of bench.py that need no GPU (package bench/): the command line's defaults (the driver runs `python bench.py --gpus N
--steps K --warmup W`; with no flags N = 1), the roofline objects built from event times + the committed counter / peak files, the solve's flop
model.  The GPU run of the whole line is tests/test_gpu_bench_contract.py."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_command_line_defaults(monkeypatch):
    from bench.args import parse
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = parse()
    assert a.gpus == 1 and a.steps == 200 and a.warmup == 5 and a.pairs == 512 and a.workload == "full"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "3"])
    a = parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 3)


def test_roofline_objects_from_event_times():
    """rooflines(): the dominant kernel of the ORB chain is the one with the largest VALU volume in the committed counter summary (FAST), its HBM
    view = algorithmic bytes per launch / event-timed launch duration, its VALU view = counted wave-instructions x 64 / duration against the
    measured packed-16 peak; conv2 against the dense 16-bit matrix-core peak."""
    from bench.config import ALGO_BYTES, HBM_PEAK_GBS
    from bench.report import ba_solve_roofline, rooflines
    steps, P = 10, 512
    # {slot: (total ms, launches)}: two FAST launches of 512 images per step, each 3.0 ms; conv2 one launch of 3.1 ms per step
    prof = {"fast": (2 * 3.0 * steps, 2 * steps), "resize": (14 * 0.1 * steps, 14 * steps), "describe": (2 * 1.6 * steps, 2 * steps),
            "calc_conv2": (3.1 * steps, steps), "octree": (0.7 * steps, 2 * steps), "unused": (0.0, 0)}
    alone = {"fast": (3 * 1.4, 3)}
    roof, valu, mfma, busy, peaks = rooflines(prof, alone, steps, P, 3)
    assert "unused" not in busy and roof["stage"] == "fast" and roof["kernel"].startswith("k_fast_strip")
    assert roof["images_per_launch"] == 512 and roof["avg_launch_ms"] == pytest.approx(3.0)
    assert roof["achieved"] == pytest.approx(ALGO_BYTES["fast"] * 512 / 3.0e-3 / 1e9) and roof["frac"] == pytest.approx(roof["achieved"] / HBM_PEAK_GBS)
    assert roof["alone"]["avg_launch_ms"] == pytest.approx(1.4) and roof["alone"]["frac"] > roof["frac"]
    assert roof["traffic"] is not None and 0.9 < roof["traffic"] / roof["algorithmic_bytes_per_launch"] < 1.2          # counter traffic ~ algorithmic bytes
    assert valu["bound"] == "valu" and 1.3e6 < valu["valu_wave_insts_per_image"] < 1.7e6
    assert valu["achieved"] == pytest.approx(valu["valu_wave_insts_per_image"] * 512 * 64 / 3.0e-3 / 1e12) and 0.3 < valu["frac"] < 0.6 and 0.8 < valu["frac_alone"] < 1.1
    assert mfma["kernel"] == "k_conv2_f16x3" and mfma["partial_products"] == 3
    assert mfma["achieved"] == pytest.approx(3 * 2 * 176160768 * P / 3.1e-3 / 1e12) and mfma["frac"] == pytest.approx(mfma["achieved"] / 2500.0)
    assert roof["traffic_stale"] is None and roof["traffic_build"]            # no library named: staleness unknown, the counter file still names its build
    roof_s = rooflines(prof, alone, steps, P, 3, build_id="not-the-profiled-build")[0]
    assert roof_s["traffic_stale"] is True
    json.dumps([roof, valu, mfma])                                            # everything in the line is JSON
    # nothing profiled: the contract's object with nulls, no exception
    roof0, valu0, mfma0, _, _ = rooflines({}, {}, steps, P, 3)
    assert roof0["bound"] == "hbm" and roof0["achieved"] is None and valu0 is None and mfma0 is None
    # the solve's flop model: rounds EXECUTED = rounds failed + 1, at most 5
    sizes = np.tile(np.array([[10, 300, 2950]]), (4, 1))
    r = ba_solve_roofline(np.array([0, 0, 4, 9]), sizes, 4, 2.0)
    assert r["rounds_executed_mean"] == pytest.approx((1 + 1 + 5 + 5) / 4) and r["windows_per_launch"] == 4 and 0 < r["frac"] < 1


def test_committed_profiles_are_the_newest_of_their_kind():
    from bench.profiles import peaks_file, pmc_file
    pmc, path = pmc_file()
    assert pmc is not None and path.startswith(("profiles/r05_pmc_", "profiles/r06_pmc_")) and any(k.startswith("k_fast_strip<32, 4, 40") for k in pmc["kernels"])      # (the instance gained a template argument at the end of round 6: "<32, 4, 40, 1>")
    assert abs(pmc["calibration"]["fetch_scale"] - 1.93) < 0.05                # gfx950 counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md)
    peaks, ppath = peaks_file()
    assert peaks and os.path.exists(os.path.join(ROOT, ppath)) and 5500 < peaks["hbm_copy_GBps"] < 8000
