"""bench/report.py — the roofline objects of bench.py's JSON line, built from the profiled pass (HIP-event time per launch, on the launch's own stream), the
extractor-alone pass, the committed counter summary (profiles/r<NN>_pmc_*.json) and the committed machine peaks (profiles/r<NN>_peaks.json):
  roofline       the contract's object: dominant kernel of the ORB chain against HBM (algorithmic bytes per launch / launch duration; counter traffic)
  roofline_valu  the same kernel against the measured packed-16 issue rate — its real bound
  roofline_mfma  CALC conv2 against the dense 16-bit matrix-core peak
and the flop model of k_ba_optimize (roofline_ba_optimize)."""
import numpy as np

from .config import ALGO_BYTES, HBM_PEAK_GBS, MFMA_BF16_PEAK_TFLOPS, SYMBOL, VALU_PEAK_TLANEOPS
from .profiles import peaks_file, pmc_file, pmc_lookup


def rooflines(prof, alone, steps, P, conv2_products, build_id=None):
    """-> (roofline, roofline_valu, roofline_mfma, busy, peaks).  prof / alone: {slot: (total ms, launches)} of the profiled pass / the extractor-alone pass"""
    pmc, pmc_path = pmc_file()
    peaks, peaks_path = peaks_file()
    valu_peak = (peaks or {}).get("valu_packed16_tlaneops") or VALU_PEAK_TLANEOPS
    busy = {k: v for k, v in prof.items() if v[1] > 0}
    roof = {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "peak_measured": (peaks or {}).get("hbm_copy_GBps"), "peaks_source": peaks_path}
    roof_valu = None
    if busy:
        # the dominant kernel of the critical (ORB) chain: the one with the largest event-timed total among the chain's stages
        chain = [k for k in busy if k in ("resize", "fast", "octree", "blur7", "describe", "hamming_match", "triangulate")]
        # event-timed durations of overlapped launches say how long a kernel was resident, not how much of the chip it used (the
        # latency-bound oct-tree runs under FAST for as long as FAST takes): among the chain's stages the dominant kernel is the
        # one with the largest VALU instruction volume (counter summary) when that is known, else the largest duration
        def volume(k):
            _, rec = pmc_lookup(pmc, k)
            return ((rec or {}).get("valu_wave_insts_per_image", 0.0), busy[k][0])
        dom = max(chain or busy, key=volume)
        dom_ms, dom_n = busy[dom]
        per_launch_ms = dom_ms / dom_n
        launches_per_step = dom_n / steps
        imgs_per_launch = 2 * P / launches_per_step
        sym, rec = pmc_lookup(pmc, dom)
        roof.update({"kernel": sym or SYMBOL.get(dom, dom), "stage": dom, "avg_launch_ms": per_launch_ms, "images_per_launch": imgs_per_launch})
        alone_ms = alone[dom][0] / alone[dom][1] if dom in alone else None      # the same launch (same images per launch) on an idle chip
        if dom in ALGO_BYTES:
            algo = ALGO_BYTES[dom] * imgs_per_launch                        # bytes per launch
            achieved = algo / (per_launch_ms * 1e-3) / 1e9
            roof.update({"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo})
            if roof["peak_measured"]:
                roof["frac_of_measured"] = achieved / roof["peak_measured"]
            if alone_ms:
                roof["alone"] = {"avg_launch_ms": alone_ms, "achieved": algo / (alone_ms * 1e-3) / 1e9,
                                 "frac": algo / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "the same launch with nothing else on the chip (pass 6)"}
        if rec:
            # HBM bytes per launch from the counter summary: FETCH_SIZE scaled by the factor that makes k_ingest's FETCH_SIZE equal
            # the bytes it provably reads (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 bytes), + WRITE_SIZE
            roof["traffic"] = (rec["fetch_bytes_per_image_corrected"] + rec["write_bytes_per_image"]) * imgs_per_launch
            # the counters are a COMMITTED measurement (separate --pmc passes cannot run inside this process): name the build they were taken on and say
            # so when it is not the library running now (round 6; counter files older than round 6 carry no digest: stale by definition)
            roof["traffic_build"] = pmc.get("build_id") or pmc.get("build")
            roof["traffic_stale"] = (pmc.get("build_id") != build_id) if build_id else None
            roof["traffic_detail"] = {"source": pmc_path, "fetch_scale": pmc["calibration"]["fetch_scale"],
                                      "fetch_bytes_per_image_corrected": rec["fetch_bytes_per_image_corrected"],
                                      "write_bytes_per_image": rec["write_bytes_per_image"]}
            v = rec.get("valu_wave_insts_per_image")
            if v:
                ach = v * imgs_per_launch * 64 / (per_launch_ms * 1e-3) / 1e12
                roof_valu = {"bound": "valu", "kernel": roof["kernel"], "unit": "Tlane-op/s", "peak": valu_peak, "achieved": ach,
                             "frac": ach / valu_peak, "frac_alone": (v * imgs_per_launch * 64 / (alone_ms * 1e-3) / 1e12 / valu_peak) if alone_ms else None,
                             "peak_spec_16_lanes_per_cycle": VALU_PEAK_TLANEOPS, "peaks_source": peaks_path,
                             "valu_wave_insts_per_image": v, "source": pmc_path,
                             "note": "peak = measured issue rate of v_pk_max_i16 / v_pk_min_i16 / v_pk_maximum3_f16 / v_pk_minimum3_f16 (4 cycles "
                                     "per wave64 instruction per SIMD; v_perm_b32, v_dot4, v_alignbyte and every VOP3 integer class measure the same; "
                                     "only VOP2 add / and / or / lshr / 16-bit min-max and f32 add / mul / fma issue in 2 cycles) — the classes "
                                     "k_fast_strip's scoring network consists of"}
        roof["note"] = ("avg_launch_ms = HIP-event duration of one launch on its own stream in the profiled pass (same schedule as the timed "
                        "region); a launch covers images_per_launch images and shares the chip with the other streams' launches; the kernel "
                        "is packed-integer VALU bound in practice (roofline_valu, DESIGN.md section 6)")
    mf = None
    if "calc_conv2" in busy:
        c2 = busy["calc_conv2"][0] / busy["calc_conv2"][1]
        f32eq = 2 * 176160768 * P / (c2 * 1e-3) / 1e12
        nprod = conv2_products          # 3 = f16 x 3 (the default model), 6 = the bf16 x 6 kernel a model outside f16's range falls back to
        mf = {"bound": "mfma", "kernel": SYMBOL["calc_conv2"] if nprod == 3 else "k_conv2_bf16x6", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s (bf16 dense)",
              "partial_products": nprod,
              "achieved": nprod * f32eq, "frac": nprod * f32eq / MFMA_BF16_PEAK_TFLOPS, "peak_measured": (peaks or {}).get("mfma_bf16_tflops"),
              "peaks_source": peaks_path, "f32_equivalent_tflops": f32eq, "avg_launch_ms": c2,
              "note": "CALC conv2 as an implicit GEMM on the 16-bit matrix cores with f32 accuracy (every f32 operand split exactly into two f16 "
                      "pieces, 3 partial products per useful f32 multiply-add; f16 and bf16 run at the same dense rate): `achieved` counts the "
                      "flops the matrix pipe executes, priced against the 16-bit dense peak; f32_equivalent_tflops counts the USEFUL f32 flops "
                      "(the f32-input MFMA peak would be 157.3)"}
    return roof, roof_valu, mf, busy, peaks


def ba_solve_roofline(rounds_failed, sizes, P, solve_ms):
    """rounds_failed: per-window `rounds` output of myslam_ba_optimize_active_map_batch (host array); sizes: [P, 3] = poses, landmarks, edges per window"""
    # k_ba_optimize against the f64 peaks: a flop MODEL of what one Levenberg iteration of a window executes (not a counter): edge evaluation +
    # Jacobians + block products ~410 flop per edge, Schur complement sum_l W_l Hll^-1 W_l^T = (108 k + 216 k^2) flop
    # for a landmark seen by k key-frames, 6x6-blocked Cholesky n^3 / 3 and two triangular solves 2 n^2 with n = 6 P; rounds x 10 iterations
    # (optimize(10), backend.cpp:212-214: every round runs its iteration budget unless a Levenberg trial fails ten times)
    # *rounds = the reference's `iteration` counter = rounds that FAILED the inlier test (backend.cpp:212-232): a window runs that many + 1
    # rounds of optimize(10), at most max_rounds = 5 (round 5 fix: the model multiplied by the counter itself, 0 for well-posed windows)
    rounds_mean = float(np.minimum(np.asarray(rounds_failed, np.float64) + 1.0, 5.0).mean())
    szs = sizes                                                    # [P, 3] = poses, landmarks, edges per window
    npo, nla, ned = [float(np.mean(szs[:, i])) for i in range(3)]
    kobs = ned / max(1.0, nla)
    flop_it = 410.0 * ned + nla * (108.0 * kobs + 216.0 * kobs * kobs) + (6 * npo) ** 3 / 3.0 + 2 * (6 * npo) ** 2
    flops = P * rounds_mean * 10 * flop_it
    solve_roof = {"bound": "f64 (vector + matrix cores)", "kernel": "k_ba_optimize", "unit": "TFLOP/s (f64, modelled flops)", "avg_launch_ms": solve_ms,
                  "windows_per_launch": P, "rounds_executed_mean": rounds_mean, "modelled_flop_per_iteration": flop_it, "achieved": flops / (solve_ms * 1e-3) / 1e12,
                  "peak": 78.6, "peak_f64_mfma_measured": 48.0, "frac": flops / (solve_ms * 1e-3) / 1e12 / 78.6,
                  "note": "one 512-thread block per window with the window's state in 133 KB of LDS (one block per CU): iterations are chains of barrier-separated "
                          "phases (pose blocks, landmark blocks, Schur chunks on v_mfma_f64_16x16x4_f64, 6x6-blocked Cholesky, back-substitution, update, chi2); "
                          "peak = MI355X f64 vector 78.6 TFLOP/s, measured f64 MFMA 48 (profiles/r03_peaks.json)"}

    return solve_roof
