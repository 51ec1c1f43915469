#!/usr/bin/env python3
"""bench.py — stereo frames/s of the per-frame dense path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch (default 512) of synthetic 1241x376 stereo pairs resident in HBM:
  ORB DetectAndCompute (2000 features) on left+right -> L/R 256-bit Hamming match -> stereo triangulation
  -> DeepLCD descriptor of the left image -> cosine scan of the key-frame database -> local-BA block build
  (one DISTINCT 10 KF x 300 landmark window per frame).  [BASELINE.json configs[3]; --workload orb_match = configs[1]]

One process per GPU.  `python bench.py --gpus N` launches its N ranks itself when it is not already running under
torch.distributed.run (which works as before: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Frames shard
across ranks (stream r on rank r, no data-path collective); the loop database is sharded by key-frame id range and the only
exchange is an all-gather of the query descriptors and of the 16-byte per-shard candidate records (RCCL).

Passes (all after the warm-up, each bracketed by barrier + synchronize):
  1. the TIMED region: K steps, per-kernel event profiling OFF            -> value, ms_per_step
  2. a profiled pass: the same K steps with a HIP event pair around every launch (on the launch's stream) -> roofline
  3. (full workload) K steps with the OptimizeActiveMap solve on 1 frame in 6 (the reference's key-frame cadence) -> full_solve_cadence6
Prints ONE JSON line (rank 0).  PyTorch is used for device memory, streams and torch.distributed only.
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

# The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) HSA hardware queues, and streams that share a queue
# serialise — a host->device copy on one of them stalls the kernels of the other (measured: the streamed pass ran at copy + compute
# instead of max(copy, compute)).  The default schedule uses 6 streams (two extractor handles, match + BA build, DeepLCD + DB scan, two
# input-copy streams: the left and the right images of a streamed step cross PCIe as two concurrent copies): one hardware queue each
# (4, 5 and 6 queues measure the same on the resident pass).  A runtime setting of the application, stated in the JSON (config.hip_hw_queues); the library reads no
# environment variable.  Must be set before the HIP runtime initialises (i.e. before `import torch`).
def _ranks_wanted():
    if "WORLD_SIZE" in os.environ:
        return int(os.environ["WORLD_SIZE"])
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            return int(sys.argv[i + 1])
        if a.startswith("--gpus=") and a[7:].isdigit():
            return int(a[7:])
    return 1


# N > 1: torch's process group brings one more stream (the collectives' own) — one more queue
def _small_batch():
    for i, a in enumerate(sys.argv):
        if a == "--pairs" and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            return int(sys.argv[i + 1]) <= 64
    return False


# N > 1: torch's process group brings one more stream (the collectives' own) — one more queue; small batches run on up to 16 lanes, and
# lanes that share a hardware queue serialise (8 pairs per step, 16 lanes: 17.1 k frames/s on 8 queues, 19.0 k on 16, 26.5 k on 24)
HW_QUEUES = os.environ.setdefault("GPU_MAX_HW_QUEUES", ("24" if _small_batch() else "6") if _ranks_wanted() == 1 else "7")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

H, W = 376, 1241
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TLANEOPS = 39.3      # spec-derived: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz (packed-16 / VOP3 integer classes: 4 cycles per wave64)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense
# The MEASURED peaks of the same machine class live in profiles/r<NN>_peaks.json (tools/peaks.hip via tools/peaks.py): HBM copy rate,
# issue rate of the instruction classes the FAST kernel is made of, bf16 MFMA rate.  The roofline objects carry both: `peak` is the
# vendor figure the contract names (HBM) or the measured ceiling (VALU: no vendor figure exists), the other one sits beside it.
PYR_PX = 1444097               # sum of the 8 level areas
# algorithmic bytes per IMAGE of each ORB stage (SURVEY.md §8(d) accounting)
ALGO_BYTES = {
    "resize": 1407767 + 977481,            # read levels 0-6, write levels 1-7
    "fast": PYR_PX + 4 * 20000,            # read every level once + candidate list
    "blur7": 2 * PYR_PX,                   # read + write every level
    "describe": 2000 * (749 + 512 + 60),   # IC patch + BRIEF samples + outputs
    "octree": 2 * 4 * 56000,               # candidates in, selected out (latency bound in practice)
}
# profiling slot (csrc/prof.hip) -> kernel symbol prefix as rocprofv3 prints it
SYMBOL = {"resize": "k_resize_strip", "fast": "k_fast_strip", "octree": "k_octree", "blur7": "k_blur7", "describe": "k_describe2",
          "hamming_match": "k_hamming_fp4", "triangulate": "k_triangulate", "lcd_preproc": "k_lcd_input_fused",
          "calc_conv1": "k_conv1_f16x3_pool_lrn", "calc_conv2": "k_conv2_f16x3", "calc_pool2": "k_pool_lrn128_2x2", "calc_conv3": "k_conv3_norm", "lcddb_scan": "k_db_scan_bf16x6",
          "ba_build": "k_ba_build", "screen": "k_screen"}


_masked = []


def masked_stream(n_cus, first=0):
    """A HIP stream whose kernels may only run on `n_cus` compute units (hipExtStreamCreateWithCUMask; bits first .. first + n_cus - 1 of the
    device's CU mask) as a torch stream — an experiment: does confining the latency-bound side chains to a few CUs keep their long-lived blocks
    out of FAST's way?  (--side-cus / --match-cus; DESIGN_APPENDIX.md section 8 has the result.)"""
    import ctypes
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    words = 8                                              # 256 CUs
    mask = (ctypes.c_uint32 * words)()
    for b in range(first, first + n_cus):
        mask[(b // 32) % words] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0 and st.value, f"hipExtStreamCreateWithCUMask failed: {rc}"
    _masked.append(st)
    return torch.cuda.ExternalStream(st.value)


def pmc_file():
    """The newest committed counter summary of this build family (tools/pmc_collect.py writes it): profiles/r<NN>_pmc_<tag>.json."""
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")) if re.search(r"r(\d+)_pmc_[^/]*\.json$", f) and "traffic" not in f and "mfma" not in f]
    if not files:
        return None, None

    def key(f):
        m = re.search(r"r(\d+)_pmc_.*?(\d+)\.json$", f)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    f = max(files, key=key)
    try:
        d = json.load(open(f))
    except Exception:
        return None, None
    return (d, os.path.relpath(f, ROOT)) if "kernels" in d and "calibration" in d else (None, None)


def peaks_file():
    """The newest committed machine-peak measurement (tools/peaks.py): profiles/r<NN>_peaks.json -> (summary dict, relative path)."""
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_peaks.json"))
    best = (None, None, -1)
    for f in files:
        m = re.search(r"r(\d+)_peaks\.json$", f)
        if not m or int(m.group(1)) <= best[2]:
            continue
        try:
            d = json.load(open(f))
            best = (d["summary"], os.path.relpath(f, ROOT), int(m.group(1)))
        except Exception:
            pass
    return best[0], best[1]


def pmc_lookup(pmc, slot):
    """(full kernel symbol, per-image record) of the profiling slot's kernel in the counter summary"""
    if not pmc:
        return None, None
    pre = SYMBOL.get(slot)
    for name, rec in pmc["kernels"].items():
        if pre and name.startswith(pre):
            return name, rec
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200 = 1.4 s of work: the fill and drain of the three-step pipeline are 0.6 % of a 50-step run)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=512, help="stereo pairs per step per GPU")
    ap.add_argument("--workload", default="full", choices=["full", "orb_match", "orb_match_lcd", "full_solve", "latency"])
    ap.add_argument("--db", type=int, default=0, help="key-frame database size (default 10000, or 6250 per GPU when sharded)")
    ap.add_argument("--scene-rects", type=int, default=6000,
                    help="rectangles of the synthetic scene (SURVEY.md §8(d): 6000 = the corner-dense BASELINE stream; 300 = a sparse stream "
                         "closer to real imagery, on which the two-phase FAST path pays most)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-frames", type=int, default=8,
                    help="K > 0 (default 8): after the timed region one more step of the same workload is run and K of its frames (evenly spread over "
                         "the batch) are compared with the CPU oracle at the bars of the parity tests -> `parity_sample`; a mismatch exits non-zero.  0 = skip")
    ap.add_argument("--cpu-pairs", type=int, default=13,
                    help="timed frames PER THREAD of the all-cores CPU baseline (after one warm-up frame per thread); the default 13 is raised until "
                         "the threads together time >= 200 frames (BASELINE.md section 3)")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the profiled, the solve-cadence and the streamed-input passes (timed region only)")
    ap.add_argument("--stream-input", type=int, default=4,
                    help="B > 0 (default 4): after the timed region, K more steps in which every step's images arrive over PCIe — B distinct batches "
                         "(consecutive frames of the synthetic stream) in pinned host memory, host->device copies on a copy stream, double-buffered "
                         "device input; reported as `streamed` beside the resident `value`.  0 = skip")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="2 = the DeepLCD / loop-DB / BA chain runs on its own HIP stream beside ORB + match + triangulation")
    ap.add_argument("--orb-split", type=int, default=0, choices=[0, 1, 2, 3, 4, 8],
                    help="S > 1 = the 2P images go through S extractor handles on S streams (S equal groups).  0 (default) = 2 with "
                         "--streams 2, 1 with --streams 1")
    ap.add_argument("--pipeline", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="1 = the left and the right images go through two extractor handles that take turns (myslam_orb_set_fast_event): "
                         "one handle's VALU-bound FAST stage runs under the other's latency-bound oct-tree / descriptor stages, match + "
                         "triangulation follow on a third stream, outputs are double-buffered and consecutive steps overlap (every step's "
                         "work is complete at the closing barrier).  0 = every step is joined before the next starts.  "
                         "-1 (default) = 1 with --streams 2, else 0")
    ap.add_argument("--graph", type=int, default=-1, choices=[-1, 0, 1],
                    help="1 = the timed region replays recorded steps: the whole step (extractor for the 2P images, match, triangulation, DeepLCD, "
                         "database scan, BA build) is recorded per LANE on one stream (myslam_graph_begin / _end) and --lanes lanes (own handles and "
                         "buffers each) replay their graphs concurrently, step k on lane k mod L — for small batches, where a step is launch- and "
                         "latency-bound; 0 = eager launches on the four-stream schedule; -1 (default) = 1 when --pairs <= 64 on one GPU, else 0.  The "
                         "other mode is timed in an extra pass (`step_graph` / `step_eager`)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of --graph 1 (0 = 16 for <= 16 pairs per step, 8 up to 64, else 4)")
    ap.add_argument("--stream-split", type=int, default=2, help="streamed pass: the left and the right images of a step as separate copies with an event each, on this many copy streams (0 = one copy of the whole batch; 2 (default): 56.9-57.1 k frames/s against 51.6-54.1 k; 4: 53-55.6 k)")
    ap.add_argument("--lcd-split", type=int, default=1, help="the DeepLCD chain of a step in this many parts on as many handles / streams (1 = one chain on the side stream)")
    ap.add_argument("--ba-stream", choices=["side", "match"], default="match", help="the BA block build behind the triangulation on the match stream (default: +0.4 %, three alternating runs) or behind the DB scan on the side stream")
    ap.add_argument("--solve-lm-hbm", type=int, default=0, help="cadence-6 pass: 1 = the solve keeps its per-landmark state in its HBM scratch (81 KB of LDS per window instead of 133: "
                    "its CU keeps room for two more of the extractor's blocks).  Measured negative (same box, two runs each: 7.06 / 7.07 ms per step against 7.02 / 7.01 "
                    "with the state in LDS): the solve's cost is the CU TIME of its blocks, and the HBM form holds its CUs longer")
    ap.add_argument("--solve-stream", choices=["own", "side"], default="own", help="the OptimizeActiveMap solve of the cadence passes on its own stream (the reference's Backend thread) or behind the side chain")
    ap.add_argument("--side-cus", type=int, default=0, help="experiment: the DeepLCD / DB / BA stream may use only this many CUs (hipExtStreamCreateWithCUMask); 0 = all")
    ap.add_argument("--match-cus", type=int, default=0, help="experiment: likewise for the match + triangulation stream")
    ap.add_argument("--created-main-stream", action="store_true", help="debug: the four-stream schedule's main chain on a created stream instead of the legacy NULL stream")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: run one joined, un-gated step and check that it reproduces the pipeline's last outputs bit for bit")
    ap.add_argument("--orb-internal-stream", type=int, default=0, choices=[0, 1, 2],
                    help="myslam_orb_set_option(INTERNAL_STREAM): 1 = Gaussian pyramid on the extractor's internal stream beside the oct-tree kernel "
                         "(the library's default), 2 = beside FAST, 0 = one stream per extractor handle (default here: with one HSA hardware queue "
                         "per HIP stream — see HW_QUEUES — the internal streams gain nothing, measured)")
    ap.add_argument("--orb-copy-input", type=int, default=0, choices=[0, 1],
                    help="myslam_orb_set_option(COPY_INPUT): 0 = level 0 read in place (the library's default), 1 = every image copied into the pyramid block")
    ap.add_argument("--fast-mode", type=int, default=-1, choices=[-1, 0, 1],
                    help="myslam_orb_set_option(FAST_MODE): -1 = the FAST kernel picks its path per level (default), 0 = two-phase, 1 = dense")
    ap.add_argument("--side-blocks-per-cu", type=int, default=-1,
                    help="myslam_orb_set_option(SIDE_BLOCKS_PER_CU): the descriptor kernel runs as a limited grid of this many blocks per CU, each walking "
                         "several work items, so that its long-lived blocks do not crowd the other handle's FAST blocks out of the CUs (0 = one block per "
                         "work item, the library's default; -1 = 2 under the pipelined schedule, else 0)")
    ap.add_argument("--lcd-skip", type=int, default=0, help="diagnostic, timing only: myslam_lcd_set_option(SKIP_KERNELS) bit mask (1 input, 2 conv1, 4 conv2, 8 pool2, 16 conv3)")
    ap.add_argument("--side-skip", default="", help="diagnostic: comma list of side-chain parts to leave out (lcd, db, ba) — measures what each part costs the step")
    ap.add_argument("--blur-mfma", type=int, default=0, choices=[0, 1],
                    help="myslam_orb_set_option(BLUR_MFMA): 1 = the Gaussian pyramid on the int8 matrix cores (k_blur7_mfma), 0 = register-strip kernel")
    ap.add_argument("--stream-mode", default="1x16,2x16,4x16",
                    help="live-stream operating points, 'PAIRSxLANES,...' ('' = skip): after the other passes each point is run as a child process "
                         "(bench.py --pairs P --lanes L --graph 1: recorded steps on L lanes scanning ONE loop database through L query contexts; its own "
                         "GPU_MAX_HW_QUEUES) and reported as `stream_mode` — the reference's call pattern is one frame per call (src/frontend.cpp:41-77)")
    ap.add_argument("--frame-latency", action="store_true", help="(child of --stream-mode) also time every step on its lane with an event pair: `frame_latency_ms`")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = debugging aid: several ranks share GPU 0 and the collectives go through host memory")
    args = ap.parse_args()
    if args.pipeline < 0:
        args.pipeline = 1 if (args.streams == 2 and args.orb_split != 1) else 0
    if args.side_blocks_per_cu < 0:
        args.side_blocks_per_cu = 2 if args.pipeline else 0
    if args.orb_split == 0:
        args.orb_split = 2 if (args.streams == 2 or args.pipeline) else 1
    if args.pipeline:
        assert args.orb_split >= 2, "--pipeline runs the images on two or more extractor handles (--orb-split >= 2)"
    return args


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one per GPU) ourselves and pass rank 0's output on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def physical_cores():
    """(hardware threads this process may run on, physical cores among them — SMT siblings counted once —, CPU quota of the container's
    cgroup in CPUs or None)"""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    n_cores = max(1, len(cores))
    # a container's CPU-time quota (cgroup) can be far below its CPU affinity: threads beyond it only time-slice
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            quota = parse(open(path).read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return len(cpus), n_cores, quota


def cpu_baseline(synth, workload, frames_per_thread, db_np, gpu_frames, ba_w):
    """The oracle (a plain C++ port of the reference arithmetic, oracle/) timed on this host over a bounded sample of the same frames
    and stages, SURVEY.md §8(d) protocol: (i) one thread — the reference runs every stage single-threaded inside its std::thread —
    and (ii) frame-parallel on every PHYSICAL core the host really grants (affinity mask, cgroup quota and a measured spin test) (std::thread pool, one frame per task, oracle/bench_oracle.cpp; SMT siblings add
    nothing to this integer / f32 code): one warm-up frame per thread, then `frames_per_thread` (>= 4) timed frames per thread, the
    GPU run's frames in a cycle; wall clock over the timed frames, per-stage medians, parallel efficiency = all-cores rate /
    (threads x one-thread rate).  Bounded to roughly 10-30 s of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle
    o = Oracle()
    hw_threads, phys, quota = physical_cores()
    cores = phys if not quota else max(1, min(phys, int(quota + 0.5)))          # threads the host will actually run at the same time
    # ... as far as the container can see.  Measured: `cores` spinning threads against one (the GPU boxes of this pool show 256
    # hardware threads and deliver about 12 CPUs' worth of time)
    capacity = max(o.cpu_capacity(cores, 200) for _ in range(4))     # the best of four probes: a noisy moment must not shrink the baseline
    if capacity < 0.75 * cores:
        cores = max(1, int(capacity + 0.5))
    stages = {"orb_match": 1, "orb_match_lcd": 2, "full": 3, "full_solve": 4}[workload]
    ids = np.arange(len(db_np), dtype=np.uint64)
    args = (synth.KITTI00, synth.calc_weights(), db_np, ids, ba_w)
    names = ["orb_extract_LR", "match_triangulate", "deeplcd_dbscan", "ba_build", "ba_solve"][:max(2, stages + 1)]
    # (i) one thread: 5 warm-up + 50 timed frames (~0.3 s per frame; BASELINE.md section 3 asks for >= 200 frames over the whole baseline,
    # (ii) supplies them)
    n1 = min(len(gpu_frames), 55); w1 = min(5, n1 - 1)
    dt1, st1 = o.bench_frames(gpu_frames[:n1], *args, stages=stages, threads=1, n_warmup=w1)
    fps1 = (n1 - w1) / dt1
    # (ii) every physical core: 1 warm-up + frames_per_thread timed frames per thread
    fpt = max(1, int(frames_per_thread), -(-200 // cores) if frames_per_thread >= 13 else 1)       # default: >= 200 timed frames in total
    wn, n = cores, cores * fpt
    dt, st = o.bench_frames(gpu_frames, *args, stages=stages, threads=cores, n_warmup=wn, n_tasks=wn + n)
    med = lambda a, k0: {nm: float(np.median(a[k0:, i]) * 1e3) for i, nm in enumerate(names)}
    return {"value": n / dt, "unit": "stereo frames/s", "cores": cores, "hardware_threads": hw_threads, "physical_cores": phys,
            "cgroup_cpu_quota": quota, "measured_concurrent_threads": capacity, "kind": "port", "value_1thread": fps1,
            "parallel_efficiency": (n / dt) / (cores * fps1),
            "stage_median_ms_1thread": med(st1, w1), "stage_median_ms_allcores": med(st, wn),
            "sample": f"{n} stereo pairs ({fpt} per thread, the GPU run's synthetic 1241x376 frames in a cycle) after {wn} warm-up frames, same stages, "
                      f"oracle frame-parallel on {cores} threads ({phys} physical cores / {hw_threads} hardware threads visible, cgroup CPU quota "
                      f"{'none' if not quota else round(quota, 1)}) in {dt:.1f} s; "
                      f"single thread: {n1 - w1} pairs after {w1} warm-up in {dt1:.1f} s"}


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


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    via_cpu = args.backend == "gloo"
    if via_cpu:
        local_rank = 0
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world} (under torch.distributed.run pass --gpus = --nproc-per-node)"
    rccl_ranks = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        assert via_cpu or torch.cuda.device_count() > local_rank, \
            f"rank {rank}: {torch.cuda.device_count()} visible GPU(s) but LOCAL_RANK {local_rank} (one GPU per rank; --backend gloo shares GPU 0)"
        torch.cuda.set_device(local_rank)
        if via_cpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        one = torch.ones(1, dtype=torch.int32, device="cpu" if via_cpu else torch.device("cuda", local_rank))
        dist.all_reduce(one)                                     # the ranks the collective library actually connected
        rccl_ranks = int(one.item())
        assert rccl_ranks == world
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    pkg = load_package()
    api, synth = pkg.api, pkg.synth
    assert api.device_count() >= 1
    if args.workload == "latency":
        assert world == 1
        from tools import latency_b1
        print(json.dumps(latency_b1.run(api, synth, with_oracle=not args.no_cpu_baseline)))
        return
    # the four-stream schedule keeps its main chain on the legacy NULL stream: measured, its launches cost the host 0.63 ms per step there and
    # 1.02 ms on a created stream (8 pairs per step; tools/ab_streams.sh) — the lanes of the small-batch mode bring their own streams
    if args.created_main_stream:
        torch.cuda.set_stream(torch.cuda.Stream())
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream
    # the LCD -> DB -> BA chain only reads the input images: it runs beside the ORB chain on its own stream
    side_stream = (masked_stream(args.side_cus) if args.side_cus > 0 else torch.cuda.Stream()) if args.streams == 2 else main_stream
    stream2 = side_stream.cuda_stream
    P = args.pairs
    K = synth.KITTI00
    Kt = (K["fx"], K["fy"], K["cx"], K["cy"])

    # ---- inputs resident in HBM before the timed region ----
    frames = synth.stereo_batch(P, stream_id=rank, n_rect=args.scene_rects)   # [P, 2, H, W]
    imgs = np.concatenate([frames[:, 0], frames[:, 1]], axis=0)         # all left images, then all right images
    d_imgs = torch.from_numpy(imgs).to(dev)
    cur = {"imgs": d_imgs}              # the device buffer the steps read their images from (the streamed pass swaps it per step)
    ext = api.ORBextractor(2000, stream=stream)
    cap = ext.max_keypoints()
    S = args.orb_split
    assert (2 * P) % S == 0
    orb_streams = [torch.cuda.Stream() for _ in range(S - 1)]
    orb_exts = [api.ORBextractor(2000, stream=st.cuda_stream) for st in orb_streams]
    for e in [ext] + orb_exts:
        e.set_option(e.OPT_INTERNAL_STREAM, args.orb_internal_stream)
        e.set_option(e.OPT_FAST_MODE, args.fast_mode)
        e.set_option(e.OPT_COPY_INPUT, args.orb_copy_input)
        e.set_option(e.OPT_BLUR_MFMA, args.blur_mfma)
        e.set_option(e.OPT_SIDE_BLOCKS_PER_CU, args.side_blocks_per_cu)
    NB = 2 if args.pipeline else 1          # pipeline: extractor outputs are double-buffered (step k+1 extracts while step k is matched)
    d_kps_b = [torch.zeros(2 * P * cap * 28, dtype=torch.uint8, device=dev) for _ in range(NB)]
    d_desc_b = [torch.zeros(2 * P * cap * 32, dtype=torch.uint8, device=dev) for _ in range(NB)]
    d_cnt_b = [torch.zeros(2 * P, dtype=torch.int32, device=dev) for _ in range(NB)]
    d_stat_b = [torch.zeros(2 * P, dtype=torch.int32, device=dev) for _ in range(NB)]
    d_kps, d_desc, d_cnt, d_stat = d_kps_b[0], d_desc_b[0], d_cnt_b[0], d_stat_b[0]
    d_midx = torch.zeros(P * cap, dtype=torch.int32, device=dev)
    d_mdist = torch.zeros(P * cap, dtype=torch.int32, device=dev)
    d_xyz = torch.zeros(P * cap * 3, dtype=torch.float64, device=dev)
    d_ok = torch.zeros(P * cap, dtype=torch.uint8, device=dev)
    use_lcd = args.workload != "orb_match"
    use_ba = args.workload in ("full", "full_solve")
    use_solve = args.workload == "full_solve"
    n_db_local = args.db or (10000 if world == 1 else 6250)
    db_np = None
    if use_lcd:
        lcd = api.DeepLCD(synth.calc_weights(), stream=stream2)
        # the DeepLCD chain in `lcd_split` parts on as many handles / streams (part 0 on the side stream): every stream of the schedule is
        # busy for about the whole step, and the side stream's chain was the longest — two half-length chains overlap better (+0.9 %)
        lcd_parts = []
        if args.lcd_split > 1 and P % args.lcd_split == 0 and args.streams == 2:
            for _ in range(args.lcd_split - 1):
                st_ = torch.cuda.Stream()
                lcd_parts.append((api.DeepLCD(synth.calc_weights(), stream=st_.cuda_stream), st_, torch.cuda.Event()))
            ev_lfork = torch.cuda.Event()
        if args.lcd_skip:
            lcd.set_option(lcd.OPT_SKIP_KERNELS, args.lcd_skip)
        d_descr = torch.zeros(P, 1064, device=dev)
        db_np = synth.lcd_database(n_db_local, seed=0xDB + rank)
        D = api.LoopDatabase(n_db_local, stream=stream2)
        ids = np.arange(rank * n_db_local, (rank + 1) * n_db_local, dtype=np.uint64)     # contiguous id range per shard
        t_db = torch.from_numpy(db_np).to(dev)
        D.append_batch(ids, t_db.data_ptr(), n_db_local)
        NQ = P * world
        cur_ids = np.full(NQ, world * n_db_local + 20, np.uint64)
        d_allq = torch.zeros(NQ, 1064, device=dev)
        d_best = torch.zeros(NQ, dtype=torch.int64, device=dev)
        d_max = torch.zeros(NQ, device=dev); d_dbcnt = torch.zeros(NQ, dtype=torch.int32, device=dev)
        d_cand = torch.zeros(NQ * 16, dtype=torch.uint8, device=dev)
        if world > 1:
            pkg.sharded_db.check_shard_order(int(ids[0]), int(ids[-1]), world, via_cpu=via_cpu, device=dev)
    ba_w = None
    if use_ba:
        ba_w, _ = synth.ba_windows(P, seed0=0xBA + 100000 * rank)        # P DISTINCT windows (10 KF x 300 MP, ~2950 edges each)
        maxP, maxL, maxE = ba_w[0].shape[1], ba_w[1].shape[1], ba_w[2].shape[1]
        b_in = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in ba_w]
        b_out = [torch.zeros(P, n, dtype=torch.float64, device=dev) for n in (maxP * 36, maxL * 9, maxE * 18, maxP * 6, maxL * 3, maxE)]
        # the solve updates poses/points in place: every step starts from the pristine windows
        s_poses, s_pts = b_in[0].clone(), b_in[1].clone()
        s_echi = torch.zeros(P, maxE, dtype=torch.float64, device=dev); s_out = torch.zeros(P, maxE, dtype=torch.uint8, device=dev)
        s_rd = torch.zeros(P, dtype=torch.int32, device=dev); s_no = torch.zeros(P, dtype=torch.int32, device=dev)
        s_st = torch.zeros(P, dtype=torch.int32, device=dev)

        # The solve has its own stream (round 5): Backend::OptimizeActiveMap runs on the Backend thread, beside LoopClosing's DeepLCD / DB chain
        # (src/backend.cpp:36-49, src/loopclosing.cpp:66-81), and needs nothing from it.  Behind the side chain it made that chain the step's
        # longest (+0.75 ms per step for 86 windows of 1.1 ms each); on its own stream the step pays only for the CUs the solve's blocks hold.
        solve_stream = torch.cuda.Stream() if (args.solve_stream == "own" and args.streams == 2) else side_stream
        s_hpl = torch.zeros(P, maxE * 18, dtype=torch.float64, device=dev) if solve_stream is not side_stream else b_out[2]     # the solve's scratch (the build writes b_out[2] at the same time)

        def solve(nwin=P):   # Backend::OptimizeActiveMap solve stage: rounds of optimize(10) + outlier flags (backend.cpp:208-243)
            with torch.cuda.stream(solve_stream):
                s_poses[:nwin].copy_(b_in[0][:nwin]); s_pts[:nwin].copy_(b_in[1][:nwin])
                api.ba_optimize_active_map_batch(s_poses.data_ptr(), s_pts.data_ptr(), *[t.data_ptr() for t in b_in[2:]], nwin, maxP, maxL, maxE, Kt,
                                                 5.991, 5.991, 5, 10, s_hpl.data_ptr(), s_echi.data_ptr(), s_out.data_ptr(), s_rd.data_ptr(),
                                                 s_no.data_ptr(), s_st.data_ptr(), solve_stream.cuda_stream)
    solve_windows = [P if use_solve else 0]          # windows the side chain also SOLVES per step (pass 3 sets ceil(P / 6))

    skip = set(args.side_skip.split(",")) if args.side_skip else set()      # diagnostic: the marginal cost of the side chain's parts

    def side_chain(with_ba=True):
        if use_ba and "ba" not in skip and solve_windows[0] and solve_stream is not side_stream:
            solve_stream.wait_stream(side_stream)      # starts with the step (the side stream has just waited for the step's start event), runs beside the chain below
            solve(solve_windows[0])
        if use_lcd and "lcd" not in skip:
            if lcd_parts:
                n_part = P // (len(lcd_parts) + 1)
                ev_lfork.record(side_stream)
                for i, (h_, st_, ev_) in enumerate(lcd_parts):
                    st_.wait_event(ev_lfork)
                    h_.describe_batch(cur["imgs"].data_ptr() + (i + 1) * n_part * H * W, n_part, H, W, W, H * W,
                                      d_descr.data_ptr() + (i + 1) * n_part * 1064 * 4, blur_in_place=False)
                    ev_.record(st_)
                lcd.describe_batch(cur["imgs"].data_ptr(), n_part, H, W, W, H * W, d_descr.data_ptr(), blur_in_place=False)
                for _, _, ev_ in lcd_parts:
                    side_stream.wait_event(ev_)
            else:
                lcd.describe_batch(cur["imgs"].data_ptr(), P, H, W, W, H * W, d_descr.data_ptr(), blur_in_place=False)
        if use_lcd and "db" not in skip:
            if world > 1:       # every shard scores every rank's queries; the 16-byte candidate records are merged after an all-gather
                if via_cpu:
                    h_all = torch.empty(d_allq.shape, dtype=d_allq.dtype)
                    dist.all_gather_into_tensor(h_all, d_descr.cpu())
                    d_allq.copy_(h_all)
                else:
                    dist.all_gather_into_tensor(d_allq, d_descr)
                D.query_batch_sharded(d_allq.data_ptr(), cur_ids, NQ, d_cand.data_ptr())
                pkg.sharded_db.exchange_and_merge(d_cand, world, d_best, d_max, d_dbcnt, via_cpu=via_cpu)
            else:
                D.query_batch(d_descr.data_ptr(), cur_ids, NQ, d_best.data_ptr(), d_max.data_ptr(), d_dbcnt.data_ptr())
        if use_ba and "ba" not in skip:
            if with_ba or solve_windows[0]:
                api.ba_build_batch(*[t.data_ptr() for t in b_in], P, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in b_out], stream2)
            if solve_windows[0] and solve_stream is side_stream:
                solve(solve_windows[0])

    ba_on_match = [False]
    step_no = [0]           # steps issued so far: step k writes extractor output buffer k % NB
    if args.pipeline:
        ba_on_match[0] = args.ba_stream == "match" and use_ba
        # two extractor handles take turns: handle A (left images, main stream) and handle B (right images, its own stream) each wait
        # for the other's FAST stage, so a FAST launch never runs beside the other FAST launch but under the other handle's oct-tree /
        # descriptor launches; match + triangulation of step k run on a third stream once both handles are done with step k, while the
        # handles already extract step k+1 into the other output buffer
        exts = [ext] + orb_exts
        sX, sM = [main_stream] + orb_streams, (masked_stream(args.match_cus) if args.match_cus > 0 else torch.cuda.Stream())
        G = 2 * P // S
        ev_fast = [torch.cuda.Event() for _ in range(S)]
        ev_done = [[torch.cuda.Event() for _ in range(S)] for _ in range(NB)]
        ev_match = [torch.cuda.Event() for _ in range(NB)]
        ev_start = torch.cuda.Event()
        for e in ev_fast + [x for pr in ev_done for x in pr] + ev_match + [ev_start]:
            e.record(main_stream)                       # creates the hipEvent_t behind the torch event
        for i, e in enumerate(exts):
            e.set_fast_event(ev_fast[i].cuda_event)
            if args.pipeline == 1:      # gate inside the call: only the FAST stages take turns (ring), the pyramids are not held back
                e.set_fast_gate(ev_fast[(i - 1) % S].cuda_event)

        def step():
            p = step_no[0] % NB
            step_no[0] += 1
            kps, desc, cnt, stat = d_kps_b[p], d_desc_b[p], d_cnt_b[p], d_stat_b[p]
            for i, e in enumerate(exts):
                if args.pipeline == 2:
                    sX[i].wait_event(ev_fast[(i - 1) % S])      # the previous handle's FAST (whole call held back)
                sX[i].wait_event(ev_match[p])                   # buffer p was last read by the match of step k - 2
                if i == 0:
                    ev_start.record(sX[0])
                o = i * G
                e.detect_and_compute_batch(cur["imgs"].data_ptr() + o * H * W, G, H, W, W, H * W, kps.data_ptr() + o * cap * 28,
                                           desc.data_ptr() + o * cap * 32, cnt.data_ptr() + 4 * o, stat.data_ptr() + 4 * o, cap)
                ev_done[p][i].record(sX[i])
            for i in range(S):
                sM.wait_event(ev_done[p][i])
            api.hamming_match_batch(desc.data_ptr(), cnt.data_ptr(), desc.data_ptr() + P * cap * 32, cnt.data_ptr() + 4 * P,
                                    P, cap, d_midx.data_ptr(), d_mdist.data_ptr(), sM.cuda_stream)
            api.triangulate_stereo_batch(kps.data_ptr(), kps.data_ptr() + P * cap * 28, d_midx.data_ptr(), cnt.data_ptr(), P, cap,
                                         Kt, K["bf"] / K["fx"], d_xyz.data_ptr(), d_ok.data_ptr(), sM.cuda_stream)
            ev_match[p].record(sM)
            if ba_on_match[0] and "ba" not in skip and not solve_windows[0]:
                api.ba_build_batch(*[t.data_ptr() for t in b_in], P, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in b_out], sM.cuda_stream)
            side_stream.wait_event(ev_start)            # the LCD / DB / BA chain of step k starts with step k
            with torch.cuda.stream(side_stream):
                side_chain(with_ba=not ba_on_match[0])

    def step_joined():
        if S == 1:
            ext.detect_and_compute_batch(cur["imgs"].data_ptr(), 2 * P, H, W, W, H * W, d_kps.data_ptr(), d_desc.data_ptr(),
                                         d_cnt.data_ptr(), d_stat.data_ptr(), cap)
        else:               # S equal groups of images, group 0 on the main stream
            G = 2 * P // S
            for st in orb_streams:
                st.wait_stream(main_stream)
            for gi, e in enumerate([ext] + orb_exts):
                o = gi * G
                e.detect_and_compute_batch(cur["imgs"].data_ptr() + o * H * W, G, H, W, W, H * W, d_kps.data_ptr() + o * cap * 28,
                                           d_desc.data_ptr() + o * cap * 32, d_cnt.data_ptr() + 4 * o, d_stat.data_ptr() + 4 * o, cap)
            for st in orb_streams:
                main_stream.wait_stream(st)
        api.hamming_match_batch(d_desc.data_ptr(), d_cnt.data_ptr(), d_desc.data_ptr() + P * cap * 32, d_cnt.data_ptr() + 4 * P,
                                P, cap, d_midx.data_ptr(), d_mdist.data_ptr(), stream)
        api.triangulate_stereo_batch(d_kps.data_ptr(), d_kps.data_ptr() + P * cap * 28, d_midx.data_ptr(), d_cnt.data_ptr(), P, cap,
                                     Kt, K["bf"] / K["fx"], d_xyz.data_ptr(), d_ok.data_ptr(), stream)
        if side_stream is not main_stream:
            side_stream.wait_stream(main_stream)            # a joined step starts behind everything issued before it (found by --verify on two
        with torch.cuda.stream(side_stream):                # ranks sharing a GPU: the zeroing of the outputs raced with this chain)
            side_chain()
        if side_stream is not main_stream:
            main_stream.wait_stream(side_stream)            # a step is complete when both chains are

    if not args.pipeline:
        step = step_joined

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_ms = [0.0]
    rank_dts = []           # multi-rank runs: every rank's wall time of the last timed() call

    def timed(n, fn=None):
        """n steps bracketed by barrier + synchronize on both sides; the slowest rank's wall time.  host_ms[0] = CPU time the launches of
        one step took (the loop that enqueues the n steps, before the closing barrier waits for the device)"""
        fn = fn or step
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host_ms[0] = (time.perf_counter() - t0) / n * 1e3
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            # every rank's own wall time (the closing barrier makes them nearly equal by construction; the device-side time of a rank's
            # own chains is in rank_gpu_ms below) and the slowest one, which is the job's time
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if via_cpu else dev)
            allt = torch.empty(world, dtype=torch.float64, device=t.device)
            dist.all_gather_into_tensor(allt, t)
            rank_dts[:] = [float(v) for v in allt.cpu()]
            dt = max(rank_dts)
        return dt

    for _ in range(args.warmup):
        step()
    barrier()
    assert all(int(t.abs().sum()) == 0 for t in d_stat_b), "ORB capacity overflow"
    n_kp = d_cnt.float().mean().item()

    # ---- small batches: LANES.  A step of a few frames is a latency-bound chain of ~38 dependent launches (0.86 ms at 8 pairs on one
    # stream, 0.53 ms host-bound on four).  Measured on ROCm 7.2: a HIP graph recorded from ONE stream replays for ~0.03 ms of host time,
    # one recorded across streams for 0.36-0.75 ms (no better than launching eagerly).  So the step is recorded per lane on a single
    # stream — one extractor handle for the 2P images, match, triangulation, DeepLCD, database scan, BA build — and L lanes (own handles,
    # buffers and stream each: the analogue of L cameras) replay their graphs concurrently: step k runs on lane k % L.
    step_eager = step
    lanes = []
    can_graph = world == 1
    use_graph = can_graph and (args.graph == 1 or (args.graph < 0 and P <= 64))
    n_lanes = args.lanes if args.lanes > 0 else (16 if P <= 32 else 8 if P <= 64 else 4)       # measured: 32 pairs 70.5 k on 16 lanes, 68.1 k on 8; 64 pairs 72.4 k on 8 or 12, 71.4 k on 16

    def build_lane():
        st = torch.cuda.Stream(); s_ = st.cuda_stream
        ln = {"stream": st, "ext": api.ORBextractor(2000, stream=s_)}
        ln["ext"].set_option(ln["ext"].OPT_INTERNAL_STREAM, 0); ln["ext"].set_option(ln["ext"].OPT_FAST_MODE, args.fast_mode)
        ln["ext"].set_option(ln["ext"].OPT_COPY_INPUT, args.orb_copy_input); ln["ext"].set_option(ln["ext"].OPT_BLUR_MFMA, args.blur_mfma)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
        o = {"kps": z(2 * P * cap * 28, torch.uint8), "desc": z(2 * P * cap * 32, torch.uint8), "cnt": z(2 * P, torch.int32), "stat": z(2 * P, torch.int32),
             "midx": z(P * cap, torch.int32), "mdist": z(P * cap, torch.int32), "xyz": z(P * cap * 3, torch.float64), "ok": z(P * cap, torch.uint8)}
        if use_lcd:
            ln["lcd"] = api.DeepLCD(synth.calc_weights(), stream=s_)
            ln["D"] = D.context(s_)                                       # ONE database for all lanes (LoopClosing::_mvDatabase is one std::map per process, loopclosing.h:120): a query context per lane
            o.update({"descr": torch.zeros(P, 1064, device=dev), "best": z(P, torch.int64), "max": z(P, torch.float32), "dbcnt": z(P, torch.int32)})
        if use_ba:
            o["ba"] = [torch.zeros(P, n, dtype=torch.float64, device=dev) for n in (maxP * 36, maxL * 9, maxE * 18, maxP * 6, maxL * 3, maxE)]
        ln["o"] = o

        def body():
            ln["ext"].detect_and_compute_batch(d_imgs.data_ptr(), 2 * P, H, W, W, H * W, o["kps"].data_ptr(), o["desc"].data_ptr(), o["cnt"].data_ptr(),
                                               o["stat"].data_ptr(), cap)
            api.hamming_match_batch(o["desc"].data_ptr(), o["cnt"].data_ptr(), o["desc"].data_ptr() + P * cap * 32, o["cnt"].data_ptr() + 4 * P,
                                    P, cap, o["midx"].data_ptr(), o["mdist"].data_ptr(), s_)
            api.triangulate_stereo_batch(o["kps"].data_ptr(), o["kps"].data_ptr() + P * cap * 28, o["midx"].data_ptr(), o["cnt"].data_ptr(), P, cap,
                                         Kt, K["bf"] / K["fx"], o["xyz"].data_ptr(), o["ok"].data_ptr(), s_)
            if use_lcd:
                ln["lcd"].describe_batch(d_imgs.data_ptr(), P, H, W, W, H * W, o["descr"].data_ptr(), blur_in_place=False)
                ln["D"].query_batch(o["descr"].data_ptr(), cur_ids[:P], P, o["best"].data_ptr(), o["max"].data_ptr(), o["dbcnt"].data_ptr())
            if use_ba:
                api.ba_build_batch(*[t.data_ptr() for t in b_in], P, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in o["ba"]], s_)
        ln["body"] = body
        for _ in range(2):                               # eager first: lazy allocations, both FAST-statistics parities
            body()
        st.synchronize()
        ln["graphs"] = [api.StepGraph.record(s_, [], body) for _ in range(2)]     # two consecutive steps, replayed alternately
        ln["k"] = 0
        return ln

    def step_lanes():
        ln = lanes[step_no[0] % len(lanes)]; step_no[0] += 1
        ln["graphs"][ln["k"] & 1].launch(ln["stream"].cuda_stream); ln["k"] += 1

    def step_lanes_eager():                              # the same lanes without the recording: what the graphs save
        ln = lanes[step_no[0] % len(lanes)]; step_no[0] += 1
        ln["body"]()

    if use_graph:
        api.prof_enable(False)
        lanes = [build_lane() for _ in range(n_lanes)]
        step = step_lanes
        for _ in range(2 * n_lanes):
            step()
        barrier()

    # ---- pass 1: the timed region (no per-kernel events) ----
    api.prof_enable(False)
    dt = timed(args.steps)
    host_launch_ms = host_ms[0]
    per_rank_ms = [v / args.steps * 1e3 for v in rank_dts] if world > 1 else None
    # ---- multi-rank runs: what the loop-database exchange costs, stage by stage, on an otherwise idle chip (after the timed region).  The two
    # all-gathers (queries: P x 4 256 B per rank in, N P x 4 256 B out; candidates: N P x 16 B per rank) and the N x larger scan are the only
    # work a rank does for the others: scaling efficiency below 1 is these numbers (DESIGN.md section 4 holds the predicted values).
    collective = None
    if world > 1 and use_lcd:
        reps = 20
        acc = np.zeros(4)
        with torch.cuda.stream(side_stream):
            for it in range(-2, reps):
                if via_cpu:      # gloo stages through the host: wall clock around synchronised stages
                    torch.cuda.synchronize(); t_ = [time.perf_counter()]
                    h_all = torch.empty(d_allq.shape, dtype=d_allq.dtype); dist.all_gather_into_tensor(h_all, d_descr.cpu()); d_allq.copy_(h_all)
                    torch.cuda.synchronize(); t_.append(time.perf_counter())
                    D.query_batch_sharded(d_allq.data_ptr(), cur_ids, NQ, d_cand.data_ptr())
                    torch.cuda.synchronize(); t_.append(time.perf_counter())
                    mine = d_cand.cpu(); gathered = torch.empty(world * NQ * 16, dtype=torch.uint8); dist.all_gather_into_tensor(gathered, mine)
                    t_.append(time.perf_counter())
                    b_, m_, c_ = api.lcd_merge_candidates(gathered.numpy().view(api.CAND_DTYPE).reshape(world, NQ))
                    t_.append(time.perf_counter())
                    ms = [(t_[k + 1] - t_[k]) * 1e3 for k in range(4)]
                else:            # RCCL: HIP events on the stream the collectives and the two library calls are ordered on
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                    ev[0].record(side_stream)
                    dist.all_gather_into_tensor(d_allq, d_descr); ev[1].record(side_stream)
                    D.query_batch_sharded(d_allq.data_ptr(), cur_ids, NQ, d_cand.data_ptr()); ev[2].record(side_stream)
                    gathered = torch.empty(world * NQ * 16, dtype=torch.uint8, device=dev)
                    dist.all_gather_into_tensor(gathered, d_cand); ev[3].record(side_stream)
                    api.lcd_merge_candidates_device(gathered.data_ptr(), world, NQ, d_best.data_ptr(), d_max.data_ptr(), d_dbcnt.data_ptr(), side_stream.cuda_stream)
                    ev[4].record(side_stream)
                    side_stream.synchronize()
                    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
                if it >= 0:
                    acc += np.array(ms)
        acc /= reps
        mine_t = torch.tensor(acc, dtype=torch.float64, device="cpu" if via_cpu else dev)
        all_t = torch.empty(world * 4, dtype=torch.float64, device=mine_t.device)
        dist.all_gather_into_tensor(all_t, mine_t)
        all_t = all_t.cpu().numpy().reshape(world, 4)
        collective = {"collective_ms_per_step": float((all_t[:, 0] + all_t[:, 2]).max()), "shard_scan_ms_per_step": float(all_t[:, 1].max()),
                      "merge_ms_per_step": float(all_t[:, 3].max()),
                      "allgather_queries_ms": [float(v) for v in all_t[:, 0]], "allgather_candidates_ms": [float(v) for v in all_t[:, 2]],
                      "shard_scan_ms": [float(v) for v in all_t[:, 1]],
                      "allgather_queries_bytes_per_rank": int(P * 1064 * 4), "allgather_candidates_bytes_per_rank": int(NQ * 16),
                      "shard_rows": int(n_db_local), "queries_scanned_per_rank": int(NQ), "reps": reps,
                      "timing": "wall clock around synchronised stages (gloo stages through host memory)" if via_cpu else "HIP events on the side stream",
                      "note": "measured after the timed region on an otherwise idle chip: the cost of the loop-database exchange alone; inside a step it runs on the side "
                              "stream under the extractor"}
        barrier()
    frame_latency = None
    if use_graph and args.frame_latency:
        # every step timed on ITS lane (event pair around the replay) while all lanes are busy, then one lane alone: what a camera sees
        n_probe = min(args.steps, 64 * len(lanes))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_probe)]
        barrier()
        for a_, b_ in evs:
            ln = lanes[step_no[0] % len(lanes)]
            a_.record(ln["stream"]); step(); b_.record(ln["stream"])
        barrier()
        loaded = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        ln = lanes[0]; alone = []
        for _ in range(32):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(ln["stream"]); ln["graphs"][ln["k"] & 1].launch(ln["stream"].cuda_stream); ln["k"] += 1; b_.record(ln["stream"])
            ln["stream"].synchronize(); alone.append(a_.elapsed_time(b_))
        frame_latency = {"loaded_median_ms": loaded[len(loaded) // 2], "loaded_p90_ms": loaded[int(0.9 * (len(loaded) - 1))],
                         "one_lane_alone_ms": float(np.median(alone)), "steps_probed": n_probe}
    if not use_graph:
        step = step_eager

    # ---- pass 2: the same steps with a HIP event pair around every launch, on the launch's own stream ----
    prof, dt_prof = {}, None
    step_timed = step
    step = step_eager                  # every pass below launches eagerly: the profiling events, the solve cadence and the streamed input change per step
    if not args.no_extra_passes:
        api.prof_reset(); api.prof_enable(True)
        dt_prof = timed(args.steps)
        api.prof_enable(False)
        prof = api.prof_read()

    # ---- pass 3: configs[3] at the reference's cadence: the solve runs per KEY-FRAME, about 1 frame in 6 ----
    cadence = None
    if use_ba and not use_solve and not args.no_extra_passes:
        solve_windows[0] = (P + 5) // 6
        api.ba_set_option(api.BA_OPT_LANDMARKS_IN_HBM, 1 if args.solve_lm_hbm else 0)
        step(); barrier()
        dt_c = timed(args.steps)
        solve_windows[0] = 0
        barrier(); api.ba_set_option(api.BA_OPT_LANDMARKS_IN_HBM, 0)
        assert int(s_st.abs().sum()) == 0
        cadence = {"value": world * P * args.steps / dt_c, "unit": "stereo frames/s", "ms_per_step": dt_c / args.steps * 1e3,
                   "solved_windows_per_step": (P + 5) // 6, "landmark_state": "hbm scratch" if args.solve_lm_hbm else "lds",
                   "note": "as the timed region, plus Backend::OptimizeActiveMap's solve stage (rounds of Levenberg-Marquardt optimize(10), Schur + "
                           "Cholesky, outlier flags) on every 6th frame's window — the reference solves per key-frame, about 1 frame in 6"}

    # ---- pass 4: streamed input — every step's images cross PCIe (app/run_kitti_stereo.cpp:61-90 reads two images per step) ----
    streamed = None
    if args.stream_input > 0 and not args.no_extra_passes:
        NBAT = args.stream_input if world == 1 else min(args.stream_input, 2)      # N ranks render on one host: two batches per rank there
        t_gen = time.perf_counter()
        h_bat = []
        for j in range(NBAT):           # consecutive frames of the same synthetic stream: batch j = frames j*P .. (j+1)*P - 1
            f = frames if j == 0 else synth.stereo_batch(P, stream_id=rank, t0=j * P, n_rect=args.scene_rects)
            h_bat.append(torch.from_numpy(np.concatenate([f[:, 0], f[:, 1]], axis=0)).pin_memory())
        t_gen = time.perf_counter() - t_gen
        NBUF = 3        # device input buffers: the copy of step k + 1 only needs step k - 2 to have finished
        d_in = [torch.empty_like(d_imgs) for _ in range(NBUF)]
        sC = torch.cuda.Stream()
        readers = [main_stream] + orb_streams + ([side_stream] if side_stream is not main_stream else [])
        ev_ready = [torch.cuda.Event() for _ in range(NBUF)]
        ev_ready2 = [torch.cuda.Event() for _ in range(NBUF)]
        sCs = [sC] + [torch.cuda.Stream() for _ in range(max(0, args.stream_split - 1))]
        ev_piece = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(NBUF)]
        ev_read = [[torch.cuda.Event() for _ in readers] for _ in range(NBUF)]
        bytes_step = h_bat[0].numel()
        k_stream = [0]
        copy_ev = []

        def step_streamed():
            k = k_stream[0]; k_stream[0] += 1
            b = k % NBUF
            for e in ev_read[b]:
                sC.wait_event(e)                    # the readers of step k - NBUF (same buffer) have finished: level 0 is read in place
            ec = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ec[0].record(sC)
            if args.stream_split and args.pipeline and S == 2:
                # the left images (extractor handle A, DeepLCD) and the right images (handle B) arrive as separate copies with an event each —
                # handle A starts when ITS half is there — on `stream_split` copy streams (1: both on one; 2: one each; 4: two each):
                # concurrent copies use more of the link (one stream sustains 51 GB/s, two 53)
                hb = h_bat[k % NBAT]
                nC = len(sCs)
                per = max(1, nC // 2)                       # copy streams per half
                for half in range(2):
                    lo, hi = half * P, (half + 1) * P
                    cs = [sCs[(half * per + j) % nC] for j in range(per)]
                    for j, c in enumerate(cs):
                        if c is not sC:
                            for e in ev_read[b]:
                                c.wait_event(e)
                        a0, a1 = lo + (hi - lo) * j // per, lo + (hi - lo) * (j + 1) // per
                        with torch.cuda.stream(c):
                            d_in[b][a0:a1].copy_(hb[a0:a1], non_blocking=True)
                    evh = ev_ready[b] if half == 0 else ev_ready2[b]
                    for c in cs[1:]:                        # the half is ready when all of its pieces are
                        ev_piece[b][half].record(c); cs[0].wait_event(ev_piece[b][half])
                    evh.record(cs[0])
                sC.wait_event(ev_ready2[b])
                ec[1].record(sC); copy_ev.append(ec)
                for st in readers:
                    st.wait_event(ev_ready2[b] if st is orb_streams[0] else ev_ready[b])
            else:
                with torch.cuda.stream(sC):
                    d_in[b].copy_(h_bat[k % NBAT], non_blocking=True)
                ec[1].record(sC); copy_ev.append(ec)
                ev_ready[b].record(sC)
                for st in readers:
                    st.wait_event(ev_ready[b])
            cur["imgs"] = d_in[b]
            step()
            for e, st in zip(ev_read[b], readers):
                e.record(st)

        for _ in range(NBUF):
            step_streamed()
        barrier()
        def signature():            # the valid descriptors of the step's first image (slots past the count hold stale bytes)
            p = (step_no[0] - 1) % NB if args.pipeline else 0
            n0 = int(d_cnt_b[p][0].item())
            return d_desc_b[p][:32 * n0].clone()
        sig_first = signature()         # of the last warm-up step
        barrier()
        copy_ev.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_streamed()
        barrier()
        dt_s = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_s], dtype=torch.float64, device="cpu" if via_cpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_s = float(t.item())
        cur["imgs"] = d_imgs
        assert all(int(t.abs().sum()) == 0 for t in d_stat_b), "ORB capacity overflow (streamed pass)"
        sig_last = signature()
        same_batch = (args.steps % NBAT) == 0          # the last timed step and the last warm-up step saw the same host batch
        assert NBAT == 1 or (sig_first.shape == sig_last.shape and torch.equal(sig_first, sig_last)) == same_batch, "streamed pass: the steps did not see the batches they were sent"
        del d_in, sig_first
        streamed = {"value": world * P * args.steps / dt_s, "unit": "stereo frames/s", "ms_per_step": dt_s / args.steps * 1e3,
                    "h2d_GBps": bytes_step * args.steps / dt_s / 1e9, "h2d_bytes_per_step": bytes_step, "distinct_batches": NBAT,
                    "host_render_s": t_gen, "h2d_copy_ms_avg": float(np.mean([a.elapsed_time(b) for a, b in copy_ev])),
                    "note": "every step extracts images that crossed PCIe for that step: consecutive frames of the synthetic stream in pinned host "
                            "memory, the left and the right images as two concurrent host->device copies per step (two copy streams), three device input buffers (a buffer is overwritten only after "
                            "every reader of the step that used it has finished — level 0 is read in place)"}

    # ---- pass 5: configs[3] read strictly — every frame is a key-frame, the solve runs on every frame's window ----
    every = None
    if use_ba and not use_solve and not args.no_extra_passes:
        solve_windows[0] = P
        step(); barrier()
        n_e = max(5, args.steps // 2)
        dt_e = timed(n_e)
        solve_windows[0] = 0
        assert int(s_st.abs().sum()) == 0
        every = {"value": world * P * n_e / dt_e, "unit": "stereo frames/s", "ms_per_step": dt_e / n_e * 1e3, "steps": n_e,
                 "note": "as the timed region, plus the OptimizeActiveMap solve stage on EVERY frame's window (configs[3] read as 'every frame is a "
                         "key-frame'; = --workload full_solve)"}

    # ---- pass 5b: the same step on a KITTI-LIKE scene.  The headline stream is deliberately hard on FAST (6 000 rectangles: 30 % of all
    # pixels are FAST-7 corners, 76 % of the pixel pairs survive the compass pre-test); street scenes are sparse (a few % corners, ~14 % of
    # the pairs survive), which is what 300 rectangles give — there FAST takes its two-phase path (pre-test, compaction, scoring of the
    # survivors only).  Reported beside `value`, never instead of it ----
    sparse = None
    if not args.no_extra_passes and world == 1 and args.scene_rects > 1000 and args.pipeline:
        fr2 = synth.stereo_batch(P, stream_id=rank, n_rect=300)
        d_sparse = torch.from_numpy(np.concatenate([fr2[:, 0], fr2[:, 1]], axis=0)).to(dev)
        keep_imgs = cur["imgs"]
        cur["imgs"] = d_sparse
        for _ in range(4):                  # the per-level path statistics of both handles settle on the new scene
            step()
        barrier()
        dt_sp = timed(args.steps)
        kp_sp = float(torch.stack([c.float().mean() for c in d_cnt_b]).mean().item())
        assert all(int(t.abs().sum()) == 0 for t in d_stat_b)
        sparse = {"value": world * P * args.steps / dt_sp, "unit": "stereo frames/s", "ms_per_step": dt_sp / args.steps * 1e3, "scene_rects": 300,
                  "keypoints_per_image": kp_sp,
                  "note": "the timed region on a sparse, street-like scene (300 rectangles instead of 6 000): FAST chooses its two-phase path per level "
                          "from the previous launch's statistics; same kernels, same outputs (tests/test_gpu_fallbacks.py)"}
        cur["imgs"] = keep_imgs
        for _ in range(4):
            step()
        barrier()
        del d_sparse

    # ---- pass 6: the extractor ALONE on an otherwise idle chip (three calls of one handle, event pair around every launch): what a launch
    # of each ORB kernel takes when nothing shares its CUs — under the pipeline a launch is resident for longer BY DESIGN (blocks of the
    # other streams move in beside FAST's), so the per-launch durations of pass 2 price the schedule, these price the kernel ----
    alone = {}
    if not args.no_extra_passes:
        barrier()
        api.prof_reset(); api.prof_enable(True)
        n_alone = 2 * P // S
        for _ in range(3):
            ext.detect_and_compute_batch(cur["imgs"].data_ptr(), n_alone, H, W, W, H * W, d_kps_b[0].data_ptr(), d_desc_b[0].data_ptr(),
                                         d_cnt_b[0].data_ptr(), d_stat_b[0].data_ptr(), cap)
        torch.cuda.synchronize()
        api.prof_enable(False)
        alone = {k: v for k, v in api.prof_read().items() if v[1] > 0}

    # (this pass comes last of the timed ones: the lanes bring their own streams, and on this runtime streams beyond GPU_MAX_HW_QUEUES share
    # hardware queues — created before the streamed-input pass they cost it a third of its rate, 38.5 k instead of 57 k frames/s)
    # the other launch mode over the same steps (eager when the timed region replayed graphs, graphs when it launched eagerly)
    other_mode = None
    lanes_eager = None
    if can_graph and not args.no_extra_passes:
        if not use_graph:
            api.prof_enable(False)
            lanes = [build_lane() for _ in range(n_lanes)]
        fn_o = step_eager if use_graph else step_lanes
        for _ in range(2 * n_lanes):
            fn_o()
        barrier()
        dt_o = timed(args.steps, fn_o)
        other_mode = {"mode": "eager launches, two extractor handles + match + side chain on four streams, consecutive steps overlap" if use_graph else
                              f"HIP graph replay on {n_lanes} single-stream lanes",
                      "value": world * P * args.steps / dt_o, "unit": "stereo frames/s", "ms_per_step": dt_o / args.steps * 1e3,
                      "host_launch_ms_per_step": host_ms[0], "graph_nodes": lanes[0]["graphs"][0].node_count()}
        for _ in range(2 * n_lanes):
            step_lanes_eager()
        barrier()
        dt_l = timed(args.steps, step_lanes_eager)
        lanes_eager = {"mode": f"the same {n_lanes} lanes launched eagerly (no recording)", "value": world * P * args.steps / dt_l,
                       "ms_per_step": dt_l / args.steps * 1e3, "host_launch_ms_per_step": host_ms[0]}
    if args.verify and args.pipeline:
        # the overlapped schedule must not change a single output: one plain step (handles un-gated, joined) against the last pipelined one
        outs = lambda p: [d_kps_b[p], d_desc_b[p], d_cnt_b[p], d_stat_b[p], d_midx, d_mdist, d_xyz, d_ok] + \
            ([d_descr, d_best, d_max, d_dbcnt] if use_lcd else []) + (list(b_out) if use_ba else [])
        torch.cuda.synchronize()
        for p in range(NB):             # both sides start from cleared buffers: slots behind an image's count keep whatever an earlier pass
            for t in outs(p):           # (the streamed one extracts other frames) left there, and whole buffers are compared
                t.zero_()
        step(); torch.cuda.synchronize()
        p_last = (step_no[0] - 1) % NB
        ref = [t.clone() for t in outs(p_last)]
        for e in exts:
            e.set_fast_event(0); e.set_fast_gate(0)
        for t in outs(0):
            t.zero_()
        step_joined()
        torch.cuda.synchronize()
        for i, (a, r) in enumerate(zip(outs(0), ref)):            # every output, the f64 BA blocks included, is bit-reproducible
            assert torch.equal(a.view(torch.uint8), r.view(torch.uint8)), f"pipeline output {i} differs from the joined step"
        for i, e in enumerate(exts):
            e.set_fast_event(ev_fast[i].cuda_event)
            if args.pipeline == 1:
                e.set_fast_gate(ev_fast[(i - 1) % S].cuda_event)
    if args.verify and lanes:           # a recorded step on a lane against the joined eager step: every output bit for bit
        torch.cuda.synchronize()
        step_joined(); torch.cuda.synchronize()
        refs = {"kps": d_kps, "desc": d_desc, "cnt": d_cnt, "midx": d_midx, "mdist": d_mdist, "xyz": d_xyz, "ok": d_ok}
        if use_lcd:
            refs.update({"descr": d_descr, "best": d_best, "max": d_max, "dbcnt": d_dbcnt})
        n0 = [int(v) for v in d_cnt.cpu()]
        for ln in lanes[:2]:
            for rep in range(2):
                ln["graphs"][rep].launch(ln["stream"].cuda_stream); ln["stream"].synchronize()
                lo = ln["o"]
                assert torch.equal(lo["cnt"], d_cnt) and int(lo["stat"].abs().sum()) == 0
                kl = lo["kps"].view(2 * P, cap * 28); kr = d_kps.view(2 * P, cap * 28); dl = lo["desc"].view(2 * P, cap * 32); dr = d_desc.view(2 * P, cap * 32)
                for b_ in range(2 * P):             # slots behind an image's count are never written
                    assert torch.equal(kl[b_, :28 * n0[b_]], kr[b_, :28 * n0[b_]]) and torch.equal(dl[b_, :32 * n0[b_]], dr[b_, :32 * n0[b_]]), f"lane key-points / descriptors, image {b_}"
                for b_ in range(P):
                    sl = slice(b_ * cap, b_ * cap + n0[b_])
                    assert torch.equal(lo["midx"][sl], d_midx[sl]) and torch.equal(lo["mdist"][sl], d_mdist[sl]) and torch.equal(lo["ok"][sl], d_ok[sl])
                    assert torch.equal(lo["xyz"].view(-1, 3)[sl].view(torch.uint8), d_xyz.view(-1, 3)[sl].view(torch.uint8))
                for k_ in ("descr", "best", "max", "dbcnt"):
                    if k_ in lo:
                        assert torch.equal(lo[k_].view(torch.uint8), refs[k_].view(torch.uint8)), f"lane output {k_}"
                if use_ba:
                    for a_, r_ in zip(lo["ba"], b_out):
                        assert torch.equal(a_.view(torch.uint8), r_.view(torch.uint8)), "lane BA blocks"

    # ---- parity sample: K frames of one more step of the timed workload against the oracle (untimed; beside cpu_baseline) ----
    parity = None
    if rank == 0 and args.parity_frames > 0 and world == 1:
        barrier()
        step_timed(); barrier()                         # the launch mode of the timed region (a graph replay on a lane for small batches)
        host = lambda t: t.cpu().numpy()
        if use_graph:
            lo = lanes[(step_no[0] - 1) % len(lanes)]["o"]
            bufs = {k: (host(v) if k != "ba" else [host(t) for t in v]) for k, v in lo.items() if k != "stat"}
        else:
            p_last = (step_no[0] - 1) % NB
            bufs = {"kps": host(d_kps_b[p_last]), "desc": host(d_desc_b[p_last]), "cnt": host(d_cnt_b[p_last]), "midx": host(d_midx), "mdist": host(d_mdist),
                    "xyz": host(d_xyz), "ok": host(d_ok)}
            if use_lcd:
                bufs.update({"descr": host(d_descr), "best": host(d_best), "max": host(d_max), "dbcnt": host(d_dbcnt)})
            if use_ba:
                bufs["ba"] = [host(t) for t in b_out]
        kf = min(args.parity_frames, P)
        sample = sorted({int(round(j * (P - 1) / max(1, kf - 1))) for j in range(kf)})
        parity = parity_sample(api, synth, frames, sample, cap, Kt, K, bufs, db_np, ids if use_lcd else None, int(cur_ids[0]) if use_lcd else 0,
                               ba_w, synth.calc_weights())

    solve_ms = None; solve_roof = None
    if use_ba and not use_solve and not args.no_extra_passes:        # the solve of ALL P windows, timed on its own (not part of `value`)
        with torch.cuda.stream(side_stream):
            solve(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                solve()
            torch.cuda.synchronize()
        solve_ms = (time.perf_counter() - t1) / 5 * 1e3
        assert int(s_st.abs().sum()) == 0
        # k_ba_optimize against the f64 peaks: a flop MODEL of what one Levenberg iteration of a window executes (not a counter): edge evaluation +
        # Jacobians + block products ~410 flop per edge, Schur complement sum_l W_l Hll^-1 W_l^T = (108 k + 216 k^2) flop
        # for a landmark seen by k key-frames, 6x6-blocked Cholesky n^3 / 3 and two triangular solves 2 n^2 with n = 6 P; rounds x 10 iterations
        # (optimize(10), backend.cpp:212-214: every round runs its iteration budget unless a Levenberg trial fails ten times)
        # *rounds = the reference's `iteration` counter = rounds that FAILED the inlier test (backend.cpp:212-232): a window runs that many + 1
        # rounds of optimize(10), at most max_rounds = 5 (round 5 fix: the model multiplied by the counter itself, 0 for well-posed windows)
        rounds_mean = float((s_rd.float() + 1.0).clamp(max=5.0).mean().item())
        szs = ba_w[6]                                                    # [P, 3] = poses, landmarks, edges per window
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

    stream_mode = None
    if rank == 0 and world == 1 and args.stream_mode and not args.no_extra_passes and args.workload in ("full", "orb_match_lcd"):
        barrier()
        pts = []
        for spec in args.stream_mode.split(","):
            pp, ll = [int(v) for v in spec.lower().split("x")]
            cmd = [sys.executable, os.path.abspath(__file__), "--pairs", str(pp), "--lanes", str(ll), "--graph", "1", "--steps", str(max(400, 1600 // pp)), "--warmup", "2",
                   "--workload", args.workload, "--no-extra-passes", "--no-cpu-baseline", "--parity-frames", str(min(2, pp)), "--frame-latency", "--stream-mode", "",
                   "--scene-rects", str(args.scene_rects)]
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, GPU_MAX_HW_QUEUES="24"), timeout=600)
            try:
                cd = json.loads(r.stdout.strip().splitlines()[-1])
                pts.append({"pairs_per_step": pp, "lanes": ll, "value": cd["value"], "ms_per_step": cd["ms_per_step"], "frame_latency_ms": cd["frame_latency"],
                            "host_launch_ms_per_step": cd["host_launch_ms_per_step"], "graph_nodes": cd["graph_nodes"], "parity_ok": (cd["parity_sample"] or {}).get("ok")})
            except Exception as e:               # a failed point is reported, not hidden
                pts.append({"pairs_per_step": pp, "lanes": ll, "error": f"{type(e).__name__}: {e}", "rc": r.returncode, "stderr_tail": r.stderr[-300:]})
        if pts:
            head = dict(pts[0])
            stream_mode = dict(head, unit="stereo frames/s", sweep=pts,
                               note="recorded steps (HIP graph replay) on L lanes, step k on lane k mod L; every lane has its own extractor / DeepLCD handles and a QUERY CONTEXT "
                                    "of the ONE shared loop database (myslam_lcddb_query_ctx); child processes of this run, GPU_MAX_HW_QUEUES=24.  The GPU runs ~4.4 in-order "
                                    "chains side by side whatever the queue count (profiles/r05_queue_concurrency.json), so frames/s ~ 4.4 x pairs_per_step / chain latency: lanes "
                                    "beyond ~16 add nothing, batching frames of several cameras into one step does")
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * P * args.steps / dt
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
            launches_per_step = dom_n / args.steps
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
            nprod = lcd.conv2_products()          # 3 = f16 x 3 (the default model), 6 = the bf16 x 6 kernel a model outside f16's range falls back to
            mf = {"bound": "mfma", "kernel": SYMBOL["calc_conv2"] if nprod == 3 else "k_conv2_bf16x6", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s (bf16 dense)",
                  "partial_products": nprod,
                  "achieved": nprod * f32eq, "frac": nprod * f32eq / MFMA_BF16_PEAK_TFLOPS, "peak_measured": (peaks or {}).get("mfma_bf16_tflops"),
                  "peaks_source": peaks_path, "f32_equivalent_tflops": f32eq, "avg_launch_ms": c2,
                  "note": "CALC conv2 as an implicit GEMM on the 16-bit matrix cores with f32 accuracy (every f32 operand split exactly into two f16 "
                          "pieces, 3 partial products per useful f32 multiply-add; f16 and bf16 run at the same dense rate): `achieved` counts the "
                          "flops the matrix pipe executes, priced against the 16-bit dense peak; f32_equivalent_tflops counts the USEFUL f32 flops "
                          "(the f32-input MFMA peak would be 157.3)"}
        n_internal = S if args.orb_internal_stream else 0        # every extractor handle runs its Gaussian pyramid on an internal stream
        out = {
            "metric": "stereo frames/sec (ORB+match+LCD+BA-build) @1241x376",
            "value": value, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (ORB, Hamming), f32 via f16x3 / bf16x6 split products on the matrix cores with f32 accumulate (CALC conv1 / conv2, DB scan), "
                     "f64 (triangulation, BA)", "data": "synthetic",
            "config": {"workload": {"full": "configs[3]: ORB extract L+R (2000 feats) + L/R Hamming match + triangulation + DeepLCD descriptor + "
                                            f"{n_db_local * world}-KF cosine DB scan + local-BA (10 KF x 300 MP, one distinct window per frame) block build",
                                    "full_solve": "configs[3] incl. solve on EVERY frame: as 'full' + the Backend::OptimizeActiveMap solve stage (rounds of "
                                                  "Levenberg-Marquardt optimize(10) with Schur + Cholesky, outlier flags) of the 10 KF x 300 MP window",
                                    "orb_match": "configs[1]: ORB extract L+R (2000 feats) + L/R Hamming match + triangulation",
                                    "orb_match_lcd": f"configs[2]: configs[1] + DeepLCD descriptor + {n_db_local * world}-KF cosine DB scan"}[args.workload],
                       "pairs_per_step_per_gpu": P, "image": "1241x376 u8", "scene_rects": args.scene_rects, "keypoints_per_image": n_kp,
                       "hip_streams": {"caller": len({stream, stream2} | {s.cuda_stream for s in orb_streams}) + (1 if args.pipeline else 0),
                                       "extractor_internal": n_internal},
                       "hip_hw_queues": int(HW_QUEUES), "orb_extractor_handles": S, "pipelined_steps": bool(args.pipeline),
                       "input_level0": "read in place (resident input images; the last image of each extractor call is copied)" if not args.orb_copy_input
                                       else "copied into the pyramid block",
                       "parallelism": f"frame-sharded x{world}" + (", id-range sharded DB + all-gather of 16-byte candidate records" if world > 1 else "")},
            "rccl_ranks": rccl_ranks if not via_cpu else None, "collective_backend": (args.backend if world > 1 else None),
            "collective_ranks": rccl_ranks,
            "per_rank_ms_per_step": per_rank_ms,
            "collective_ms_per_step": None if collective is None else collective["collective_ms_per_step"],
            "shard_scan_ms_per_step": None if collective is None else collective["shard_scan_ms_per_step"],
            "db_exchange": collective,
            "roofline": roof, "roofline_valu": roof_valu, "roofline_mfma": mf,
            "profiled_pass": None if dt_prof is None else {"ms_per_step": dt_prof / args.steps * 1e3,
                                                           "kernel_ms_per_step": {SYMBOL.get(k, k): v[0] / args.steps for k, v in busy.items()}},
            "extractor_alone": None if not alone else {"images_per_call": 2 * P // S, "calls": 3,
                                                       "kernel_ms_per_call": {SYMBOL.get(k, k): v[0] / 3 for k, v in alone.items()},
                                                       "note": "pass 6: one extractor handle on an otherwise idle chip; event-timed launches"},
            "streamed": None if streamed is None else dict(streamed, ratio_to_resident=streamed["value"] / value,
                                                           pcie_h2d_measured_GBps=(peaks or {}).get("h2d_GBps"),
                                                           pcie_bound_frames_per_s=None if not (peaks or {}).get("h2d_GBps") else
                                                           world * (peaks["h2d_GBps"] * 1e9) / (2 * H * W)),
            "full_solve_cadence6": cadence, "full_solve_every_frame": every, "kitti_like_scene": sparse,
            "launch_mode": f"HIP graph replay on {n_lanes} single-stream lanes (step k on lane k mod {n_lanes})" if use_graph else
                           "eager launches (two extractor handles + match + side chain on four streams, consecutive steps overlap)",
            "host_launch_ms_per_step": host_launch_ms,      # CPU time of enqueuing one step of the timed region
            "graph_nodes": lanes[0]["graphs"][0].node_count() if (use_graph and lanes) else None,
            ("step_eager" if use_graph else "step_graph"): other_mode, "step_lanes_eager": lanes_eager,
            "ba_solve_all_windows_ms": solve_ms,     # OptimizeActiveMap solve stage for all P windows, outside the timed region
            "roofline_ba_optimize": solve_roof,
            "parity_sample": parity,
            "stream_mode": stream_mode, "frame_latency": frame_latency,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(synth, args.workload, args.cpu_pairs, db_np if db_np is not None else synth.lcd_database(16), frames,
                                               ba_w if ba_w is not None else synth.ba_windows(2)[0])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        if parity is not None and not parity["ok"]:
            print("parity_sample FAILED: " + "; ".join(parity["mismatches"][:8]), file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
