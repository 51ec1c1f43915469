"""bench/args.py — the command line of bench.py (the driver uses --gpus / --steps / --warmup only; every other flag is a measurement knob whose default is
the schedule DESIGN.md section 4 describes)."""
import argparse


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200 = 1.4 s of work: the fill and drain of the three-step pipeline are 0.6 %% of a 50-step run)")
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
    ap.add_argument("--no-repeats", action="store_true", help="skip the two repeat regions behind the timed one (`repeats_ms_per_step` then holds one value)")
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
    ap.add_argument("--ba-stream", choices=["side", "match"], default="match", help="the BA block build behind the triangulation on the match stream (default: +0.4 %%, three alternating runs) or behind the DB scan on the side stream")
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
                         "work item, the library's default; -1 = 0.  Up to build v68 the pipelined schedule ran 2: with the kernel's direct-to-LDS form — three times "
                         "the windows in flight per block — the unlimited grid is the faster one, profiles/r06_ab_describe_glds.json)")
    ap.add_argument("--lcd-skip", type=int, default=0, help="diagnostic, timing only: myslam_lcd_set_option(SKIP_KERNELS) bit mask (1 input, 2 conv1, 4 conv2, 8 pool2, 16 conv3)")
    ap.add_argument("--side-skip", default="", help="diagnostic: comma list of side-chain parts to leave out (lcd, db, ba) — measures what each part costs the step")
    ap.add_argument("--blur-mfma", type=int, default=0, choices=[0, 1],
                    help="myslam_orb_set_option(BLUR_MFMA): 1 = the Gaussian pyramid on the int8 matrix cores (k_blur7_mfma), 0 = register-strip kernel")
    ap.add_argument("--stream-mode", default="1x16,1x4,2x16,4x16",
                    help="live-stream operating points, 'PAIRSxLANES,...' ('' = skip): after the other passes each point is run as a child process "
                         "(bench.py --pairs P --lanes L --graph 1: recorded steps on L lanes scanning ONE loop database through L query contexts; its own "
                         "GPU_MAX_HW_QUEUES) and reported as `stream_mode` — the reference's call pattern is one frame per call (src/frontend.cpp:41-77)")
    ap.add_argument("--stream-mode-late", action="store_true", help="A/B: run the --stream-mode children after the other passes, beside this process's idle GPU context (the order up to build v67)")
    ap.add_argument("--gap-before-timed", action="store_true", help="A/B: the order up to build v67 (warm-up steps, then the run's checks / lane set-up / clock probe, then the timed region)")
    ap.add_argument("--frame-latency", action="store_true", help="(child of --stream-mode) also time every step on its lane with an event pair: `frame_latency_ms`")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="N > 1 on ONE GPU, one process, no collective: per step this rank scans N x pairs queries (its own + (N - 1) x pairs resident ones) against a "
                         "6 250-row shard and merges N candidate sets — the per-rank compute of an N-GPU job inside the pipeline (`emulated_world` in the line; "
                         "the two all-gathers are what a real job adds)")
    ap.add_argument("--block-trace", default="", help="profiling builds only (a library built with -DMYSLAM_BLOCK_TRACE, tools/build_variants.sh): after the timed region run 4 more "
                    "steps with the block trace on and write the records (npy, 2 x u64 per block that ran on XCD 0) to this file; tools/block_trace_report.py reads it")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = debugging aid: several ranks share GPU 0 and the collectives go through host memory")
    args = ap.parse_args()
    if args.pipeline < 0:
        args.pipeline = 1 if (args.streams == 2 and args.orb_split != 1) else 0
    if args.side_blocks_per_cu < 0:
        args.side_blocks_per_cu = 0
    if args.orb_split == 0:
        args.orb_split = 2 if (args.streams == 2 or args.pipeline) else 1
    if args.pipeline:
        assert args.orb_split >= 2, "--pipeline runs the images on two or more extractor handles (--orb-split >= 2)"
    return args
