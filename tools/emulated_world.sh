#!/bin/bash
# Per-rank compute of an N-GPU job measured on ONE GPU (bench.py --emulate-world N, no collective): N = 1 (the plain bench step), 2, 4, 8, same box, back to back.
#   tools/emulated_world.sh <tag>  -> gpurun_out/r06_emulated_world_<tag>.json
TAG=${1:-x}
for n in 1 2 4 8; do
  timeout 600 python bench.py --no-cpu-baseline --parity-frames 8 --stream-input 0 --stream-mode "" --steps 100 --emulate-world $n > gpurun_out/emu_${TAG}_$n.json 2> gpurun_out/emu_${TAG}_$n.err; echo "N=$n rc=$?"
done
python - <<PY
import json
out = {"build": "$TAG", "points": []}
for n in (1, 2, 4, 8):
    try:
        d = json.load(open(f"gpurun_out/emu_${TAG}_{n}.json"))
    except Exception as e:
        print(n, "failed", e); continue
    e = d.get("emulated_world") or {}
    km = (d.get("profiled_pass") or {}).get("kernel_ms_per_step") or {}
    out["points"].append({"world": n, "ms_per_step": d["ms_per_step"], "repeats_ms_per_step": d.get("repeats_ms_per_step"), "per_rank_frames_per_s": d["value"],
                          "shard_rows": e.get("shard_rows", 10000), "queries_scanned_per_step": e.get("queries_scanned_per_step", 512),
                          "shard_scan_ms_per_step_in_pipeline": e.get("shard_scan_ms_per_step", km.get("k_db_scan_bf16x6")),
                          "shard_scan_alone_ms": e.get("shard_scan_alone_ms"), "merge_alone_ms": e.get("merge_alone_ms"),
                          "parity_ok": (d.get("parity_sample") or {}).get("ok"), "cadence6_ms_per_step": (d.get("full_solve_cadence6") or {}).get("ms_per_step"),
                          "clock_mhz": d.get("clock_mhz")})
    print(out["points"][-1])
out["note"] = "bench.py --emulate-world N on one MI355X, same box, back to back; N = 1 is the plain 10 000-row step.  No collective ran: a real N-GPU job adds two all-gathers per step"
json.dump(out, open("gpurun_out/r06_emulated_world_$TAG.json", "w"), indent=1)
PY
