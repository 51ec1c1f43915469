#!/bin/bash
# counters of the CALC kernels (conv2 in particular): effective clock, matrix-pipe busy, LDS conflicts, wave wait breakdown
cd "$(dirname "$0")/.."; R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
run() { tag=$1; shift; ( cd /tmp && rm -rf /tmp/c2pmc_$tag && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/c2pmc_$tag -o a -- python $R/tools/conv2_time.py > /dev/null 2>&1 )
  f=$(find /tmp/c2pmc_$tag -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("myslam_hip::", "")
    if "conv2" not in k: continue
    d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in d.items():
    print(k, {a: round(v / n[(k, a)]) for a, v in c.items()})
PY
}
run a GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM
