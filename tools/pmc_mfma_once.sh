export TMPDIR=/tmp; cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL"; do
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --pairs 64 --workload orb_match_lcd --streams 1 --orb-internal-stream 0 --no-cpu-baseline --no-extra-passes > /dev/null 2>/tmp/pm.err || tail -3 /tmp/pm.err
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f: print("no output for $set")
else:
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('myslam_hip::','')
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k in ('k_conv2_f16x3','k_conv1_f16x3_pool_lrn','k_hamming_fp4'):
        print(k, {c: round(v/4) for c,v in agg[k].items()}, "(per launch of 64 frames)")
PY
done
