export TMPDIR=/tmp; cd /tmp
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --pairs 64 --workload orb_match --streams 1 --orb-internal-stream 0 --no-cpu-baseline --no-extra-passes > /dev/null 2>/tmp/pm.err || tail -3 /tmp/pm.err
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f: print("no output for $set")
else:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('myslam_hip::','')
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k in ('k_describe2','k_fast_strip<40, 4>','k_octree<256>','k_blur7_strip','k_resize_strip'):
        print(k, {c: round(v/4/128) for c,v in agg[k].items()}, "(per image)")
PY
done
