# Instruction / scalar-data cache and wave counters of the one-launch streaming hop (GPU box): rocprofv3 --pmc passes over tools/bench_streaming.py
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_h2_$i -- python3 tools/bench_streaming.py 256 5 kernel) > /tmp/pmc_h2_$i.log 2>&1
  python3 - /tmp/pmc_h2_$i <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "stream_hop" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / cnt[(k, c)]:16.0f}  (mean of {cnt[(k, c)]} launches)")
PY
  tail -2 /tmp/pmc_h2_$i.log | grep -i "error\|invalid" 
done
