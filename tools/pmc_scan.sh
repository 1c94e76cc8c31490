#!/bin/bash
# SQ counters of the scan kernels (GPU box): two rocprofv3 --pmc passes over tools/bench_scan.py, summaries to gpurun_out/
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
cp $GRAFT_REPO_ROOT/tools/bench_scan.py /tmp/bench_scan.py
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_scan_$i -- python3 /tmp/bench_scan.py "$@" > $OUT/pmc_scan_$i.log 2>&1
  python3 - /tmp/pmc_scan_$i <<'PY' > $OUT/pmc_scan_$i.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "scan" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / cnt[(k, c)]:16.0f}  (mean of {cnt[(k, c)]} launches)")
PY
done
