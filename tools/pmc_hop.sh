#!/bin/bash
# SQ counters of the one-launch streaming hop (GPU box): rocprofv3 --pmc passes over tools/bench_streaming.py 256 5 kernel
# (16 hops per launch, 256 workgroups of 8 waves); means per launch to gpurun_out/pmc_hop_<pass>.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA" \
           "SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_hop_$i -- python3 tools/bench_streaming.py 256 5 kernel) > $OUT/pmc_hop_$i.log 2>&1
  python3 - /tmp/pmc_hop_$i <<'PY' > $OUT/pmc_hop_$i.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "stream_hop" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / cnt[(k, c)]:16.0f}  (mean of {cnt[(k, c)]} launches)")
PY
  cat $OUT/pmc_hop_$i.txt; tail -3 $OUT/pmc_hop_$i.log
done
