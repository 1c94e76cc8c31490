#!/bin/bash
# SQ counters of the 256x256 NT GEMM on a plain 8192^3 problem, gemm_nt8_kernel (CUM_NT9=0) vs gemm_nt9_kernel (CUM_NT9=1)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
cp $GRAFT_REPO_ROOT/tools/gemm_plain.py /tmp/gemm_plain.py
for v in 0 1; do
  export CUM_NT9=$v
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_gp_${v}_$i -- python3 /tmp/gemm_plain.py > $OUT/pmc_gp.log 2>&1)
    python3 - /tmp/pmc_gp_${v}_$i $v <<'PY' >> $OUT/pmc_gemm_plain.txt
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter(); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("CUM_NT9=%s  kernel us (median of %d): %.1f" % (sys.argv[2], len(dur), sorted(dur)[len(dur) // 2] / 1e3 if dur else -1))
for c, v in sorted(acc.items()): print(f"   {c:30s} {v / cnt[c]:16.0f}")
PY
  done
done
cat $OUT/pmc_gemm_plain.txt
