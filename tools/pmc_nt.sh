#!/bin/bash
# PMC counters of the gemm_nt kernels inside a short train-step run (GPU box).  One rocprofv3 --pmc pass per set.
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_nt_$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > $OUT/pmc_nt_$i.log 2>&1
  python3 - /tmp/pmc_nt_$i <<'PY' > $OUT/pmc_nt_$i.txt
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt" not in k and "gemm_tn" not in k: continue
        m = re.search(r"Li(\d)ELi(\d+)ELi(\d+)E", k)
        k = "nt epi%s %sx%s" % m.groups() if m else ("tn" if "gemm_tn" in k else "nt other")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:34s} {v / cnt[(k, c)]:16.0f}  (mean of {cnt[(k, c)]} launches)")
PY
done
tail -3 $OUT/pmc_nt_3.log
