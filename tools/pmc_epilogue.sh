#!/bin/bash
# Counter passes over tools/one_gemm.py: the same deep-layer GEMM with the plain and with the GLU-backward epilogue
# (separate rocprofv3 --pmc passes, --kernel-trace only).  Output: gpurun_out/pmc_epilogue.txt (per kernel, mean per launch).
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
 "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES"
 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ"
 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST"
 "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_EA0_RDREQ TCC_EA0_WRREQ_64B"
 "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL"
)
rm -f $OUT/pmc_epilogue.txt
i=0
for s in "${SETS[@]}"; do
  rm -rf /tmp/pe_$i
  (cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $s --kernel-trace --output-format csv -d /tmp/pe_$i -- python3 tools/one_gemm.py 3) > $OUT/pmc_epilogue_$i.log 2>&1
  python3 - /tmp/pe_$i <<'PY' >> $OUT/pmc_epilogue.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt" in k:
            tag = "GLU_BWD" if "Li4E" in k else "BIAS"
            acc[r["Counter_Name"]][tag].append(float(r["Counter_Value"]))
for c, d in sorted(acc.items()):
    print(c, " ".join("%s=%.4g" % (t, sum(v) / len(v)) for t, v in sorted(d.items())))
PY
  i=$((i+1))
done
cat $OUT/pmc_epilogue.txt
