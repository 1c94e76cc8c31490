#!/bin/bash
# Same-box A/B of library builds on the train step (GPU box): per-step kernel sums from rocprofv3 traces, A B A B.
#   tools/ab_libs.sh <libA.so> <libB.so>     (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$1" "$2"; do
    export CUM_LIB=$GRAFT_REPO_ROOT/$v
    rm -rf /tmp/ab
    (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d /tmp/ab -o step -- python3 bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline) > /tmp/ab_out.txt 2>&1
    python3 $GRAFT_REPO_ROOT/tools/ab_kernel_sums.py /tmp/ab/step_results.db $(basename $v)
    grep -o '"ms_per_step": [0-9.]*' /tmp/ab_out.txt
  done
done
