#!/bin/bash
# Same-box A/B of two builds of the library on the train step: per-step kernel sums from rocprofv3 traces.
#   tools/ab_step.sh <libA.so> <libB.so>   (paths relative to the repo root; run on the GPU box)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$1" "$2"; do
    export CUM_LIB=$GRAFT_REPO_ROOT/$v
    rm -rf /tmp/ab
    rocprofv3 --kernel-trace --stats -d /tmp/ab -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
    python3 $GRAFT_REPO_ROOT/tools/ab_kernel_sums.py /tmp/ab/step_results.db $(basename $v)
  done
done
