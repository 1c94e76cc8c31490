#!/bin/bash
# per-kernel times of tools/bench_scan.py (rocprofv3 --kernel-trace --stats) -> gpurun_out/prof_scan_stats.csv
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
cp $GRAFT_REPO_ROOT/tools/bench_scan.py /tmp/bench_scan.py
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_scan -- python3 /tmp/bench_scan.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_scan.log 2>&1
f=$(find /tmp/prof_scan -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_scan_stats.csv
head -25 "$f" | cut -c1-200
