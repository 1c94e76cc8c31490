#!/bin/bash
# HBM traffic of the one-launch streaming hop (GPU box): two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# MI355X_MICROARCH.md: FETCH_SIZE doubled for wide streaming reads) over tools/bench_streaming.py 256 5 kernel.
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmch_$c
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmch_$c -- python3 tools/bench_streaming.py 256 5 kernel) > /tmp/pmch_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"/tmp/pmch_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stream_hop" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    out[c] = v
n = len(out["FETCH_SIZE"])
f = 2 * sum(out["FETCH_SIZE"]) / max(n, 1) / 1e3
w = sum(out["WRITE_SIZE"]) / max(len(out["WRITE_SIZE"]), 1) / 1e3
print(f"stream_hop_kernel: {n} launches (16 hops of 256 streams each, the last one shorter); per launch FETCH_SIZE x 2 = {f:.1f} MB, WRITE_SIZE = {w:.1f} MB")
print(f"per hop of 256 streams: {f / 16:.2f} MB read, {w / 16:.2f} MB written")
PY
