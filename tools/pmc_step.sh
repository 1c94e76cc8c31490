#!/bin/bash
# HBM traffic of EVERY kernel of the E8 B = 16 f16 train step (GPU box): two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; counters with --kernel-trace only, MI355X_MICROARCH.md) over three eager steps of bench.py, then the
# time-parallel scan kernels (tools/bench_scan_tp.py).  Summary -> gpurun_out/${1:-r06}_step_pmc.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c /tmp/pmct_$c
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$c -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-roofline --no-cpu-baseline) > $OUT/pmc_step_$c.log 2>&1
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmct_$c -- python3 tools/bench_scan_tp.py) > $OUT/pmc_scantp_$c.log 2>&1
done
python3 - <<'PY' > $OUT/${1:-r06}_step_pmc.txt
import csv, glob, collections, re
def load(prefix):
    acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    for c in acc:
        for f in glob.glob(f"/tmp/{prefix}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[c][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
def short(k):
    k = re.sub(r"^void ", "", k)
    return k[:96]
for title, prefix, per in (("train step (E8, B = 16, f16 autocast, eager; settle + warm-up + 3 steps: launches are totals)", "pmcs", None),
                           ("time-parallel scan (tools/bench_scan_tp.py)", "pmct", None)):
    acc = load(prefix)
    print("== " + title)
    print("kernel | launches | FETCH_SIZE MB per launch (x2 corrected: MI355X_MICROARCH.md, wide streaming reads) | WRITE_SIZE MB per launch")
    rows = []
    for k in set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"]):
        f, w = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
        n = max(len(f), len(w))
        fm = 2 * sum(f) / max(len(f), 1) / 1e3
        wm = sum(w) / max(len(w), 1) / 1e3
        rows.append((n * (fm + wm), short(k), n, fm, wm))
    for _, k, n, fm, wm in sorted(rows, reverse=True)[:45]:
        print(f"{k} | {n} | {fm:.1f} | {wm:.1f}")
PY
cat $OUT/${1:-r06}_step_pmc.txt | head -70
