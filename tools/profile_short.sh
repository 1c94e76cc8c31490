#!/bin/bash
# Per-step kernel sums of a short bench.py run (part 1 of profile_bench.sh alone).  Output: gpurun_out/prof_short/.
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_short
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d /tmp/short -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $OUT/short.json 2> $OUT/short.err
python3 - /tmp/short/step_results.db <<'PY' > $OUT/short_gaps.txt
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "optim_adam_kernel" in r[2]]
a, b = idx[-11], idx[-1]
seg = rows[a + 1:b + 1]
wall = (seg[-1][1] - seg[0][0]) / 1e6
busy = sum(e - s for s, e, _ in seg) / 1e6
print("10 steps: wall %.2f ms, kernel busy %.2f ms, idle %.2f ms, launches %d" % (wall, busy, wall - busy, len(seg)))
per = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in seg:
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)[:90]
    per[k][0] += 1; per[k][1] += (e - s) / 1e6
print("per step, by kernel (ms, launches):")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
    print("  %7.3f  %5.1f  %s" % (t / 10, n / 10, k))
PY
