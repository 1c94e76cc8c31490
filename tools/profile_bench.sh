#!/bin/bash
# Kernel trace of the default bench.py run + a short run for per-step sums (GPU box).  Outputs under gpurun_out/prof_round/.
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
rm -rf $OUT && mkdir -p $OUT
# (1) short run, no roofline / cpu legs: kernel-time sum per step against the wall-clock step time
rocprofv3 --kernel-trace --stats -d /tmp/short -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $OUT/short.json 2> $OUT/short.err
python3 $GRAFT_REPO_ROOT/tools/ab_kernel_sums.py /tmp/short/step_results.db short > $OUT/short_sums.txt 2>&1
python3 - /tmp/short/step_results.db <<'PY' > $OUT/short_gaps.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# the last 10 steps: split by the Adam kernel (csrc/optim.hip: one optim_adam_kernel launch per step)
idx = [i for i, r in enumerate(rows) if "optim_adam_kernel" in r[2]]
print("kernels", len(rows), "adam launches", len(idx))
if len(idx) >= 12:
    ends = idx
    a, b = ends[-11], ends[-1]
    seg = rows[a + 1:b + 1]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1))
    print("10 steps: wall %.2f ms, kernel busy %.2f ms, idle %.2f ms, launches %d" % (wall, busy, wall - busy, len(seg)))
    print("largest gaps (us):")
    for g in gaps[-15:]: print("  %.1f  %s -> %s" % g)
    import collections
    hist = collections.Counter(min(int(g[0] // 2) * 2, 40) for g in gaps)
    print("gap histogram (us bucket: count):", sorted(hist.items()))
    import re
    per = collections.defaultdict(lambda: [0, 0.0])
    for s, e, k in seg:
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"\(.*$", "", k)[:90]
        per[k][0] += 1; per[k][1] += (e - s) / 1e6
    print("per step, by kernel (ms, launches):")
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
        print("  %7.3f  %5.1f  %s" % (t / 10, n / 10, k))
PY
# (2) the default run, csv stats for profiles/
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/full -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/full.err
cp /tmp/full/bench_kernel_stats.csv $OUT/ 2>/dev/null || find /tmp/full -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
# (3) agreement of roofline.launch_ms with the trace: the roofline loop is the last thing bench.py runs at N = 1
#     before the cpu baseline (16 shapes x (3 warm-up + 10 timed) cum_gemm_tn calls; roofline.launch_ms covers the last 10)
python3 - /tmp/full $OUT/bench_under_rocprof.json <<'PY' > $OUT/tn_agreement.txt
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
is_tn = lambda k: "gemm_tn" in k and "reduce" not in k        # gemm_tn_kernel / gemm_tn8_kernel / gemm_tn9_kernel
tn = [i for i, r in enumerate(rows) if is_tn(r[2])]
n = 10 * 13          # the MFMA-bound group the roofline object is quoted on: the last 10 of the 16 shapes (enc3-enc7)
first = tn[-n]
seg = rows[first:]
g = [e - s for s, e, k in seg if is_tn(k)]
red = [e - s for s, e, k in seg if "tn_reduce_kernel" in k]
d = json.load(open(sys.argv[2]))
print("last %d gemm_tn launches: mean %.1f us; tn_reduce launches in that span: %d, %.1f us per gemm_tn call" %
      (len(g), sum(g) / len(g) / 1e3, len(red), sum(red) / len(g) / 1e3))
print("kernel + reduce per call: %.1f us; roofline.launch_ms of the same run: %.1f us" %
      ((sum(g) + sum(red)) / len(g) / 1e3, 1e3 * d["roofline"]["launch_ms"]))
PY
