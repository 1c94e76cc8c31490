#!/bin/bash
# Kernel composition of a streaming hop (GPU box): rocprofv3 kernel stats of tools/bench_streaming.py 256 5 <variant>
# (variant: kernel = the one-launch hop (default path), fused = the per-layer hop, cached, bf16)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/strm -o s -- python3 tools/bench_streaming.py 256 5 ${1:-fused} > gpurun_out/stream_prof.json 2> gpurun_out/stream_prof.err
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/strm/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
hops = 5 * 16000 // 256
tot = 0
print("per hop (%d hops):" % hops)
for r in rows[:28]:
    ms = float(r["TotalDurationNs"]) / 1e3 / hops
    tot += ms
    print(f'{r["Name"][:80]:80s} {int(r["Calls"]) / hops:6.1f}/hop {ms:7.1f} us/hop  {float(r["AverageNs"]) / 1e3:6.1f} us')
print("sum of all kernels per hop: %.1f us, launches per hop: %.2f" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / hops, sum(int(r["Calls"]) for r in rows) / hops))
vendor = [r["Name"] for r in rows if r["Name"].startswith("Cijk_")]
aten = sum(int(r["Calls"]) for r in rows if "at::native" in r["Name"])
print("vendor GEMM kernels in the trace: %d names; at::native launches: %d in %d hops (first frame of the streams, state import, flush drain)" % (len(vendor), aten, hops))
PY
cat gpurun_out/stream_prof.json
