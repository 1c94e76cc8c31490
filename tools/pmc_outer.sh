#!/bin/bash
# HBM traffic of the fused outer-layer kernels (GPU box): two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE with
# --kernel-trace only) over tools/outer_layers.py.  Summary -> gpurun_out/pmc_outer.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmco_$c
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmco_$c -- python3 tools/outer_layers.py 5) > $OUT/pmc_outer_$c.log 2>&1
done
python3 - <<'PY' > $OUT/pmc_outer.txt
import csv, glob, collections
acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob(f"/tmp/pmco_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for name in ("enc0_fwd_kernel", "enc0_bwd_kernel", "dec7_fwd_kernel", "dec7_bwd_kernel"):
                if name in k:
                    acc[c][name].append(float(r["Counter_Value"]))
M = 16 * 80128
alg = {"enc0_fwd_kernel": 2 * (2 * M) * 8 + 2 * M * 64 * 2,                  # input rows [2M][8] 16-bit + out + gate
       "enc0_bwd_kernel": 2 * M * 128 + 2 * (2 * M) * 8,                      # dZ [M][128] + input rows
       "dec7_fwd_kernel": 2 * M * 64 + 2 * (2 * M) * 8,                       # u + out rows [2M][8]
       "dec7_bwd_kernel": 2 * M * 64 + 2 * (2 * M) * 8 + M * 16 + 2 * 2 * M * 64}   # u + dY rows + mask + dU + dpre
print("kernel | algorithmic MB | FETCH_SIZE MB (x2 corrected, MI355X_MICROARCH.md) | WRITE_SIZE MB | traffic / algorithmic")
for name in alg:
    f = 2 * sum(acc["FETCH_SIZE"][name]) / max(len(acc["FETCH_SIZE"][name]), 1) / 1e3
    w = sum(acc["WRITE_SIZE"][name]) / max(len(acc["WRITE_SIZE"][name]), 1) / 1e3
    a = alg[name] / 1e6
    print(f"{name} | {a:.1f} | {f:.1f} | {w:.1f} | {(f + w) / a:.2f}   ({len(acc['FETCH_SIZE'][name])} launches)")
PY
cat $OUT/pmc_outer.txt
# third pass: SQ counters of the same kernels (where the wave time goes)
rm -rf /tmp/pmco_sq
(cd $GRAFT_REPO_ROOT && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmco_sq -- python3 tools/outer_layers.py 5) > $OUT/pmc_outer_sq.log 2>&1
python3 - <<'PY' > $OUT/pmc_outer_sq.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmco_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for name in ("enc0_fwd_kernel", "enc0_bwd_kernel", "dec7_fwd_kernel", "dec7_bwd_kernel"):
            if name in r["Kernel_Name"]:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(name)
    for c in sorted(m):
        print(f"   {c:24s} {m[c]:16.0f}")
    if m.get("SQ_WAVE_CYCLES"):
        print(f"   -> VALU active / wave cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']:.2f}, any instruction active {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.2f}, "
              f"waiting {m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']:.2f}, waiting on an instruction slot {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.2f}")
PY
cat $OUT/pmc_outer_sq.txt
