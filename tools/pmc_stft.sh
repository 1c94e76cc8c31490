#!/bin/bash
# HBM traffic of the STFT loss, fused kernels against the rocFFT route (GPU box): per route two SEPARATE rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE, with --kernel-trace only) over tools/stft_only.py.  Summary -> gpurun_out/pmc_stft.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
for fused in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcs_${fused}_$c
    (cd $GRAFT_REPO_ROOT && CUM_STFT_FUSED=$fused rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_${fused}_$c -- python3 tools/stft_only.py 5) > $OUT/pmc_stft_${fused}_$c.log 2>&1
  done
done
python3 - <<'PY' > $OUT/pmc_stft.txt
import csv, glob, collections
print("route | kernel | launches per step | FETCH_SIZE MB per step (x2 corrected, MI355X_MICROARCH.md) | WRITE_SIZE MB per step")
for fused in (1, 0):
    acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    for c in acc:
        for f in glob.glob(f"/tmp/pmcs_{fused}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "stft" in k or "fft" in k.lower():
                    k = k.split("(")[0].replace("void ", "")[:70]
                    acc[c][k].append(float(r["Counter_Value"]))
    tot_f = tot_w = 0.0
    for k in sorted(acc["FETCH_SIZE"]):
        n = len(acc["FETCH_SIZE"][k]) / 5.0
        f = 2 * sum(acc["FETCH_SIZE"][k]) / 5.0 / 1e3
        w = sum(acc["WRITE_SIZE"].get(k, [0.0])) / 5.0 / 1e3
        tot_f += f; tot_w += w
        print(f"{'fused' if fused else 'rocFFT'} | {k} | {n:.0f} | {f:.1f} | {w:.1f}")
    print(f"{'fused' if fused else 'rocFFT'} | TOTAL | | {tot_f:.1f} | {tot_w:.1f}")
PY
cat $OUT/pmc_stft.txt
