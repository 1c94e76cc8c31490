#!/bin/bash
# HBM traffic of the GEMM kernels (GPU box): two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters with
# --kernel-trace only) over tools/bench_gemm.py [tn].  Summaries -> gpurun_out/pmc_traffic_<mode>.txt
#   bash tools/pmc_traffic.sh nt      forward conv / 1x1+GLU launches of the E8 encoder (gemm_nt_kernel)
#   bash tools/pmc_traffic.sh tn      the 16 weight-gradient shapes (gemm_tn8_kernel / gemm_tn_kernel + tn_reduce_kernel)
MODE=${1:-nt}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARG=""; [ "$MODE" = "tn" ] && ARG="tn"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 tools/bench_gemm.py $ARG) > $OUT/pmc_traffic_${MODE}_$c.log 2>&1
done
python3 - $MODE <<'PY' > $OUT/pmc_traffic_$MODE.txt
import csv, glob, sys, collections
mode = sys.argv[1]
# launches in dispatch order per counter pass; bench_gemm.py issues 12 launches of cum_gemm_* per shape (2 warm-up + 10)
seq = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt" in k or "gemm_tn" in k or "tn_reduce" in k:
                rows.append((int(r["Dispatch_Id"]), k, float(r["Counter_Value"])))
    rows.sort()
    seq[c] = rows
B, Ts, Cs = 16, [160254, 80126, 40062, 20030, 10014, 5006, 2502, 1250, 624], [1, 64, 128, 256, 512, 768, 768, 768, 768]
rup = lambda x, m: (x + m - 1) // m * m
shapes = []
for i in range(8):
    M, Cin, H = B * (Ts[i + 1] + 2), rup(Cs[i], 8), Cs[i + 1]
    if mode == "tn":
        # operands once (X rows overlap: unique bytes = M * ldx) + f32 result
        shapes.append((f"enc{i}.conv.w", 2 * M * (H + 2 * Cin) + 4 * H * 4 * Cin, 2.0 * M * H * 4 * Cin))
        shapes.append((f"enc{i}.1x1.w", 2 * M * (2 * H + H) + 4 * 2 * H * H, 2.0 * M * 2 * H * H))
    else:
        shapes.append((f"enc{i}.conv+relu", 2 * M * (2 * Cin + H) + 2 * H * 4 * Cin, 2.0 * M * H * 4 * Cin))
        shapes.append((f"enc{i}.1x1+glu", 2 * M * (H + H) + 2 * 2 * H * H, 2.0 * M * 2 * H * H))
main = ("gemm_tn_kernel", "gemm_tn8_kernel", "gemm_tn9_kernel") if mode == "tn" else ("gemm_nt",)
def per_shape(c):
    out, cur, cnt = [], 0.0, 0
    for _, k, v in seq[c]:
        if any(m in k for m in main):
            if cnt == 12:
                out.append(cur / 12); cur, cnt = 0.0, 0
            cnt += 1
        cur += v           # tn: the reduce launches that follow a gemm_tn launch belong to it
    out.append(cur / max(cnt, 1))
    return out
fe, wr = per_shape("FETCH_SIZE"), per_shape("WRITE_SIZE")
print("shape | algorithmic MB | FETCH_SIZE MB (x2 corrected) | WRITE_SIZE MB | traffic / algorithmic")
tf = tw = ta = 0.0
for (name, alg, fl), f, w in zip(shapes, fe, wr):
    fm, wm, am = 2 * f / 1e3, w / 1e3, alg / 1e6
    tf, tw, ta = tf + fm, tw + wm, ta + am
    print(f"{name} | {am:.1f} | {fm:.1f} | {wm:.1f} | {(fm + wm) / am:.2f}")
print(f"mean over the {len(shapes)} shapes | {ta / len(shapes):.1f} | {tf / len(shapes):.1f} | {tw / len(shapes):.1f} | {(tf + tw) / ta:.2f}")
PY
cat $OUT/pmc_traffic_$MODE.txt
