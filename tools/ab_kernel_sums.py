"""Per-step kernel-time sums from a rocprofv3 rocpd database (on the GPU box): the same-box A/B harness.

  for v in A B A B; do cp tools/_ab/lib_$v.so cleanumamba_amd/libcleanumamba_hip.so
    rocprofv3 --kernel-trace --stats -d /tmp/ab -o step -- python3 bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline
    python3 tools/ab_kernel_sums.py /tmp/ab/step_results.db $v; done

Boxes (and DVFS states) differ by several per cent, so variants are only compared inside one gpurun call, with the
scan backward kernel as the reference that the change did not touch."""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()
by = {}
for n, c, s in rows:
    if "gemm_nt" not in n:
        continue
    m = re.search(r"Li(\d)ELi", n)
    e = m.group(1) if m else "0/1"
    by[e] = by.get(e, 0) + s / 1e6 / 13
sb = [s for n, c, s in rows if "scan_bwd_kernel" in n][0] / 1e6 / 13
tot = sum(s for n, c, s in rows) / 1e6 / 13
print(sys.argv[2], "total %.2f" % tot, "NT %.2f" % sum(by.values()), {k: round(v, 2) for k, v in sorted(by.items())},
      "scan_bwd %.2f" % sb, "launches/step %.0f" % (sum(c for n, c, s in rows) / 13))
