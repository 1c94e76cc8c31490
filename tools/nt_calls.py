"""Durations (us) of the gemm_nt launches of the LAST train step in a rocpd database, in launch order."""
import json
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, grid_x, workgroup_x, (end-start)/1000.0 from kernels where name like '%gemm_nt%' order by start").fetchall()
per = len(rows) // 13
out = []
for n, g, w, d in rows[-per:]:
    m = re.search(r"Li(\d)ELi(\d+)ELi(\d+)E", n)
    if not m:
        m2 = re.search(r"E, (\d+), (\d+)>", n)
        tag = ("r", m2.group(1), m2.group(2))
    else:
        tag = m.groups()
    out.append([tag[0], int(tag[1]) * int(tag[2]) // 64, round(d, 1)])
print(json.dumps(out))
