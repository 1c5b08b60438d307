"""Summarise a rocprofv3 --kernel-trace sqlite db: per-kernel totals (and optional per-grid breakdown)."""
import re
import sqlite3
import sys

db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
pat = sys.argv[3] if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
if pat:
    rows = c.execute("select grid_x/workgroup_x, grid_y, grid_z, count(*), avg(end-start)/1e3, sum(end-start)/1e3 from kernels "
                     "where name like ? group by 1,2,3 order by 6 desc", (f"%{pat}%",)).fetchall()
    for gx, gy, gz, n, avg, tot in rows[:20]:
        print(f"   grid=({gx},{gy},{gz}) n/step={n / steps:.1f} avg={avg:8.1f}us  per-step={tot / steps:8.1f}us")
    sys.exit(0)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time per step: {tot / steps / 1e3:.3f} ms ({steps:g} steps)")
print(f"{'%':>6} {'us/step':>10} {'calls/step':>10} {'avg us':>9}  kernel")
for n, cnt, s, a in rows[:40]:
    name = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "").replace("unsigned short", "bf16")[:90]
    print(f"{s / tot * 100:6.2f} {s / steps:10.1f} {cnt / steps:10.1f} {a:9.2f}  {name}")
