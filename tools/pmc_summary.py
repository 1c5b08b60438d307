"""Aggregate rocprofv3 --pmc results (sqlite db) per kernel: mean counter value per launch.

    python tools/pmc_summary.py <db> [name-substring]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced
streaming reads by exactly 2x (MI355X_MICROARCH.md, HBM section), so `fetch_bytes_corrected` doubles it.
"""
import json
import re
import sqlite3
import sys

db = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
# pmc_events view: one row per (dispatch, counter)
name_col = "counter_name" if "counter_name" in cols else ("name" if "name" in cols else cols[0])
val_col = "value" if "value" in cols else "counter_value"
q = (f"select k.name, p.{name_col}, count(*), avg(p.{val_col}) from pmc_events p join kernels k "
     f"on p.event_id = k.id group by 1,2") if "event_id" in cols else None
rows = []
try:
    rows = c.execute(q).fetchall() if q else []
except Exception as e:  # fall back to dispatch-id join variants
    for key in ("dispatch_id", "id"):
        try:
            rows = c.execute(f"select k.name, p.{name_col}, count(*), avg(p.{val_col}) from pmc_events p join kernels k "
                             f"on p.{key} = k.dispatch_id group by 1,2").fetchall()
            break
        except Exception:
            continue
out = {}
for name, ctr, n, avg in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "")
    if (pat and pat not in short) or short.startswith(("Cijk_", "at::", "__amd")):
        continue
    out.setdefault(short, {})[ctr] = {"launches": n, "mean": avg}
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from yolo_master_amd.build import source_hash  # noqa: E402

print(json.dumps({"csrc_sha16": source_hash(), "kernels": out}, indent=1))
