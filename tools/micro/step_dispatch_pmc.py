"""Per-dispatch fabric counters of ONE eager bench step, in launch order (the summaries under profiles/ average per kernel name):
   python tools/micro/step_dispatch_pmc.py <fetch.db> <write.db> <dispatches per step>"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_master_amd.build import source_hash  # noqa: E402


def load(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    name_col = "counter_name" if "counter_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    rows = c.execute(f"select k.id, k.name, p.{val_col}, (k.end - k.start) from pmc_events p join kernels k on p.event_id = k.id order by k.id").fetchall()
    return [r for r in rows if not r[1].startswith(("Cijk", "void at::", "__amd"))]


f, w = load(sys.argv[1]), load(sys.argv[2])
names = [r[1] for r in f]
# one step = the span between two consecutive stem kernels
starts = [i for i, n in enumerate(names) if "stem_pair" in n]
a, b = starts[-2], starts[-1]
wn = [r[1] for r in w]
ws = [i for i, n in enumerate(wn) if "stem_pair" in n]
wa = ws[-2]
print(f"{'#':>3s} {'kernel':60s} {'us':>8s} {'fetch MB':>9s} {'write MB':>9s}")
tf = tw = 0.0
for j in range(b - a):
    r, q = f[a + j], w[wa + j]
    assert r[1] == q[1], (r[1], q[1])
    fm, wm = 2 * r[2] * 1024 / 1e6, q[2] * 1024 / 1e6
    tf += fm; tw += wm
    print(f"{j:3d} {r[1].replace('void ', '').split('(')[0][:60]:60s} {r[3] / 1e3:8.1f} {fm:9.1f} {wm:9.1f}")
print(f"step: fetch {tf / 1e3:.2f} GB (2x FETCH_SIZE), write {tw / 1e3:.2f} GB")
print(f"# csrc_sha16 {source_hash()} weights {os.environ.get('YMK_BENCH_WEIGHTS', 'cond')}")   # bench.py quotes the total only for matching kernel sources
