"""One line per kernel from the committed SQ / PMC counter summaries of a round (profiles/<tag>_sq_{1,2}.json, <tag>_pmc_{FETCH,WRITE}_SIZE.json,
written by tools/pmc_summary.py): matrix-core busy share, VALU issue rate, parked-wave share, LDS conflict share, HBM bytes per launch.
The SQ counters of this rocprofv3 sample ONE of the 32 shader engines: busy fractions are MFMA_BUSY_CYCLES / (32 SIMDs of an engine x
SQ_BUSY_CYCLES); HBM bytes = 2 x FETCH_SIZE (gfx950 counts wide reads at half size) + WRITE_SIZE, KiB.    python tools/sq_summary.py r03"""
import json
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = Path(__file__).resolve().parent.parent / "profiles"
raw = lambda n: json.loads((P / f"{tag}_{n}.json").read_text())   # noqa: E731
sha = raw("sq_1").get("csrc_sha16")
ld = lambda n: raw(n).get("kernels", raw(n))   # noqa: E731
s1, s2, fe, wr = ld("sq_1"), ld("sq_2"), ld("pmc_FETCH_SIZE"), ld("pmc_WRITE_SIZE")
g = lambda d, k, c: d.get(k, {}).get(c, {}).get("mean", 0.0) if isinstance(d.get(k), dict) else 0.0   # noqa: E731
rows = []
for k, v in s1.items():
    if not isinstance(v, dict) or "SQ_BUSY_CYCLES" not in v:
        continue
    n = v["SQ_BUSY_CYCLES"].get("launches", 0)
    busy, wave = g(s1, k, "SQ_BUSY_CYCLES"), g(s1, k, "SQ_WAVE_CYCLES")
    mfma = g(s1, k, "SQ_VALU_MFMA_BUSY_CYCLES") / (32.0 * busy) if busy else 0.0
    valu = g(s1, k, "SQ_ACTIVE_INST_VALU") / wave if wave else 0.0
    parked = g(s2, k, "SQ_WAIT_ANY") / wave if wave else 0.0
    conf = g(s2, k, "SQ_LDS_BANK_CONFLICT") / g(s2, k, "SQ_LDS_IDX_ACTIVE") if g(s2, k, "SQ_LDS_IDX_ACTIVE") else 0.0
    hbm = (2.0 * g(fe, k, "FETCH_SIZE") + g(wr, k, "WRITE_SIZE")) * 1024 / 1e6
    rows.append((busy * n, k, n, mfma, valu, parked, conf, hbm))
print(f"csrc_sha16 {sha}")
print("kernel | launches (3 eager steps) | MFMA busy | VALU inst / wave-cycle | waves parked | LDS conflict cycles / LDS cycles | HBM counter MB per launch (2*FETCH+WRITE)")
for _, k, n, mfma, valu, parked, conf, hbm in sorted(rows, reverse=True)[:30]:
    print(f"{k[:62]:62s} {n:7d} {mfma * 100:6.1f} % {valu:6.3f} {parked * 100:5.0f} % {conf * 100:5.0f} % {hbm:9.1f}")
