"""Per-shape HBM-side fetch counters of the convolution cores (one process, every shape launched REPS times in a row):
   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- python tools/micro/conv_shape_pmc.py run
   python tools/micro/conv_shape_pmc.py report <db>
Answers "which shapes re-fetch their input" (the aggregated summaries under profiles/ average over shapes of one instantiation)."""
import sqlite3
import sys

REPS = 5
SHAPES = [  # cin, cout, k, s, hw (batch 64): the 3x3 shapes of YOLO-Master-S at 640^2 + three 1x1
    (128, 128, 3, 2, 160), (256, 256, 3, 2, 80), (256, 512, 3, 2, 40), (128, 128, 3, 2, 80), (256, 256, 3, 2, 40),
    (128, 64, 3, 1, 80), (256, 64, 3, 1, 40), (256, 64, 3, 1, 20), (64, 64, 3, 1, 80), (64, 64, 3, 1, 40), (64, 64, 3, 1, 20),
    (192, 256, 1, 1, 80), (384, 256, 1, 1, 40), (256, 768, 1, 1, 20)]


def run():
    import torch
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from yolo_master_amd import ops
    bf = torch.bfloat16
    for cin, cout, k, s, hw in SHAPES:
        g = torch.Generator().manual_seed(cin + cout + k)
        x = torch.randn(64, hw, hw, cin, generator=g).to(bf).cuda()
        w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, bf).cuda()
        bias = (torch.randn(cout, generator=g) * 0.1).cuda()
        torch.cuda.synchronize()
        for _ in range(REPS):
            y = ops.conv2d(x, w, bias, k, s, True)
        torch.cuda.synchronize()
        del x, w, y


def report(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    name_col = "counter_name" if "counter_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    dur = "(k.end - k.start)" if "end" in kcols and "start" in kcols else "0"
    rows = c.execute(f"select k.id, k.name, p.{name_col}, p.{val_col}, {dur} from pmc_events p join kernels k on p.event_id = k.id "
                     f"order by k.id").fetchall()
    rows = [r for r in rows if "conv" in r[1] and not r[1].startswith(("Cijk", "void at::"))]
    print(f"{'shape':28s} {'kernel':44s} {'x MB':>8s} {'y MB':>8s} {'counter MB (2x FETCH KiB)':>26s} {'ratio to x':>10s}")
    i = 0
    for cin, cout, k, s, hw in SHAPES:
        grp = rows[i: i + REPS]
        i += REPS
        if len(grp) < REPS:
            break
        ho = (hw + 2 * (k // 2) - k) // s + 1
        xb, yb = 64 * hw * hw * cin * 2 / 1e6, 64 * ho * ho * cout * 2 / 1e6
        vals = sorted(r[3] for r in grp[1:])
        fetch = 2 * vals[len(vals) // 2] * 1024 / 1e6
        name = grp[-1][1].replace("void ", "").split("(")[0][:44]
        print(f"{'%d->%d k%d s%d @%d' % (cin, cout, k, s, hw):28s} {name:44s} {xb:8.1f} {yb:8.1f} {fetch:26.1f} {fetch / xb:10.2f}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
