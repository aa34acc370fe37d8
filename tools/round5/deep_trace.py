"""Per-layer kernel durations of tools/round5/deep_layers.py from a rocprofv3 --kernel-trace rocpd file: consecutive dispatches are
grouped by (kernel, grid); the first three of a group (warm-up) are dropped. usage: deep_trace.py results.db"""
import re
import sqlite3
import sys
rows = sqlite3.connect(sys.argv[1]).execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
want = [r for r in rows if re.search(r"conv_deep|conv_pipe|splitk_finish|conv_glds", r[0])]
groups = []
i = 0
while i + 1 < len(want):                                        # (conv, finish) pairs repeat REPS + 3 times per layer
    key = (want[i][0], want[i][3], want[i + 1][0], want[i + 1][3])
    j = i
    durs = []
    while j + 1 < len(want) and (want[j][0], want[j][3], want[j + 1][0], want[j + 1][3]) == key:
        durs.append(((want[j][2] - want[j][1]) / 1e3, (want[j + 1][2] - want[j + 1][1]) / 1e3, (want[j + 1][2] - want[j][1]) / 1e3))
        j += 2
    if durs:
        d = durs[3:] if len(durs) > 3 else durs
        nm = re.sub(r"void |mpu::|\(anonymous namespace\)::|unsigned short", "", key[0])[:40]
        print("%-42s grid %5d x%2d: conv %6.2f us, finish %6.2f us, first start -> finish end %6.2f us" % (
            nm, key[1] // max(want[i][4], 1), len(d), sum(x[0] for x in d) / len(d), sum(x[1] for x in d) / len(d), sum(x[2] for x in d) / len(d)))
        i = j
    else:
        i += 1
