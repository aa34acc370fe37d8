"""Per-dispatch table of one train step from a rocprofv3 --kernel-trace rocpd SQLite file (dev tool): the kernels between two
adam launches in dispatch order, durations averaged over the steps of the run. usage: rocpd_sequence.py results.db"""
import re
import sqlite3
import sys

rows = sqlite3.connect(sys.argv[1]).execute(
    "select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
# a step starts at its cast_pad launch (round 6: the optimizer is several launches on two streams, no longer a delimiter)
idx = [i for i, r in enumerate(rows) if "cast_pad" in r[0]]
steps = [rows[idx[i]:idx[i + 1]] for i in range(len(idx) - 1)]
L = len(steps[-1])
steps = [s for s in steps if len(s) == L]
tot = 0.0
for k in range(L):
    name = re.sub(r"void |mpu::|\(anonymous namespace\)::|unsigned short", "", steps[0][k][0])[:52]
    us = sum(s[k][2] - s[k][1] for s in steps) / len(steps) / 1e3
    tot += us
    print("%3d %-54s grid=%6d %8.2f" % (k, name, steps[0][k][3] // max(steps[0][k][4], 1), us))
span = sum(max(r[2] for r in s) - s[0][1] for s in steps) / len(steps) / 1e3
print("total", tot, "steps", len(steps), "| first launch -> last completion of a step: %.1f us (launches of the optimizer branch run beside the others)" % span)
