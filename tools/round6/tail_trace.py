"""Tail of one train step (from the first wgrad group launch on) with start / end times: shows which kernels overlap. Dev tool.
usage: tail_trace.py results.db [graph]"""
import re, sqlite3, sys
rows = sqlite3.connect(sys.argv[1]).execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "cast_pad" in r[0]]
i0, i1 = starts[-2], starts[-1]
step = rows[i0:i1]
t0 = step[0][1]
k0 = next(i for i, r in enumerate(step) if "wgrad_c8_kernel" in r[0])
for r in step[k0:]:
    n = re.sub(r"void |mpu::|\(anonymous namespace\)::|unsigned short|\(.*", "", r[0])[:44]
    print("%9.1f -> %9.1f  (%7.1f us) grid %6d  %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), n))
print("step span %.1f us; sum of kernel durations %.1f us" % ((step[-1][2] - t0) / 1e3, sum(r[2] - r[1] for r in step) / 1e3))
per = [(rows[starts[i + 1]][1] - rows[starts[i]][1]) / 1e3 for i in range(len(starts) - 1)]
gaps = [(rows[starts[i + 1]][1] - max(r[2] for r in rows[starts[i]:starts[i + 1]])) / 1e3 for i in range(len(starts) - 1)]
print("step periods (us):", " ".join("%.0f" % p for p in per[-8:]), "| gaps between steps (us):", " ".join("%.1f" % g for g in gaps[-8:]))
