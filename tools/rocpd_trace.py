"""List the dispatches of one train step (kernel, grid, duration) from a rocpd db. Dev tool."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
scol = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scol else "display_name"
gx = "grid_size_x" if "grid_size_x" in cols else None
q = "select s.%s, d.start, d.end, %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start" % (name_col, ("d.grid_size_x, d.workgroup_size_x" if gx else "0,0"))
rows = cur.execute(q).fetchall()
# find the last occurrence of cast_pad (start of a step) and print until the next adam
starts = [i for i, r in enumerate(rows) if "cast_pad" in r[0]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = rows[i0][1]
for r in rows[i0:i1]:
    n = re.sub(r"\(.*", "", r[0]).replace("_ZN3mpu", "")
    print("%9.1f us  +%7.1f  grid %7d  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[3] // max(r[4], 1)) if r[4] else 0, n[:90]))
print("step span %.1f us, sum of kernel time %.1f us" % ((rows[i1 - 1][2] - t0) / 1e3, sum(r[2] - r[1] for r in rows[i0:i1]) / 1e3))
