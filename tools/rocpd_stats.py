"""Per-kernel stats (count, total, avg, %) from a rocprofv3 rocpd SQLite file. Dev tool."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
scol = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
rows = cur.execute("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.%s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows)
span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
print("total kernel time %.3f ms over %d kernels; first->last dispatch span %.3f ms" % (tot / 1e6, sum(r[1] for r in rows), (span[1] - span[0]) / 1e6))
print("%-110s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for n, c, t, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = re.sub(r"\(.*", "", n)
    print("%-110s %8d %12.1f %10.2f %6.2f" % (n[:110], c, t / 1e3, t / c / 1e3, 100.0 * t / tot))
