"""Per-kernel PMC counter sums from a rocprofv3 rocpd db. Dev tool."""
import sqlite3, sys, re
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pc = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
ic = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
ec = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
sc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in sc else "display_name"
q = """select s.%s, p.name, sum(e.value), count(distinct d.id), sum(d.end-d.start)
       from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id
       join rocpd_kernel_dispatch d on e.event_id=d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.%s, p.name""" % (name_col, name_col)
try:
    rows = cur.execute(q).fetchall()
except Exception as ex:
    print("query failed:", ex); print(pc, ic, ec); sys.exit(1)
by = defaultdict(dict)
for n, c, v, k, t in rows:
    by[re.sub(r"\(.*", "", n).replace("_ZN3mpu", "")][c] = (v, k, t)
for n, d in sorted(by.items(), key=lambda kv: -max(x[2] for x in kv[1].values())):
    if not any(k in n for k in ("igemm", "conv_", "wgrad_")) and len(sys.argv) < 3: continue
    k = max(x[1] for x in d.values()); t = max(x[2] for x in d.values())
    print("%s  (%d dispatches, %.1f us avg)" % (n[:100], k, t / k / 1e3))
    print("    " + "  ".join("%s=%.4g" % (c, v[0] / k) for c, v in sorted(d.items())))
