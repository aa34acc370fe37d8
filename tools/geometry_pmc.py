"""HBM traffic of the geometry kernels from two rocprofv3 PMC passes of tools/bench_geometry.py (dev tool).
usage: geometry_pmc.py FETCH_SIZE.db WRITE_SIZE.db out.json
bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B; the factor 2 was confirmed for
12-byte-per-lane reads with mpu_probe_gather12 in round 3: 0.805 GB counted for 1.611 GB read)."""
import json, os, re, sqlite3, sys


def sums(path, counter):
    db = sqlite3.connect(path); cur = db.cursor()
    sc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sc else "display_name"
    q = """select s.%s, sum(e.value), count(distinct d.id), sum(d.end - d.start) from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id=s.id where p.name=? group by s.%s""" % (name_col, name_col)
    return {re.sub(r"\(.*", "", n): (v, k, t) for n, v, k, t in cur.execute(q, (counter,))}


fetch, write = sums(sys.argv[1], "FETCH_SIZE"), sums(sys.argv[2], "WRITE_SIZE")
out = {"note": __doc__.strip().split("\n", 2)[2], "kernels": {}}
for name, (v, k, t) in fetch.items():
    if not any(s in name for s in ("map_fuse", "map_view", "sample_f", "sample_view")):
        continue
    mt = re.search(r"(map_fuse_fast_kernel|map_fuse_fixup_kernel|map_view_fast_kernel|map_view_fixup_kernel|sample_fast_kernel|sample_fixup_kernel|map_fuse_kernel|sample_view_planes_kernel)ILi(\d+)E?L?[ib]?(\d*)", name)
    short = name.replace("_ZN3mpu", "")[:70]
    if mt and mt.group(1) == "map_fuse_fast_kernel":
        short = "map_fuse_fast_kernel<%s,%s>" % (mt.group(2), mt.group(3) or "2")
    wv = write.get(name, (0.0, k, 0))[0]
    out["kernels"][short] = {"launches": k, "avg_us_profiled": round(t / k / 1e3, 1), "fetch_KB_raw_per_launch": round(v / k, 1),
                             "write_KB_per_launch": round(wv / k, 1), "hbm_bytes_per_launch": int((2 * v + wv) * 1024 / k)}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiplanarunet_amd.srchash import source_sha16, GEOMETRY_SOURCES
out["source_sha16"] = source_sha16(GEOMETRY_SOURCES)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
