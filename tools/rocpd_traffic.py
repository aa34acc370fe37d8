"""HBM bytes per launch of the conv / wgrad kernel families from two rocprofv3 PMC passes (dev tool).
usage: rocpd_traffic.py FETCH_SIZE.db WRITE_SIZE.db out.json
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section)."""
import os, sqlite3, sys, re, json

CLASSES = {
    "conv_igemm": (("conv_glds", "conv_halo", "conv_ws", "conv_c8", "conv_igemm", "conv_pipe"), ("splitk_finish",)),
    "wgrad_igemm": (("wgrad_glds", "wgrad_taps", "wgrad_c8_kernel", "wgrad_igemm"), ("wgrad_reduce", "colsum_finalize", "wgrad_c8_finalize")),
}

def sums(path, counter):
    db = sqlite3.connect(path); cur = db.cursor()
    sc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sc else "display_name"
    q = """select s.%s, sum(e.value), count(distinct d.id) from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id=s.id where p.name=? group by s.%s""" % (name_col, name_col)
    return {re.sub(r"\(.*", "", n): (v, k) for n, v, k in cur.execute(q, (counter,))}

fetch, write = sums(sys.argv[1], "FETCH_SIZE"), sums(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 2 --no-graph "
               "--no-kernel-events`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B, "
               "MI355X_MICROARCH.md section HBM); per launch of the kernel class incl. its split-K finish / reduce passes",
       "classes": {}, "kernels": {}}
for cls, (main, aux) in CLASSES.items():
    f = w = 0.0; n = 0
    for name, (v, k) in fetch.items():
        if any(m in name for m in main): f += v; n += k
        elif any(m in name for m in aux): f += v
    for name, (v, k) in write.items():
        if any(m in name for m in main + aux): w += v
    out["classes"][cls] = {"launches": n, "fetch_KB_raw_per_launch": f / max(n, 1), "write_KB_per_launch": w / max(n, 1),
                           "hbm_bytes_per_launch": (2 * f + w) * 1024 / max(n, 1)}
for name, (v, k) in sorted(fetch.items(), key=lambda kv: -kv[1][0]):
    short = name.replace("_ZN3mpu", "")[:80]
    wv = write.get(name, (0.0, k))[0]
    out["kernels"][short] = {"launches": k, "hbm_MB_per_launch": round((2 * v + wv) * 1024 / k / 1e6, 2)}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiplanarunet_amd.srchash import source_sha16, CONV_SOURCES
out["source_sha16"] = source_sha16(CONV_SOURCES)           # bench.py reports these bytes only for the sources they were measured on
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["classes"], indent=1))
