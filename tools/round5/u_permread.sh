# R5u: what HBM delivers to coalesced requests whose DRAM pages are opened out of order (mpu_probe_permuted_read: 1 GiB read once in
# permuted runs of 128 B ... 4 KB) beside the float4 copy probe -- the bound of the gather kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5u; mkdir -p $O
cd $R
python bench.py --no-predict --no-cpu-baseline --no-e2e --no-kernel-events --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").readline())
p = d.get("measured_peaks") or d.get("roofline", {}).get("measured_peaks")
print(d["ms_per_step"], json.dumps(p))
PY
