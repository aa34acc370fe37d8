#!/bin/bash
# round 4 call G: shader clock during the train step / predict / probes (bench.py's new clock fields); full bench line
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4g; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/R4g/bench_line.json").read())
print("ms/step", d["ms_per_step"], "clock during step", d.get("shader_clock_mhz_during_step"), "conv frac", d["roofline"]["frac"], "wgrad frac", d["wgrad"]["frac"])
p=d["predict_fuse"]; print("predict", p["seconds"], "unet_ms", p["unet_ms"], "clock", p.get("shader_clock_mhz_during_predict"), "map_fuse_ms", p["map_fuse_ms"])
print(d["measured_peaks"])
PY
