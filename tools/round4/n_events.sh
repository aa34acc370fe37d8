#!/bin/bash
# R4n: roofline leg with dispatch-bound events (hipExtLaunchKernel start/stop) vs marker events (MPU_PROF_MARKERS=1)
mkdir -p gpurun_out/R4n
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/R4n/bench_ext.json 2> gpurun_out/R4n/bench_ext.err
MPU_PROF_MARKERS=1 timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/R4n/bench_markers.json 2> gpurun_out/R4n/bench_markers.err
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/R4n/pytest.log
for f in ext markers; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/R4n/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"], d.get("wgrad_roofline"))
except Exception as e:
    print(f, "ERR", e); print(open(f"gpurun_out/R4n/bench_{f}.err").read()[-1500:])
PY
done
cat gpurun_out/R4n/pytest.log
