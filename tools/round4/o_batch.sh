#!/bin/bash
# R4o: predict batch-size sweep (planes per forward launch): grid-quantisation tails at the deep levels
mkdir -p gpurun_out/R4o
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python - > gpurun_out/R4o/sweep.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
from multiplanarunet_amd.predict import multi_view_predict
dev = torch.device("cuda:0")
quiet = lambda *a, **k: None
vol, views, model, fm = bench._predict_setup(dev, quiet, 256, 6, 3, 1)
for rnd in range(2):
    for b in (None, 92, 96, 128, 112, 69, 138, 160, 276):
        try:
            multi_view_predict(model, vol, views, 256, 256.0, fm, batch_size=b, want_probs=False)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t = {}
                t0 = time.perf_counter()
                multi_view_predict(model, vol, views, 256, 256.0, fm, batch_size=b, want_probs=False, timings=t)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0, t["unet_ms"]))
            ts.sort()
            print("batch", b, "total %.1f ms unet %.1f ms" % (ts[1][0] * 1e3, ts[1][1]), flush=True)
        except Exception as e:
            print("batch", b, "ERR", str(e)[:200], flush=True)
PY
cat gpurun_out/R4o/sweep.txt
