R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6p; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_baseline_shapes.py tests/test_gpu_unet.py tests/test_gpu_replay.py -q -s -k "split_bf16" > $O/pytest.log 2>&1; grep -E "passed|failed|max .err|Error|assert|bf16x3 step|replay:" $O/pytest.log | tail -12
python - <<'PY' 2>&1 | tail -4
import time, numpy as np, torch
from multiplanarunet_amd.unet import UNet
q = lambda *a, **k: None
rng = np.random.RandomState(0)
x = torch.tensor(rng.randn(16, 128, 128, 1).astype(np.float32), device="cuda")
y = torch.tensor(rng.randint(0, 3, (16, 128 * 128, 1)).astype(np.uint8), device="cuda")
for dt in ("bf16x3",):
    m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype=dt, logger=q, seed=0)
    m.compile("Adam", "SparseCategoricalCrossentropy")
    for _ in range(3): m.train_step(x, y, None, want_loss=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.train_step(x, y, None, want_loss=False)
    torch.cuda.synchronize(); print("%-7s train step %.3f ms" % (dt, (time.perf_counter() - t0) / 10 * 1e3))
PY
