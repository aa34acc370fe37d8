# R6aa: bf16x3 with the BatchNorm accumulators and the bias gradient out of the dz split pass: tests, step time, kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6aa; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_conv.py tests/test_gpu_baseline_shapes.py tests/test_gpu_replay.py -q -x -k "split_bf16 or x3 or fused_adam or graphed_train_step" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/x3step.py <<PY
import sys, time, numpy as np, torch
sys.path.insert(0, "$R")
from multiplanarunet_amd.unet import UNet
q = lambda *a, **k: None
rng = np.random.RandomState(0)
x = torch.tensor(rng.randn(16, 128, 128, 1).astype(np.float32), device="cuda")
y = torch.tensor(rng.randint(0, 3, (16, 128 * 128, 1)).astype(np.uint8), device="cuda")
m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype="bf16x3", logger=q, seed=0)
m.compile("Adam", "SparseCategoricalCrossentropy")
for _ in range(6): m.train_step(x, y, None, want_loss=False)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    rep = m.make_graphed_train_step(x, y, None); rep(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rep()
    torch.cuda.synchronize(); print("bf16x3 graphed step %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
PY
python /tmp/x3step.py time; MPU_BN_ATOMIC=0 python /tmp/x3step.py time
rocprofv3 --kernel-trace --stats -d $O/x3 -o s -- python /tmp/x3step.py > /dev/null 2>&1
S=$(find $O/x3 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $S 40 > $O/stats_bf16x3.txt; head -34 $O/stats_bf16x3.txt | cut -c1-150
rm -rf $O/x3
