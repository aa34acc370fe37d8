# R6q: where the bf16x3 (and f32) step spends its time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in bf16x3; do
cat > /tmp/x3step.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from multiplanarunet_amd.unet import UNet
q = lambda *a, **k: None
rng = np.random.RandomState(0)
x = torch.tensor(rng.randn(16, 128, 128, 1).astype(np.float32), device="cuda")
y = torch.tensor(rng.randint(0, 3, (16, 128 * 128, 1)).astype(np.uint8), device="cuda")
m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype="$dt", logger=q, seed=0)
m.compile("Adam", "SparseCategoricalCrossentropy")
for _ in range(6): m.train_step(x, y, None, want_loss=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $O/$dt -o s -- python /tmp/x3step.py > /dev/null 2>&1
S=$(find $O/$dt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $S 28 > $O/stats_$dt.txt; echo "== $dt"; head -30 $O/stats_$dt.txt | cut -c1-150
rm -rf $O/$dt
done
