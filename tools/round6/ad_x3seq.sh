# R6ad: launch sequence of one bf16x3 step (which split / reduce launches are slow)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ad; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
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
for _ in range(8): m.train_step(x, y, None, want_loss=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $O/x3 -o s -- python /tmp/x3step.py > /dev/null 2>&1
S=$(find $O/x3 -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $S > $O/x3_sequence.txt 2>&1; tail -3 $O/x3_sequence.txt
rm -rf $O/x3
