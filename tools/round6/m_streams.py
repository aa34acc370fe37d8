"""Which producer streams let the mp-train loop run at the step's rate? One process, one captured graph, N candidate streams,
30 pipeline steps on each. Dev tool (round 6)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.pipeline import TrainPipeline
quiet = lambda *a, **k: None
dev = torch.device("cuda")
B, dim = 16, 128
m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16", logger=quiet, seed=0, device=dev)
m.compile("Adam", "SparseCategoricalCrossentropy")
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
smp = TrainSampler([vol], random_views(6, 60.0, 0), dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=7)
pipe = TrainPipeline(m, smp)
pipe.run_epoch(12)
def rate(n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pipe.run_epoch(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("picked stream (probe %.0f us): %.3f ms per step" % (pipe.side_latency_us, rate()))
cands = [torch.cuda.Stream(device=dev, priority=-1 if k % 2 == 0 else 0) for k in range(12)]
for k, st in enumerate(cands):
    pipe._pending = None; pipe.side = st
    rate(4)
    print("candidate %2d prio %2d: %.3f ms per step" % (k, -1 if k % 2 == 0 else 0, rate()))
