"""R6ag: where the graphed bf16x3 pipeline's epoch loss differs from the serial loop's although the parameters agree bit for bit:
per-step history of both device-side loss sums (stream-ordered copies, no host synchronisation inside an epoch). Dev tool."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.pipeline import TrainPipeline
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0"); B, dim = 16, 128
quiet = lambda *a, **k: None
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
views = random_views(6, 60.0, 0)
def mk():
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype=dtype, logger=quiet, seed=0, device=dev)
    m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-4})
    return m, TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=3)
m0, s0 = mk(); m1, s1 = mk()
p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
p1 = TrainPipeline(m1, s1)
N = 60
for ep in range(10):
    h0 = torch.zeros(N, dtype=torch.float64, device=dev); h1 = torch.zeros(N, dtype=torch.float64, device=dev)
    for i in range(N):
        p0.step(); h0[i:i + 1].copy_(p0.loss_sum)
        p1.step(); h1[i:i + 1].copy_(p1.loss_sum)
    a, b = p0.epoch_loss(), p1.epoch_loss()
    d0 = torch.diff(h0, prepend=h0.new_zeros(1)).cpu().numpy(); d1 = torch.diff(h1, prepend=h1.new_zeros(1)).cpu().numpy()
    bad = [i for i in range(N) if d0[i] != d1[i]]
    print(ep, "%.6f %.6f" % (a, b), "params", "EQ" if torch.equal(m0.params, m1.params) else "DIFF", "steps that differ:", bad[:12],
          [(round(float(d0[i]), 5), round(float(d1[i]), 5)) for i in bad[:6]], flush=True)
    if ep % 2 == 1:
        for m in (m0, m1):
            m.optimizer_kwargs["lr"] *= 0.9
