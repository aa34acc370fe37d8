"""R6al: inside the failing configuration (graphed bf16x3 pipeline): WHICH tensor of `loss_sum.add_(loss.mean().double())` is stale
when the added value repeats -- the per-pixel loss the library wrote, torch's captured mean of it, or the cast. Dev tool."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.pipeline import TrainPipeline
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0"); B, dim = 16, 128
quiet = lambda *a, **k: None
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
views = random_views(6, 60.0, 0)
m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype=dtype, logger=quiet, seed=0, device=dev)
m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-4})
s = TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=3)
keep = {}
orig_fb = m.forward_backward
def fb(*a, **k):
    r = orig_fb(*a, **k)
    if torch.cuda.is_current_stream_capturing():
        keep["loss"] = r[1]
    return r
m.forward_backward = fb
orig_mean = torch.Tensor.mean
def mean(self, *a, **k):
    r = orig_mean(self, *a, **k)
    if torch.cuda.is_current_stream_capturing() and self is keep.get("loss"):
        keep["mean"] = r
    return r
torch.Tensor.mean = mean
p = TrainPipeline(m, s, overlap=False)
p.step()                                                     # capture (+ warm-up step)
torch.Tensor.mean = orig_mean
assert "loss" in keep and "mean" in keep, keep.keys()
N = 600
h_sum = torch.zeros(N, dtype=torch.float64, device=dev); h_mean = torch.zeros(N, device=dev); h_eager = torch.zeros(N, dtype=torch.float64, device=dev)
h_px = torch.zeros(N, device=dev)
for i in range(N):
    p.step()
    h_sum[i:i + 1].copy_(p.loss_sum); h_mean[i:i + 1].copy_(keep["mean"].reshape(1))
    h_eager[i:i + 1].copy_(keep["loss"].double().mean().reshape(1)); h_px[i:i + 1].copy_(keep["loss"].reshape(-1)[12345:12346])
torch.cuda.synchronize()
hs, hm, he, hp = h_sum.cpu().numpy(), h_mean.cpu().numpy(), h_eager.cpu().numpy(), h_px.cpu().numpy()
inc = np.diff(hs, prepend=0.0)
stale_mean = [i for i in range(1, N) if hm[i] == hm[i - 1]]
stale_loss = [i for i in range(1, N) if he[i] == he[i - 1]]
mism = [i for i in range(N) if abs(he[i] - hm[i]) > 1e-5 * abs(he[i])]
print("captured mean repeats at %d steps %s" % (len(stale_mean), stale_mean[:8]))
print("eager mean of the captured per-pixel loss repeats at %d steps %s" % (len(stale_loss), stale_loss[:8]))
print("captured mean != eager mean of the same tensor at %d steps %s" % (len(mism), mism[:8]))
for i in (mism[:3] if mism else []):
    print("  step %d: added %.6f, captured mean %.6f, eager mean %.6f, pixel %.6f (previous step: %.6f %.6f %.6f)" % (i, inc[i], hm[i], he[i], hp[i], hm[i - 1], he[i - 1], hp[i - 1]))
