"""R6an: run-to-run reproducibility of the 6-view predict + fuse (same weights, same volume, twice in one process and once with another
plane batch): identical label volumes and probabilities. Dev tool."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch, hashlib
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views
from multiplanarunet_amd.predict import multi_view_predict
dev = torch.device("cuda:0")
quiet = lambda *a, **k: None
m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype="bf16", logger=quiet, seed=0, device=dev)
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
views = random_views(6, 60.0, 0)
outs = []
keepv = []
for bs in (None, None, 37):
    probs, labels = multi_view_predict(m, vol, views, 128, 128.0, sum_fusion=True, batch_size=bs)
    torch.cuda.synchronize()
    keepv.append((probs.clone(), labels.clone()))
    outs.append(hashlib.sha256(labels.cpu().numpy().tobytes()).hexdigest()[:16] + "/" +
                hashlib.sha256(probs.cpu().numpy().tobytes()).hexdigest()[:16])
print("label / probability volume hashes (automatic batch twice, then batch 37):", outs, "IDENTICAL" if len(set(outs)) == 1 else "DIFFERENT (labels: %s)" % ("same" if len(set(o.split("/")[0] for o in outs)) == 1 else "differ"))
d = (keepv[0][1] != keepv[2][1])
pd = (keepv[0][0] - keepv[2][0]).abs()
top2 = keepv[0][0].topk(2, dim=-1).values
margin = (top2[..., 0] - top2[..., 1])
print("automatic batch vs batch 37: %d of %d labels differ (%.4f %%), fused probabilities differ by at most %.2e (mean %.2e); the margin between the "
      "two best classes at the differing voxels is at most %.2e" % (int(d.sum()), d.numel(), 100.0 * float(d.float().mean()), float(pd.max()), float(pd.mean()),
                                                                    float(margin[d].max()) if int(d.sum()) else 0.0))
