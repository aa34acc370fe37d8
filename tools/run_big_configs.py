"""BASELINE.json configs[3] / configs[4] at full size (dev tool): one train step of B=32 256x256, one 6-view predict+fuse of
a 512^3 x 2-channel volume with 5 classes (dim 512 -> 532 planes per view)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.fusion_model import FusionModel
from multiplanarunet_amd.interpolation import Volume
from multiplanarunet_amd.predict import multi_view_predict
q = lambda *a, **k: None
ONLY = os.environ.get("BIG_ONLY", "")
# cfg4: train step B=32, 256x256
if ONLY in ("", "cfg4"):
  m = UNet(n_classes=3, dim=256, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16", logger=q, seed=0)
  m.compile("Adam", "SparseCategoricalCrossentropy")
  x = torch.randn(32, 256, 256, 1, device="cuda"); y = torch.randint(0, 3, (32, 256 * 256, 1), device="cuda", dtype=torch.uint8)
  w = torch.ones(32, device="cuda")
  for _ in range(3): m.train_step(x, y, w, want_loss=False)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(10): l = m.train_step(x, y, w, want_loss=False)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
  print("cfg4 train B=32 256^2: %.2f ms/step = %.0f slices/s" % (dt * 1e3, 32 / dt), "finite:", bool(torch.isfinite(m.params).all()))
  del m, x, y
if ONLY == "cfg4": sys.exit(0)
# cfg5: predict 512^3 x 2, K=5, V=6
D, K = 512, 5
rng = np.random.RandomState(0)
vol = Volume(rng.randn(D, D, D, 2).astype(np.float32), None, np.eye(4), bg_value=[0.0, 0.0],
             scaler=(np.array([0.0, 0.0]), np.array([1.3, 1.3])), device="cuda")
views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.707], [-0.6, 0.64, 0.48], [0.7, -0.5, 0.5]], float)
m = UNet(n_classes=K, dim=D, n_channels=2, depth=4, complexity_factor=1, dtype="bf16", logger=q, seed=0)
fm = FusionModel(6, K, verbose=False)
print("auto batch:", m.auto_batch(D + 20), "max batch:", m.max_batch())
multi_view_predict(m, vol, views[:1], D, float(D), FusionModel(1, K, verbose=False), want_probs=False)   # warm-up (1 view)
torch.cuda.synchronize(); t0 = time.perf_counter()
t = {}
probs, labels = multi_view_predict(m, vol, views, D, float(D), fm, want_probs=False, timings=t)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("cfg5 predict 512^3x2 K=5 V=6: %.2f s = %.1f Mvox/s" % (dt, D ** 3 / dt / 1e6), {k: round(v, 1) for k, v in t.items()},
      "labels", tuple(labels.shape), labels.dtype, "max label", int(labels.max()))
