"""Throughput of the train-time plane sampler (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.augmentation import build_augmenters
vols = []
for i in range(4):
    img, lab, aff = make_toy_volume(128, i)
    vols.append(as_volume(img, lab, aff, "1pct", "RobustScaler", "cuda", "toy%d" % i))
views = random_views(6, 60.0, 0)
for augs in (None, [{"cls_name": "Elastic2D", "kwargs": {"alpha": [0, 450], "sigma": [20, 30], "apply_prob": 0.333}}]):
    s = TrainSampler(vols, views, 128, 128.0, 16, 3, noise_sd=0.1, seed=0, augmenters=build_augmenters(augs, 1))
    for _ in range(3): s()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): x, y, w = s()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("augmenters=%s: %.1f ms / batch of 16 -> %.0f slices/s" % (bool(augs), dt / n * 1e3, 16 * n / dt))
