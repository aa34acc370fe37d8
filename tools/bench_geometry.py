"""Geometry kernels alone at the predict shape (dev tool): sampling of V views and the fused back-mapping with random
per-view predictions; fast paths timed and checked for equality against the exact search (MPU_GEOM_FAST=0 path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiplanarunet_amd import _lib
from multiplanarunet_amd.interpolation import Volume, ViewGeometry, sample_view, map_and_fuse, map_accumulate

D = int(os.environ.get("D", 256)); K = int(os.environ.get("K", 3)); C = int(os.environ.get("C", 1))
check = int(os.environ.get("CHECK", 1)); reps = int(os.environ.get("REPS", 5))
lib = _lib.load()
rng = np.random.RandomState(0)
vol = Volume(rng.randn(D, D, D, C).astype(np.float32), None, np.eye(4), bg_value=[0.0] * C,
             scaler=(np.zeros(C), np.full(C, 1.349)))
views = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.707], [-0.6, 0.64, 0.48], [0.7, -0.5, 0.5]], float)
geoms = [ViewGeometry(v, D, float(D), "same+20") for v in views]
P = geoms[0].n_planes
W = torch.tensor(rng.uniform(.5, 1.5, (len(views), K)).astype(np.float32), device="cuda")
b = torch.tensor(rng.uniform(-.1, .1, (K,)).astype(np.float32), device="cuda")
preds = [(torch.rand((g.n_planes, D, D, K), device="cuda"), (g.real_axis, g.real_axis, g.offsets), g.inv_basis,
          g.device_axes(vol.device)) for g in geoms]
outs = [torch.empty((P, D, D, C), device="cuda") for _ in geoms]


def run_sample():
    for g, o in zip(geoms, outs):
        sample_view(vol, g, want_labels=False, out=o)


def run_fuse():
    return map_and_fuse(vol, preds, W, b, want_probs=False)


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, r


res = {}
for fast in (1, 0) if check else (1,):
    _lib.check(lib.mpu_geometry_set_fast_path(fast), "set_fast")
    ts, _ = timed(run_sample)
    tf, r = timed(run_fuse)
    res[fast] = ([o.clone() for o in outs], r[1].clone())
    zacc = torch.zeros((D, D, D, K), device="cuda")
    ta, _ = timed(lambda: map_accumulate(vol, preds[3][0], preds[3][1], preds[3][2], W[3], 0, P, True, zacc))
    samp_bytes = len(views) * (4 * D ** 3 * C + 4 * P * D * D * C)
    fuse_bytes = D ** 3 * (len(views) * K * 4 + 1)
    print("fast=%d  sample %.3f ms (%.0f GB/s compulsory)   map_fuse %.3f ms (%.0f GB/s algorithmic, %.3f of 8 TB/s)"
          % (fast, ts, samp_bytes / ts / 1e6, tf, fuse_bytes / tf / 1e6, fuse_bytes / tf / 1e6 / 8000), flush=True)
    print("        map_accumulate of one oblique view: %.3f ms" % ta, flush=True)
_lib.check(lib.mpu_geometry_set_fast_path(1), "set_fast")
if check:
    ok = all(torch.equal(a, b_) for a, b_ in zip(res[1][0], res[0][0])) and torch.equal(res[1][1], res[0][1])
    print("fast == exact:", ok)
    if not ok:
        for vi, (a, b_) in enumerate(zip(res[1][0], res[0][0])):
            print("  view", vi, "sample mismatches", int((a != b_).sum()))
        print("  label mismatches", int((res[1][1] != res[0][1]).sum()))
        sys.exit(1)
