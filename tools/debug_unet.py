import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from multiplanarunet_amd.unet import UNet
from oracle import unet_ref as U
from test_gpu_unet import rand_weights
quiet = lambda *a, **k: None
def run(K, C, D, cf, H, W, B, dtype):
    w = rand_weights(U, K, C, D, cf, seed=9)
    rng = np.random.RandomState(2)
    x = rng.randn(B, H, W, C).astype(np.float32)
    y = rng.randint(0, K, (B, H * W, 1)).astype(np.uint8)
    m = UNet(n_classes=K, dim=H, n_channels=C, depth=D, complexity_factor=cf, dtype=dtype, logger=quiet)
    m.set_weights_dict(w)
    r = U.train_step(w, x, y, np.ones(B, np.float32), depth=D)
    m.forward_backward(x, y, None)
    g = m.grads.cpu().numpy()
    print("==", (K, C, D, cf, H, W, B, dtype))
    for name in m._keras_order():
        if "moving" in name: continue
        kind, off, ps, ls = m._tensors[name]
        a = m._from_stored(name, g[off:off + int(np.prod(ps))].reshape(ps), ps, ls)
        gr = r["grads"][name]
        rel = np.abs(a - gr).max() / (np.abs(gr).max() + 1e-12)
        cos = (a * gr).sum() / (np.linalg.norm(a) * np.linalg.norm(gr) + 1e-30)
        print("%-28s rel-max-err %.4f cos %.5f  |g| %.3e" % (name, rel, cos, np.abs(gr).max()), flush=True)
run(3, 1, 3, 0.25, 64, 64, 4, "bf16")
