import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_unet import rand_weights, CFGS, quiet
from multiplanarunet_amd.unet import UNet
from oracle import unet_ref as U
import os
for cfg in CFGS[:2]:
    K, C, D, cf, H, W, B = cfg
    w = rand_weights(U, K, C, D, cf, seed=5)
    rng = np.random.RandomState(1)
    x = rng.randn(B, H, W, C).astype(np.float32)
    y = rng.randint(0, K, (B, H * W, 1)).astype(np.uint8)
    sw = np.array([1.0, 0.33, 1.0][:B], np.float32)
    ref = U.train_step(w, x, y, sw, depth=D, dtype=torch.float64)
    for dt in ("f32", "bf16x3"):
        m = UNet(n_classes=K, img_rows=H, img_cols=W, n_channels=C, depth=D, complexity_factor=cf, dtype=dt, logger=quiet, flatten_output=True)
        m.set_weights_dict(w)
        probs, loss = m.forward_backward(x, y, sw)
        pe = np.abs(probs.cpu().numpy() - ref["probs"]).max()
        g = m.grads.cpu().numpy()
        out = []
        for name, gr in ref["grads"].items():
            kind, off, ps, ls = m._tensors[name]
            a = m._from_stored(name, g[off:off + int(np.prod(ps))].reshape(ps), ps, ls)
            out.append("%s=%.0e" % (name.replace("upsample", "up").replace("encoder", "enc").replace("/kernel", "/k").replace("/bias", "/b").replace("/gamma", "/g").replace("/beta", "/be"), np.abs(a - gr).max() / (np.abs(gr).max() + 1e-12)))
        print(cfg, dt, "probs %.1e" % pe, " ".join(out))
