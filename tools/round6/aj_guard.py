"""R6aj: guard bands around the workspace of the graphed bf16x3 pipeline (the one whose epoch loss goes stale for stretches): does a
kernel write outside the planned workspace? 256 MB of 0xA5 before and behind it, checked after every epoch. Dev tool."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from multiplanarunet_amd import _lib
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.pipeline import TrainPipeline
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0"); B, dim = 16, 128
quiet = lambda *a, **k: None
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
views = random_views(6, 60.0, 0)
G = 256 << 20
def mk(guard):
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype=dtype, logger=quiet, seed=0, device=dev)
    m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-4})
    if guard:
        n = _lib.load().mpu_unet_workspace_bytes(m._h, B)
        full = torch.full((n + 2 * G,), 0xA5, dtype=torch.uint8, device=dev)
        m._ws_full, m._ws, m._ws_batch, m._ws_n = full, full[G:G + n], B, n
        for name in ("grads", "packed", "params"):                 # the other buffers kernels write: copies with guard bands
            t = getattr(m, name)
            nb = t.numel() * t.element_size()
            f2 = torch.full((nb + 2 * (G // 8),), 0xA5, dtype=torch.uint8, device=dev)
            f2[G // 8:G // 8 + nb].copy_(t.reshape(-1).view(torch.uint8))
            setattr(m, "_g_" + name, f2)
            setattr(m, name, f2[G // 8:G // 8 + nb].view(t.dtype).reshape(t.shape))
    return m, TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=3)
def check(m):
    out = []
    f, n = m._ws_full, m._ws_n
    for nm, reg in (("before ws", f[:G]), ("behind ws", f[G + n:])):
        bad = (reg != 0xA5).nonzero()
        if bad.numel(): out.append("%s: %d bytes changed, first at %+d" % (nm, bad.numel(), int(bad[0]) - (G if nm == "before ws" else 0)))
    for name in ("grads", "packed", "params"):
        f2 = getattr(m, "_g_" + name); nb = f2.numel() - 2 * (G // 8)
        for nm, reg, base in ((name + " before", f2[:G // 8], G // 8), (name + " behind", f2[G // 8 + nb:], 0)):
            bad = (reg != 0xA5).nonzero()
            if bad.numel(): out.append("%s: %d bytes changed, first at %+d" % (nm, bad.numel(), int(bad[0]) - base))
    return out
m0, s0 = mk(False); m1, s1 = mk(True)
p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
p1 = TrainPipeline(m1, s1, overlap=False)
for ep in range(8):
    a, b = p0.run_epoch(60), p1.run_epoch(60)
    torch.cuda.synchronize()
    print(ep, "%.6f %.6f" % (a, b), "EQ" if a == b and torch.equal(m0.params, m1.params) else "DIFF", check(m1), flush=True)
