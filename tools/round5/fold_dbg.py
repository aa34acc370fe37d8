"""dev: first divergence between the folded and the separate BatchNorm launches (one forward / one backward at configs[1])"""
import os, subprocess, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from multiplanarunet_amd.unet import UNet
quiet = lambda *a, **k: None
B, K, D = 16, 3, 4
m = UNet(n_classes=K, dim=128, n_channels=1, depth=D, complexity_factor=1, dtype="bf16", logger=quiet, flatten_output=True, seed=5)
rng = np.random.RandomState(7)
x = torch.tensor(rng.randn(B, 128, 128, 1).astype(np.float32), device="cuda")
y = torch.tensor(rng.randint(0, K, (B, 128 * 128, 1)).astype(np.uint8), device="cuda")
sw = torch.ones(B, device="cuda")
probs, loss = m.forward_backward(x, y, sw)
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([m.bn_state.cpu().numpy(), probs.float().cpu().numpy().ravel(), m.grads.cpu().numpy()]))
""" % R
out = {}
for tag, env in (("fold", {}), ("separate", {"MPU_BN_FOLD": "0"})):
    f = "/tmp/fold1_%s.npy" % tag
    r = subprocess.run([sys.executable, "-c", SCRIPT, f], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:])
    out[tag] = np.load(f)
sys.path.insert(0, R)
from multiplanarunet_amd.unet import UNet
m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype="bf16", logger=lambda *a, **k: None, flatten_output=True, seed=5)
a, b = out["fold"], out["separate"]
nst = m.bn_state.numel(); npr = 16 * 128 * 128 * 3
d = a != b
print("bn_state", d[:nst].sum(), "probs", d[nst:nst + npr].sum(), "grads", d[nst + npr:].sum())
for nm in m._order:
    kind, off, ps, ls = m._tensors[nm]
    n = int(np.prod(ps))
    if kind == 1:
        c = d[off:off + n].sum()
        print("%-34s %5d / %5d  max|d| %.3e  rel %.3e" % (nm, c, n, np.abs(a[off:off+n] - b[off:off+n]).max(), (np.abs(a[off:off+n] - b[off:off+n]) / (np.abs(b[off:off+n]) + 1e-30)).max()))
