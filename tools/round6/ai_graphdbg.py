"""R6ai: the graphed bf16x3 step adds a STALE loss for stretches of replays (parameters stay right): is the per-pixel loss tensor
stale, or the captured reduction of it? Own capture of body(): keeps `loss` and the reduction's result, compares after every replay
with an eager reduction of the same tensor. Dev tool."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from multiplanarunet_amd.unet import UNet
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0"); B, dim = 16, 128
quiet = lambda *a, **k: None
m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype=dtype, logger=quiet, seed=0, device=dev)
m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-4})
rng = np.random.RandomState(0)
xs = [torch.tensor(rng.randn(B, dim, dim, 1).astype(np.float32), device=dev) for _ in range(8)]
ys = [torch.tensor(rng.randint(0, 3, (B, dim * dim, 1)).astype(np.uint8), device=dev) for _ in range(8)]
x = xs[0].clone(); y = ys[0].clone()
m.train_step(x, y, None)                                     # lazy buffers
m._ensure_adam()
step_dev = torch.tensor([m.iterations], dtype=torch.int64, device=dev)
loss_sum = torch.zeros(1, dtype=torch.float64, device=dev)
keep = {}
def body():
    _, loss = m.forward_backward(x, y, None, want_loss=True, adam=(0, step_dev))
    keep["loss"] = loss
    keep["mean"] = loss.mean()
    loss_sum.add_(keep["mean"].double())
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
mode = sys.argv[2] if len(sys.argv) > 2 else "sync"
if mode == "sync":
    bad = 0
    for i in range(400):
        x.copy_(xs[i % 8]); y.copy_(ys[i % 8])
        g.replay()
        torch.cuda.synchronize()
        eager_mean = float(keep["loss"].mean().item()); cap_mean = float(keep["mean"].item())
        bad += abs(eager_mean - cap_mean) > 1e-6 * abs(eager_mean)
    print("REPLAYS with a wrong captured reduction:", bad, "of 400 (%s, a synchronisation after every replay)" % dtype)
else:
    N = 400
    hc = torch.zeros(N, device=dev); he = torch.zeros(N, device=dev); hs = torch.zeros(N, dtype=torch.float64, device=dev)
    for i in range(N):
        x.copy_(xs[i % 8]); y.copy_(ys[i % 8])
        g.replay()
        if mode == "async_eager":
            he[i:i + 1].copy_(keep["loss"].mean().reshape(1))      # eager reduction right behind the replay, same stream
        hc[i:i + 1].copy_(keep["mean"].reshape(1)); hs[i:i + 1].copy_(loss_sum)
    torch.cuda.synchronize()
    hc, he, hs = hc.cpu().numpy(), he.cpu().numpy(), hs.cpu().numpy()
    inc = np.diff(hs, prepend=0.0)
    rep = [i for i in range(1, N) if hc[i] == hc[i - 1]]
    print("mode %s: captured mean repeats its previous value at %d of %d replays (first: %s); |added - captured mean| max %.2e"
          % (mode, len(rep), N, rep[:10], np.abs(inc - hc).max()))
    if mode == "async_eager":
        d = [i for i in range(N) if abs(he[i] - hc[i]) > 1e-6 * abs(he[i])]
        print("   eager reduction of the captured loss tensor differs from the captured reduction at %d replays (first: %s)" % (len(d), d[:10]))
