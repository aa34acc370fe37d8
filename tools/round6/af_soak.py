"""R6af: soak of the round-6 step on the REAL network (configs[1]: depth 4 / 64 filters / 16 x 128^2 -- the shapes that take the
overlapped tail, the BatchNorm accumulators and the grouped weight gradients): the overlapped, graphed `mp train` pipeline (producer on
a side stream, optimizer branch of the captured graph) against the serial eager loop, 600 steps, parameters / BatchNorm state / Adam
moments compared bit for bit every 60 steps, a learning-rate change (re-capture) every 120. A race between the two branches of the
tail, or between the producer stream and the step, shows up as a difference. Dev tool (round 6)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.pipeline import TrainPipeline
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0"); B, dim = 16, 128
quiet = lambda *a, **k: None
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
views = random_views(6, 60.0, 0)
def mk():
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype=dtype, logger=quiet, seed=0, device=dev)
    m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-4})
    s = TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=3)
    return m, s
m0, s0 = mk(); m1, s1 = mk()
p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
kw = {}
if len(sys.argv) > 2:                                          # e.g. "graphed=1,overlap=0"
    kw = {k: bool(int(v)) for k, v in (t.split("=") for t in sys.argv[2].split(","))}
p1 = TrainPipeline(m1, s1, **kw)
ok = True
for ep in range(10):
    a, b = p0.run_epoch(60), p1.run_epoch(60)
    torch.cuda.synchronize()
    same = (a == b) and torch.equal(m0.params, m1.params) and torch.equal(m0.bn_state, m1.bn_state) \
        and torch.equal(m0._adam_m, m1._adam_m) and torch.equal(m0._adam_v, m1._adam_v) \
        and torch.equal(m0.packed.view(torch.uint8), m1.packed.view(torch.uint8))
    print(ep, "%.6f %.6f" % (a, b), "EQ" if same else "DIFF", "producer windows %s" % p1.side_loop_ms if ep == 0 else "", flush=True)
    ok = ok and same
    if ep % 2 == 1:
        for m in (m0, m1):
            m.optimizer_kwargs["lr"] *= 0.9
import hashlib
h = hashlib.sha256(m1.params.cpu().numpy().tobytes() + m1.bn_state.cpu().numpy().tobytes()).hexdigest()[:16]
print("SOAK", dtype, kw, os.environ.get("MPU_BN_ATOMIC", ""), "OK" if ok else "FAILED", "parameters after 600 steps: sha256", h)
