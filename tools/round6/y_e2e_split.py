"""R6y: where the mp-train loop loses against the bare step: (1) bare graphed step, (2) the pipeline with a STUB producer (fixed
tensors, no GPU work: copies + replay + host only), (3) the real sampler; host time inside the producer and inside its reads.
Dev tool (round 6)."""
import time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiplanarunet_amd.unet import UNet
from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
from multiplanarunet_amd.pipeline import TrainPipeline
dev = torch.device("cuda:0"); B, dim = 16, 128
quiet = lambda *a, **k: None
m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16", logger=quiet, seed=0, device=dev)
m.compile("Adam", "SparseCategoricalCrossentropy")
img, lab, aff = make_toy_volume(128, 77)
vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
views = random_views(6, 60.0, 0)
mk = lambda seed: TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=seed)
x, y, w = mk(1)()
def timed(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
rep = m.make_graphed_train_step(x, y, w)
def bare(n):
    for _ in range(n): rep()
bare(20); print("bare graphed step      %.3f ms" % timed(bare, 200))
class Stub:
    batch_size, dim, volumes = B, dim, [vol]
    def __call__(self): return x, y, w
for name, s in (("stub producer", Stub()), ("real sampler", mk(7)), ("real sampler", mk(8)), ("real sampler", mk(9)), ("real sampler", mk(10))):
    p = TrainPipeline(m, s)
    p.run_epoch(60)
    print("   calibration windows (ms per step):", p.side_loop_ms)
    t_prod = [0.0]
    orig = p._produce
    def prod():
        t0 = time.perf_counter(); r = orig(); t_prod[0] += time.perf_counter() - t0; return r
    p._produce = prod
    ms = timed(lambda n: p.run_epoch(n), 120)
    print("%-22s %.3f ms per step; host inside the producer %.3f ms per step; stream latency %.0f us" % (name, ms, t_prod[0] / 120 * 1e3, p.side_latency_us))
# the sampler alone, host time split
s = mk(9); s(); s()
t = timed(lambda n: [s() for _ in range(n)], 30); print("sampler alone          %.3f ms per batch, rounds %s" % (t, s.rounds))
