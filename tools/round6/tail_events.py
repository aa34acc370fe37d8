"""Timeline of the step's tail from the library's own timing events (mpu_debug_tail_events), eager and graph-replayed. Dev tool."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from multiplanarunet_amd import _lib
from multiplanarunet_amd.unet import UNet
quiet = lambda *a, **k: None
m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype="bf16", logger=quiet, seed=0)
m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs=dict(lr=5e-5))
rng = np.random.RandomState(0)
x = torch.tensor(rng.randn(16, 128, 128, 1).astype(np.float32), device="cuda")
y = torch.tensor(rng.randint(0, 3, (16, 128 * 128, 1)).astype(np.uint8), device="cuda")
lib = _lib.load()
names = ["t0 before glds", "fork (glds+reduce done)", "side: adam starts", "side: adam done", "taps done", "taps reduce done", "rest adam done", "joined"]
def show(tag, out):
    print(tag, " | ".join("%s %.1f" % (n.split(":")[-1].strip(), 1e3 * v) for n, v in zip(names, out)))
for _ in range(3):
    m.train_step(x, y, None, want_loss=False)
for rep in range(3):
    lib.mpu_debug_tail_events(1, None)
    m.train_step(x, y, None, want_loss=False)
    out = (C.c_float * 8)()
    lib.mpu_debug_tail_events(0, out)
    show("eager ", list(out))
lib.mpu_debug_tail_events(1, None)
replay = m.make_graphed_train_step(x, y)
for rep in range(3):
    for _ in range(5):
        replay()
    out = (C.c_float * 8)()
    torch.cuda.synchronize()
    lib.mpu_debug_tail_events(1, None)        # (stays armed: the events are nodes of the graph)
    import time
    t0 = time.perf_counter(); [replay() for _ in range(20)]; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    for k in range(8):
        pass
    # read the events of the LAST replay
    lib.mpu_debug_tail_events(0, out)
    show("graph ", list(out)); print("   ms per step over 20 replays: %.4f" % (dt * 1e3))
    lib.mpu_debug_tail_events(1, None)
