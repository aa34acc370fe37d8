"""Socket power (sysfs hwmon power1_input of the busiest card) and shader clock while one workload runs back to back (dev tool).
usage: python tools/round4/power_layers.py what[,what...]   what = enc1c2 | up2c2 | enc2c2 | mfma | mfma_random | triad | step
env: MPU_LIB_PATH / MPU_HALO_KNOCKOUT / MPU_HALO16 / ZERO_DATA as for clock_layers.py"""
import sys, os, time, glob, threading, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiplanarunet_amd import ops, _lib
LAYERS = {"enc1c2": (0, 128, 128, 0, 128), "enc2c2": (0, 64, 256, 0, 256), "up2c2": (0, 128, 128, 128, 128)}
B = int(os.environ.get("BENCH_B", "138")); dt = torch.bfloat16
ZERO = os.environ.get("ZERO_DATA") == "1"
lib = _lib.load()
HW = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))


def read_power():
    out = []
    for p in HW:
        try:
            out.append(int(open(p).read()) / 1e6)
        except Exception:
            out.append(0.0)
    return out


def make(what):
    st = _lib.stream_ptr()
    if what in LAYERS:
        mode, H, C0, C1, Cout = LAYERS[what]
        mk = (lambda *s: torch.zeros(*s, device="cuda")) if ZERO else (lambda *s: torch.randn(*s, device="cuda"))
        x0 = mk(B, H, H, C0).to(dt); x1 = mk(B, H, H, C1).to(dt) if C1 else None
        w = mk(3, 3, C0 + C1, Cout) * 0.05
        bias = torch.zeros(Cout, device="cuda")
        wp, _ = ops.pack_weights(w, mode, dt)
        return lambda: ops.conv2d(mode, x0, wp, Cout, (H, H), bias=bias, x1=x1, relu=1)
    sink = torch.zeros(16, device="cuda"); fl = C.c_double()
    if what == "mfma":
        return lambda: lib.mpu_probe_mfma_bf16(1024, 2000, _lib.ptr(sink), C.byref(fl), st)
    if what == "mfma_random":
        return lambda: lib.mpu_probe_mfma_bf16_random(1024, 2000, _lib.ptr(sink), C.byref(fl), st)
    if what == "triad":
        n = 1 << 28
        a = torch.empty(n, device="cuda"); b = torch.ones(n, device="cuda"); c = torch.ones(n, device="cuda")
        return lambda: lib.mpu_probe_stream_triad(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), n, st)
    if what == "step":
        from multiplanarunet_amd.unet import UNet
        m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16", logger=lambda *a, **k: None, seed=0)
        m.compile("Adam", "SparseCategoricalCrossentropy")
        x = torch.randn(16, 128, 128, 1, device="cuda"); y = torch.randint(0, 3, (16, 128 * 128, 1), device="cuda", dtype=torch.uint8)
        w = torch.ones(16, device="cuda")
        for _ in range(3): m.train_step(x, y, w, want_loss=False)
        return m.make_graphed_train_step(x, y, w)
    raise SystemExit("unknown workload " + what)


idle = read_power()
for what in sys.argv[1].split(","):
    run = make(what)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    one_ms = e0.elapsed_time(e1)
    reps = max(8, int(1500.0 / max(one_ms, 1e-3)))                 # ~1.5 s of back-to-back launches
    n, naps = 300, max(1, int(1.2e9 / 300 / 8192 / 2))              # clock sampler spanning ~0.6 s at 2 GHz
    buf = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    samples = []
    stop = threading.Event()
    def poll():
        while not stop.is_set():
            samples.append(read_power()); time.sleep(0.02)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps // 4): run()
    with torch.cuda.stream(side):
        _lib.check(lib.mpu_probe_clock(_lib.ptr(buf), n, naps, _lib.stream_ptr()), "clock")
    th = threading.Thread(target=poll); th.start()
    for _ in range(reps - reps // 4): run()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    s = buf.cpu().numpy().reshape(n, 2)
    mhz = float(s[-1, 0] - s[n // 8, 0]) / float(s[-1, 1] - s[n // 8, 1]) * 100.0
    P = np.array(samples[2:-2]) if len(samples) > 6 else np.array(samples)
    card = int(np.argmax(P.mean(0)))
    pw = P[:, card]
    per = e0.elapsed_time(e1) / reps
    print("%-12s %9.1f us per launch  clock %5.0f MHz  power mean %6.0f W  max %6.0f W  (idle %4.0f W, %d samples, card %d)  energy %.3f J per launch"
          % (what, per * 1e3, mhz, pw.mean(), pw.max(), idle[card], len(pw), card, pw.mean() * per * 1e-3), flush=True)
