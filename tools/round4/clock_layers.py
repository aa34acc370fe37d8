"""Shader clock while ONE conv layer runs back to back (dev tool): which part of a kernel's activity costs the power budget?
usage: [MPU_LIB_PATH=..knock.so MPU_HALO_KNOCKOUT=mask | MPU_HALO16=1] python tools/round4/clock_layers.py layer[,layer]"""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiplanarunet_amd import ops, _lib
LAYERS = {"enc1c2": (0, 128, 128, 0, 128), "enc2c2": (0, 64, 256, 0, 256), "up2c2": (0, 128, 128, 128, 128)}
B = int(os.environ.get("BENCH_B", "138")); dt = torch.bfloat16
ZERO = os.environ.get("ZERO_DATA") == "1"
lib = _lib.load()
for name in sys.argv[1].split(","):
    mode, H, C0, C1, Cout = LAYERS[name]
    Cin = C0 + C1
    mk = (lambda *s: torch.zeros(*s, device="cuda")) if ZERO else (lambda *s: torch.randn(*s, device="cuda"))
    x0 = mk(B, H, H, C0).to(dt)
    x1 = mk(B, H, H, C1).to(dt) if C1 else None
    w = mk(3, 3, Cin, Cout) * 0.05
    bias = torch.zeros(Cout, device="cuda")
    wp, _ = ops.pack_weights(w, mode, dt)
    out = torch.empty(B, H, H, Cout, device="cuda", dtype=dt)
    run = lambda: ops.conv2d(mode, x0, wp, Cout, (H, H), bias=bias, x1=x1, relu=1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    n, naps = 300, 6
    buf = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        _lib.check(lib.mpu_probe_clock(_lib.ptr(buf), n, naps, _lib.stream_ptr()), "clock")
    reps = 30
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(n, 2)
    lo, hi = n // 8, n - 1
    mhz = float(s[hi, 0] - s[lo, 0]) / float(s[hi, 1] - s[lo, 1]) * 100.0
    span_ms = float(s[hi, 1] - s[0, 1]) / 1e5
    print("%-7s %8.1f us per launch   shader clock %5.0f MHz   (sampler span %.1f ms of %.1f ms)" %
          (name, e0.elapsed_time(e1) * 1e3 / reps, mhz, span_ms, e0.elapsed_time(e1)), flush=True)
