"""Micro-benchmark of the conv kernels on the cfg2 layer shapes (dev tool). usage: bench_conv.py [fwd|wgrad] [reps]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiplanarunet_amd import ops
CONV3, UPCONV2, CONV3S2, CONV1 = 0, 1, 2, 3
B = int(os.environ.get("BENCH_B", "16"))
SCALE = int(os.environ.get("BENCH_SCALE", "1"))          # multiplies every layer's H (cfg4 / predict shapes)
ONLY = os.environ.get("BENCH_ONLY")                      # comma-separated layer names
# name, mode, H(out), C0, C1, Cout
LAYERS = [("enc0c1", 0, 128, 8, 0, 64), ("enc0c2", 0, 128, 64, 0, 64), ("enc1c1", 0, 64, 64, 0, 128), ("enc1c2", 0, 64, 128, 0, 128),
          ("enc2c1", 0, 32, 128, 0, 256), ("enc2c2", 0, 32, 256, 0, 256), ("enc3c1", 0, 16, 256, 0, 512), ("enc3c2", 0, 16, 512, 0, 512),
          ("botc1", 0, 8, 512, 0, 1024), ("botc2", 0, 8, 1024, 0, 1024),
          ("up0c1", 1, 16, 1024, 0, 512), ("up0c2", 0, 16, 512, 512, 512), ("up0c3", 0, 16, 512, 0, 512),
          ("up1c2", 0, 32, 256, 256, 256), ("up2c2", 0, 64, 128, 128, 128), ("up3c1", 1, 128, 128, 0, 64), ("up3c2", 0, 128, 64, 64, 64),
          ("dg_up3", 2, 64, 64, 0, 128), ("dg_up0", 2, 8, 512, 0, 1024), ("dg_botc1", 0, 8, 1024, 0, 512)]
import os
RELU = int(os.environ.get("RELU_BITS", "1"))
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.bfloat16
tot_t = tot_f = 0
for name, mode, H, C0, C1, Cout in LAYERS:
    if ONLY and name not in ONLY.split(","):
        continue
    H *= SCALE
    k = {0: 3, 1: 2, 2: 3, 3: 1}[mode]
    Hi = H // 2 if mode == 1 else (2 * H if mode == 2 else H)
    Cin = C0 + C1
    x0 = torch.randn(B, Hi, Hi, C0, device="cuda").to(dt)
    x1 = torch.randn(B, Hi, Hi, C1, device="cuda").to(dt) if C1 else None
    w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
    bias = torch.zeros(Cout, device="cuda")
    if mode == 2:
        wp = torch.randn(9 * Cin * Cout, device="cuda").to(dt)
    else:
        wp, _ = ops.pack_weights(w, mode, dt)
    dz = torch.randn(B, H, H, Cout, device="cuda").to(dt)
    ws = torch.empty(8 * B * H * H * Cout, dtype=torch.float32, device="cuda")     # split-K partials (as the U-Net passes)
    def run():
        if what == "fwd":
            return ops.conv2d(mode, x0, wp, Cout, (H, H), bias=bias, x1=x1, relu=RELU, workspace=ws)
        return ops.conv2d_wgrad(mode, x0, dz, x1=x1)
    if what == "wgrad" and mode == 2:
        continue
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    lib.mpu_profile_enable(1)          # per-launch HIP events: pure kernel time (the python loop is launch-bound)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms, fl_, n_ = C.c_double(), C.c_double(), C.c_int64()
    lib.mpu_profile_summary(0 if what == "fwd" else 1, C.byref(ms), C.byref(fl_), C.byref(n_))
    lib.mpu_profile_enable(0)
    wall = e0.elapsed_time(e1) * 1e3 / reps
    us = ms.value * 1e3 / max(n_.value, 1) * (n_.value / reps)
    taps = {0: 9, 1: 4, 2: 9, 3: 1}[mode]
    fl = 2.0 * B * H * H * Cout * taps * Cin
    tot_t += us; tot_f += fl
    print("%-8s M=%7d N=%5d K=%6d  %8.1f us  %7.1f TF/s   (loop wall %.1f us)" % (name, B * H * H, Cout, taps * Cin, us, fl / us / 1e6, wall), flush=True)
print("total %.1f us  %.1f TF/s" % (tot_t, tot_f / tot_t / 1e6))
