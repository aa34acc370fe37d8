"""In-kernel phase stamps of conv_halo8 / wgrad_taps on the cfg1 layer shapes (dev tool; needs MPU_STAMPS=1).
usage: MPU_STAMPS=1 python tools/stamps.py [fwd|wgrad] layer[,layer...]"""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiplanarunet_amd import ops, _lib
LAYERS = {"enc0c2": (0, 128, 64, 0, 64), "enc1c1": (0, 64, 64, 0, 128), "enc1c2": (0, 64, 128, 0, 128), "enc2c1": (0, 32, 128, 0, 256),
          "enc2c2": (0, 32, 256, 0, 256), "up1c2": (0, 32, 256, 256, 256), "up2c2": (0, 64, 128, 128, 128),
          "up3c2": (0, 128, 64, 64, 64), "up2c1": (1, 64, 256, 0, 128)}
what = sys.argv[1]; names = sys.argv[2].split(",")
B = 16; dt = torch.bfloat16
lib = _lib.load()
buf = (C.c_uint64 * 512)()
for name in names:
    mode, H, C0, C1, Cout = LAYERS[name]
    k = 2 if mode == 1 else 3
    Hi = H // 2 if mode == 1 else H
    Cin = C0 + C1
    x0 = torch.randn(B, Hi, Hi, C0, device="cuda").to(dt)
    x1 = torch.randn(B, Hi, Hi, C1, device="cuda").to(dt) if C1 else None
    w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
    bias = torch.zeros(Cout, device="cuda")
    wp, _ = ops.pack_weights(w, mode, dt)
    dz = torch.randn(B, H, H, Cout, device="cuda").to(dt)
    ws = torch.empty(8 * B * H * H * Cout, dtype=torch.float32, device="cuda")
    run = (lambda: ops.conv2d(mode, x0, wp, Cout, (H, H), bias=bias, x1=x1, relu=1, workspace=ws)) if what == "fwd" \
        else (lambda: ops.conv2d_wgrad(mode, x0, dz, x1=x1))
    for _ in range(5): run()
    torch.cuda.synchronize()
    lib.mpu_debug_stamps_read(buf, 512)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.mpu_profile_enable(1)
    run()
    ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
    lib.mpu_profile_summary(0 if what == "fwd" else 1, C.byref(ms), C.byref(fl), C.byref(n))
    lib.mpu_profile_enable(0)
    lib.mpu_debug_stamps_read(buf, 512)
    s = np.array(buf[:], dtype=np.uint64).reshape(32, 16).astype(np.int64)
    s = s[s[:, 0] > 0]
    if not len(s):
        print(name, "no stamps (schedule not instrumented?)"); continue
    s = s[s[:, 0] >= s[:, 0].max() - 200000]                    # (drop stale rows of earlier, larger grids)
    t0 = s[:, 0].min()
    span = s[:, 5].max() - t0
    print("%s %s: kernel %.1f us (events); %d stamped WGs; first entry -> last done = %d ticks; steps %d" %
          (name, what, ms.value * 1e3, len(s), span, s[0, 6]))
    d = np.stack([s[:, 0] - t0, s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]], 1)
    lab = ["entry-skew", "prologue", "main-loop", "epi-A", "epi-B(stores issued)", "store-drain"]
    for j, l in enumerate(lab):
        print("   %-22s mean %8.0f  min %8.0f  max %8.0f ticks" % (l, d[:, j].mean(), d[:, j].min(), d[:, j].max()))
    if s[:, 8].min() > 0:       # inside the loop, step 4: compute part, vmcnt wait, lgkmcnt wait, barrier; step period
        e = np.stack([s[:, 9] - s[:, 8], s[:, 10] - s[:, 9], s[:, 11] - s[:, 10], s[:, 12] - s[:, 8]], 1)
        for j, l in enumerate(["step4 vmcnt wait", "step4 lgkmcnt wait", "step4 barrier", "step period (4->5)"]):
            print("   %-22s mean %8.0f  min %8.0f  max %8.0f ticks" % (l, e[:, j].mean(), e[:, j].min(), e[:, j].max()))
        if s[:, 7].min() > 0:   # MPU_HALO8_SCHED=1: stamp 8 = start of the load phase, 11 = start of the compute phase, 7 = its end
            c = s[:, 7] - s[:, 11]
            print("   %-22s mean %8.0f  min %8.0f  max %8.0f ticks" % ("step4 compute phase", c.mean(), c.min(), c.max()))
