"""In-kernel phase stamps of conv_halo16 on predict-size layers (dev tool; needs MPU_STAMPS=1).
usage: MPU_STAMPS=1 python tools/round4/stamps16.py layer[,layer...]   (B from BENCH_B, default 138)"""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiplanarunet_amd import ops, _lib
LAYERS = {"enc1c2": (0, 128, 128, 0, 128), "enc2c2": (0, 64, 256, 0, 256), "up2c2": (0, 128, 128, 128, 128), "up1c2": (0, 64, 256, 256, 256)}
names = sys.argv[1].split(",")
B = int(os.environ.get("BENCH_B", "138")); dt = torch.bfloat16
lib = _lib.load()
buf = (C.c_uint64 * 512)()
for name in names:
    mode, H, C0, C1, Cout = LAYERS[name]
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, device="cuda").to(dt)
    x1 = torch.randn(B, H, H, C1, device="cuda").to(dt) if C1 else None
    w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    bias = torch.zeros(Cout, device="cuda")
    wp, _ = ops.pack_weights(w, mode, dt)
    run = lambda: ops.conv2d(mode, x0, wp, Cout, (H, H), bias=bias, x1=x1, relu=1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    lib.mpu_debug_stamps_read(buf, 512)
    lib.mpu_profile_enable(1)
    run()
    ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
    lib.mpu_profile_summary(0, C.byref(ms), C.byref(fl), C.byref(n))
    lib.mpu_profile_enable(0)
    lib.mpu_debug_stamps_read(buf, 512)
    s = np.array(buf[:], dtype=np.uint64).reshape(32, 16).astype(np.int64)
    s = s[s[:, 0] > 0]
    if not len(s):
        print(name, "no stamps"); continue
    print("%s: kernel %.1f us (events); %d stamped WGs (first round); taps %d" % (name, ms.value * 1e3, len(s), s[0, 6]))
    def row(label, d):
        print("   %-44s mean %8.0f  min %8.0f  max %8.0f" % (label, d.mean(), d.min(), d.max()))
    row("entry -> prologue landed (1-0)", s[:, 1] - s[:, 0])
    row("main loop (2-1)", s[:, 2] - s[:, 1])
    row("   per interval (32 MFMAs per wave)", (s[:, 2] - s[:, 1]) / s[:, 6])
    row("accumulators -> staging (3-2)", s[:, 3] - s[:, 2])
    row("stores issued (4-3)", s[:, 4] - s[:, 3])
    row("stores drained (5-4)", s[:, 5] - s[:, 4])
    row("whole workgroup (5-0)", s[:, 5] - s[:, 0])
    rt = (s[:, 15] - s[:, 14]).astype(float)
    print("   shader clock from s_memtime / s_memrealtime (100 MHz): %.0f MHz; workgroup life %.1f us; entry spread of the window %.1f us"
          % ((s[:, 5] - s[:, 0]).mean() / rt.mean() * 100.0, rt.mean() / 100.0, (s[:, 14].max() - s[:, 14].min()) / 100.0))
    row("L(2): reads issued (13-8)", s[:, 13] - s[:, 8])
    row("L(2): weight request + vmcnt wait (9-13)", s[:, 9] - s[:, 13])
    row("L(2): lgkmcnt wait (10-9)", s[:, 10] - s[:, 9])
    row("L(2): barrier (11-10)", s[:, 11] - s[:, 10])
    row("C(2): 32 MFMAs + 4 reads (7-11)", s[:, 7] - s[:, 11])
    row("C(2): barrier (12-7)", s[:, 12] - s[:, 7])
    row("tap period L(2) start -> L(3) start (12-8)", s[:, 12] - s[:, 8])
