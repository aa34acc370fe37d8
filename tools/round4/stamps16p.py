"""In-kernel stamps of conv_halo16p (persistent) on predict-size layers (dev tool; needs MPU_STAMPS=1 MPU_HALO16=1).
usage: MPU_STAMPS=1 MPU_HALO16=1 python tools/round4/stamps16p.py layer[,layer...]   (B from BENCH_B, default 138)"""
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
    print("%s: kernel %.1f us (events); %d stamped WGs" % (name, ms.value * 1e3, len(s)))
    def row(label, d):
        print("   %-52s mean %8.0f  min %8.0f  max %8.0f" % (label, d.mean(), d.min(), d.max()))
    row("entry -> prologue landed (1-0)", s[:, 1] - s[:, 0])
    row("tile 0: main loop (2-1)", s[:, 2] - s[:, 1])
    for k in range(6):
        row("tile %d: epilogue of wave 0 (%d-%d)" % (k, 3 + 2 * k, 2 + 2 * k), s[:, 3 + 2 * k] - s[:, 2 + 2 * k])
        if k < 5:
            row("tile %d: main loop (%d-%d)" % (k + 1, 4 + 2 * k, 3 + 2 * k), s[:, 4 + 2 * k] - s[:, 3 + 2 * k])
    row("tile period (C(8) to C(8), tiles 1..5)", (s[:, 12] - s[:, 2]) / 5.0)
    rt = (s[:, 15] - s[:, 14]).astype(float)
    print("   workgroup life %.1f us" % (rt.mean() / 100.0))
