"""The deep-level 3x3 layers of configs[1] (16 slices of 128 x 128: 16 x 16 and 8 x 8 maps), forward launches through the C-ABI, each
REPS times in a row -- run under rocprofv3 --kernel-trace (tools/round5/deep_trace.py groups the dispatches per layer) and, with
MPU_STAMPS=1, print conv_deep's in-kernel phase stamps. usage: deep_layers.py [stamps]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiplanarunet_amd import ops, _lib

LAYERS = [("L3c1", 16, 16, 256, 0, 512), ("L3c2", 16, 16, 512, 0, 512), ("botc1", 16, 8, 512, 0, 1024),
          ("botc2", 16, 8, 1024, 0, 1024), ("up3c2", 16, 16, 512, 512, 512), ("up3c3", 16, 16, 512, 0, 512)]
REPS = 12
lib = _lib.load()
dt = torch.bfloat16
buf = (C.c_uint64 * 512)()
for name, B, H, C0, C1, Cout in LAYERS:
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, device="cuda").to(dt)
    x1 = torch.randn(B, H, H, C1, device="cuda").to(dt) if C1 else None
    w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.02
    bias = torch.zeros(Cout, device="cuda")
    wp, _ = ops.pack_weights(w, 0, dt)
    ws = torch.empty(8 * B * H * H * Cout, dtype=torch.float32, device="cuda")
    run = lambda: ops.conv2d(0, x0, wp, Cout, (H, H), bias=bias, x1=x1, relu=1, workspace=ws)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        run()
    e1.record(); torch.cuda.synchronize()
    print("%s B=%d %dx%d %d+%d->%d: %.2f us per launch pair (events, back to back)" % (name, B, H, H, C0, C1, Cout, e0.elapsed_time(e1) * 1e3 / REPS))
    if len(sys.argv) > 1 and sys.argv[1] == "stamps":
        lib.mpu_debug_stamps_read(buf, 512)
        run(); torch.cuda.synchronize()
        lib.mpu_debug_stamps_read(buf, 512)
        s = np.array(buf[:], dtype=np.uint64).reshape(32, 16).astype(np.int64)
        s = s[s[:, 0] > 0]
        if not len(s):
            print("   no stamps"); continue
        t0 = s[:, 0].min()
        d = np.stack([s[:, 0] - t0, s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]], 1)
        print("   %d stamped workgroups, %d intervals each; first entry -> last drained %d cycles" % (len(s), s[0, 6], s[:, 5].max() - t0))
        for j, l in enumerate(["entry-skew", "prologue", "main-loop", "staging", "stores issued", "store-drain"]):
            print("   %-16s mean %8.0f  min %8.0f  max %8.0f cycles" % (l, d[:, j].mean(), d[:, j].min(), d[:, j].max()))
        if (s[:, 13] > 0).all():
            f = np.stack([s[:, 13] - s[:, 2], s[:, 14] - s[:, 13], s[:, 3] - s[:, 14]], 1)
            for j, l in enumerate(["  sync wait", "  lds writes", "  reads issued"]):
                print("   %-16s mean %8.0f  min %8.0f  max %8.0f cycles" % (l, f[:, j].mean(), f[:, j].min(), f[:, j].max()))
            w0, w7 = s[:16], s[16:]
            if len(w0) == len(w7) == 16:
                print("   wave 7 - wave 0 at: loop end %+.0f, after sync %+.0f, stores issued %+.0f, drained %+.0f" % (
                    (w7[:, 2] - w0[:, 2]).mean(), (w7[:, 13] - w0[:, 13]).mean(), (w7[:, 4] - w0[:, 4]).mean(), (w7[:, 5] - w0[:, 5]).mean()))
        print("   main loop per interval: %.0f cycles" % (d[:, 2].mean() / s[0, 6]))
        ok = s[:, 8] > 0
        if ok.any():
            e = np.stack([s[ok, 9] - s[ok, 8], s[ok, 10] - s[ok, 9], s[ok, 11] - s[ok, 10], s[ok, 12] - s[ok, 11], s[ok, 11] - s[ok, 7]], 1)
            for j, l in enumerate(["iv4 lgkm wait", "iv4 vmcnt wait", "iv4 barrier", "barrier4 -> barrier5", "barrier3 -> barrier4"]):
                print("   %-20s mean %8.0f  min %8.0f  max %8.0f cycles" % (l, e[:, j].mean(), e[:, j].min(), e[:, j].max()))
