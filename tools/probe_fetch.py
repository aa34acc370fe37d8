"""FETCH_SIZE calibration (dev tool; run under rocprofv3 --pmc FETCH_SIZE): a 16-byte streaming read of known size
(stream triad) next to a 12-byte-per-lane read of known size (mpu_probe_gather12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiplanarunet_amd import _lib
lib = _lib.load()
n = 1 << 27                                         # 128 M outputs: 1.6 GB read by the gather probe, 1 GiB per triad array
x = torch.ones(3 * n, device="cuda"); out = torch.empty(n, device="cuda")
a = torch.empty(1 << 28, device="cuda"); b = torch.ones(1 << 28, device="cuda"); c = torch.ones(1 << 28, device="cuda")
st = _lib.stream_ptr()
for _ in range(3):
    lib.mpu_probe_gather12(_lib.ptr(x), _lib.ptr(out), n, st)
    lib.mpu_probe_stream_triad(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), 1 << 28, st)
torch.cuda.synchronize()
print("gather12 reads %d bytes per launch; triad reads %d bytes per launch" % (12 * n, 8 * (1 << 28)))
