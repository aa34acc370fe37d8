"""
The measurement probes behind bench.py's `measured_peaks` (csrc/probe.hip) move the bytes they are priced on: a probe that
skipped or repeated part of its array would report a bandwidth nobody can reach. Byte / integer work: exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from multiplanarunet_amd import _lib
    return _lib, _lib.load()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_stream_copy_copies_every_element(variant):
    L, lib = _lib()
    n = 4096 * 37
    src = torch.arange(n, device="cuda", dtype=torch.float32)
    dst = torch.full((n,), -1.0, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.mpu_probe_stream_copy(L.ptr(dst), L.ptr(src), n, variant, st), "mpu_probe_stream_copy")
    torch.cuda.synchronize()
    assert torch.equal(dst, src)


@pytest.mark.parametrize("run_bytes", [128, 256, 1024, 4096, 1 << 30])
def test_permuted_read_visits_every_float_exactly_once(run_bytes):
    """The permutation of the runs is a bijection: the per-thread sums add up to the sum of the source for two unrelated integer
    patterns (small integers: every partial sum is exact in f32), and a thread's sum is a sum over whole source runs."""
    L, lib = _lib()
    n = 1 << 18
    st = torch.cuda.current_stream().cuda_stream
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    for pat in (idx % 251, (idx * idx + 3 * idx) % 241):
        src = pat.to(torch.float32)
        out = torch.full((n // 32,), -1.0, device="cuda")
        L.check(lib.mpu_probe_permuted_read(L.ptr(src), L.ptr(out), n, run_bytes, st), "mpu_probe_permuted_read")
        torch.cuda.synchronize()
        assert int(out.double().sum().item()) == int(pat.sum().item())
    # one marked float: exactly one thread sum sees it, wherever its run lands
    src = torch.zeros(n, device="cuda")
    src[123457] = 5.0
    out = torch.empty(n // 32, device="cuda")
    L.check(lib.mpu_probe_permuted_read(L.ptr(src), L.ptr(out), n, run_bytes, st), "mpu_probe_permuted_read")
    torch.cuda.synchronize()
    assert int((out != 0).sum().item()) == 1 and float(out.sum().item()) == 5.0


def test_permuted_read_rejects_sizes_it_cannot_permute():
    L, lib = _lib()
    src = torch.zeros(1 << 14, device="cuda")
    out = torch.zeros(1 << 9, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mpu_probe_permuted_read(L.ptr(src), L.ptr(out), 12288, 128, st) != 0       # not a power of two
    assert lib.mpu_probe_permuted_read(L.ptr(src), L.ptr(out), 1 << 14, 96, st) != 0      # run not a power of two
