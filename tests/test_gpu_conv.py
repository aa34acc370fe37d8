"""
GPU parity of the implicit-GEMM convolution kernels (forward, data gradient,
weight gradient; f32 and bf16) against torch-CPU convolutions with Keras/TF
padding semantics. Tolerances: f32 path rtol 1e-5 (+atol 1e-5*|scale|); bf16
path compares against a reference fed the same bf16-rounded operands, so only
accumulation order and the final bf16 rounding differ: atol/rtol 1.2e-2.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CONV3, UPCONV2, CONV3S2, CONV1 = 0, 1, 2, 3


def ref_forward(mode, x, w, b=None):
    """x [B,H,W,C] f64, w HWIO f64 -> NHWC f64 (pre-activation)."""
    xt = x.permute(0, 3, 1, 2)
    wt = w.permute(3, 2, 0, 1)
    if mode == CONV3:
        y = F.conv2d(xt, wt, b, padding=1)
    elif mode == UPCONV2:
        up = F.interpolate(xt, scale_factor=2, mode="nearest")
        y = F.conv2d(F.pad(up, (0, 1, 0, 1)), wt, b)
    elif mode == CONV1:
        y = F.conv2d(xt, wt, b)
    return y.permute(0, 2, 3, 1)


X3_TOL = [None]          # set by the split-bf16 cases: (rtol, atol / max|ref|) of their products (2^-16 relative each)


def tol(dtype, ref):
    s = float(ref.abs().max())
    if X3_TOL[0] is not None and dtype == torch.float32:
        return (X3_TOL[0], X3_TOL[0] * s)
    return (1e-5, 1e-5 * s) if dtype == torch.float32 else (1.2e-2, 1.2e-2 * s)


def rnd(t, dtype):
    """round to the storage dtype, return f64"""
    return t.to(dtype).to(torch.float64)


CASES = [
    # mode,   B, H,  W,  C0, C1, Cout
    (CONV3,   2, 16, 16, 64, 0, 64),
    (CONV3,   1, 12, 20, 8, 0, 72),        # ragged M / N, tiny Cin
    (CONV3,   2, 8, 8, 64, 64, 128),       # concat of two sources
    (CONV3,   3, 32, 32, 128, 0, 136),     # 128-wide tiles with ragged N (LDS-resident patch kernel)
    (CONV3,   2, 36, 40, 64, 64, 64),      # patch kernel: ragged H and W tiles, concat, 64-channel tile
    (CONV3,   2, 36, 40, 72, 72, 64),      # concat whose first source is NOT a multiple of 64 channels: the weight gradient runs
                                           #   as one job per source (strip kernel), the reduction writes each source's rows of dW
    (CONV3,   4, 16, 16, 136, 136, 200),   # ... the same on the per-tap kernel (W < 32), ragged N
    (CONV3,   1, 64, 96, 72, 0, 40),       # patch kernel: channel tail (72 = 64 + 8), ragged N
    (CONV3,   2, 8, 64, 192, 0, 128),      # patch kernel: three channel chunks (patch reloaded twice)
    (CONV3,   33, 8, 32, 128, 0, 128),     # 33 strips of 32 pixels (odd): the last workgroup of the strip weight-gradient kernel has ONE
                                           #   busy 4-wave group, the other only keeps the barriers (both staggered and lockstep variants)
    (CONV3,   4, 130, 250, 8, 0, 24),      # register-stationary-weights kernel (>= 1024 tiles): ragged H, W, N; tiny Cin
    (CONV3,   2, 256, 256, 64, 0, 64),     # ... full 64 -> 64 layer, 4 tiles per persistent workgroup
    (CONV3,   6, 128, 128, 16, 0, 256),    # large grid of 128-channel tiles: the 8-row patch variant (768 tiles)
    (UPCONV2, 2, 32, 64, 72, 0, 136),      # low-resolution patch variant of the up-conv: channel tail, ragged N
    (UPCONV2, 2, 36, 40, 64, 0, 64),       # ... ragged W tile, 64-channel tile, far-edge zero padding
    (UPCONV2, 4, 128, 96, 16, 0, 24),      # all-taps weight gradient of the up-conv (192 strips), ragged channel tiles
    (UPCONV2, 2, 16, 16, 128, 0, 64),
    (UPCONV2, 1, 8, 12, 72, 0, 40),
    (CONV1,   2, 16, 16, 64, 0, 8),
    (CONV3,   5, 1, 1, 128, 0, 128),       # 1 x 1 maps (the bottom of a depth-6 network on 64 x 64): every divisor of the pixel decode is 1
    (CONV3,   3, 2, 1, 136, 0, 72),        # N x 1 maps, channel tails
    (CONV3,   8, 512, 32, 8, 0, 24),       # ONE column of 32-pixel tiles, 1024 tiles: the persistent level-0 kernel's tile decode divides by 1
]

# BASELINE configs[1] deep-level layers at their real shapes (B=16): few pixels, long reductions. With a workspace
# they take the one-workgroup-per-CU split-K schedule (conv_pipe_kernel) the train step uses.
DEEP_CASES = [
    # mode,   B, H,  W,  C0,  C1,  Cout
    (CONV3,   16, 16, 16, 256, 0, 512),     # encoder_L3_conv1
    (CONV3,   16, 8, 8, 1024, 0, 1024),     # bottom_conv2: K = 9216, 19 MB of weights
    (CONV3,   16, 16, 16, 512, 512, 512),   # upsample_L0_conv2 (concat)
    (UPCONV2, 16, 16, 16, 1024, 0, 512),    # upsample_L0_conv1 (+ its stride-2 data gradient)
    (CONV3,   3, 8, 12, 136, 72, 200),      # ragged: M = 288 (tail tile), channel tails in both sources, N tail
    (CONV3,   8, 48, 44, 256, 0, 1024),     # > 512 tiles of 256 x 128 with 1024 filters (predict-batch shape: the two-workgroup schedules take it)
    (UPCONV2, 4, 128, 130, 512, 0, 256),    # > 512 tiles, wide transposed conv (same)
]


# Grids of about one workgroup per CU with 8-row tiles: the 8-wave kernel with the double-buffered patch (conv_halo8,
# round 3) -- the BASELINE configs[1] level-1 / level-2 layers at their real shapes, and ragged variants.
HALO8_CASES = [
    # mode,   B, H,  W,  C0,  C1,  Cout
    (CONV3,   16, 64, 64, 128, 0, 128),     # encoder_L1_conv2: 256 tiles of 128 channels, two chunks (one patch prefetch)
    (CONV3,   16, 32, 32, 256, 0, 256),     # encoder_L2_conv2: 64-channel tiles (TM = 1), four chunks
    (CONV3,   12, 64, 64, 128, 128, 136),   # concat (second source = chunks 2..3) + ragged N (two 128-channel tiles: 384 workgroups)
    (CONV3,   13, 64, 40, 72, 0, 64),       # ragged W tile, channel tail (72 = 64 + 8)
    (UPCONV2, 16, 64, 64, 256, 0, 128),     # upsample_L2_conv1: low-resolution patch, 128-channel tiles
    (UPCONV2, 9, 64, 72, 72, 0, 40),        # up-conv: ragged W tile, channel tail, ragged N (216 tiles)
]


@pytest.mark.parametrize("case", HALO8_CASES)
def test_halo8_double_buffered_patch_schedule(case):
    import ctypes as C
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    lib.mpu_schedule_log_enable(1)
    try:
        _run_case(case, torch.bfloat16)
        n = lib.mpu_schedule_log_read(None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        lib.mpu_schedule_log_read(buf, n + 1)
    finally:
        lib.mpu_schedule_log_enable(0)
    conv = [l.split()[1] for l in buf.value.decode().splitlines() if l.startswith("conv ")]
    assert conv and conv[0] == "halo8", conv                    # the forward launch of the case took the new schedule


# Large inference grids of 128-channel tiles: the persistent 16-row staggered kernel (conv_halo16p, round 4). The first case
# reaches the dispatcher's grid bound (512 tiles) at its real size; the others run in a subprocess with MPU_HALO16_MIN=1 (read
# once per process) so that small shapes exercise the schedule: chunk boundaries, concat sources, channel tails (k-step
# guards), ragged W / N tiles, one-tile grids.
HALO16_BIG = (CONV3, 12, 128, 128, 64, 0, 256)
HALO16_CASES = [
    # mode,   B, H,  W,  C0,  C1,  Cout          (sources in multiples of 32 channels, an even number of 32-channel chunks)
    (CONV3,   2, 32, 64, 128, 0, 128),      # two blocks: patch prefetch of the next block's chunk A
    (CONV3,   1, 16, 40, 64, 0, 136),       # one block; ragged W tile, ragged N (two n-tiles)
    (CONV3,   2, 48, 32, 64, 64, 256),      # concat: the second block comes from the second source
    (CONV3,   1, 32, 96, 256, 0, 128),      # four blocks
    (CONV3,   3, 16, 32, 32, 32, 72),       # ONE block whose chunk B is the second source; 72 of 128 channels
    (CONV3,   1, 16, 32, 96, 32, 128),      # three chunks of source 0 + one of source 1: a block that straddles the sources
    (CONV3,   1, 16, 32, 192, 0, 128),      # three blocks
]


def _schedules_of(fn):
    import ctypes as C
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    lib.mpu_schedule_log_enable(1)
    try:
        fn()
        n = lib.mpu_schedule_log_read(None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        lib.mpu_schedule_log_read(buf, n + 1)
    finally:
        lib.mpu_schedule_log_enable(0)
    return [l.split()[1] for l in buf.value.decode().splitlines() if l.startswith("conv ")]


@pytest.mark.parametrize("case", HALO16_CASES + [HALO16_BIG])
def test_halo16_cases(case):
    """(meaningful under MPU_HALO16_MIN=1: test_halo16_subprocess runs it that way and asserts the schedule)"""
    import os
    conv = _schedules_of(lambda: _run_case(case, torch.bfloat16))
    if os.environ.get("MPU_HALO16_MIN") == "1" or case == HALO16_BIG:     # forward: the persistent form; the masked data gradient: conv_halo
        assert conv and conv[0] == "halo16p", conv
    # the same forward WITHOUT the ReLU (the clamp's lower bound is -inf then) and without a bias
    from multiplanarunet_amd import ops
    mode, B, H, W, C0, C1, Cout = case
    g = torch.Generator().manual_seed(7)
    x = rnd(torch.randn(B, H, W, C0 + C1, generator=g), torch.bfloat16)
    w = rnd(torch.randn(3, 3, C0 + C1, Cout, generator=g) / np.sqrt(9 * (C0 + C1)), torch.bfloat16)
    ref = ref_forward(mode, x, w, torch.zeros(Cout, dtype=torch.float64))
    xd = x.to("cuda", torch.bfloat16)
    wf, _ = ops.pack_weights(w.to("cuda", torch.float32), mode, torch.bfloat16)
    y = ops.conv2d(mode, xd[..., :C0].contiguous(), wf, Cout, (H, W), x1=xd[..., C0:].contiguous() if C1 else None, relu=False)
    rt, at = tol(torch.bfloat16, ref)
    assert float(ref.min()) < -0.1                                   # (negative outputs exist and survive)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), rtol=rt, atol=at)


def test_halo16_subprocess():
    """conv_halo16p with the grid bound lowered (MPU_HALO16_MIN=1, read once per process) so that every eligible small shape of its
    cases -- and of the whole layer suite -- takes it, in a fresh interpreter."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    # two passes: one tile per workgroup (grid = tiles), and 3 workgroups walking many tiles each (tile boundaries: the
    # look-ahead into the next tile, the store-aware waits)
    for extra in ({}, {"MPU_HALO16P_WGS": "3"}):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_conv.py"), "-x", "-q", "-k",
                            "halo16_cases or forward_dgrad_wgrad"],
                           env=dict(os.environ, MPU_HALO16_MIN="1", **extra),
                           capture_output=True, text=True, cwd=os.path.dirname(here))
        assert r.returncode == 0, str(extra) + r.stdout[-2000:] + r.stderr[-2000:]


# More 3x3 layers on square 8 / 16-pixel maps (whole 256-pixel tiles, sources in multiples of 64 channels, filters in
# multiples of 128, fewer tiles than CUs: the split-K schedules). Written for conv_deep, the whole-image halo-patch split-K
# kernel of round 5 that lost to conv_pipe inside the step and was removed; kept as cases of conv_pipe.
DEEPH_CASES = [
    # mode,   B, H,  W,  C0,  C1,  Cout
    (CONV3,   16, 16, 16, 512, 0, 512),     # encoder_L3_conv2: 64 tiles, K split 4 ways, two chunks per workgroup
    (CONV3,   16, 8, 8, 512, 0, 1024),      # bottom_conv1: 32 tiles, 8 ways, ONE chunk per workgroup (no patch prefetch at all)
    (CONV3,   4, 16, 16, 192, 0, 128),      # 4 tiles, three chunks split three ways
    (CONV3,   4, 8, 8, 320, 0, 256),        # 2 tiles, five chunks split five ways (four 8x8 images per tile)
    (CONV3,   8, 8, 8, 64, 128, 128),       # concat, three chunks over the two sources
    (CONV3,   1, 16, 16, 448, 64, 384),     # one image; eight chunks, the last one from the second source; three n-tiles
    (CONV3,   12, 8, 8, 448, 0, 128),       # 3 tiles, seven chunks split seven ways
    (CONV3,   32, 16, 16, 128, 0, 128),     # 32 tiles, two chunks: K split two ways (the minimum)
]


def _conv_schedules_of(fn):
    import ctypes as C
    from multiplanarunet_amd import _lib
    lib = _lib.load()
    lib.mpu_schedule_log_enable(1)
    try:
        fn()
        n = lib.mpu_schedule_log_read(None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        lib.mpu_schedule_log_read(buf, n + 1)
    finally:
        lib.mpu_schedule_log_enable(0)
    return [l.split()[1] for l in buf.value.decode().splitlines() if l.startswith("conv ")]


@pytest.mark.parametrize("case", DEEPH_CASES)
def test_more_deep_level_shapes_on_the_split_k_schedules(case):
    """Forward, data gradients and weight gradient of the case against the fp64 layer (odd K splits, one chunk per workgroup,
    concat chunks over two sources, three filter tiles)."""
    conv = _conv_schedules_of(lambda: _run_case(case, torch.bfloat16, workspace=True))
    assert conv and conv[0] in ("pipe", "glds", "deepk"), conv


# 3x3 layers on 16-pixel maps with K split over the WAVES of a workgroup (conv_deepk, round 5): 128-pixel x 64-filter tiles,
# no split-K partials, epilogue in the kernel. Eligible: 192..512 tiles, an even number of 64-channel chunks, plain epilogue.
DEEPK_CASES = [
    # mode,   B, H,  W,  C0,  C1,  Cout
    (CONV3,   16, 16, 16, 256, 0, 512),     # encoder_L3_conv1: 256 tiles, two chunk pairs
    (CONV3,   16, 16, 16, 512, 0, 512),     # encoder_L3_conv2 shape: the data gradient (ReLU mask) is eligible too
    (CONV3,   16, 16, 16, 256, 256, 512),   # concat: four pairs over two sources (reductions over more than 512 channels stay on conv_pipe)
    (CONV3,   12, 16, 16, 128, 0, 512),     # 192 tiles (the lower bound), ONE chunk pair
    (CONV3,   8, 16, 16, 64, 64, 1024),     # one pair made of the two sources; 16 filter tiles
    (CONV3,   5, 16, 16, 384, 0, 1280),     # 10 x 20 = 200 tiles: odd image count (a tile is half an image), three pairs
]


@pytest.mark.parametrize("case", DEEPK_CASES)
def test_k_split_over_the_waves_schedule_of_the_16_pixel_maps(case):
    """conv_deepk: the forward launch of every case takes it (schedule log) and matches the fp64 layer, as does every
    eligible data gradient (ReLU mask in the kernel's epilogue)."""
    import os
    conv = _conv_schedules_of(lambda: _run_case(case, torch.bfloat16, workspace=True))
    if os.environ.get("MPU_CONV_DEEPK") != "0":
        assert conv and conv[0] == "deepk", conv
        mode, B, H, W, C0, C1, Cout = case
        tiles_d = (B * 2) * ((C0 + C1) // 64)
        if 192 <= tiles_d <= 512 and (Cout // 64) % 2 == 0 and Cout <= 512:
            assert conv.count("deepk") >= 2, conv


@pytest.mark.parametrize("case", DEEP_CASES)
def test_deep_level_layers_split_k_schedule(case):
    """bf16, with the split-K workspace (the path mpu_unet_forward / backward take at the deep levels)."""
    _run_case(case, torch.bfloat16, workspace=True)


@pytest.mark.parametrize("dtype", (torch.float32, torch.bfloat16))
@pytest.mark.parametrize("case", CASES)
def test_conv_forward_dgrad_wgrad(case, dtype):
    _run_case(case, dtype)


def _run_case(case, dtype, workspace=False, x3=False, report=None):
    from multiplanarunet_amd import ops
    if x3:                                                       # dtype "bf16x3": f32 tensors, split-bf16 products (round 6)
        assert dtype == torch.float32
        _conv2d, _wgrad = ops.conv2d, ops.conv2d_wgrad
        _pack = ops.pack_weights
        ops = type("OpsX3", (), dict(pack_weights=staticmethod(lambda *a, **k: _pack(*a, x3=True, **k)),
                                     conv2d=staticmethod(lambda *a, **k: _conv2d(*a, x3=True, **k)),
                                     conv2d_wgrad=staticmethod(lambda *a, **k: _wgrad(*a, x3=True, **k))))
    mode, B, H, W, C0, C1, Cout = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    k = {CONV3: 3, UPCONV2: 2, CONV1: 1}[mode]
    Hi, Wi = (H // 2, W // 2) if mode == UPCONV2 else (H, W)
    Cin = C0 + C1
    x = rnd(torch.randn(B, Hi, Wi, Cin, generator=g), dtype)
    w = rnd(torch.randn(k, k, Cin, Cout, generator=g) / np.sqrt(k * k * Cin), dtype)
    b = torch.randn(Cout, generator=g).to(torch.float64) * 0.1
    dz = rnd(torch.randn(B, H, W, Cout, generator=g), dtype)

    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    pre = ref_forward(mode, xr, wr, b)
    y_ref = torch.relu(pre)
    pre.backward(dz)
    dx_ref, dw_ref = xr.grad, wr.grad

    dev = "cuda"
    xd = x.to(dev, dtype)
    x0 = xd[..., :C0].contiguous()
    x1 = xd[..., C0:].contiguous() if C1 else None
    wf, wd = ops.pack_weights(w.to(dev, torch.float32), mode, dtype)
    ws = None
    if workspace:
        ws = torch.empty(8 * B * H * W * max(Cout, Cin), dtype=torch.float32, device=dev)
    y = ops.conv2d(mode, x0, wf, Cout, (H, W), bias=b.to(dev, torch.float32), x1=x1, relu=True, workspace=ws)
    rt, at = tol(dtype, y_ref)
    np.testing.assert_allclose(y.cpu().double().numpy(), y_ref.detach().numpy(), rtol=rt, atol=at)

    # data gradient (+ ReLU mask of the consumer's input)
    dzd = dz.to(dev, dtype)
    dmode = CONV3S2 if mode == UPCONV2 else (CONV1 if mode == CONV1 else CONV3)
    wdg = wd if mode != CONV1 else w.to(dev, dtype).reshape(-1)
    mask = (torch.rand(B, Hi, Wi, Cin, generator=g) > 0.3).to(torch.float64)
    dx = ops.conv2d(dmode, dzd, wdg, Cin, (Hi, Wi), mask=mask.to(dev, dtype),
                    w_tap_stride=Cin * Cout, w_row_stride=Cout, workspace=ws)
    ref = dx_ref * mask
    rt, at = tol(dtype, ref)
    np.testing.assert_allclose(dx.cpu().double().numpy(), ref.numpy(), rtol=rt, atol=at)
    if C1:   # channel-sliced data gradient (how the concat gradient is split)
        import ctypes
        esz = 2 if dtype == torch.bfloat16 else 4
        dx1 = ops.conv2d(dmode, dzd, wdg[C0 * Cout:], C1, (Hi, Wi), w_tap_stride=Cin * Cout, w_row_stride=Cout,
                         workspace=ws)
        np.testing.assert_allclose(dx1.cpu().double().numpy(), dx_ref[..., C0:].numpy(), rtol=rt, atol=at)

    # weight gradient
    dW = ops.conv2d_wgrad(mode, x0, dzd, x1=x1)
    ref = dw_ref.reshape(k * k, Cin, Cout)
    rt, at = tol(dtype, ref)
    if dtype == torch.bfloat16:
        rt, at = 2e-3, 2e-3 * float(ref.abs().max())      # f32 accumulate of exact bf16 products
    np.testing.assert_allclose(dW.cpu().double().numpy(), ref.numpy(), rtol=rt, atol=at)


X3_CASES = [CASES[0], CASES[2], CASES[3], CASES[4], CASES[7], CASES[8], CASES[13], CASES[14], CASES[16], CASES[20]]   # (3x3 and up-convs: the 1x1 head has its own kernels)


@pytest.mark.parametrize("case", X3_CASES + DEEP_CASES[:5])
def test_split_bf16_products_against_fp64(case):
    """dtype "bf16x3" (MPU_F32X3, round 6): f32 tensors, every product of forward / data gradient / weight gradient as three
    bf16 MFMAs on operands split hi + lo in registers. Each product is good to ~2^-16 relative, so the layer outputs are held to
    5e-5 of the tensor maximum against the fp64 layer -- 240x tighter than the bf16 mode's bound, 5x looser than exact f32."""
    X3_TOL[0] = 5e-5
    try:
        _run_case(case, torch.float32, workspace=case in DEEP_CASES, x3=True)
    finally:
        X3_TOL[0] = None


def _rand_shapes(n, seed):
    """Random layer shapes incl. the degenerate extents the multiply-high decodes must survive (1, 2, 3, one short of /
    exactly / one past a 32-pixel tile), batches up to 40 and grids past 1024 tiles."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        mode = [CONV3, CONV3, UPCONV2][rng.randint(0, 3)]
        B = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 40]))
        if mode == UPCONV2:
            H = int(rng.choice([2, 4, 8, 16, 32, 34, 64, 128]))
            W = int(rng.choice([2, 4, 32, 34, 40, 64, 66, 96]))
        else:
            H = int(rng.choice([1, 2, 3, 4, 8, 16, 31, 32, 33, 64, 130]))
            W = int(rng.choice([1, 2, 3, 31, 32, 33, 40, 64, 96, 160]))
        C0 = int(rng.choice([8, 16, 24, 64, 72, 128]))
        C1 = int(rng.choice([0, 0, 0, C0])) if mode == CONV3 and C0 % 64 == 0 else 0
        Cout = int(rng.choice([8, 24, 40, 64, 72, 128, 136]))
        if B * H * W * (C0 + C1) * Cout > 1.2e9:
            continue
        out.append((mode, B, H, W, C0, C1, Cout))
    return out


SWEEP = _rand_shapes(34, 11) + [
    (CONV3, 40, 33, 31, 8, 0, 24),          # ragged both ways, 40 images
    (CONV3, 24, 64, 32, 64, 0, 64),         # ONE 32-pixel tile wide, 384 x 4-row tiles
    (CONV3, 40, 130, 32, 8, 0, 40),         # one tile wide, > 1024 tiles (persistent level-0 kernel), ragged H
    (CONV3, 13, 3, 33, 128, 0, 128),        # 3 rows, 33 columns: a second tile column of one pixel
    (UPCONV2, 40, 2, 2, 64, 0, 72),         # 1 x 1 low-resolution maps
    (UPCONV2, 5, 34, 66, 72, 0, 40),        # ragged low-resolution patch in both directions
    (CONV3, 8, 31, 33, 64, 64, 72),         # concat on ragged tiles
    (CONV3, 2, 2, 3, 128, 128, 136),        # concat on a 2 x 3 map
]


@pytest.mark.parametrize("dtype", (torch.float32, torch.bfloat16))
@pytest.mark.parametrize("case", SWEEP)
def test_shape_sweep_against_fp64_reference(case, dtype):
    """Shape sweep (ragged tiles, strips, channel tails, degenerate extents): every schedule the dispatcher picks, in the
    exact-f32 AND the bf16 mode, against the independent fp64 convolution of _run_case (round 3 compared the bf16 kernels
    with the f32 kernels of the same library, which share the index decodes: the divisor-1 bug lived there)."""
    _run_case(case, dtype)


@pytest.mark.parametrize("case", [
    # B, H,   W,   Cout, image channels
    (16, 128, 128, 64, 1),        # BASELINE configs[1] first layer: dedicated HBM-bound kernel, 1024 strips
    (3, 20, 50, 64, 2),           # two image channels, ragged row passes (50 = 32 + 18)
    (2, 16, 24, 128, 1),          # 16 channel groups: 16 pixel lanes per pass
    (5, 9, 7, 8, 2),              # one channel group: 256 pixel lanes, rows shorter than a pass
    (2, 12, 40, 24, 1),           # 3 channel groups (not a power of two): general path, same entry point
    (2, 16, 16, 64, 3),           # three image channels: general path
])
def test_first_layer_weight_gradient(case):
    """mpu_conv2d_wgrad_first_layer (wgrad_c8.hip) vs an f64 evaluation of dW[t][ci][co] = sum x[px+t][ci] dz[px][co]
    and db = sum dz on the same bf16-rounded operands; padding channels of dW must be exactly zero."""
    from multiplanarunet_amd import ops
    B, H, W, Cout, CI = case
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + CI)
    x = torch.zeros(B, H, W, 8)
    x[..., :CI] = torch.randn(B, H, W, CI, generator=g)
    x = x.bfloat16()
    dz = torch.randn(B, H, W, Cout, generator=g).bfloat16()
    dW, db = ops.conv2d_wgrad_first_layer(x.cuda(), CI, dz.cuda())
    xd, zd = x.double(), dz.double()
    xp = F.pad(xd.permute(0, 3, 1, 2), (1, 1, 1, 1)).permute(0, 2, 3, 1)        # [B, H+2, W+2, 8]
    ref = torch.stack([torch.einsum("bhwi,bhwo->io", xp[:, ky:ky + H, kx:kx + W], zd)
                       for ky in range(3) for kx in range(3)])                   # [9, 8, Cout]
    dWc = dW.cpu().double()
    assert torch.count_nonzero(dWc[:, CI:]) == 0
    s = float(ref.abs().max())
    assert float((dWc - ref).abs().max()) <= 2e-3 * s
    rb = zd.sum((0, 1, 2))
    assert float((db.cpu().double() - rb).abs().max()) <= 2e-3 * float(rb.abs().max()) + 1e-3
    # the general-purpose entry point agrees (it has no channel-count hint and runs the MFMA kernels)
    dWg = ops.conv2d_wgrad(CONV3, x.cuda(), dz.cuda()).cpu().double()
    assert float((dWg - ref).abs().max()) <= 2e-3 * s


def test_upconv_8row_tiles_subprocess():
    """The up-conv halo kernels switch to 8-row pixel tiles on large grids (predict batches). MPU_HALO_UP8_MIN=1 (read
    once per process, hence a fresh interpreter) selects them for every eligible shape: the layer parity tests of this
    file must pass with them too (both the 64- and the 128-channel tile variants)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_conv.py"), "-x", "-q", "-k",
                        "deep_level or forward_dgrad_wgrad"], env=dict(os.environ, MPU_HALO_UP8_MIN="1"),
                       capture_output=True, text=True, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
