"""
`mp train`'s producer / consumer loop (multiplanarunet_amd/pipeline.py; reference: mpunet/train/trainer.py:238-257, five loader
threads ahead of model.fit). Overlap changes WHEN a batch is cut, never WHICH batch a step sees: the overlapped, graphed
pipeline must end with the weights of the serial eager loop, bit for bit.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def quiet(*a, **k):
    pass


def _model_and_sampler(seed, l2_reg=None, elastic=False, dtype="bf16"):
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
    from multiplanarunet_amd.augmentation import build_augmenters
    dev = torch.device("cuda")
    dim, B = 64, 8
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=3, complexity_factor=0.25, flatten_output=True, dtype=dtype,
             l2_reg=l2_reg, logger=quiet, seed=0, device=dev)
    m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-3})
    img, lab, aff = make_toy_volume(64, 5)
    vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy64")
    augs = build_augmenters([{"cls_name": "Elastic2D", "kwargs": {"alpha": [0, 100], "sigma": [6, 9], "apply_prob": 0.5}}],
                            seed=4) if elastic else None
    s = TrainSampler([vol], random_views(3, 60.0, 0), dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=seed,
                     augmenters=augs)
    return m, s


@pytest.mark.parametrize("kw", [dict(l2_reg=1e-4), dict(elastic=True), dict(dtype="f32")], ids=["l2", "elastic", "f32"])
def test_pipeline_variants_equal_the_serial_eager_loop_bitwise(kw):
    """The same identity with the l2 term summed into the device-side loss, with the Elastic2D augmenter running on the producer
    stream, and in the f32 mode."""
    from multiplanarunet_amd.pipeline import TrainPipeline
    m0, s0 = _model_and_sampler(21, **kw)
    m1, s1 = _model_and_sampler(21, **kw)
    p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
    p1 = TrainPipeline(m1, s1)
    for n in (4, 3):
        a, b = p0.run_epoch(n), p1.run_epoch(n)
        torch.cuda.synchronize()
        assert np.isfinite(a) and a == b, (kw, a, b)
        assert torch.equal(m0.params, m1.params) and torch.equal(m0.bn_state, m1.bn_state)


def test_side_stream_runs_beside_a_busy_main_stream():
    """The fill probe of pipeline.pick_side_stream in a FRESH process (its figure depends on the streams a process has created before:
    bench.py's e2e leg reads 6 ms for every candidate after the other legs although the loop then runs at the step's rate -- the
    stream that matters is chosen under the real loop, test_mp_train_pipeline_delivers_at_least_090_of_the_step_rate)."""
    import subprocess, sys, os
    code = ("import torch\n"
            "from multiplanarunet_amd.pipeline import pick_side_stream\n"
            "st, lat = pick_side_stream(torch.device('cuda'), busy_ms=3.0)\n"
            "assert isinstance(st, torch.cuda.Stream) and st != torch.cuda.current_stream()\n"
            "print('LAT %.1f' % lat)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lat = float([l for l in r.stdout.splitlines() if l.startswith("LAT ")][-1].split()[1])
    assert np.isfinite(lat) and lat > 0
    print("producer stream probe latency %.0f us" % lat)
    # a stream queued BEHIND the probe's 3 ms of fills answers after >= 3000 us. Round 6: FAIL, do not skip --
    # `mp train` on such a stream delivers 0.64 of the step rate (gpurun R5p), and a runtime update that changes the stream -> queue
    # mapping must be seen here, not in a throughput regression nobody attributes
    assert lat < 1500.0, "no hardware queue beside the main stream after 4 batches of candidates (best latency %.0f us)" % lat


def test_mp_train_pipeline_delivers_at_least_090_of_the_step_rate():
    """`train_e2e` as a guarded property: the producer / consumer loop of `mp train` (sampler of a 128^3 volume on
    the measured side stream, one batch ahead of the graphed configs[1] step) must deliver >= 0.90 of the rate of the bare step
    replayed on a fixed batch -- bench.py reports 0.97; 0.64 is what a producer stream on the training stream's queue gives."""
    import time
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
    from multiplanarunet_amd.pipeline import TrainPipeline
    dev = torch.device("cuda")
    B, dim = 16, 128
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16", logger=quiet,
             seed=0, device=dev)
    m.compile("Adam", "SparseCategoricalCrossentropy")
    img, lab, aff = make_toy_volume(128, 77)
    vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
    smp = TrainSampler([vol], random_views(6, 60.0, 0), dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=7)
    x, y, w = smp()
    rep = m.make_graphed_train_step(x, y.reshape(B, dim * dim, 1), w)
    for _ in range(10):
        rep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60):
        rep()
    torch.cuda.synchronize()
    bare = (time.perf_counter() - t0) / 60
    pipe = TrainPipeline(m, smp)
    pipe.run_epoch(4 + pipe.CAL_STREAMS * pipe.CAL_WINDOW)       # capture + the producer-stream windows (pipeline.TrainPipeline)
    assert pipe._cal is None and len(pipe.side_loop_ms) >= 1, pipe.side_loop_ms
    epochs = []
    for _ in range(3):                                    # three epochs of 90 steps; the best one is the loop's rate (the first
        torch.cuda.synchronize(); t0 = time.perf_counter()   # still carries one-off costs of the freshly captured graph)
        loss = pipe.run_epoch(90)
        torch.cuda.synchronize()
        epochs.append((time.perf_counter() - t0) / 90)
    e2e = min(epochs)
    frac = bare / e2e
    print("bare step %.3f ms, mp-train loop %s ms per step, fraction %.3f, producer stream latency %.0f us (candidates under the loop: "
          "%s ms), loss %.4f" % (bare * 1e3, " / ".join("%.3f" % (e * 1e3) for e in epochs), frac, pipe.side_latency_us,
                                 pipe.side_loop_ms, loss))
    assert np.isfinite(loss)
    assert frac >= 0.90, "mp train's loop delivers %.2f of the step rate (producer stream latency %.0f us)" % (frac, pipe.side_latency_us)


def test_overlapped_graphed_pipeline_equals_the_serial_eager_loop_bitwise():
    from multiplanarunet_amd.pipeline import TrainPipeline
    steps = 7
    m0, s0 = _model_and_sampler(11)
    m1, s1 = _model_and_sampler(11)
    p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
    p1 = TrainPipeline(m1, s1)                                   # defaults: graphed step, sampler on a side stream
    assert p1.graphed and p1.overlap and p1.side is not None
    l0 = p0.run_epoch(steps)
    l1 = p1.run_epoch(steps)
    torch.cuda.synchronize()
    assert np.isfinite(l0) and l0 == l1, (l0, l1)
    w0, w1 = m0.get_weights_dict(), m1.get_weights_dict()
    assert sorted(w0) == sorted(w1)
    for k in w0:
        np.testing.assert_array_equal(w0[k], w1[k], err_msg=k)
    # a second epoch continues from the pending batch: still identical, and the device-side loss sum was reset by the read
    assert p0.run_epoch(3) == p1.run_epoch(3)


def test_epoch_loss_of_the_graphed_pipeline_on_the_real_network_split_bf16():
    """Round 6 (found by tools/round6/af_soak.py): in the graphed bf16x3 step the epoch loss went STALE for stretches of replays -- the
    parameters stayed bit-identical to the serial loop's, the per-pixel loss the library wrote was fresh, but torch's captured
    `loss.mean()` (a multi-block reduction with semaphores) repeated an old value for up to 50 consecutive replays. The step's mean
    loss now comes out of the backward pass itself (UNet.loss_mean: head_bwd_finalize_kernel), so no torch reduction is captured.
    configs[1] network, 240 steps: epoch losses and every parameter equal the serial eager loop's, and loss_mean() agrees with the
    mean of the per-pixel loss tensor."""
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
    from multiplanarunet_amd.pipeline import TrainPipeline
    dev = torch.device("cuda")
    B, dim = 16, 128
    img, lab, aff = make_toy_volume(128, 77)
    vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy128")
    views = random_views(6, 60.0, 0)

    def mk():
        m = UNet(n_classes=3, dim=dim, n_channels=1, depth=4, complexity_factor=1, flatten_output=True, dtype="bf16x3", logger=quiet,
                 seed=0, device=dev)
        m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-4})
        return m, TrainSampler([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=3)
    m0, s0 = mk()
    m1, s1 = mk()
    x, y, w = s0.__class__([vol], views, dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=9)()
    _, loss = m0.forward_backward(x, y, w)
    lm, ref = float(m0.loss_mean().item()), float(loss.double().mean().item())
    assert abs(lm - ref) <= 2e-6 * abs(ref), (lm, ref)
    m0.grads.zero_()
    m0.bn_state.copy_(m1.bn_state)                               # (the probe above moved the moving statistics)
    p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
    p1 = TrainPipeline(m1, s1)
    for ep in range(4):
        a, b = p0.run_epoch(60), p1.run_epoch(60)
        torch.cuda.synchronize()
        assert np.isfinite(a) and a == b, (ep, a, b)
        assert torch.equal(m0.params, m1.params) and torch.equal(m0.bn_state, m1.bn_state), ep


def test_a_learning_rate_change_recaptures_the_graph():
    from multiplanarunet_amd.pipeline import TrainPipeline
    m0, s0 = _model_and_sampler(3)
    m1, s1 = _model_and_sampler(3)
    p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
    p1 = TrainPipeline(m1, s1)
    for p in (p0, p1):
        p.run_epoch(3)
        p.model.optimizer_kwargs["lr"] = 2.5e-4                  # what ReduceLROnPlateau does between epochs
    a, b = p0.run_epoch(3), p1.run_epoch(3)
    torch.cuda.synchronize()
    assert a == b, (a, b)
    w0, w1 = m0.get_weights_dict(), m1.get_weights_dict()
    for k in w0:
        np.testing.assert_array_equal(w0[k], w1[k], err_msg=k)
