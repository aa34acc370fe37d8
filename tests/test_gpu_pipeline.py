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


def _model_and_sampler(seed):
    from multiplanarunet_amd.unet import UNet
    from multiplanarunet_amd.data import make_toy_volume, as_volume, random_views, TrainSampler
    dev = torch.device("cuda")
    dim, B = 64, 8
    m = UNet(n_classes=3, dim=dim, n_channels=1, depth=3, complexity_factor=0.25, flatten_output=True, dtype="bf16",
             logger=quiet, seed=0, device=dev)
    m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs={"lr": 1e-3})
    img, lab, aff = make_toy_volume(64, 5)
    vol = as_volume(img, lab, aff, "1pct", "RobustScaler", dev, "toy64")
    s = TrainSampler([vol], random_views(3, 60.0, 0), dim, float(dim), B, 3, noise_sd=0.1, fg_batch_fraction=0.5, seed=seed)
    return m, s


def test_side_stream_runs_beside_a_busy_main_stream():
    from multiplanarunet_amd.pipeline import pick_side_stream
    st, lat = pick_side_stream(torch.device("cuda"), busy_ms=3.0)
    assert isinstance(st, torch.cuda.Stream) and st != torch.cuda.current_stream()
    assert lat < 1500.0, lat        # a stream queued BEHIND the probe's 3 ms of fills answers after >= 3000 us


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
