"""
GPU parity of the Elastic2D augmentation kernels: BIT-EXACT against the outputs of the reference's own
elastic_transform_2d (tests/golden/elastic_golden.npz) given the same noise fields, and against the oracle on
larger seeded cases (incl. the full 256x256 training size and the YAML's alpha/sigma ranges).
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "elastic_golden.npz"))


def test_bit_exact_vs_reference_goldens():
    from multiplanarunet_amd.augmentation import elastic_transform_2d
    for k in range(int(G["n_cases"])):
        alpha, sigma, seed = G["c%d_params" % k]
        image, labels = G["c%d_image" % k], G["c%d_labels" % k]
        np.random.seed(int(seed))
        noise = np.stack([np.random.rand(*image.shape[:2]), np.random.rand(*image.shape[:2])])
        x, y = elastic_transform_2d(torch.tensor(image, device="cuda"), torch.tensor(labels, device="cuda"), alpha, sigma,
                                    list(G["c%d_bg" % k]), noise=noise)
        assert np.array_equal(y.cpu().numpy(), G["c%d_out_labels" % k]), k
        assert np.array_equal(x.cpu().numpy(), G["c%d_out_image" % k]), (k, np.abs(x.cpu().numpy() - G["c%d_out_image" % k]).max())


@pytest.mark.parametrize("H,W,C,alpha,sigma", [(256, 256, 1, 450.0, 20.0), (128, 128, 2, 300.0, 30.0), (96, 160, 1, 50.0, 2.0)])
def test_bit_exact_vs_oracle_training_sizes(H, W, C, alpha, sigma):
    from multiplanarunet_amd.augmentation import elastic_transform_2d
    from oracle import augmentation as A
    rng = np.random.RandomState(H + W)
    image = rng.randn(H, W, C).astype(np.float32)
    labels = rng.randint(0, 5, (H, W)).astype(np.uint8)
    noise = (rng.rand(H, W), rng.rand(H, W))
    bg = [0.25] * C
    ref_x, ref_y = A.elastic_transform_2d(image, labels, alpha, sigma, bg, noise=noise)
    x, y = elastic_transform_2d(torch.tensor(image, device="cuda"), torch.tensor(labels, device="cuda"), alpha, sigma, bg,
                                noise=np.stack(noise))
    assert np.array_equal(y.cpu().numpy(), ref_y)
    assert np.array_equal(x.cpu().numpy(), ref_x)
    assert (ref_y != labels).any()                                  # the field does move pixels


def test_elastic2d_augmenter_semantics():
    from multiplanarunet_amd.augmentation import Elastic2D, build_augmenters
    aug = build_augmenters([{"cls_name": "Elastic2D", "kwargs": {"alpha": [0, 450], "sigma": [20, 30], "apply_prob": 0.5}}], seed=3)[0]
    B, H = 16, 64
    x = torch.randn(B, H, H, 1, device="cuda"); y = torch.randint(0, 3, (B, H, H), device="cuda", dtype=torch.uint8)
    w = torch.ones(B, device="cuda")
    x0, y0 = x.clone(), y.clone()
    x, y, w = aug(x, y, [[0.0]] * B, w)
    changed = (x != x0).reshape(B, -1).any(1).cpu().numpy()
    wn = w.cpu().numpy()
    assert 0 < changed.sum() < B                                    # some, not all, elements were deformed
    np.testing.assert_allclose(wn[changed], 0.33, rtol=1e-6)        # augmented elements carry aug_weight ...
    assert (wn[~changed] == 1.0).all() and (x[~torch.tensor(changed)] == x0[~torch.tensor(changed)]).all()
    with pytest.raises(ValueError):
        Elastic2D([5, 1], 20, 0.3)
    with pytest.raises(NotImplementedError):
        build_augmenters([{"cls_name": "Elastic3D", "kwargs": {}}])
