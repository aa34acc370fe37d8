"""The elastic-deformation oracle against outputs of the reference's own function (tests/golden/elastic_golden.npz)."""
import os
import numpy as np
import scipy.ndimage

from oracle import augmentation as A

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "elastic_golden.npz"))


def test_gaussian_filter_restatement_is_bit_exact_vs_scipy():
    rng = np.random.RandomState(0)
    a = rng.rand(37, 52) * 2 - 1
    for sigma in (2.5, 7.0, 20.0, 30.0):
        ref = scipy.ndimage.gaussian_filter(a, sigma, mode="constant", cval=0.)
        got = A.gaussian_filter_zero_pad(a, sigma)
        assert np.array_equal(got, ref), (sigma, np.abs(got - ref).max())


def test_elastic_transform_2d_bit_exact_vs_reference_goldens():
    for k in range(int(G["n_cases"])):
        alpha, sigma, seed = G["c%d_params" % k]
        image, labels = G["c%d_image" % k], G["c%d_labels" % k]
        np.random.seed(int(seed))
        noise = (np.random.rand(*image.shape[:2]), np.random.rand(*image.shape[:2]))
        im2, lab2 = A.elastic_transform_2d(image, labels, alpha, sigma, list(G["c%d_bg" % k]), noise=noise)
        assert np.array_equal(lab2, G["c%d_out_labels" % k]), k
        assert np.array_equal(im2, G["c%d_out_image" % k]), (k, np.abs(im2 - G["c%d_out_image" % k]).max())
