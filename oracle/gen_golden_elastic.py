"""
TEST INFRASTRUCTURE ONLY -- generates tests/golden/elastic_golden.npz by running the reference's own,
unmodified elastic_transform_2d (mpunet/augmentation/elastic_deformation.py:6-69, mpunet 0.2.12 under
/root/reference) through oracle/ref_shim.py with seeded NumPy RNG state. Run by hand in the build container:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_elastic.py

The file holds only data: seeded inputs (image, labels, alpha, sigma, bg values, the RNG seed that reproduces
the two uniform noise fields the reference draws) and the reference's outputs.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from mpunet.augmentation.elastic_deformation import elastic_transform_2d  # noqa: E402


def main():
    out = {}
    cases = [  # H, W, C, alpha, sigma, seed
        (40, 48, 1, 120.0, 6.0, 1),
        (33, 29, 2, 450.0, 20.0, 2),      # kernel radius (80) beyond the image: zero padding dominates
        (64, 64, 1, 0.0, 25.0, 3),        # alpha 0: identity sampling at integer points (edge semantics)
        (24, 56, 3, 900.0, 3.0, 4),       # strong, rough field: many samples leave the image (fill values)
    ]
    for k, (H, W, C, alpha, sigma, seed) in enumerate(cases):
        rng = np.random.RandomState(100 + seed)
        image = rng.randn(H, W, C).astype(np.float32)
        labels = rng.randint(0, 4, (H, W)).astype(np.uint8)
        bg = [float(b) for b in rng.randn(C)]
        np.random.seed(seed)                                    # the reference draws np.random.rand(H, W) twice
        im2, lab2 = elastic_transform_2d(image.copy(), labels.copy(), alpha, sigma, bg)
        out["c%d_image" % k] = image; out["c%d_labels" % k] = labels
        out["c%d_params" % k] = np.array([alpha, sigma, seed], np.float64)
        out["c%d_bg" % k] = np.array(bg, np.float64)
        out["c%d_out_image" % k] = im2; out["c%d_out_labels" % k] = lab2
    out["n_cases"] = np.array(len(cases))
    dst = os.path.join(HERE, "..", "tests", "golden", "elastic_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
