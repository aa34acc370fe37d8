"""
Third-party PUBLISHED known answers for the TensorFlow half of the oracle (VERDICT r5 task 5).

tensorflow==2.3.2 (the reference's requirements.txt:10) is not in this image, so oracle/unet_ref.py cannot be
compared with TensorFlow itself. What TensorFlow publishes are the numeric outputs printed in the docstring
examples of its own API pages (tf.keras 2.3, `Usage:` / `Standalone usage:` blocks, which TF's doctest runner
executes against the library). Those values are used here as known answers for the pieces of the restatement
whose semantics were otherwise *inferred* (SURVEY.md 8a row a7):

  * tf.keras.losses.SparseCategoricalCrossentropy / tf.keras.losses.sparse_categorical_crossentropy and
    tf.keras.metrics.SparseCategoricalCrossentropy (the latter's docstring spells the arithmetic out: a zero
    probability becomes EPSILON, its log is -16.1181 = log(1e-7): the CLIPPED-probability form);
  * tf.keras.optimizers.Adam ("the first step is -learning_rate * sign(grad)": 10.0 -> 9.9);
  * tf.keras.layers.MaxPool2D, UpSampling2D, tf.keras.activations.relu (layer semantics used by unet.py:114-216).

This narrows, but does not remove, the "PARITY UNPINNED" status of oracle/unet_ref.py: BatchNormalization's
moving-variance update and Conv2D have no published numeric example. CPU only.
"""
import numpy as np
import torch
import torch.nn.functional as F
from oracle import unet_ref as U

# tf.keras.losses.SparseCategoricalCrossentropy, "Standalone usage" (TF 2.3 API page)
Y_TRUE = np.array([1, 2])
Y_PRED = np.array([[0.05, 0.95, 0.0], [0.1, 0.8, 0.1]], np.float32)


def _ce(sample_w=(1.0, 1.0)):
    # the oracle's layout: probs [B,H,W,K], y [B,H,W], one weight per image -> two "images" of one pixel
    l = U.keras_sparse_ce(torch.tensor(Y_PRED).reshape(2, 1, 1, 3), torch.tensor(Y_TRUE).reshape(2, 1, 1),
                          torch.tensor(np.asarray(sample_w, np.float32)))
    return l.numpy().reshape(2).astype(np.float64)


def test_sparse_ce_reduction_none_published_vector():
    # >>> scce = SparseCategoricalCrossentropy(reduction=tf.keras.losses.Reduction.NONE)
    # >>> scce(y_true, y_pred).numpy()   ->   array([0.0513, 2.303], dtype=float32)
    # (reduction NONE is what the reference compiles with: mpunet/train/trainer.py:78-97)
    l = _ce()
    np.testing.assert_allclose(l, [0.0513, 2.303], atol=5e-4)         # the docstring prints 3-4 significant digits
    np.testing.assert_allclose(l, [-np.log(0.95), -np.log(0.1)], atol=5e-7)      # f32 arithmetic + the second softmax (sum q = 1 + 1e-7)


def test_sparse_ce_mean_sum_and_sample_weight_published_values():
    l = _ce()
    assert abs(l.mean() - 1.177) < 5e-4                                # default reduction: 1.177
    assert abs(l.sum() - 2.354) < 5e-4                                 # Reduction.SUM: 2.354
    lw = _ce((0.3, 0.7))
    assert abs(lw.mean() - 0.814) < 5e-4                               # sample_weight=[0.3, 0.7]: 0.814


def test_sparse_ce_metric_docstring_spells_out_the_clipping():
    # tf.keras.metrics.SparseCategoricalCrossentropy docstring:
    #   softmax = [[0.05, 0.95, EPSILON], [0.1, 0.8, 0.1]]
    #   log(softmax) = [[-2.9957, -0.0513, -16.1181], [-2.3026, -0.2231, -2.3026]]
    #   xent = [0.0513, 2.3026];  m.result().numpy() -> 1.1769392
    q = torch.clamp(torch.tensor(Y_PRED), U.CE_EPS, 1 - U.CE_EPS)
    np.testing.assert_allclose(torch.log(q).numpy(), [[-2.9957, -0.0513, -16.1181], [-2.3026, -0.2231, -2.3026]], atol=6e-5)
    assert abs(_ce().mean() - 1.1769392) < 2e-6
    # and the loss of the class whose probability is exactly 0 is the clipped one, -log(1e-7), not infinity
    l0 = U.keras_sparse_ce(torch.tensor(Y_PRED[:1]).reshape(1, 1, 1, 3), torch.tensor([[[2]]]), torch.ones(1))
    assert abs(float(l0) - 16.1181) < 1e-3


def test_adam_first_step_published_value():
    # >>> opt = tf.keras.optimizers.Adam(learning_rate=0.1); var1 = tf.Variable(10.0)
    # >>> loss = lambda: (var1 ** 2) / 2.0          # d(loss)/d(var1) == var1
    # >>> opt.minimize(loss, [var1]);  # "The first step is `-learning_rate*sign(grad)`"
    # >>> var1.numpy()   ->   9.9
    th, m, v = U.adam_update(np.float64(10.0), np.float64(10.0), 0.0, 0.0, 1, lr=0.1, b1=0.9, b2=0.999, eps=1e-7)
    assert abs(th - 9.9) < 1e-6
    # the same with the reference's epsilon (train_hparams.yaml:126): still lr * sign(g) to 1e-8
    th2, _, _ = U.adam_update(np.float64(10.0), np.float64(10.0), 0.0, 0.0, 1, lr=0.1, eps=1e-8)
    assert abs(th2 - 9.9) < 1e-8
    assert abs(m - 1.0) < 1e-12 and abs(v - 0.1) < 1e-12               # m = (1-b1) g, v = (1-b2) g^2


def test_maxpool2d_published_example():
    # tf.keras.layers.MaxPool2D docstring: x = [[1,2,3,4],[5,6,7,8],[9,10,11,12]] reshaped [1,3,4,1],
    # pool_size=(2,2), strides=(2,2), padding='valid'  ->  [[[[6.],[8.]]]]   (the reference's pooling: unet.py:126)
    x = torch.arange(1.0, 13.0).reshape(1, 1, 3, 4)
    np.testing.assert_array_equal(F.max_pool2d(x, 2, 2).numpy().reshape(-1), [6.0, 8.0])
    # ... and strides (1,1) on the 3x3 example  ->  [[5,6],[8,9]]
    x3 = torch.arange(1.0, 10.0).reshape(1, 1, 3, 3)
    np.testing.assert_array_equal(F.max_pool2d(x3, 2, 1).numpy().reshape(2, 2), [[5.0, 6.0], [8.0, 9.0]])


def test_upsampling2d_published_example():
    # tf.keras.layers.UpSampling2D docstring: x = arange(12).reshape(2,2,1,3); size=(1,2) ->
    # [[[[0 1 2] [0 1 2]] [[3 4 5] [3 4 5]]] [[[6 7 8] [6 7 8]] [[9 10 11] [9 10 11]]]]   (nearest: rows/cols repeated)
    x = np.arange(12, dtype=np.float32).reshape(2, 2, 1, 3)
    y = F.interpolate(torch.tensor(x).permute(0, 3, 1, 2), scale_factor=(1, 2), mode="nearest").permute(0, 2, 3, 1).numpy()
    want = np.array([[[[0, 1, 2], [0, 1, 2]], [[3, 4, 5], [3, 4, 5]]], [[[6, 7, 8], [6, 7, 8]], [[9, 10, 11], [9, 10, 11]]]], np.float32)
    np.testing.assert_array_equal(y, want)
    # the oracle's own call (scale 2 on both axes, unet.py:151) repeats every pixel 2 x 2
    z = F.interpolate(torch.tensor(x).permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(z, np.repeat(np.repeat(x, 2, 1), 2, 2))


def test_relu_published_example():
    # tf.keras.activations.relu docstring: foo = [-10, -5, 0.0, 5, 10] -> [0., 0., 0., 5., 10.]
    np.testing.assert_array_equal(torch.relu(torch.tensor([-10.0, -5.0, 0.0, 5.0, 10.0])).numpy(), [0, 0, 0, 5, 10])
