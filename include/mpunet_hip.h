/*
 * mpunet_hip.h -- C ABI of libmpunet_hip.so: the MI355X (gfx950) hot path of
 * perslev/MultiPlanarUNet (mpunet 0.2.12).
 *
 * The reference reaches its arithmetic through tf.keras / NumPy; it has no FFI
 * of its own. Each entry point below therefore cites the reference call site
 * (file:line under /root/reference) whose work it replaces. INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller;
 *     everything else is host memory read before the call returns;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues
 *     work on it (no synchronisation, no allocation) unless stated;
 *   - return 0 on success, negative mpu_status otherwise; the message of the
 *     last failure on the calling thread is mpu_last_error();
 *   - activations are NHWC; `dtype` selects storage/arithmetic:
 *       MPU_F32  : f32 storage, exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
 *       MPU_BF16 : bf16 storage, v_mfma_f32_32x32x16_bf16, f32 accumulate.
 *       MPU_F32X3: f32 storage, every product as three bf16 MFMAs on operands split hi + lo in registers (x ~ bf16(x) +
 *                  bf16(x - bf16(x)): ~2^-16 relative per product) -- the tolerance-grade mode at bf16-class matrix rates
 *                  (round 6; Python dtype "bf16x3"). Elementwise kernels, buffers and layouts are those of MPU_F32.
 */
#ifndef MPUNET_HIP_H
#define MPUNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MPU_OK = 0, MPU_EINVAL = -1, MPU_EHIP = -2, MPU_EUNSUPPORTED = -3 } mpu_status;
typedef enum { MPU_F32 = 0, MPU_BF16 = 1, MPU_F32X3 = 2 } mpu_dtype;

int         mpu_abi_version(void);
const char* mpu_last_error(void);
/* 16 hex digits: sha256 over every source, header and compiler flag this library was built from (compiled in by
 * multiplanarunet_amd/build.py). The ctypes host layer refuses a library whose hash is not that of the sources beside it. */
const char* mpu_build_hash(void);

/* ------------------------------------------------------------------------ *
 * Predict-time geometry (HBM-bound kernels)
 * ------------------------------------------------------------------------ */

/* A strictly increasing coordinate axis that also lives in device memory as an f64 array.
 * When the host has verified that a closed form reproduces the array BIT FOR BIT it says so
 * here and the kernels evaluate the form in registers instead of loading the array:
 *   kind 0: no closed form; only `step` is used (to seed the cell search)
 *   kind 1: value(i) = (i == n-1) ? last : fl(fl(i*step) + start)        (np.linspace)
 *   kind 2: value(i) = fl((i - start) * step)      (voxel axes: (arange(n)-(n-1)/2)*pixdim) */
typedef struct {
    int32_t kind;
    int32_t n;
    double  start, step, last;
} mpu_axis;

/* One view of IsotrophicLiveViewSequence2D.get_view_from
 * (mpunet/sequences/isotrophic_live_view_sequence_2d.py:29-117) expressed as
 * numbers: the plane basis of sample_plane_at (mpunet/interpolation/
 * sample_grid.py:192-244), the in-plane mgrid axis, and the optional rot_mat of
 * ViewInterpolator.apply_rotation (mpunet/interpolation/view_interpolator.py:54-60).
 * Matrices are row-major 3x3. */
typedef struct {
    double  basis[9];      /* columns u, v, n_hat                              */
    double  rot[9];        /* rot_mat; ignored unless has_rot                  */
    int32_t has_rot;
    int32_t dim;           /* sample_dim                                       */
    int32_t n_planes;      /* P = len(offsets)                                 */
    int32_t _pad;
    double  g_start;       /* np.mgrid[-hd:hd:dim*1j]: value(i) = i*g_step+g_start */
    double  g_step;
    mpu_axis vol_axis[3];  /* closed forms / spacings of the voxel axes d_ax, d_ay, d_az */
} mpu_view_geom;

/* get_view_from / sample_at: trilinear image planes (+ nearest label planes)
 * for every offset of one view, then MultiChannelScaler.transform.
 *   d_vol      f32 [X,Y,Z,C]            ImagePair.image
 *   d_labels   u8  [X,Y,Z] or NULL      ImagePair.labels
 *   d_ax,d_ay,d_az  f64 voxel axes of get_voxel_axes_real_space (sample_grid.py:63-98)
 *   d_offsets  f64 [P]                  np.linspace(-bounds, bounds, P)
 *   d_bg       f32 [C]                  RegularGridInterpolator fill_value per channel
 *   d_center,d_scale f64 [C] or NULL    sklearn scaler center_/scale_ (NULL = identity)
 *   d_out      f32 [P,dim,dim,C]        == np.moveaxis(Xs, 2, 0) of the reference
 *   d_out_lab  u8  [P,dim,dim] or NULL
 * Replaces isotrophic_live_view_sequence_2d.py:64-101 + view_interpolator.py:62-133
 * + regular_grid_interpolator.py:152-270 + preprocessing/scaling.py:75-89. */
int mpu_sample_view_planes(const float* d_vol, const uint8_t* d_labels,
                           const int32_t vol_shape[4],
                           const double* d_ax, const double* d_ay, const double* d_az,
                           const mpu_view_geom* geom, const double* d_offsets,
                           const float* d_bg, uint8_t bg_class,
                           const double* d_center, const double* d_scale,
                           float* d_out, uint8_t* d_out_lab, void* stream);

/* One view's prediction volume as seen by the back-mapping
 * (mpunet/utils/fusion/fuse_and_predict.py:92-137). */
typedef struct {
    double       inv_basis[9];   /* np.linalg.inv(basis) of the view            */
    const float* d_pred;         /* f32 [P,dim,dim,K] (model.predict output order) */
    const double* d_g;           /* f64 [dim]  real_axis = np.linspace(-hd,hd,dim) */
    const double* d_offsets;     /* f64 [P]                                     */
    int32_t      dim;
    int32_t      n_planes;
    mpu_axis     g_axis;         /* closed form / spacing of d_g       */
    mpu_axis     o_axis;         /* closed form / spacing of d_offsets */
} mpu_view_pred;

/* Voxel grid of get_voxel_grid_real_space (sample_grid.py:101-130), computed on
 * the fly: p = A*(i,j,k) - center. */
typedef struct {
    double  A[9];          /* affine[:3,:3]                                     */
    double  center[3];     /* mean over all voxels of A*(i,j,k)                 */
    int32_t shape[3];      /* X, Y, Z                                           */
    int32_t _pad;
} mpu_voxel_grid;

/* map_real_space_pred(method="nearest") for ONE view: d_mapped f32 [X,Y,Z,K],
 * OOB voxels -> [1,0,...,0] (fuse_and_predict.py:99-100). */
int mpu_map_view_nearest(const mpu_voxel_grid* grid, const mpu_view_pred* view,
                         int32_t n_classes, float* d_mapped, void* stream);

/* Fused _multi_view_predict_on + merge_multi_view_preds
 * (mpunet/bin/predict.py:294-366): for every voxel, nearest lookup in each of
 * the V view predictions, FusionLayer.call = softmax_k(sum_v W[v,k] x[v,k] + b[k])
 * (mpunet/models/fusion_model.py:38-39) or, with sum_fusion, sum_v x[v,k]
 * (predict.py:364); then pred_to_class argmax -> uint8 (utils/utils.py:326-328).
 * Nothing of `combined[V,X,Y,Z,K]` is materialised.
 *   views      host array of V mpu_view_pred
 *   d_W f32 [V,K], d_b f32 [K]  (ignored when sum_fusion)
 *   d_probs    f32 [X,Y,Z,K] or NULL ; d_labels u8 [X,Y,Z] or NULL */
int mpu_map_fuse_views(const mpu_voxel_grid* grid, const mpu_view_pred* views,
                       int32_t n_views, int32_t n_classes,
                       const float* d_W, const float* d_b, int32_t sum_fusion,
                       float* d_probs, uint8_t* d_labels, void* stream);

/* FusionModel.predict on an explicit x[N,V,K] (predict.py:358-361 layout). */
int mpu_fusion_forward(const float* d_x, int64_t n, int32_t n_views, int32_t n_classes,
                       const float* d_W, const float* d_b,
                       float* d_probs, uint8_t* d_labels, void* stream);

/* Multi-GPU predict: accumulate W[v,:] * nearest(x_v) of one
 * view's plane chunk [p_lo, p_hi) into d_z f32 [X,Y,Z,K]; the rank that owns
 * p_lo == 0 also adds the OOB contribution W[v,:]*[1,0..]. Then
 * mpu_fusion_finalize applies +b, softmax, argmax on a voxel slab. */
int mpu_map_accumulate_view(const mpu_voxel_grid* grid, const mpu_view_pred* view,
                            int32_t n_classes, const float* d_Wv,
                            int32_t p_lo, int32_t p_hi, int32_t owns_oob,
                            float* d_z, void* stream);
int mpu_fusion_finalize(const float* d_z, int64_t n, int32_t n_classes,
                        const float* d_b, int32_t sum_fusion,
                        float* d_probs, uint8_t* d_labels, void* stream);


/* ------------------------------------------------------------------------ *
 * 2-D U-Net (MFMA-bound): mpunet.models.UNet (mpunet/models/unet.py:20-251)
 * ------------------------------------------------------------------------ */

/* conv modes of the implicit-GEMM kernels */
enum { MPU_CONV3 = 0,      /* Conv2D(k=3, padding="same")                 unet.py:120-179 */
       MPU_UPCONV2 = 1,    /* UpSampling2D(2) + Conv2D(k=2, "same")       unet.py:159-163 */
       MPU_CONV3S2 = 2,    /* 3x3 stride-2 pad-1: data gradient of MPU_UPCONV2           */
       MPU_CONV1 = 3 };    /* Conv2D(k=1)                                 unet.py:211     */

/* What UNet.__init__ / init_model (unet.py:26-112,182-216) derive from the
 * constructor arguments. filters[l] = int(64 * 2^l * sqrt(complexity_factor)),
 * l = 0..depth (unet.py:91,120,132). padding="same", activation="relu",
 * kernel_size=3 are the only supported values. */
typedef struct {
    int32_t n_classes;     /* 1..8                                              */
    int32_t n_channels;
    int32_t depth;         /* 1..6                                              */
    int32_t H, W;          /* img_rows, img_cols: multiples of 2^depth          */
    int32_t dtype;         /* mpu_dtype of activations / MFMA operands          */
    int32_t softmax;       /* out_activation: 1 = "softmax", 0 = "linear"       */
    int32_t filters[8];
} mpu_unet_config;

typedef struct mpu_unet mpu_unet;   /* host-only layer table; owns no device memory */

mpu_unet* mpu_unet_create(const mpu_unet_config* cfg);      /* NULL + mpu_last_error() on bad config */
void      mpu_unet_destroy(mpu_unet* m);

/* Flat fp32 buffers the caller allocates (channel counts are padded to multiples
 * of 8; padded entries are zero and stay zero under training):
 *   params / grads / adam m / adam v : mpu_unet_param_floats()   floats each
 *   BN moving statistics             : mpu_unet_bn_state_floats() floats
 *   packed MFMA weight operands      : mpu_unet_packed_bytes()    bytes
 *   activations + scratch            : mpu_unet_workspace_bytes(batch) bytes
 * mpu_unet_tensor_info enumerates "<keras layer name>/<kernel|bias|gamma|beta|
 * moving_mean|moving_variance>" with its offset, stored (padded) and logical
 * Keras shape (kernels are HWIO), kind 0 = params buffer, 1 = BN-state buffer. */
int64_t mpu_unet_param_floats(const mpu_unet* m);
int64_t mpu_unet_bn_state_floats(const mpu_unet* m);
int64_t mpu_unet_packed_bytes(const mpu_unet* m);
int64_t mpu_unet_logical_param_count(const mpu_unet* m);   /* trainable, unpadded */
int64_t mpu_unet_workspace_bytes(const mpu_unet* m, int32_t batch);
int32_t mpu_unet_num_tensors(const mpu_unet* m);
int     mpu_unet_tensor_info(const mpu_unet* m, int32_t idx, char* name, int32_t name_cap,
                             int32_t* kind, int64_t* offset,
                             int32_t stored_shape[4], int32_t logical_shape[4]);

/* fp32 master weights -> MFMA operands (forward [tap][co][ci], data-gradient
 * [tap][ci][co] incl. the tap-combined 3x3 stride-2 form of the up-conv).
 * Call after every change of d_params (set_weights, load_weights, Adam). */
int mpu_unet_pack_weights(const mpu_unet* m, const float* d_params, void* d_packed, void* stream);

/* Inference-mode BatchNormalization is folded into the producing convolutions: this computes the
 * per-channel scale/shift (gamma, beta, moving statistics, epsilon 1e-3) into the tail of d_packed.
 * Call after every change of d_params / d_bn_state and before mpu_unet_forward(training = 0). */
int mpu_unet_prepare_inference(const mpu_unet* m, const float* d_params, const float* d_bn_state,
                               void* d_packed, void* stream);

/* model.predict_on_batch / the forward half of a Model.fit step
 * (mpunet/utils/fusion/fuse_and_predict.py:88, mpunet/train/trainer.py:246).
 *   d_x   f32 [B,H,W,n_channels] ; d_out f32 [B,H,W,n_classes] (probabilities or
 *   logits). training != 0: BatchNormalization uses batch statistics and updates
 *   the moving statistics in d_bn_state; activations stay in d_workspace for
 *   mpu_unet_backward. */
/* d_out may be NULL in training mode: the probabilities then stay in the workspace only, at byte offset
 * mpu_unet_workspace_probs_offset(m, batch) (f32 [B,H,W,n_classes]; valid until the next forward on that workspace). */
int64_t mpu_unet_workspace_probs_offset(const mpu_unet* m, int32_t batch);
/* Byte offset (inside the workspace of `batch`) of ONE float: the mean over the B*H*W pixels of the weighted per-pixel loss of the
 * last backward pass (= mean of the d_loss tensor mpu_unet_backward fills; written by the pass whether d_loss is NULL or not).
 * It is what a training loop accumulates per step without a reduction of its own (reference: the `loss` Keras logs per batch,
 * mpunet/train/trainer.py:246-257). Valid until the next backward pass on that workspace. */
int64_t mpu_unet_workspace_loss_mean_offset(const mpu_unet* m, int32_t batch);
int mpu_unet_forward(const mpu_unet* m, int32_t batch, const float* d_x, const float* d_params,
                     const void* d_packed, float* d_bn_state, void* d_workspace,
                     int32_t training, float* d_out, void* stream);

/* Backward half of the Keras train step compiled at mpunet/train/trainer.py:78-97
 * (SparseCategoricalCrossentropy(reduction=NONE) on clipped probabilities,
 * per-image sample weights, gradient of the SUM over batch and pixels).
 *   d_y u8 [B,H*W] ; d_sample_weight f32 [B] ; d_grads f32 [param_floats]
 *   d_loss f32 [B,H*W] per-pixel weighted loss or NULL.
 * Must follow mpu_unet_forward(training=1) on the same workspace -- ONE backward pass per training forward: the forward's first
 * launch zeroes the fixed-point BatchNorm accumulators of both passes (round 6, bf16 / bf16x3), a second backward pass on the same
 * forward would add to the first one's sums. The handle remembers which form of the head the last training forward ran (with a
 * 64-channel last block and a softmax head it leaves NO post-BatchNorm tensor of that block in the workspace; the backward pass
 * then recomputes it): forward and backward of one step run on the same handle, not interleaved with another step's. */
int mpu_unet_backward(const mpu_unet* m, int32_t batch, const uint8_t* d_y,
                      const float* d_sample_weight, const float* d_params, const void* d_packed,
                      float* d_bn_state, void* d_workspace, float* d_grads, float* d_loss,
                      void* stream);

/* Accept / reject statistics of one sampled training slice (mpunet/sequences/isotrophic_live_view_sequence.py:91-128:
 * np.isin(fg_classes, lab), np.any(~np.isclose(im, bg))): d_out2[0] = OR of (1 << label) over the labels (< 32),
 * d_out2[1] = 1 when some pixel differs from its channel's background value d_bg[c]. Either input may be NULL. */
int mpu_plane_stats(const uint8_t* d_labels, const float* d_image, int64_t n_pixels, int32_t n_channels,
                    const float* d_bg, uint32_t* d_out2, void* stream);

/* mpu_sample_view_planes for ONE plane followed by mpu_plane_stats of it: one candidate slice of the train-time sampler per call
 * (arguments as the two; d_bg_scaled = the background value as the scaled plane shows it). */
int mpu_sample_plane_stats(const float* d_vol, const uint8_t* d_labels, const int32_t vol_shape[4],
                           const double* d_ax, const double* d_ay, const double* d_az,
                           const mpu_view_geom* geom, const double* d_offset,
                           const float* d_bg, uint8_t bg_class, const double* d_center, const double* d_scale,
                           float* d_out, uint8_t* d_out_lab, const float* d_bg_scaled, uint32_t* d_stats2, void* stream);

/* Elastic2D augmentation of one training slice (mpunet/augmentation/elastic_deformation.py:6-69, applied by
 * mpunet/augmentation/augmenters.py:87-107 after scaling): image [H][W][C] f32 bilinear with fill d_bg[c],
 * labels [H][W] u8 nearest with fill 0 (either pair may be NULL), displaced by alpha * gaussian_filter(2*noise-1)
 * of the two uniform [0,1) fields d_noise [2][H][W] (f64). d_gauss_w: the 2*radius+1 normalised Gaussian weights
 * (f64; radius = int(4*sigma + .5)); d_workspace: mpu_elastic_workspace_doubles(H, W) doubles. */
int64_t mpu_elastic_workspace_doubles(int32_t H, int32_t W);
int mpu_elastic_transform_2d(const float* d_image, const uint8_t* d_labels, int32_t H, int32_t W, int32_t C,
                             const double* d_noise, const double* d_gauss_w, int32_t radius, double alpha,
                             const float* d_bg, double* d_workspace, float* d_out_image,
                             uint8_t* d_out_labels, void* stream);

/* Fusion-model training (mpunet/bin/train_fusion.py:327-362 `fusion_model.fit`): one Adam step of the FusionLayer
 * (mpunet/models/fusion_model.py:14-39) on a batch of points x[n][V][K] with integer targets y[n] under the
 * reference's per-point generalized Dice loss (mpunet/evaluate/loss_functions.py:207-246, SUM_OVER_BATCH_SIZE) plus
 * its 1e-6 * mean(square) regularisers. t >= 1: Keras Adam step t is applied to d_W [V][K] / d_b [K] (moments in
 * d_adam_m / d_adam_v, V*K+K floats); t == 0: gradients and loss only. d_grads_out (V*K+K floats) and d_loss_out
 * (1 float, the batch loss BEFORE the update) may be NULL. */
int64_t mpu_fusion_train_workspace_floats(int32_t n_views, int32_t n_classes);
int mpu_fusion_train_step(const float* d_x, const uint8_t* d_y, int64_t n, int32_t n_views,
                          int32_t n_classes, float* d_W, float* d_b, float* d_adam_m, float* d_adam_v,
                          int64_t t, double lr, double beta1, double beta2, double eps,
                          float* d_workspace, float* d_grads_out, float* d_loss_out, void* stream);
/* The same step in two halves for data-parallel fitting (the reference builds its models under
 * tf.distribute.MirroredStrategy, mpunet/bin/train_fusion.py:336): every rank turns ITS share of the batch (n >= 0 points)
 * into d_sums[V*K + K + 2] doubles -- the gradient SUMS of the data term, the summed per-point loss and, last, its point count
 * -- the host SUM-all-reduces the doubles over the ranks, and mpu_fusion_apply_sums divides by the total count, adds the
 * regularisers and applies the Adam step. With one rank the pair gives bit for bit what mpu_fusion_train_step gives. */
int mpu_fusion_grad_sums(const float* d_x, const uint8_t* d_y, int64_t n, int32_t n_views, int32_t n_classes,
                         const float* d_W, const float* d_b, float* d_workspace, double* d_sums, void* stream);
int mpu_fusion_apply_sums(const double* d_sums, int32_t n_views, int32_t n_classes, float* d_W, float* d_b,
                          float* d_adam_m, float* d_adam_v, int64_t t, double lr, double beta1, double beta2,
                          double eps, float* d_grads_out, float* d_loss_out, void* stream);

/* Data-parallel training: the backward pass finishes the flat gradient buffer from its END towards its start
 * (head, up blocks, bottom, encoder levels). mpu_unet_grad_ready_points returns the number of ready points and
 * writes their float offsets (descending): after point k every gradient at offset >= offsets[k] is final.
 * mpu_unet_backward_events is mpu_unet_backward that also records ready_events[k] (hipEvent_t, NULL = skip) on
 * `stream` at point k, so the caller can start the all-reduce of a bucket (tf.distribute.MirroredStrategy's
 * gradient aggregation, mpunet/bin/train.py:349) while the rest of the backward pass is still running. */
int32_t mpu_unet_grad_ready_points(const mpu_unet* m, int64_t* offsets, int32_t cap);
int mpu_unet_backward_events(const mpu_unet* m, int32_t batch, const uint8_t* d_y,
                             const float* d_sample_weight, const float* d_params, const void* d_packed,
                             float* d_bn_state, void* d_workspace, float* d_grads, float* d_loss,
                             void* const* ready_events, int32_t n_events, void* stream);

/* kernel_regularizer=regularizers.l2(l2) of the encoder / bottom / up-sampling conv kernels (mpunet/models/unet.py:
 * 122-177,189; the 1x1 output conv at :211, biases and BatchNorm carry none): d_grads += 2*l2*W over those tensors,
 * to be called after mpu_unet_backward (after the replica all-reduce under data parallelism) and before the Adam
 * step. d_reg_loss (optional, 1 float) receives l2 * sum W^2, the term Keras adds to the reported loss; it needs
 * d_partial = mpu_unet_l2_workspace_doubles() doubles of scratch. Deterministic summation order. */
int mpu_unet_l2_regularizer(const mpu_unet* m, const float* d_params, float* d_grads, double l2, double* d_partial,
                            float* d_reg_loss, void* stream);
int64_t mpu_unet_l2_workspace_doubles(void);

/* The optimizer step of a U-Net in ONE launch: the Adam update of mpu_adam_step over the model's whole flat
 * parameter buffer AND the refresh of the packed MFMA operands (mpu_unet_pack_weights), tile by tile -- the updated
 * kernel tile is written to d_params and, converted, to both packed copies while it is in registers / LDS. Bit-identical
 * to mpu_adam_step followed by mpu_unet_pack_weights (tests/test_gpu_unet.py), 0.25 GB less HBM traffic per step of the
 * configs[1] network. d_step != NULL: step count t-1 in device memory, incremented by the call (graph replay, as
 * mpu_adam_step_device_counter); else t is the 1-based step. Replaces, for this path, Keras' optimizer.apply_gradients
 * inside Model.fit (mpunet/train/trainer.py:246). */
int mpu_unet_adam_pack(const mpu_unet* m, float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t t,
                       int64_t* d_step, double lr, double beta1, double beta2, double eps, void* d_packed, void* stream);

/* One Model.fit inner step's backward half AND its optimizer (mpunet/train/trainer.py:246: Keras runs the tape's
 * gradients and optimizer.apply_gradients inside one fit step): mpu_unet_backward followed by mpu_unet_adam_pack, with the
 * same results bit for bit, scheduled so that the HBM-bound optimizer runs BESIDE the MFMA-bound weight gradients
 * instead of behind them (round 6). The deep levels' weight gradients (every layer that is not on the strip-resident
 * wgrad_taps schedule: 86 % of the parameters of the configs[1] network) are finished first; their Adam update + operand
 * refresh then run on a library-owned side stream, as a register- and LDS-lean kernel that shares the compute units with
 * the grouped weight-gradient launch of the high-resolution levels; the remaining parameters follow that launch. Under
 * stream capture the fork / join become parallel branches of the graph (run one eager call first: the side stream and
 * its two events are created at the first use). MPU_TAIL_OVERLAP=0: the serial order. Single GPU, no l2 term (both need
 * the gradients between the two halves: use the two calls). Arguments as mpu_unet_backward + mpu_unet_adam_pack;
 * d_params / d_packed are read by the backward pass and updated by the optimizer. */
int mpu_unet_backward_adam(const mpu_unet* m, int32_t batch, const uint8_t* d_y, const float* d_sample_weight,
                           float* d_params, void* d_packed, float* d_bn_state, void* d_workspace, float* d_grads,
                           float* d_loss, float* d_m, float* d_v, int64_t t, int64_t* d_step, double lr, double beta1,
                           double beta2, double eps, void* stream);

/* Dev aid: on != 0 arms eight timing events that mpu_unet_backward_adam records at the branch points of its tail (0 before the
 * deep levels' weight gradients, 1 after their reductions = the fork, 2 / 3 side stream before / after its optimizer launch, 4 after
 * the grouped wgrad_taps launch, 5 after its reductions, 6 after the remaining optimizer launch, 7 after the join); on == 0
 * disarms, synchronises and writes the milliseconds of each since event 0 into ms_out[8] (-1: not recorded). Timestamps of BOTH
 * streams without rocprofv3, whose per-dispatch signals change how the two queues interleave. */
int mpu_debug_tail_events(int32_t on, float* ms_out);

/* Keras Adam (TF ApplyAdam form), t = 1-based step; YAML defaults
 * lr 5e-5, beta_1 .9, beta_2 .999, epsilon 1e-8
 * (mpunet/bin/defaults/MultiPlanar/train_hparams.yaml:126). */
int mpu_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n,
                  int64_t t, double lr, double beta1, double beta2, double eps, void* stream);

/* Same update with the step count t-1 held in device memory (*d_step, incremented by the call): the form a
 * captured HIP graph can replay, since every replay must see a new bias-correction factor. */
int mpu_adam_step_device_counter(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n,
                                 int64_t* d_step, double lr, double beta1, double beta2, double eps,
                                 void* stream);

/* Single-layer entry points (unit tests / layer-wise integration). Channels
 * must be multiples of 8. d_w is the fp32 Keras HWIO kernel; d_w_dgrad may be
 * NULL; sizes: fwd taps*Cin*Cout elements, dgrad 9*Cin*Cout elements.
 * mpu_conv2d_igemm computes out[m][n] = act(sum_k in[m@tap][k] w[tap][n][k] + bias[n])
 * (* (mask[m][n] > 0) when d_mask is given) over the channel-concatenation of
 * in0 and in1. */
int mpu_conv2d_pack_weights(int32_t dtype, int32_t mode, const float* d_w, int32_t Cin, int32_t Cout,
                            void* d_w_fwd, void* d_w_dgrad, void* stream);
int mpu_conv2d_igemm(int32_t dtype, int32_t mode, const void* d_in0, int32_t C0,
                     const void* d_in1, int32_t C1, const void* d_w_packed,
                     int64_t w_tap_stride, int32_t w_row_stride, const float* d_bias,
                     const void* d_mask, void* d_out, int32_t B, int32_t Ho, int32_t Wo,
                     int32_t Cout, int32_t relu, void* stream);
/* Same, with a caller-owned scratch buffer: layers with few output tiles and a long reduction (the deep U-Net levels)
 * split K over workgroups and need room for their f32 partial sums (8 * B*Ho*Wo * Cout floats always suffice; with
 * less the split is reduced). mpu_unet_forward / backward pass their workspace the same way. */
int mpu_conv2d_igemm_ws(int32_t dtype, int32_t mode, const void* d_in0, int32_t C0,
                     const void* d_in1, int32_t C1, const void* d_w_packed,
                     int64_t w_tap_stride, int32_t w_row_stride, const float* d_bias,
                     const void* d_mask, void* d_out, int32_t B, int32_t Ho, int32_t Wo,
                     int32_t Cout, int32_t relu, float* d_workspace, int64_t workspace_floats, void* stream);
int64_t mpu_conv2d_wgrad_workspace_floats(int32_t mode, int32_t Cin, int32_t Cout, int64_t M);
/* Planning queries (host arithmetic only, no GPU needed): the floats ONE layer's weight-gradient job writes into its scratch
 * region (K-split / strip partials of dW + bias-gradient partial rows) when it runs stand-alone (grouped = 0) or inside a
 * grouped launch (grouped = 1: fewer strips / splits but other schedule thresholds, so either can be the larger), and the
 * region mpu_unet_workspace_bytes() reserves for that layer (>= both). tests/test_host_geometry.py sweeps shapes over them. */
int64_t mpu_conv2d_wgrad_job_floats(int32_t dtype, int32_t mode, int32_t B, int32_t Ho, int32_t Wo, int32_t C0, int32_t C1,
                                    int32_t Cout, int32_t grouped);
int64_t mpu_conv2d_wgrad_scratch_floats(int32_t dtype, int32_t mode, int32_t B, int32_t Ho, int32_t Wo, int32_t C0, int32_t C1,
                                        int32_t Cout);
int mpu_conv2d_wgrad(int32_t dtype, int32_t mode, const void* d_x0, int32_t C0, const void* d_x1,
                     int32_t C1, const void* d_dz, int32_t Cout, int32_t B, int32_t Ho, int32_t Wo,
                     float* d_workspace, float* d_dW, void* stream);

/* Weight and bias gradient of the FIRST layer (Conv2D(F, 3) on the n_channels-channel input image,
 * mpunet/models/unet.py:120-123): d_x holds the image in 8-channel pixel records of which the first
 * n_image_channels are real (the rest zero). d_dW is the fp32 kernel gradient [9][8][Cout] (zeros for the padding
 * channels), d_db (optional) the bias gradient [Cout]. With 1-2 image channels in bf16 this is a dedicated HBM-bound
 * kernel (wgrad_c8.hip); other shapes take the general path. Workspace: ..._workspace_floats(Cout, B*H*W) floats. */
int64_t mpu_conv2d_wgrad_first_layer_workspace_floats(int32_t Cout, int64_t M);
int mpu_conv2d_wgrad_first_layer(int32_t dtype, const void* d_x, int32_t n_image_channels, const void* d_dz,
                                 int32_t Cout, int32_t B, int32_t H, int32_t W, float* d_workspace, float* d_dW,
                                 float* d_db, void* stream);

/* Epoch-end validation counting (Validation._count_cm_elements_from_queue, mpunet/callbacks/validation.py:115-125):
 * p = argmax over the class axis of d_pred [n][n_classes] (f32 scores; first maximum, NaN counts as the maximum, as
 * np.argmax), then per class TP = #(y == p == c), relevant = #(y == c), selected = #(p == c), ADDED to
 * d_counts [3][n_classes] (int64; the caller zeroes it at the start of an epoch and may SUM all-reduce it across
 * replicas). Targets >= n_classes count for no class. Integer work: exact and order-independent. 1 <= n_classes <= 16. */
int mpu_validation_count(const float* d_pred, const uint8_t* d_y, int64_t n, int32_t n_classes, int64_t* d_counts,
                         void* stream);

/* Measurement aid (bench.py roofline leg; no reference counterpart): when
 * enabled, every MFMA convolution launch is bracketed by HIP events recorded on
 * its own stream. mpu_profile_summary synchronises on them and returns the summed
 * kernel time, the summed ALGORITHMIC FLOPs (2*M*N*K of the reference's layer,
 * SURVEY.md 8d) and the launch count. kind 0 = conv_igemm (forward + data
 * gradient), 1 = wgrad_igemm. mpu_profile_enable(0|1) also clears the records. */
int mpu_profile_enable(int32_t on);
int mpu_profile_summary(int32_t kind, double* total_ms, double* total_flops, int64_t* launches);

/* Test aid: the geometry kernels decide cells / nearest neighbours with a closed form on uniform axes wherever that
 * is provably the exact answer (coordinate farther than 1e-6 index units from every decision boundary) and run the
 * exact NumPy-order search otherwise. on = 0 forces the exact search for every sample (equality tests, A/B);
 * the environment variable MPU_GEOM_FAST=0 sets the same at first use. */
int mpu_geometry_set_fast_path(int32_t on);

/* Test aid for the plane sampler's cell division (csrc/geometry.hip cell_div): evaluates `count` pseudo-random
 * quotients (x - g[c]) / (g[c+1] - g[c]) on the closed-form axis both ways and returns in *n_bad how many differ
 * from the IEEE `/` (must be 0). Synchronous; default stream. */
int mpu_geometry_check_cell_division(const mpu_axis* axis, int64_t count, uint64_t seed, uint64_t* n_bad);

/* Measured machine peaks quoted next to the spec peaks in bench.py's roofline objects. The caller times
 * the launches with events on `stream`. mpu_probe_mfma_bf16: `blocks` workgroups x 4 waves each issue iters x 8
 * independent v_mfma_f32_32x32x16_bf16 (no memory traffic); *flops = FLOPs executed. mpu_probe_stream_triad:
 * a = b + 1.5 c over n floats (n % 4 == 0), 12 * n bytes of HBM traffic. */
int mpu_probe_mfma_bf16(int32_t blocks, int32_t iters, float* d_sink, double* flops, void* stream);
/* the same loop on pseudo-random bf16 operands: the matrix rate the part sustains under its POWER limit on real data */
int mpu_probe_mfma_bf16_random(int32_t blocks, int32_t iters, float* d_sink, double* flops, void* stream);
int mpu_probe_stream_triad(float* d_a, const float* d_b, const float* d_c, int64_t n, void* stream);
/* dst = src over n floats (n % 4096 == 0), 16 bytes per lane, four float4s per thread in flight: 8 * n bytes of HBM traffic -- the
 * "float4 copy" MI355X_MICROARCH.md quotes at 6.29 TB/s. variant 0: default cache policy; 1: non-temporal loads and stores;
 * 2: default policy, grid-stride. */
int mpu_probe_stream_copy(float* d_dst, const float* d_src, int64_t n, int32_t variant, void* stream);
/* Measurement aid: the n floats of d_src (n a power of two >= 8192) read once, in runs of run_bytes contiguous bytes (a power of
 * two >= 16) visited in a scattered order; per-thread sums to d_out (n / 32 floats). What HBM delivers to a kernel whose requests
 * are coalesced but whose DRAM pages are opened out of order (the gather kernels of fuse_and_predict.py:92-137). */
int mpu_probe_permuted_read(const float* d_src, float* d_out, int64_t n, int32_t run_bytes, void* stream);
/* out[i] = x[3i] + x[3i+1] + x[3i+2]: 12 contiguous bytes per lane, 12 n bytes read exactly once -- the access width of
 * the fused back-mapping's K = 3 gathers; calibrates rocprofv3's FETCH_SIZE for that width (MI355X_MICROARCH.md: the
 * gfx950 x2 correction is established for 16-byte accesses only). */
int mpu_probe_gather12(const float* d_x, float* d_out, int64_t n, void* stream);
/* Shader-clock sampler (measurement aid): ONE wave writes n pairs (s_memtime = shader cycles, s_memrealtime = 100-MHz ticks)
 * into d_samples [2 n] u64, about naps x 8 k cycles apart. Launch it on a side stream next to the work of interest; the
 * effective clock of a window is d(cycles) / d(ticks) x 100 MHz. bench.py reports it for its timed regions. */
int mpu_probe_clock(uint64_t* d_samples, int32_t n, int32_t naps, void* stream);

/* Test aid (no reference counterpart): when enabled, every convolution / weight-gradient launch appends one text
 * line naming the kernel schedule the dispatcher chose for the layer shape ("conv halo mode=0 B=.. H=.. W=.. Cin=..
 * Cout=.. dgrad=0", "wgrad taps ... ksplit=.."), so that parity tests at the BASELINE shapes can assert that the
 * schedules bench.py times are the ones they checked. mpu_schedule_log_enable(0|1) clears the log;
 * mpu_schedule_log_read copies it (NUL-terminated, truncated to cap) and returns its full length. */
int mpu_schedule_log_enable(int32_t on);
int64_t mpu_schedule_log_read(char* buf, int64_t cap);

/* Environment switches (csrc/env.h: the library's only getenv; schedule / fusion on-off pairs for A/B runs, dev aids). One line
 * per switch, "NAME\tvalue in force\tdefault\twhat it does", NUL-terminated and truncated to cap; returns the full length.
 * Values are read once per process, at the first query. */
int64_t mpu_env_describe(char* buf, int64_t cap);

/* Dev aid: with MPU_STAMPS=1 in the environment a few workgroups of the instrumented kernels (conv_halo8, wgrad_taps)
 * record s_memtime stamps at their phase boundaries (entry, prologue landed, main loop done, stores issued) into a
 * 64 x 8 table of uint64 (slot = a function of the workgroup index). Reads and clears it (synchronises the device). */
int mpu_debug_stamps_read(uint64_t* host_out, int32_t n);

/* Test aid (no reference counterpart): teacher-forced replay of a train step. While a tap is installed on a model,
 * mpu_unet_forward(training) / mpu_unet_backward call it right after enqueueing every convolution launch with the
 * device pointers and shapes of that launch, so that a test can synchronise the stream, copy the launch's OWN inputs
 * and output out of the workspace and compare the output with an independent fp64 convolution of exactly those
 * inputs -- launch by launch, without the error amplification of the train-mode network between them
 * (tests/test_gpu_replay.py). kind: 0 forward conv (out = relu(conv(in0|in1) + bias)), 1 data gradient
 * (out = conv^T(dz) restricted to input channels [n_off, n_off + n_cnt), times the ReLU mask 1[mask > 0] when mask
 * is set), 2 weight gradient (x = in0|in1, dz; the result lands in the flat gradient buffer at float offset w_off
 * once the backward pass has returned). H, W = resolution of the launch's OUTPUT (kind 0/1) or of dz (kind 2).
 * Round 4: the NON-convolution launches of the step report too (their inputs and outputs are final when the callback runs):
 *   kind 3  BatchNormalization forward, training (mpunet/models/unet.py:127,144,165,176): in0 = x [B,H,W,C0], out = y
 *           (= gamma * xhat + beta), mask = the 2x2 max-pooled y of an encoder level (or NULL), aux0 / aux1 = batch mean /
 *           1/sqrt(var + eps) (f32 [C0]), w_off / b_off = float offsets of gamma / beta in the flat parameter buffer;
 *   kind 4  BatchNormalization backward: in0 = dn, in1 = x (the layer's post-ReLU input), out = dz = 1[x > 0] * d(x),
 *           aux0 / aux1 = saved mean / invstd; dgamma / dbeta land at w_off / b_off of the flat GRADIENT buffer;
 *   kind 5  MaxPooling2D backward + skip add: in0 = n (the level's BN output), in1 = dskip, dz = dp (gradient of the pooled
 *           tensor [B,H/2,W/2,C0]), out = dn = dskip + unpool(dp) (routed to the FIRST maximum of each window);
 *   kind 6  1x1 head forward: in0 = n [B,H,W,C0], out = probabilities f32 [B,H,W,Cout] (softmax of n @ Wh + bh; Wh at w_off
 *           with row stride Cout, bh at b_off);
 *   kind 7  head backward (Keras sparse CE on clipped probabilities, sum gradient): in0 = n, in1 = probabilities (f32),
 *           dz = labels (u8 [B,H*W]), mask = per-image sample weights (f32 [B]) or NULL, out = dn [B,H,W,C0]; dWh / dbh land
 *           at w_off / b_off of the gradient buffer.
 * The callback runs on the calling thread; NULL removes the tap. Not for use under graph capture. */
typedef struct mpu_launch_info {
    int32_t kind, conv_index, mode, dtype;      /* mode: 0 3x3, 1 up-conv 2x2, 3 1x1 (the LAYER's mode); kinds >= 3: conv_index = BN index */
    int32_t B, H, W, C0, C1, Cout;              /* C0/C1: channels of in0/in1 (kind 1: C0 = channels of dz) */
    int32_t n_off, n_cnt, relu, _pad;
    const void* in0; const void* in1; const void* dz; const void* mask; const void* out;
    int64_t w_off, b_off;                       /* float offsets of the layer's kernel / bias in the flat buffers */
    const void* aux0; const void* aux1;         /* kinds 3, 4: batch mean, 1/sqrt(var + eps) */
} mpu_launch_info;
typedef void (*mpu_launch_tap_fn)(void* user, const mpu_launch_info* info);
int mpu_unet_set_launch_tap(mpu_unet* m, mpu_launch_tap_fn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* MPUNET_HIP_H */
