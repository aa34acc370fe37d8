"""
ctypes binding of libmpunet_hip.so (include/mpunet_hip.h). There is no CPU
fallback anywhere in this package: if the shared library is missing the import
of any compute entry point raises.
"""
import ctypes as C
import os

# torch bundles its own HIP runtime (libamdhip64); it must be the one already loaded when
# libmpunet_hip.so resolves its dependency, so that streams / device pointers are shared.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPU_LIB_PATH") or os.path.join(_HERE, "lib", "libmpunet_hip.so")   # (override: A/B of two builds)

c_p = C.c_void_p
i32, i64, f32, f64, u8 = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint8

MPU_F32, MPU_BF16, MPU_F32X3 = 0, 1, 2


class Axis(C.Structure):                # mpu_axis
    _fields_ = [("kind", i32), ("n", i32), ("start", f64), ("step", f64), ("last", f64)]


def make_axis(values):
    """mpu_axis for a host f64 axis array: picks a closed form only if it reproduces the array bit for bit."""
    import numpy as np
    v = np.ascontiguousarray(values, dtype=np.float64)
    n = int(v.shape[0])
    a = Axis()
    a.n = n
    a.kind = 0
    a.step = float(v[1] - v[0]) if n > 1 else 1.0
    a.start = float(v[0])
    a.last = float(v[-1])
    if n > 2:
        i = np.arange(n, dtype=np.float64)
        step1 = (v[-1] - v[0]) / float(n - 1)                     # np.linspace: arange*step + start, last = stop
        lin = i * step1 + v[0]
        lin[-1] = v[-1]
        if np.array_equal(lin, v):
            a.kind, a.start, a.step, a.last = 1, float(v[0]), float(step1), float(v[-1])
            return a
        half = (n - 1) / 2.0                                       # voxel axes: (arange(n) - (n-1)/2) * pixdim
        pd = (v[-1] - v[0]) / float(n - 1)
        bases = (pd, float(v[1] - v[0]), float(v[-1] / half) if half else pd, float(v[0] / -half) if half else pd)
        for base in bases:                                         # the quotients can miss pixdim by an ulp: try the neighbours too
            for cand in (base, float(np.nextafter(base, np.inf)), float(np.nextafter(base, -np.inf))):
                if np.array_equal((i - half) * cand, v):
                    a.kind, a.start, a.step, a.last = 2, float(half), float(cand), float(v[-1])
                    return a
    return a


class ViewGeom(C.Structure):            # mpu_view_geom
    _fields_ = [("basis", f64 * 9), ("rot", f64 * 9), ("has_rot", i32), ("dim", i32),
                ("n_planes", i32), ("_pad", i32), ("g_start", f64), ("g_step", f64), ("vol_axis", Axis * 3)]


class ViewPred(C.Structure):            # mpu_view_pred
    _fields_ = [("inv_basis", f64 * 9), ("d_pred", c_p), ("d_g", c_p), ("d_offsets", c_p),
                ("dim", i32), ("n_planes", i32), ("g_axis", Axis), ("o_axis", Axis)]


class VoxelGrid(C.Structure):           # mpu_voxel_grid
    _fields_ = [("A", f64 * 9), ("center", f64 * 3), ("shape", i32 * 3), ("_pad", i32)]


class UNetConfig(C.Structure):         # mpu_unet_config
    _fields_ = [("n_classes", i32), ("n_channels", i32), ("depth", i32), ("H", i32), ("W", i32),
                ("dtype", i32), ("softmax", i32), ("filters", i32 * 8)]


class MpuError(RuntimeError):
    pass


_SIGS = {
    "mpu_abi_version": (C.c_int, []),
    "mpu_build_hash": (C.c_char_p, []),
    "mpu_last_error": (C.c_char_p, []),
    "mpu_sample_view_planes": (C.c_int, [c_p, c_p, C.POINTER(i32), c_p, c_p, c_p,
                                         C.POINTER(ViewGeom), c_p, c_p, u8, c_p, c_p,
                                         c_p, c_p, c_p]),
    "mpu_map_view_nearest": (C.c_int, [C.POINTER(VoxelGrid), C.POINTER(ViewPred), i32, c_p, c_p]),
    "mpu_map_accumulate_view": (C.c_int, [C.POINTER(VoxelGrid), C.POINTER(ViewPred), i32, c_p,
                                          i32, i32, i32, c_p, c_p]),
    "mpu_map_fuse_views": (C.c_int, [C.POINTER(VoxelGrid), C.POINTER(ViewPred), i32, i32,
                                     c_p, c_p, i32, c_p, c_p, c_p]),
    "mpu_fusion_forward": (C.c_int, [c_p, i64, i32, i32, c_p, c_p, c_p, c_p, c_p]),
    "mpu_fusion_finalize": (C.c_int, [c_p, i64, i32, c_p, i32, c_p, c_p, c_p]),
    "mpu_unet_create": (c_p, [C.POINTER(UNetConfig)]),
    "mpu_unet_destroy": (None, [c_p]),
    "mpu_unet_param_floats": (i64, [c_p]),
    "mpu_unet_bn_state_floats": (i64, [c_p]),
    "mpu_unet_packed_bytes": (i64, [c_p]),
    "mpu_unet_logical_param_count": (i64, [c_p]),
    "mpu_unet_workspace_bytes": (i64, [c_p, i32]),
    "mpu_unet_num_tensors": (i32, [c_p]),
    "mpu_unet_tensor_info": (C.c_int, [c_p, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64),
                                       C.POINTER(i32), C.POINTER(i32)]),
    "mpu_unet_workspace_probs_offset": (i64, [c_p, i32]),
    "mpu_unet_workspace_loss_mean_offset": (i64, [c_p, i32]),
    "mpu_unet_pack_weights": (C.c_int, [c_p, c_p, c_p, c_p]),
    "mpu_unet_prepare_inference": (C.c_int, [c_p, c_p, c_p, c_p, c_p]),
    "mpu_unet_forward": (C.c_int, [c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, c_p, c_p]),
    "mpu_unet_backward": (C.c_int, [c_p, i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpu_adam_step": (C.c_int, [c_p, c_p, c_p, c_p, i64, i64, f64, f64, f64, f64, c_p]),
    "mpu_plane_stats": (C.c_int, [c_p, c_p, i64, C.c_int32, c_p, c_p, c_p]),
    "mpu_elastic_workspace_doubles": (C.c_int64, [C.c_int32, C.c_int32]),
    "mpu_elastic_transform_2d": (C.c_int, [c_p, c_p, C.c_int32, C.c_int32, C.c_int32, c_p, c_p, C.c_int32, f64, c_p, c_p,
                                           c_p, c_p, c_p]),
    "mpu_fusion_train_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "mpu_fusion_train_step": (C.c_int, [c_p, c_p, i64, C.c_int32, C.c_int32, c_p, c_p, c_p, c_p, i64, f64, f64, f64, f64,
                                        c_p, c_p, c_p, c_p]),
    "mpu_fusion_grad_sums": (C.c_int, [c_p, c_p, i64, C.c_int32, C.c_int32, c_p, c_p, c_p, c_p, c_p]),
    "mpu_fusion_apply_sums": (C.c_int, [c_p, C.c_int32, C.c_int32, c_p, c_p, c_p, c_p, i64, f64, f64, f64, f64, c_p, c_p, c_p]),
    "mpu_unet_grad_ready_points": (C.c_int32, [c_p, c_p, C.c_int32]),
    "mpu_unet_backward_events": (C.c_int, [c_p, C.c_int32] + [c_p] * 8 + [c_p, C.c_int32, c_p]),
    "mpu_adam_step_device_counter": (C.c_int, [c_p, c_p, c_p, c_p, i64, c_p, f64, f64, f64, f64, c_p]),
    "mpu_unet_l2_regularizer": (C.c_int, [c_p, c_p, c_p, f64, c_p, c_p, c_p]),
    "mpu_unet_l2_workspace_doubles": (i64, []),
    "mpu_conv2d_pack_weights": (C.c_int, [i32, i32, c_p, i32, i32, c_p, c_p, c_p]),
    "mpu_conv2d_igemm": (C.c_int, [i32, i32, c_p, i32, c_p, i32, c_p, i64, i32, c_p, c_p, c_p,
                                   i32, i32, i32, i32, i32, c_p]),
    "mpu_conv2d_igemm_ws": (C.c_int, [i32, i32, c_p, i32, c_p, i32, c_p, i64, i32, c_p, c_p, c_p,
                                      i32, i32, i32, i32, i32, c_p, i64, c_p]),
    "mpu_conv2d_wgrad_workspace_floats": (i64, [i32, i32, i32, i64]),
    "mpu_conv2d_wgrad_job_floats": (i64, [i32] * 9),
    "mpu_conv2d_wgrad_scratch_floats": (i64, [i32] * 8),
    "mpu_conv2d_wgrad_first_layer_workspace_floats": (i64, [i32, i64]),
    "mpu_conv2d_wgrad_first_layer": (C.c_int, [i32, c_p, i32, c_p, i32, i32, i32, i32, c_p, c_p, c_p, c_p]),
    "mpu_profile_enable": (C.c_int, [i32]),
    "mpu_profile_summary": (C.c_int, [i32, C.POINTER(f64), C.POINTER(f64), C.POINTER(i64)]),
    "mpu_validation_count": (C.c_int, [c_p, c_p, i64, i32, c_p, c_p]),
    "mpu_geometry_set_fast_path": (C.c_int, [i32]),
    "mpu_geometry_check_cell_division": (C.c_int, [C.POINTER(Axis), i64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mpu_probe_mfma_bf16": (C.c_int, [i32, i32, c_p, C.POINTER(f64), c_p]),
    "mpu_probe_mfma_bf16_random": (C.c_int, [i32, i32, c_p, C.POINTER(f64), c_p]),
    "mpu_probe_stream_triad": (C.c_int, [c_p, c_p, c_p, i64, c_p]),
    "mpu_probe_stream_copy": (C.c_int, [c_p, c_p, i64, i32, c_p]),
    "mpu_probe_permuted_read": (C.c_int, [c_p, c_p, i64, i32, c_p]),
    "mpu_probe_gather12": (C.c_int, [c_p, c_p, i64, c_p]),
    "mpu_probe_clock": (C.c_int, [c_p, i32, i32, c_p]),
    "mpu_schedule_log_enable": (C.c_int, [i32]),
    "mpu_schedule_log_read": (i64, [C.c_char_p, i64]),
    "mpu_env_describe": (i64, [C.c_char_p, i64]),
    "mpu_conv2d_wgrad": (C.c_int, [i32, i32, c_p, i32, c_p, i32, c_p, i32, i32, i32, i32, c_p, c_p, c_p]),
    "mpu_unet_set_launch_tap": (C.c_int, [c_p, c_p, c_p]),
    "mpu_unet_adam_pack": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, c_p, f64, f64, f64, f64, c_p, c_p]),
    "mpu_debug_stamps_read": (C.c_int, [c_p, i32]),
    "mpu_debug_tail_events": (C.c_int, [i32, c_p]),
    "mpu_sample_plane_stats": (C.c_int, [c_p, c_p, C.POINTER(i32), c_p, c_p, c_p, C.POINTER(ViewGeom), c_p, c_p, u8, c_p, c_p,
                                        c_p, c_p, c_p, c_p, c_p]),
    "mpu_unet_backward_adam": (C.c_int, [c_p, i32] + [c_p] * 10 + [i64, c_p, f64, f64, f64, f64, c_p]),
}


class LaunchInfo(C.Structure):
    """mpu_launch_info (include/mpunet_hip.h): one convolution launch of a tapped train step (test aid)."""
    _fields_ = [("kind", i32), ("conv_index", i32), ("mode", i32), ("dtype", i32),
                ("B", i32), ("H", i32), ("W", i32), ("C0", i32), ("C1", i32), ("Cout", i32),
                ("n_off", i32), ("n_cnt", i32), ("relu", i32), ("_pad", i32),
                ("in0", c_p), ("in1", c_p), ("dz", c_p), ("mask", c_p), ("out", c_p),
                ("w_off", i64), ("b_off", i64), ("aux0", c_p), ("aux1", c_p)]


LAUNCH_TAP_FN = C.CFUNCTYPE(None, c_p, C.POINTER(LaunchInfo))

_lib = None


def declared_symbols():
    """Every extern "C" symbol include/mpunet_hip.h declares."""
    return sorted(_SIGS)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpuError(
                "libmpunet_hip.so not found at %s -- build it with "
                "`python -m multiplanarunet_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)         # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _check_build_hash(lib)
        _lib = lib
    return _lib


def build_hash():
    """The source hash the loaded library was built from (`mpu_build_hash()`, compiled in by build.py)."""
    return (load().mpu_build_hash() or b"").decode()


def _check_build_hash(lib):
    """A library that was not built from the sources beside it is refused (a stale .so used to run unnoticed: build() compared
    mtimes). Skipped when the sources are absent (an installed copy) and for an explicit MPU_LIB_PATH (A/B of two builds)."""
    from . import srchash
    if os.environ.get("MPU_LIB_PATH") or not srchash.sources_present():
        return
    from .build import expected_hash
    have, want = (lib.mpu_build_hash() or b"").decode(), expected_hash()
    if have != want:
        raise MpuError("libmpunet_hip.so at %s was built from other sources (library %s, tree %s): rebuild it with "
                       "`python -m multiplanarunet_amd.build`" % (LIB_PATH, have, want))


def check(status, what):
    if status != 0:
        msg = load().mpu_last_error()
        raise MpuError("%s failed (%d): %s" % (what, status, (msg or b"").decode()))


def call(name, *args):
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
