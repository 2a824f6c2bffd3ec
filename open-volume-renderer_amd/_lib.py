"""ctypes binding of the C ABI in include/ovr_hip.h (libovr_hip.so, built for gfx950 by csrc/Makefile).

There is no CPU fallback: if the library is missing, or no MI355X is present when a renderer is created, the
call fails loudly (RuntimeError), exactly like the reference's device throws std::runtime_error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OVR_HIP_LIBRARY: development override (A/B runs of differently built kernels); the default is the in-tree build
LIB_PATH = os.environ.get("OVR_HIP_LIBRARY") or os.path.join(_HERE, "libovr_hip.so")

# numeric values of ovr::ValueType (ovr/scene.h:32-53)
TYPE_UINT8, TYPE_INT8 = 100, 101
TYPE_UINT16, TYPE_INT16 = 200, 201
TYPE_UINT32, TYPE_INT32 = 300, 301
TYPE_FLOAT, TYPE_DOUBLE = 400, 500
MEM_HOST, MEM_DEVICE = 0, 1
SHADE_NONE, SHADE_GRADIENT, SHADE_FULL = 0, 1, 2
GRID_CELL_CENTRED, GRID_VERTEX_CENTRED = 0, 1
PIPELINE_AUTO, PIPELINE_IN_PLACE, PIPELINE_POOLED = 0, 1, 2
JITTER_TEA, JITTER_BLUE_NOISE = 0, 1
LAYOUT_AUTO, LAYOUT_GENERAL, LAYOUT_THIN, LAYOUT_THIN_T, LAYOUT_QUAD = -1, 0, 1, 2, 3
# the ABI these ctypes structures describe: load() refuses a library of another version (ovr_hip_get_stats would write past them)
EXPECTED_ABI = 10


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("samples", C.c_uint64),
        ("shaded_samples", C.c_uint64),
        ("shadow_samples", C.c_uint64),
        ("active_pixels", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("render_ms", C.c_double),
        ("frame_index", C.c_int32),
        ("pipeline", C.c_int32),
        ("march_ms", C.c_double),
        ("shade_ms", C.c_double),
        ("composite_ms", C.c_double),
        ("pool_chunks", C.c_uint64),
        ("skipped_samples", C.c_uint64),
        ("skipped_shadow_samples", C.c_uint64),
        ("layout", C.c_int32),
        ("stale_tiles", C.c_int32),
        ("lds_fallback_taps", C.c_uint64),
        ("lds_unstaged_rounds", C.c_uint64),
        ("lds_rounds", C.c_uint64),
        ("skipping_kernels", C.c_int32),
        ("tuning", C.c_int32),
        ("replicas_building", C.c_int32),
    ]


class VolumeInfo(C.Structure):
    _fields_ = [
        ("dims", C.c_int32 * 3),
        ("value_type", C.c_int32),
        ("resident_bytes", C.c_uint64),
        ("data_lower", C.c_float),
        ("data_upper", C.c_float),
        ("tf_lower", C.c_float),
        ("tf_upper", C.c_float),
    ]


# every symbol include/ovr_hip.h declares: name -> (restype, argtypes)
_F3 = C.POINTER(C.c_float)
_H = C.c_void_p
SYMBOLS = {
    "ovr_hip_last_error": (C.c_char_p, []),
    "ovr_hip_abi_version": (C.c_int, []),
    "ovr_hip_create": (C.c_int, [C.POINTER(_H), C.c_int]),
    "ovr_hip_destroy": (None, [_H]),
    "ovr_hip_create_group": (C.c_int, [C.POINTER(_H), C.POINTER(C.c_int32), C.c_int32]),
    "ovr_hip_group_info": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "ovr_hip_group_host_times": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "ovr_hip_get_upload_times": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "ovr_hip_pow_floats": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ovr_hip_built_for_exact_parity": (C.c_int, []),
    "ovr_hip_get_member_stats": (C.c_int, [_H, C.c_int32, C.POINTER(Stats)]),
    "ovr_hip_rccl_selftest": (C.c_int, [C.c_int]),
    "ovr_hip_set_stream": (C.c_int, [_H, C.c_void_p]),
    "ovr_hip_set_volume": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), _F3, _F3]),
    "ovr_hip_set_grid_convention": (C.c_int, [_H, C.c_int]),
    "ovr_hip_set_transfer_function": (C.c_int, [_H, _F3, C.c_int32, _F3, C.c_int32, C.c_float, C.c_float]),
    "ovr_hip_set_camera": (C.c_int, [_H, _F3, _F3, _F3, C.c_float]),
    "ovr_hip_set_fbsize": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "ovr_hip_set_sample_per_pixel": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_volume_sampling_rate": (C.c_int, [_H, C.c_float]),
    "ovr_hip_set_frame_accumulation": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_sparse_sampling": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_focus": (C.c_int, [_H, C.c_float, C.c_float, C.c_float, C.c_float]),
    "ovr_hip_set_noise_tile": (C.c_int, [_H, _F3, C.c_int32]),
    "ovr_hip_set_shading": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_shading_pipeline": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_empty_space_skipping": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_get_macrocells": (C.c_int, [_H, C.POINTER(C.c_int32), _F3, _F3, C.c_size_t]),
    "ovr_hip_set_image_shard": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "ovr_hip_commit": (C.c_int, [_H]),
    "ovr_hip_render": (C.c_int, [_H]),
    "ovr_hip_render_async": (C.c_int, [_H]),
    "ovr_hip_sync": (C.c_int, [_H]),
    "ovr_hip_mapframe": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ovr_hip_swap": (C.c_int, [_H]),
    "ovr_hip_render_time_ms": (C.c_double, [_H]),
    "ovr_hip_get_stats": (C.c_int, [_H, C.POINTER(Stats)]),
    "ovr_hip_owned_tiles": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int32)]),
    "ovr_hip_pack_tiles": (C.c_int, [_H, C.c_void_p, C.c_size_t]),
    "ovr_hip_unpack_tiles": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "ovr_hip_mapframe_rgba8": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ovr_hip_unpack_all_tiles": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]),
    "ovr_hip_sparse_mask": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]),
    "ovr_hip_tea_floats": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64]),
    "ovr_hip_set_pixel_jitter": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_lds_staging": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_phase_timing": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_query_addressing_mode": (C.c_int, [C.POINTER(C.c_int32), C.c_int, C.c_int32, C.c_int32, C.c_int32]),
    "ovr_hip_set_volume_layouts": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_set_layout_choice": (C.c_int, [_H, C.c_int32]),
    "ovr_hip_get_volume_info": (C.c_int, [_H, C.POINTER(VolumeInfo)]),
    "ovr_hip_mapframe_rgba16f": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
}

_lib = None


def load():
    """dlopen libovr_hip.so and bind every declared symbol; raises RuntimeError if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found - build it with `make -C open-volume-renderer_amd/csrc` "
            "(or __graft_entry__.build()); this backend has no CPU fallback")
    # torch ships its own copy of the HIP runtime: if libovr_hip.so brings up /opt/rocm's first, a later torch.cuda
    # initialisation in the same process finds "No HIP GPUs" (seen on the GPU box).  Loading torch's libraries first makes
    # both share one runtime; hosts without torch (the C++ plugin) have only one runtime anyway.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libovr_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    abi = lib.ovr_hip_abi_version()
    if abi != EXPECTED_ABI:
        raise RuntimeError(f"{LIB_PATH} implements ABI version {abi}, this Python package describes version {EXPECTED_ABI}: rebuild the "
                           "library (make -C open-volume-renderer_amd/csrc) or update the package")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().ovr_hip_last_error()
        raise RuntimeError((msg or b"").decode("utf-8", "replace") or f"ovr_hip error {code}")
