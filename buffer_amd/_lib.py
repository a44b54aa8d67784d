"""ctypes binding of libbuffer_hip.so (the C ABI declared in include/buffer_hip.h).

There is no CPU fallback: importing an operator without the built library, or calling one
without a HIP device, raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BUF_LIB_PATH: development aid (tools/ab.sh times two builds of the library in one GPU session); the default is the in-tree build
LIB_PATH = os.environ.get("BUF_LIB_PATH") or os.path.join(_HERE, "libbuffer_hip.so")

BUF_OK = 0


class BufferHipError(RuntimeError):
    pass


class buf_grid_t(C.Structure):
    _fields_ = [
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("ns", C.c_int), ("nb", C.c_int),
        ("cells_per_elem", C.c_int64),
        ("radius", C.c_float),
        ("supports", C.c_void_p),
        ("desc", C.c_void_p), ("s_off", C.c_void_p), ("table", C.c_void_p),
        ("sorted", C.c_void_p), ("order", C.c_void_p), ("scan_tmp", C.c_void_p),
    ]


_lib = None

_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

_SIGNATURES = {
    "buf_last_error": (C.c_char_p, []),
    "buf_version": (_i, []),
    "buf_device_count": (_i, []),
    "buf_timing_enable": (None, [_i]),
    "buf_timing_collect": (C.c_longlong, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "buf_timing_collect_kernel": (C.c_longlong, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "buf_grid_default_cells": (_i64, [_i, _i]),
    "buf_grid_ws_bytes": (_sz, [_i, _i, _i64]),
    "buf_grid_build": (_i, [C.POINTER(buf_grid_t), _vp, _i, _vp, _i, _f, _i64, _vp, _sz, _vp]),
    "buf_grid_query": (_i, [C.POINTER(buf_grid_t), _vp, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "buf_radius_neighbors": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "buf_grid_subsample_ws_bytes": (_sz, [_i, _i, _i64, _i]),
    "buf_grid_subsample_batch": (_i, [_vp, _i, _vp, _i, _f, _i, _vp, _i, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "buf_fps_ws_bytes": (_sz, [_i, _i]),
    "buf_fps": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "buf_gather": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "buf_group": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "buf_ball_query": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "buf_three_nn": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "buf_select_patches": (_i, [_vp, _vp, _i, _i, _f, _i, _vp, _vp]),
    "buf_select_patches_batched_ws_bytes": (_sz, [_i, _i]),
    "buf_select_patches_batched": (_i, [_vp, _vp, _i, _vp, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "buf_permute_clouds": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "buf_compact_ws_bytes": (_sz, [_i]),
    "buf_compact_greater": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "buf_knn_ws_bytes": (_sz, [_i, _i, _i]),
    "buf_knn1_ws_bytes": (_sz, [_i, _i, _i]),
    "buf_knn": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "buf_fps_ragged": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "buf_svd3x3_batched": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "buf_vn_gather_block": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "buf_vn_gather_pre_ws_bytes": (_sz, [_i, _i]),
    "buf_vn_gather_block_pre": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "buf_vn_pointwise": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "buf_gather_max": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "buf_vn_std": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "buf_score_head_ws_bytes": (_sz, [_i, _i]),
    "buf_score_head": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f,
                            _vp, _vp, _sz, _vp]),
    "buf_patch_voxelize_ws_bytes": (_sz, [_i]),
    "buf_patch_voxelize": (_i, [_vp, _vp, _i, _i, _f, _vp, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _sz, _vp]),
    "buf_cylindrical_net_wg": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "buf_cylindrical_net_wg_supports": (_i, [_vp, _vp]),
    "buf_cylindrical_net_split_safe": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "buf_cost_volume_net_split_safe": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "buf_cylindrical_net_split": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "buf_cylindrical_net_split_head": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "buf_split_gemm_count": (C.c_longlong, [_i, _i, _i, _i]),
    "buf_split_tile_gemm": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "buf_cost_volume_net_split": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "buf_cost_volume_net_split_gather": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "buf_split_filter_count": (C.c_longlong, [_i, _i]),
    "buf_split_tile_filters": (_i, [_vp, _i, _i, _vp]),
    "buf_winograd_tile_weights": (_i, [_vp, _i, _i, _vp]),
    "buf_winograd_group": (_i, [_i, _i]),
    "buf_cost_winograd_group": (_i, [_i]),
    "buf_winograd_tile_filters": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "buf_voxel_downsample_ws_bytes": (_sz, [_i, _i64]),
    "buf_voxel_downsample": (_i, [_vp, _vp, _i, _i, C.c_double, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "buf_voxel_downsample_batch_ws_bytes": (_sz, [_i, _i, _i64]),
    "buf_voxel_downsample_batch": (_i, [_vp, _i, _i, _vp, _i, C.c_double, _vp, _vp, _i64, _vp, _sz, _vp]),
    "buf_knn_normals": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "buf_row_linear": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "buf_segment_instance_norm_ws_bytes": (_sz, [_i, _i]),
    "buf_segment_instance_norm": (_i, [_vp, _i, _i, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "buf_descriptor_head": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "buf_cost_volume_net": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "buf_cost_volume_net_gather": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "buf_hypotheses_score": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "buf_ransac_ws_bytes": (_sz, [_i]),
    "buf_ransac_kabsch": (_i, [_vp, _vp, _vp, _i, _i, C.c_uint64, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "buf_ransac_masked_ws_bytes": (_sz, [_i, _i]),
    "buf_ransac_kabsch_masked": (_i, [_vp, _vp, _vp, _i, _i, C.c_uint64, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "buf_post_refine": (_i, [_vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp]),
    "buf_recover_poses_ws_bytes": (_sz, [_i, _i, _i]),
    "buf_recover_poses_batched": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _i, _f, _f, _f, _i, _vp, _vp, _sz, _vp]),
}


def exported_symbols():
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BufferHipError(
                f"{LIB_PATH} is missing: build it with `python -m buffer_amd.build` "
                "(hipcc --offload-arch=gfx950). buffer_amd has no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  Loading it first makes
        # our library bind to that same runtime instead of pulling a second one from /opt/rocm, so
        # torch's allocator/streams and our kernels share one context.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != BUF_OK:
        msg = lib().buf_last_error().decode(errors="replace")
        raise BufferHipError(f"{what} failed ({rc}): {msg}")
