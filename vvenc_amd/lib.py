"""ctypes binding of libvvenc_hip.so — one prototype per symbol declared in include/vvenc_hip.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VVHIP_LIB") or os.path.join(HERE, "libvvenc_hip.so")          # ($VVHIP_LIB: A/B measurements against another build of the library)


class VVHipError(RuntimeError):
    pass


vp, i32, u64p, sz = C.c_void_p, C.c_int, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); the list doubles as the export check in tests/test_abi.py
PROTOTYPES = {
    "vvhip_create": (i32, [C.POINTER(vp), i32]),
    "vvhip_destroy": (None, [vp]),
    "vvhip_last_error": (C.c_char_p, [vp]),
    "vvhip_set_stream": (i32, [vp, vp]),
    "vvhip_use_own_stream": (i32, [vp]),
    "vvhip_get_stream": (vp, [vp]),
    "vvhip_sync": (i32, [vp]),
    "vvhip_sync_all_devices": (i32, [vp]),
    "vvhip_set_blocking_sync": (i32, [vp, i32]),
    "vvhip_malloc": (i32, [vp, C.POINTER(vp), sz]),
    "vvhip_free": (i32, [vp, vp]),
    "vvhip_upload": (i32, [vp, vp, vp, sz]),
    "vvhip_download": (i32, [vp, vp, vp, sz]),
    "vvhip_download_async": (i32, [vp, vp, vp, sz]),
    "vvhip_upload_2d": (i32, [vp, vp, sz, vp, sz, sz, sz]),
    "vvhip_download_2d": (i32, [vp, vp, sz, vp, sz, sz, sz]),
    "vvhip_host_register": (i32, [vp, vp, sz]),
    "vvhip_host_unregister": (i32, [vp, vp]),
    "vvhip_host_alloc": (i32, [vp, C.POINTER(vp), sz]),
    "vvhip_host_free": (i32, [vp, vp]),
    "vvhip_event_create": (i32, [vp, C.POINTER(vp)]),
    "vvhip_event_record": (i32, [vp, vp]),
    "vvhip_event_wait": (i32, [vp, vp]),
    "vvhip_event_destroy": (i32, [vp, vp]),
    "vvhip_device_count": (i32, []),
    "vvhip_get_device": (i32, [vp]),
    "vvhip_make_current": (i32, [vp]),
    "vvhip_copy_peer": (i32, [vp, vp, vp, vp, sz]),
    "vvhip_version": (C.c_char_p, []),
    "vvhip_dist_batch": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "vvhip_dist_multi": (i32, [vp, i32, vp, i32, vp, i32, i32, vp, i32]),
    "vvhip_dist_multi_func": (i32, [vp, vp, i32, vp, i32, i32, vp, i32]),
    "vvhip_tiled8_elems": (sz, [i32, i32]),
    "vvhip_plane_tile8": (i32, [vp, vp, i32, i32, vp]),
    "vvhip_plane_shift1": (i32, [vp, vp, C.c_size_t, vp]),
    "vvhip_planes_derive": (i32, [vp, vp, i32, i32, vp, vp, i32, i32, vp, vp]),
    "vvhip_dist_multi_func_tiled": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, i32]),
    "vvhip_sad_mask_batch": (i32, [vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "vvhip_fix_weighted_sse_batch": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp]),
    "vvhip_sad_x5_batch": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "vvhip_sad_surface": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "vvhip_fwd_transform_batch": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "vvhip_inv_transform_batch": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "vvhip_quant_batch": (i32, [vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "vvhip_dequant_batch": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "vvhip_need_rdoq_batch": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "vvhip_dequant_core": (i32, [vp, i32, i32, i32, vp, sz, vp, i32, i32, i32]),
    "vvhip_quant_core": (i32, [vp, vp, i32, i32, i32, i32, C.c_int64, i32, vp, vp, vp, vp]),
    "vvhip_quant_core_lfnst": (i32, [vp, vp, i32, i32, i32, i32, C.c_int64, i32, i32, vp, vp, vp, vp]),
    "vvhip_need_rdoq_core": (i32, [vp, vp, sz, i32, C.c_int64, i32, vp]),
    "vvhip_tu_rdo_batch": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp]),
    "vvhip_tu_rdo_multi": (i32, [vp, vp, i32, i32, vp, i32]),
    "vvhip_tu_rdo_multi_strided": (i32, [vp, vp, vp, i32, vp, i32]),
    "vvhip_me_plan_create": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, C.POINTER(vp)]),
    "vvhip_me_plan_create_lists": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
    "vvhip_me_plan_destroy": (None, [vp, vp]),
    "vvhip_me_plan_run": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "vvhip_me_plan_run_parts": (i32, [vp, vp, vp, i32, vp, vp, vp, i32]),
    "vvhip_me_plan_info": (i32, [vp, vp, vp, vp, vp]),
    "vvhip_me_plan_set_timing": (i32, [vp, vp, i32]),
    "vvhip_me_plan_last_times": (i32, [vp, vp, vp]),
    "vvhip_fast_fwd_core": (i32, [vp, i32, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, i32]),
    "vvhip_fast_inv_core": (i32, [vp, i32, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint]),
    "vvhip_round_clip": (i32, [vp, vp, C.c_uint, C.c_uint, C.c_uint, i32, i32, i32, i32]),
    "vvhip_cpy_resi": (i32, [vp, vp, vp, C.c_ssize_t, C.c_uint, C.c_uint]),
    "vvhip_cpy_coeff": (i32, [vp, vp, C.c_ssize_t, vp, C.c_uint, C.c_uint]),
    "vvhip_if_filter": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, i32, vp]),
    "vvhip_if_copy": (i32, [vp, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32]),
    "vvhip_interp_luma_batch": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vvhip_subpel_dist_batch": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "vvhip_mctf_apply_plane": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, C.c_double, C.c_double, vp, i32]),
    "vvhip_mctf_filter_params": (i32, [i32, i32, C.c_double, i32, vp, vp]),
    "vvhip_dmvr_refine_batch": (i32, [vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "vvhip_alf_classify": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "vvhip_ccalf_stats_plane": (i32, [vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "vvhip_alf_stats_plane_units": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp]),
    "vvhip_alf_stats_plane": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp]),
    "vvhip_graph_begin": (i32, [vp]),
    "vvhip_graph_end": (i32, [vp, vp]),
    "vvhip_graph_launch": (i32, [vp, vp]),
    "vvhip_graph_destroy": (None, [vp]),
    "vvhip_alf_filter_plane": (i32, [vp, vp, C.c_ssize_t, vp, C.c_ssize_t, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32]),
    "vvhip_ccalf_filter_plane": (i32, [vp, vp, C.c_ssize_t, vp, C.c_ssize_t, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32]),
    "vvhip_subpel_refine_batch": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp]),
    "vvhip_get_tr_matrix_host": (i32, [i32, i32, vp]),
    "vvhip_get_scan_order_host": (i32, [i32, i32, vp]),
    "vvhip_get_me_tap_tables_host": (i32, [i32, vp]),
    "vvhip_mctf_error_batch": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "vvhip_mctf_calc_var_batch": (i32, [vp, vp, i32, i32, i32, vp, i32, vp]),
    "vvhip_mctf_subsample": (i32, [vp, vp, i32, i32, i32, vp, i32, i32]),
    "vvhip_extend_border": (i32, [vp, vp, i32, i32, i32, i32]),
    "vvhip_mctf_init_mvs": (i32, [vp, vp, i32]),
    "vvhip_mctf_me_level": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32]),
    "vvhip_mctf_motion_estimation": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vvhip_mctf_motion_estimation_async": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vvhip_tu_set_sparse_outputs": (i32, [vp, i32]),
    "vvhip_mctf_set_stats": (i32, [vp, i32]),
    "vvhip_mctf_get_stats": (i32, [vp, vp]),
    "vvhip_mctf_set_timing": (i32, [vp, i32]),
    "vvhip_mctf_last_times": (i32, [vp, vp]),
}

_lib = None


def load_library(path=None):
    """Loads libvvenc_hip.so (built in-tree by `make -C vvenc_amd/csrc` / __graft_entry__.build()).
    Raises VVHipError when it is missing: there is no fallback implementation."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # torch ships its own libamdhip64; it must be the process' (single) HIP runtime BEFORE our library is loaded so that
    # device pointers and streams are shared.  (A pure C/C++ host links the system runtime instead and never sees torch.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise VVHipError("%s not found - build it with `make -C vvenc_amd/csrc` (hipcc --offload-arch=gfx950); "
                         "vvenc_amd has no CPU fallback" % p)
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI/header drift
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
