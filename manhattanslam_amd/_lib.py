"""ctypes binding of include/msl.h.  Fails loudly when libmsl.so has not been built."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSL_LIB", os.path.join(_HERE, "libmsl.so"))   # MSL_LIB: kernel-variant experiments only


class MslError(RuntimeError):
    pass


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
SURFEL_DTYPE = np.dtype([("px", "<f4"), ("py", "<f4"), ("pz", "<f4"), ("nx", "<f4"), ("ny", "<f4"),
                         ("nz", "<f4"), ("size", "<f4"), ("color", "<f4"), ("r", "<i4"), ("g", "<i4"),
                         ("b", "<i4"), ("weight", "<f4"), ("updateTimes", "<i4"), ("lastUpdate", "<i4")])
SEED_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("normX", "<f4"), ("normY", "<f4"),
                       ("normZ", "<f4"), ("posX", "<f4"), ("posY", "<f4"), ("posZ", "<f4"),
                       ("viewCos", "<f4"), ("meanDepth", "<f4"), ("meanIntensity", "<f4"), ("r", "<i4"),
                       ("g", "<i4"), ("b", "<i4"), ("fused", "u1"), ("stable", "u1"), ("use", "u1"),
                       ("_pad", "u1")])
PEAC_STATS_DTYPE = np.dtype([(n, "<f8") for n in ("sx", "sy", "sz", "sxx", "syy", "szz", "sxy", "syz", "sxz")] + [("N", "<i4"), ("nouse", "<i4")])
PEAC_PARAMS_DTYPE = np.dtype([(n, "<i4") for n in ("window_w", "window_h", "min_support", "max_step", "do_refine", "erode_type", "init_loose", "_pad")] +
                             [(n, "<f8") for n in ("depth_sigma", "std_tol_init", "std_tol_merge", "z_near", "z_far", "angle_near", "angle_far",
                                                   "similarity_th_merge", "similarity_th_refine", "depth_alpha", "depth_change_tol")])
PEAC_BLOCK_DTYPE = np.dtype([(n, "<f8") for n in ("sx", "sy", "sz", "sxx", "syy", "szz", "sxy", "syz", "sxz")] + [("N", "<i4"), ("nouse", "<i4")] +
                            [("center", "<f8", (3,)), ("normal", "<f8", (3,)), ("mse", "<f8"), ("curvature", "<f8")])
PEAC_PLANE_DTYPE = np.dtype([("normal", "<f8", (3,)), ("center", "<f8", (3,)), ("mse", "<f8"), ("N", "<i4"), ("_pad", "<i4")])   # msl_peac_plane
FRAME_PARAMS_DTYPE = np.dtype([(n, "<f4") for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf", "minX", "maxX", "minY", "maxY")])
MATCH_PARAMS_DTYPE = np.dtype([(n, "<f4") for n in ("fx", "fy", "cx", "cy", "bf", "minX", "maxX", "minY", "maxY", "th")] +
                              [("check_orientation", "<i4"), ("nlevels", "<i4"), ("scale_factors", "<f4", (16,))])
assert KEYPOINT_DTYPE.itemsize == 28 and SURFEL_DTYPE.itemsize == 56 and SEED_DTYPE.itemsize == 64

MSL_MEM_HOST, MSL_MEM_DEVICE = 0, 1

# name -> (restype, argtypes); every symbol include/msl.h declares
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "msl_last_error": (C.c_char_p, []),
    "msl_version": (C.c_char_p, []),
    "msl_device_count": (_i, []),
    "msl_orb_create": (_vp, [_i, _f, _i, _i, _i, _i, _i, _i, _i]),
    "msl_orb_destroy": (None, [_vp]),
    "msl_orb_scale_tables": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "msl_orb_features_per_level": (_i, [_vp, _vp]),
    "msl_orb_capacity": (_i, [_vp]),
    "msl_orb_levels": (_i, [_vp]),
    "msl_orb_extract": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _i, _vp]),
    "msl_orb_extract_batch": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _vp, _vp, _i, _vp, _i]),
    "msl_frame_image_bounds": (_i, [_vp, _i, _i]),
    "msl_orb_extract_frame_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _sz, _sz, _sz, _sz, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i]),
    "msl_orb_sync": (_i, [_vp]),
    "msl_orb_set_stream": (_i, [_vp, _vp]),
    "msl_orb_debug_level_size": (_i, [_vp, _i, _vp, _vp]),
    "msl_orb_debug_level": (_i, [_vp, _i, _i, _i, _vp]),
    "msl_orb_debug_candidates": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "msl_orb_profile_enable": (_i, [_vp, _i]),
    "msl_orb_profile_read": (_i, [_vp, _vp, _vp]),
    "msl_orb_kernel_name": (C.c_char_p, [_i]),
    "msl_sf_create": (_vp, [_i, _i, _f, _f, _f, _f, _f, _f, _i]),
    "msl_sf_destroy": (None, [_vp]),
    "msl_sf_fuse": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _vp]),
    "msl_sf_fuse_ex": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _vp, C.c_uint]),
    "msl_sf_map_reserve": (_i, [_vp, _sz]),
    "msl_sf_map_upload": (_i, [_vp, _vp, _sz]),
    "msl_sf_map_download": (_i, [_vp, _vp, _sz, _vp]),
    "msl_sf_map_size": (_i, [_vp, _vp]),
    "msl_sf_map_snapshot": (_i, [_vp]),
    "msl_sf_map_restore": (_i, [_vp]),
    "msl_sf_map_detach": (_i, [_vp, _i, _vp, _sz, _vp]),
    "msl_sf_map_append": (_i, [_vp, _vp, _sz]),
    "msl_sf_map_export": (_i, [_vp, _i, _vp, _sz, _vp]),
    "msl_sf_export_ply": (_i, [_vp, _i, _vp, _sz, C.c_char_p]),
    "msl_peac_default_params": (None, [_vp]),
    "msl_peac_block_fit": (_i, [_i, _vp, _sz, _sz, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _i]),
    "msl_peac_membership_batch": (_i, [_i, _vp, _sz, _sz, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    "msl_peac_membership_from_blocks": (_i, [_vp, _vp, _sz, _sz, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    "msl_peac_extract_batch": (_i, [_i, _vp, _sz, _sz, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "msl_peac_extract_from_blocks": (_i, [_vp, _vp, _sz, _sz, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "msl_peac_block_stats": (_i, [_i, _vp, _sz, _sz, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _i, C.c_double, C.c_double, _i, _vp, _vp, _i]),
    "msl_match_by_projection_batch": (_i, [_i, _i, _i] + [_vp] * 15 + [_i, _vp, _vp, _i]),
    "msl_match_descriptor_distance": (_i, [_i, _vp, _vp, _i, _vp]),
    "msl_match_create": (_vp, [_i]),
    "msl_match_destroy": (None, [_vp]),
    "msl_match_sync": (_i, [_vp]),
    "msl_match_set_stream": (_i, [_vp, _vp]),
    "msl_match_by_projection": (_i, [_vp, _i, _i] + [_vp] * 15 + [_i, _vp, _vp, _i]),
    "msl_match_descriptor_distances": (_i, [_vp, _vp, _vp, _i, _vp]),
    "msl_sf_fuse_resident": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _sz, _i, _vp]),
    "msl_sf_set_batch_capacity": (_i, [_vp, _i]),
    "msl_sf_fuse_resident_batch": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _i, _vp]),
    "msl_sf_fuse_resident_batch_d16": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _vp, _sz, _sz, _f, _vp, _sz, _sz, _i, _vp]),
    "msl_sf_last_counters": (_i, [_vp, _vp]),
    "msl_sf_sync": (_i, [_vp]),
    "msl_sf_set_stream": (_i, [_vp, _vp]),
    "msl_sf_staged_gray": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "msl_orb_wait_event": (_i, [_vp, _vp]),
    "msl_orb_debug_stamps": (_i, [_vp, _vp, _i]),
    "msl_sf_debug_seeds": (_i, [_vp, _vp]),
    "msl_sf_debug_index": (_i, [_vp, _vp]),
    "msl_sf_debug_ctr": (_i, [_vp, _vp]),
    "msl_sf_debug_scratch": (_i, [_vp, _i, _sz, _vp, _sz]),
    "msl_sf_debug_event_overhead": (_i, [_vp, _i, _i, _vp]),
    "msl_debug_div100": (_i, [_vp, _vp, _sz]),
    "msl_debug_deal": (_i, [_vp, _i, _vp]),
    "msl_debug_chain_sum": (_i, [_vp, _vp, _i, _i, _vp]),
    "msl_debug_peac_mse": (_i, [_vp, _sz, _i, _vp]),
    "msl_debug_peac_cluster_on_device": (_i, [_i]),
    "msl_debug_peac_thread_shortfall": (C.c_longlong, []),
    "msl_sf_profile_enable": (_i, [_vp, _i]),
    "msl_sf_profile_stride": (_i, [_vp, _i]),
    "msl_sf_profile_read": (_i, [_vp, _vp, _vp]),
    "msl_sf_kernel_name": (C.c_char_p, [_i]),
    "msl_debug_throw": (_i, [_i]),
}
MSL_ORB_NKERNELS = 6
MSL_SF_NKERNELS = 12


def _load():
    # PyTorch-ROCm bundles its own libamdhip64; when torch shares the process it must be loaded first so that both
    # sides bind one HIP runtime (two runtimes in one process cannot both own the GPU).  C/C++ hosts are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise MslError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C manhattanslam_amd/csrc). There is no CPU fallback.")
    dll = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(dll, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return dll


lib = _load()


def check(rc, what="msl call"):
    if rc != 0:
        raise MslError(f"{what} failed ({rc}): {lib.msl_last_error().decode()}")


def device_count():
    return int(lib.msl_device_count())


def ptr(a):
    """void* of a numpy array (host) or anything exposing data_ptr() (torch device tensor)."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)
