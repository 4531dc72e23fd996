"""Host mirror of the PEAC front: organised point cloud + initial block statistics.

Mirrors PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:44-76) and the per-block ahc::PlaneSeg initialisation of
ahc::PlaneFitter::initGraph (include/peac/AHCPlaneFitter.hpp:756-776, include/peac/AHCPlaneSeg.hpp:237-285) through the C ABI.
"""
import numpy as np

from ._lib import lib, check, ptr, PEAC_STATS_DTYPE, PEAC_PARAMS_DTYPE, PEAC_BLOCK_DTYPE, PEAC_PLANE_DTYPE, MSL_MEM_HOST, MSL_MEM_DEVICE


def block_stats(depth_u16, fx, fy, cx, cy, depth_map_factor, window=(10, 10), depth_alpha=0.04, depth_change_tol=0.02,
                init_loose=False, want_cloud=True, device=0):
    """depth_u16: [H, W] or [F, H, W] uint16.  Returns (cloud [F, ch*cw, 3] f64 or None, stats [F, Nh*Nw] PEAC_STATS_DTYPE)."""
    d = np.asarray(depth_u16)
    if d.ndim == 2:
        d = d[None]
    d = np.ascontiguousarray(d, np.uint16)
    F, H, W = d.shape
    cw, ch = (W + 1) // 2, (H + 1) // 2
    nb = (cw // window[0]) * (ch // window[1])
    stats = np.zeros((F, nb), PEAC_STATS_DTYPE)
    cloud = np.zeros((F, cw * ch, 3), np.float64) if want_cloud else None
    check(lib.msl_peac_block_stats(device, ptr(d), d.strides[1], d.strides[0], W, H, F, MSL_MEM_HOST, fx, fy, cx, cy, depth_map_factor,
                                   window[0], window[1], depth_alpha, depth_change_tol, 1 if init_loose else 0,
                                   ptr(cloud) if want_cloud else None, ptr(stats), MSL_MEM_HOST), "msl_peac_block_stats")
    return cloud, stats


def default_params():
    """ahc::PlaneFitter / ahc::ParamSet defaults (PlaneDetection overrides none of them)."""
    p = np.zeros(1, PEAC_PARAMS_DTYPE)
    lib.msl_peac_default_params(ptr(p))
    return p


def _frames(depth_u16):
    d = np.asarray(depth_u16)
    if d.ndim == 2:
        d = d[None]
    return np.ascontiguousarray(d, np.uint16)


def block_fit(depth_u16, fx, fy, cx, cy, depth_map_factor, params=None, device=0):
    """Initial graph node of every window (Stats + PCA plane fit) on the GPU: [F, Nh * Nw] PEAC_BLOCK_DTYPE."""
    d = _frames(depth_u16)
    F, H, W = d.shape
    prm = default_params() if params is None else params
    nb = (((W + 1) // 2) // int(prm["window_w"][0])) * (((H + 1) // 2) // int(prm["window_h"][0]))
    out = np.zeros((F, nb), PEAC_BLOCK_DTYPE)
    check(lib.msl_peac_block_fit(device, ptr(d), d.strides[1], d.strides[0], W, H, F, MSL_MEM_HOST, fx, fy, cx, cy, depth_map_factor, ptr(prm), ptr(out),
                                 MSL_MEM_HOST), "msl_peac_block_fit")
    return out


def plane_membership(depth_u16, fx, fy, cx, cy, depth_map_factor, params=None, device=0):
    """PlaneDetection::readDepthImage + runPlaneDetection for [H, W] or [F, H, W] uint16 depth: (membership [F, ceil(H/2), ceil(W/2)] int32 =
    plane_filter.membershipImg, number of extracted planes [F])."""
    d = _frames(depth_u16)
    F, H, W = d.shape
    prm = default_params() if params is None else params
    member = np.zeros((F, (H + 1) // 2, (W + 1) // 2), np.int32)
    n = np.zeros(F, np.int32)
    check(lib.msl_peac_membership_batch(device, ptr(d), d.strides[1], d.strides[0], W, H, F, MSL_MEM_HOST, fx, fy, cx, cy, depth_map_factor, ptr(prm),
                                        ptr(member), ptr(n)), "msl_peac_membership_batch")
    return member, n


def plane_membership_from_blocks(blocks, depth_u16, fx, fy, cx, cy, depth_map_factor, params=None):
    """The host stage alone (clustering, erosion, region growing) on block fits [F, Nh * Nw] PEAC_BLOCK_DTYPE and the depth images they came from."""
    d = _frames(depth_u16)
    F, H, W = d.shape
    prm = default_params() if params is None else params
    b = np.ascontiguousarray(blocks, PEAC_BLOCK_DTYPE).reshape(F, -1)
    member = np.zeros((F, (H + 1) // 2, (W + 1) // 2), np.int32)
    n = np.zeros(F, np.int32)
    check(lib.msl_peac_membership_from_blocks(ptr(b), ptr(d), d.strides[1], d.strides[0], W, H, F, fx, fy, cx, cy, depth_map_factor, ptr(prm), ptr(member),
                                              ptr(n)), "msl_peac_membership_from_blocks")
    return member, n


def _split_planes(F, n, planes, offsets, indices):
    out = []
    for f in range(F):
        k = int(n[f])
        out.append((planes[f, :k].copy(), [indices[f, offsets[f, j]:offsets[f, j + 1]].copy() for j in range(k)]))
    return out


def extract(depth_u16, fx, fy, cx, cy, depth_map_factor, params=None, max_planes=64, device=0, with_cloud=False):
    """Everything PlaneDetection hands on after runPlaneDetection: (membership [F, ch, cw], n_planes [F], per frame (planes PEAC_PLANE_DTYPE [n],
    plane_vertices_: list of int32 vertex-index arrays)); with_cloud=True appends cloud.vertices [F, ch * cw, 3] (float64)."""
    d = _frames(depth_u16)
    F, H, W = d.shape
    prm = default_params() if params is None else params
    ch, cw = (H + 1) // 2, (W + 1) // 2
    member = np.zeros((F, ch, cw), np.int32); n = np.zeros(F, np.int32)
    planes = np.zeros((F, max_planes), PEAC_PLANE_DTYPE); offsets = np.zeros((F, max_planes + 1), np.int32); indices = np.zeros((F, ch * cw), np.int32)
    cloud = np.zeros((F, ch * cw, 3), np.float64) if with_cloud else None
    check(lib.msl_peac_extract_batch(device, ptr(d), d.strides[1], d.strides[0], W, H, F, MSL_MEM_HOST, fx, fy, cx, cy, depth_map_factor, ptr(prm), ptr(member),
                                     ptr(n), max_planes, ptr(planes), ptr(offsets), ptr(indices), ptr(cloud)), "msl_peac_extract_batch")
    if with_cloud:
        return member, n, _split_planes(F, n, planes, offsets, indices), cloud
    return member, n, _split_planes(F, n, planes, offsets, indices)


def extract_from_blocks(blocks, depth_u16, fx, fy, cx, cy, depth_map_factor, params=None, max_planes=64):
    """extract() from block fits the caller already has: the host stage only, no device."""
    d = _frames(depth_u16)
    F, H, W = d.shape
    prm = default_params() if params is None else params
    b = np.ascontiguousarray(blocks, PEAC_BLOCK_DTYPE).reshape(F, -1)
    ch, cw = (H + 1) // 2, (W + 1) // 2
    member = np.zeros((F, ch, cw), np.int32); n = np.zeros(F, np.int32)
    planes = np.zeros((F, max_planes), PEAC_PLANE_DTYPE); offsets = np.zeros((F, max_planes + 1), np.int32); indices = np.zeros((F, ch * cw), np.int32)
    check(lib.msl_peac_extract_from_blocks(ptr(b), ptr(d), d.strides[1], d.strides[0], W, H, F, fx, fy, cx, cy, depth_map_factor, ptr(prm), ptr(member), ptr(n),
                                           max_planes, ptr(planes), ptr(offsets), ptr(indices)), "msl_peac_extract_from_blocks")
    return member, n, _split_planes(F, n, planes, offsets, indices)


def plane_membership_device(d_depth16, frame_step, n_frames, width, height, fx, fy, cx, cy, depth_map_factor, params, member_out, nplanes_out, device=0):
    """Device-resident 16-bit depth frames (torch tensor, frame j of the call = frame j * frame_step of the tensor); membership and plane
    counts land in the caller's host arrays ([n_frames, ceil(H/2), ceil(W/2)] int32, [n_frames] int32)."""
    check(lib.msl_peac_membership_batch(device, ptr(d_depth16), 2 * width, 2 * width * height * int(frame_step), width, height, n_frames, MSL_MEM_DEVICE, fx, fy,
                                        cx, cy, depth_map_factor, ptr(params), ptr(member_out), ptr(nplanes_out)), "msl_peac_membership_batch")
