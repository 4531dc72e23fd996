"""Host mirror of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th) (reference src/ORBmatcher.cc:547-678) through the
C ABI, for a batch of independent frame pairs."""
import numpy as np

from ._lib import KEYPOINT_DTYPE, MATCH_PARAMS_DTYPE, MSL_MEM_HOST, check, lib, ptr


def match_params(frame_params, scale_factors, th, check_orientation=True):
    """msl_match_params from the frame's msl_frame_params (fx..cy, bf, image bounds) and the extractor's mvScaleFactors."""
    p = np.zeros(1, MATCH_PARAMS_DTYPE)
    for k in ("fx", "fy", "cx", "cy", "bf", "minX", "maxX", "minY", "maxY"):
        p[k] = frame_params[k][0]
    p["th"] = th
    p["check_orientation"] = 1 if check_orientation else 0
    p["nlevels"] = len(scale_factors)
    p["scale_factors"][0, :len(scale_factors)] = scale_factors
    return p


class Matcher:
    """One msl_match handle (= one ORBmatcher object, src/ORBmatcher.cc:41): own stream, own cached device buffers."""

    def __init__(self, device=0):
        self.h = lib.msl_match_create(device)
        if not self.h:
            from ._lib import MslError
            raise MslError(lib.msl_last_error().decode())
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib.msl_match_destroy(self.h)
            self.h = None

    __del__ = close

    def sync(self):
        check(lib.msl_match_sync(self.h), "msl_match_sync")

    def set_stream(self, hip_stream):
        check(lib.msl_match_set_stream(self.h, hip_stream), "msl_match_set_stream")

    def search_by_projection_batch(self, params, cur, last, Tcw_cur, Tcw_last):
        return search_by_projection_batch(params, cur, last, Tcw_cur, Tcw_last, handle=self)

    def search_by_projection_device(self, params, n_pairs, cap, arrays, match_out, nmatches):
        """Device-resident inputs and outputs (torch tensors / device pointers in msl.h's argument order): asynchronous on the handle's stream."""
        check(lib.msl_match_by_projection(self.h, n_pairs, cap, ptr(params), *[ptr(a) for a in arrays], 1, ptr(match_out), ptr(nmatches), 1), "msl_match_by_projection")

    def descriptor_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros(len(a), np.int32)
        check(lib.msl_match_descriptor_distances(self.h, ptr(a), ptr(b), len(a), ptr(out)), "msl_match_descriptor_distances")
        return out


def search_by_projection_batch(params, cur, last, Tcw_cur, Tcw_last, device=0, handle=None):
    """cur / last: lists (one entry per pair) of dicts with the arrays msl.h names:
         cur:  kps (KEYPOINT_DTYPE), un_xy (N,2) f32, uright (N,) f32, grid_cell (N,) i32, desc (N,32) u8
         last: xyz (M,3) f32, desc (M,32) u8, flags (M,) u8, octave (M,) i32, angle (M,) f32
       Tcw_*: (n_pairs, 4, 4) or (n_pairs, 3, 4) float32, row-major.  Returns (match [n_pairs][N] i32 lists, nmatches)."""
    B = len(cur)
    cap = max(max(len(c["kps"]) for c in cur), max(len(l["xyz"]) for l in last), 1)
    kps = np.zeros((B, cap), KEYPOINT_DTYPE); un = np.zeros((B, cap, 2), np.float32); ur = np.zeros((B, cap), np.float32)
    cell = np.full((B, cap), -1, np.int32); cdesc = np.zeros((B, cap, 32), np.uint8); ncur = np.zeros(B, np.int32)
    xyz = np.zeros((B, cap, 3), np.float32); ldesc = np.zeros((B, cap, 32), np.uint8); flags = np.zeros((B, cap), np.uint8)
    octv = np.zeros((B, cap), np.int32); ang = np.zeros((B, cap), np.float32); nlast = np.zeros(B, np.int32)
    for f in range(B):
        n, m = len(cur[f]["kps"]), len(last[f]["xyz"])
        ncur[f], nlast[f] = n, m
        kps[f, :n] = cur[f]["kps"]; un[f, :n] = cur[f]["un_xy"]; ur[f, :n] = cur[f]["uright"]; cell[f, :n] = cur[f]["grid_cell"]
        cdesc[f, :n] = cur[f]["desc"]
        xyz[f, :m] = last[f]["xyz"]; ldesc[f, :m] = last[f]["desc"]; flags[f, :m] = last[f]["flags"]; octv[f, :m] = last[f]["octave"]
        ang[f, :m] = last[f]["angle"]
    tc = np.ascontiguousarray(np.asarray(Tcw_cur, np.float32)[:, :3, :4].reshape(B, 12))
    tl = np.ascontiguousarray(np.asarray(Tcw_last, np.float32)[:, :3, :4].reshape(B, 12))
    match = np.zeros((B, cap), np.int32); nm = np.zeros(B, np.int32)
    args = (B, cap, ptr(params), ptr(kps), ptr(un), ptr(ur), ptr(cell), ptr(cdesc), ptr(ncur), ptr(xyz), ptr(ldesc), ptr(flags), ptr(octv), ptr(ang),
            ptr(nlast), ptr(tc), ptr(tl), MSL_MEM_HOST, ptr(match), ptr(nm), MSL_MEM_HOST)
    if handle is not None:
        check(lib.msl_match_by_projection(handle.h, *args), "msl_match_by_projection")
    else:
        check(lib.msl_match_by_projection_batch(device, *args), "msl_match_by_projection_batch")
    return [match[f, :ncur[f]].copy() for f in range(B)], nm


def descriptor_distance(a, b, device=0):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    out = np.zeros(len(a), np.int32)
    check(lib.msl_match_descriptor_distance(device, ptr(a), ptr(b), len(a), ptr(out)), "msl_match_descriptor_distance")
    return out
