"""Host mirror of SurfelFusion (reference include/SurfelFusion.h:43-139) and of the
SurfelMapping::fuseMap step (src/SurfelMapping.cpp:353-392) on a device-resident map."""
import ctypes as C

import numpy as np

from ._lib import (MSL_MEM_DEVICE, MSL_MEM_HOST, MSL_SF_NKERNELS, SEED_DTYPE, SURFEL_DTYPE, MslError, check, lib, ptr)


def _pose16(pose):
    """Accepts a 4x4 matrix (row-major numpy, Twc) or a flat column-major float32[16]."""
    pose = np.asarray(pose, np.float32)
    if pose.shape == (4, 4):
        pose = pose.T.reshape(16)  # Eigen::Matrix4f storage is column-major
    assert pose.shape == (16,)
    return np.ascontiguousarray(pose)


class SurfelFusion:
    """Same constructor arguments as the reference class (src/SurfelFusion.cpp:29-38)."""

    def __init__(self, width, height, fx, fy, cx, cy, fuseFar, fuseNear, device=0):
        self._h = lib.msl_sf_create(int(width), int(height), float(fx), float(fy), float(cx), float(cy), float(fuseFar),
                                    float(fuseNear), int(device))
        if not self._h:
            raise MslError("msl_sf_create failed: " + lib.msl_last_error().decode())
        self.width, self.height = int(width), int(height)
        self.nseeds = (self.width // 8) * (self.height // 8)

    def close(self):
        if getattr(self, "_h", None):
            lib.msl_sf_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- SurfelFusion::fuseInitializeMap, host-vector mode ----
    def fuseInitializeMap(self, referenceFrameIndex, inputImage, inputDepth, inputPlaneMembershipImg, pose, localSurfels, local_unchanged=False):
        """Updates `localSurfels` (structured SURFEL_DTYPE array) in place and returns the new surfels.  local_unchanged=True: the array is
        byte for byte what the previous call on this handle left in it (MSL_SF_LOCAL_UNCHANGED: no upload)."""
        g = inputImage if inputImage.strides[1] == 1 else np.ascontiguousarray(inputImage)
        d = inputDepth if inputDepth.strides[1] == 4 else np.ascontiguousarray(inputDepth)
        m = inputPlaneMembershipImg if inputPlaneMembershipImg.strides[1] == 4 else np.ascontiguousarray(inputPlaneMembershipImg)
        assert g.dtype == np.uint8 and d.dtype == np.float32 and m.dtype == np.int32
        assert localSurfels.dtype == SURFEL_DTYPE and localSurfels.flags.c_contiguous
        new = np.zeros(self.nseeds, SURFEL_DTYPE)
        n_new = C.c_size_t(0)
        p = _pose16(pose)
        check(lib.msl_sf_fuse_ex(self._h, int(referenceFrameIndex), ptr(g), g.strides[0], ptr(d), d.strides[0], ptr(m), m.strides[0],
                                 ptr(p), ptr(localSurfels) if len(localSurfels) else None, len(localSurfels), ptr(new), len(new),
                                 C.byref(n_new), 1 if local_unchanged else 0), "msl_sf_fuse_ex")
        return new[:n_new.value].copy()

    # ---- device-resident map ----
    def map_reserve(self, cap):
        check(lib.msl_sf_map_reserve(self._h, int(cap)), "msl_sf_map_reserve")

    def map_upload(self, surfels):
        surfels = np.ascontiguousarray(surfels, SURFEL_DTYPE)
        check(lib.msl_sf_map_upload(self._h, ptr(surfels) if len(surfels) else None, len(surfels)), "msl_sf_map_upload")

    def map_snapshot(self):
        """Device-side copy of the resident map (synchronous); map_restore() puts it back, asynchronously, device to device."""
        check(lib.msl_sf_map_snapshot(self._h), "msl_sf_map_snapshot")

    def map_restore(self):
        check(lib.msl_sf_map_restore(self._h), "msl_sf_map_restore")

    def map_size(self):
        n = C.c_size_t(0)
        check(lib.msl_sf_map_size(self._h, C.byref(n)), "msl_sf_map_size")
        return n.value

    def map_download(self):
        n = self.map_size()
        out = np.zeros(n, SURFEL_DTYPE)
        got = C.c_size_t(0)
        check(lib.msl_sf_map_download(self._h, ptr(out) if n else None, n, C.byref(got)), "msl_sf_map_download")
        return out[:got.value]

    # ---- map maintenance (SurfelMapping::moveAddSurfels / Stop on the resident map) ----
    def map_detach(self, pose_index):
        """Live surfels last updated by `pose_index`, in map order; they are marked deleted in the map."""
        return self._select(lib.msl_sf_map_detach, pose_index, "msl_sf_map_detach")

    def map_export(self, min_update_times=5):
        """Surfels with updateTimes >= min_update_times, in map order (SurfelMapping::Stop filter)."""
        return self._select(lib.msl_sf_map_export, min_update_times, "msl_sf_map_export")

    def _select(self, fn, arg, what):
        n = C.c_size_t(0)
        out = np.zeros(0, SURFEL_DTYPE)
        rc = fn(self._h, int(arg), None, 0, C.byref(n))
        if rc == 0:
            return out
        if rc != -4:
            check(rc, what)
        out = np.zeros(n.value, SURFEL_DTYPE)
        check(fn(self._h, int(arg), ptr(out), len(out), C.byref(n)), what)
        return out[:n.value]

    def export_ply(self, path, min_update_times=5, inactive=None):
        """System::saveSurfels on SurfelMapping::Stop's cloud: local surfels with updateTimes >= min_update_times, then `inactive`."""
        ina = np.zeros(0, SURFEL_DTYPE) if inactive is None else np.ascontiguousarray(inactive, SURFEL_DTYPE)
        check(lib.msl_sf_export_ply(self._h, int(min_update_times), ptr(ina) if len(ina) else None, len(ina), str(path).encode()), "msl_sf_export_ply")

    def map_append(self, surfels):
        surfels = np.ascontiguousarray(surfels, SURFEL_DTYPE)
        check(lib.msl_sf_map_append(self._h, ptr(surfels) if len(surfels) else None, len(surfels)), "msl_sf_map_append")

    def fuse_resident(self, referenceFrameIndex, gray, depth, member, pose, device=False, strides=None):
        """fuseInitializeMap + fuseMap compaction on the resident map; asynchronous."""
        if strides is None:
            strides = (gray.stride(0) if device else gray.strides[0], (depth.stride(0) * 4) if device else depth.strides[0],
                       (member.stride(0) * 4) if device else member.strides[0])
        p = _pose16(pose)
        check(lib.msl_sf_fuse_resident(self._h, int(referenceFrameIndex), ptr(gray), strides[0], ptr(depth), strides[1], ptr(member),
                                       strides[2], MSL_MEM_DEVICE if device else MSL_MEM_HOST, ptr(p)), "msl_sf_fuse_resident")

    def set_batch_capacity(self, max_frames):
        check(lib.msl_sf_set_batch_capacity(self._h, int(max_frames)), "msl_sf_set_batch_capacity")

    def fuse_resident_batch(self, refs, grays, depths, members, poses, device=False, member_shared=False, frame_step=1, member_frame_step=None,
                            depth_factor=None):
        """Keyframes in order.  grays (n,H,W) u8, depths (n,H,W) f32, members (n,H/2,W/2) i32 (or (H/2,W/2) with
        member_shared=True), poses (n,16) column-major; host numpy arrays or, with device=True, torch tensors.
        frame_step = k: keyframe j is frame j * k of the image arrays (SurfelFusion on every k-th frame of a sequence).
        depth_factor = a: depths are RAW uint16 images, converted on the device as float(raw) * a (src/Frame.cc:96-97)."""
        n = len(refs)
        k = int(frame_step)
        km = k if member_frame_step is None else int(member_frame_step)   # membership images may come one per keyframe (PEAC output)
        refs = np.ascontiguousarray(refs, np.int32)
        poses = np.ascontiguousarray(np.stack([_pose16(p) for p in poses]), np.float32)
        w, h = self.width, self.height
        mw, mh = (w + 1) // 2, (h + 1) // 2      # the membership image is ceil(h / 2) x ceil(w / 2) (PlaneDetection's cloud size)
        if depth_factor is not None:
            check(lib.msl_sf_fuse_resident_batch_d16(self._h, n, ptr(refs), ptr(grays), w, k * w * h, ptr(depths), 2 * w, 2 * k * w * h, float(depth_factor),
                                                     ptr(members), 4 * mw, 0 if member_shared else 4 * km * mw * mh,
                                                     MSL_MEM_DEVICE if device else MSL_MEM_HOST, ptr(poses)), "msl_sf_fuse_resident_batch_d16")
            return
        check(lib.msl_sf_fuse_resident_batch(self._h, n, ptr(refs), ptr(grays), w, k * w * h, ptr(depths), 4 * w, 4 * k * w * h, ptr(members),
                                             4 * mw, 0 if member_shared else 4 * km * mw * mh,
                                             MSL_MEM_DEVICE if device else MSL_MEM_HOST, ptr(poses)), "msl_sf_fuse_resident_batch")

    def staged_gray(self):
        """(device address, row stride, frame stride, hipEvent_t) of the gray images the last host-image batch staged (msl_sf_staged_gray): ONE upload
        for the ORB extractor as well (ORBextractor.wait_event + extract_batch_shared)."""
        p, rs, fs, ev = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_void_p()
        check(lib.msl_sf_staged_gray(self._h, C.byref(p), C.byref(rs), C.byref(fs), C.byref(ev)), "msl_sf_staged_gray")
        return p.value, rs.value, fs.value, ev.value

    def counters(self):
        c = np.zeros(5, np.int64)
        check(lib.msl_sf_last_counters(self._h, ptr(c)), "msl_sf_last_counters")
        return dict(zip(("n_live_before", "n_new", "n_deleted", "n_updated", "n_live_after"), (int(v) for v in c)))

    def sync(self):
        check(lib.msl_sf_sync(self._h), "msl_sf_sync")

    def set_stream(self, hip_stream):
        check(lib.msl_sf_set_stream(self._h, C.c_void_p(hip_stream)))

    # ---- debug / profiling ----
    def debug_seeds(self):
        out = np.zeros(self.nseeds, SEED_DTYPE)
        check(lib.msl_sf_debug_seeds(self._h, ptr(out)))
        return out

    def debug_ctr(self):
        out = np.zeros(16, np.int64)
        check(lib.msl_sf_debug_ctr(self._h, ptr(out)))
        return out

    def debug_scratch(self, n_words, which=0, offset=0):
        out = np.zeros(n_words, np.uint32)
        check(lib.msl_sf_debug_scratch(self._h, int(which), int(offset), ptr(out), n_words))
        return out

    def debug_event_overhead(self, grid=1, n=200):
        us = C.c_float(0)
        check(lib.msl_sf_debug_event_overhead(self._h, int(grid), int(n), C.byref(us)))
        return float(us.value)

    def debug_index(self):
        out = np.zeros((self.height, self.width), np.int32)
        check(lib.msl_sf_debug_index(self._h, ptr(out)))
        return out

    def profile_stride(self, stride=1):
        check(lib.msl_sf_profile_stride(self._h, int(stride)))

    def profile_enable(self, mode=-1):
        check(lib.msl_sf_profile_enable(self._h, int(mode)))

    @staticmethod
    def kernel_names():
        return [lib.msl_sf_kernel_name(k).decode() for k in range(MSL_SF_NKERNELS)]

    def profile_read(self):
        ms = np.zeros(MSL_SF_NKERNELS, np.float32)
        cnt = np.zeros(MSL_SF_NKERNELS, np.int32)
        check(lib.msl_sf_profile_read(self._h, ptr(ms), ptr(cnt)))
        return {lib.msl_sf_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(MSL_SF_NKERNELS)}


class SurfelMap(SurfelFusion):
    """SurfelFusion plus the resident local-surfel vector: `fuseMap` mirrors
    SurfelMapping::fuseMap(image, depth, planeMembershipImg, pose, referenceIndex)."""

    def fuseMap(self, image, depth, planeMembershipImg, pose, referenceIndex, device=False):
        self.fuse_resident(referenceIndex, image, depth, planeMembershipImg, pose, device=device)
