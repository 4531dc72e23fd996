"""Host mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:43-110)."""
import ctypes as C

import numpy as np

from ._lib import (FRAME_PARAMS_DTYPE, KEYPOINT_DTYPE, MSL_MEM_DEVICE, MSL_MEM_HOST, MSL_ORB_NKERNELS, MslError, check, lib, ptr)


def frame_params(fx, fy, cx, cy, bf, width, height, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
    """msl_frame_params with the image bounds of Frame::ComputeImageBounds filled in."""
    p = np.zeros(1, FRAME_PARAMS_DTYPE)
    for k, v in dict(fx=fx, fy=fy, cx=cx, cy=cy, k1=k1, k2=k2, p1=p1, p2=p2, k3=k3, bf=bf).items():
        p[k] = v
    check(lib.msl_frame_image_bounds(ptr(p), int(width), int(height)), "msl_frame_image_bounds")
    return p


class ORBextractor:
    """Same constructor arguments and call semantics as the reference class.

    ``extractor(image, mask=None)`` returns ``(keypoints, descriptors)``: a structured array with the
    cv::KeyPoint fields and an (N, 32) uint8 array (``None`` when N == 0, mirroring
    ``_descriptors.release()`` at src/ORBextractor.cc:832-833).  The mask is ignored, as in the reference.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width=640, max_height=480,
                 max_batch=1, device=0):
        self._h = lib.msl_orb_create(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST),
                                     int(minThFAST), int(max_width), int(max_height), int(max_batch), int(device))
        if not self._h:
            raise MslError("msl_orb_create failed: " + lib.msl_last_error().decode())
        self.nlevels = int(nlevels)
        self.scaleFactor = float(np.float32(scaleFactor))
        self.capacity = lib.msl_orb_capacity(self._h)
        self.max_batch = int(max_batch)

    def close(self):
        if getattr(self, "_h", None):
            lib.msl_orb_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- include/ORBextractor.h:58-80 ----
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def _tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        check(lib.msl_orb_scale_tables(self._h, *[ptr(a) for a in t]), "msl_orb_scale_tables")
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        check(lib.msl_orb_features_per_level(self._h, ptr(out)))
        return out

    # ---- operator() ----
    def __call__(self, image, mask=None):
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), None
        if image.dtype != np.uint8 or image.ndim != 2:
            raise MslError("image must be CV_8UC1 (2-D uint8)")  # assert at src/ORBextractor.cc:819
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int(0)
        check(lib.msl_orb_extract(self._h, ptr(image), w, h, image.strides[0], ptr(kps), ptr(desc), self.capacity,
                                  C.byref(n)), "msl_orb_extract")
        n = n.value
        return kps[:n].copy(), (desc[:n].copy() if n else None)

    def extract_batch(self, images):
        """images: (B, H, W) uint8 host array -> list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images)
        b, h, w = images.shape
        kps = np.zeros((b, self.capacity), KEYPOINT_DTYPE)
        desc = np.zeros((b, self.capacity, 32), np.uint8)
        n = np.zeros(b, np.int32)
        check(lib.msl_orb_extract_batch(self._h, ptr(images), b, w, h, w, w * h, MSL_MEM_HOST, ptr(kps), ptr(desc),
                                        self.capacity, ptr(n), MSL_MEM_HOST), "msl_orb_extract_batch")
        return [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(b)]

    def extract_frames(self, images, depths, params):
        """ORB + the Frame post-ORB steps (src/Frame.cc:107-153) for RGB-D frames: images (B,H,W) u8, depths (B,H,W) f32 metres.
        Returns per frame (mvKeys, mDescriptors, mvKeysUn xy (N,2), mvDepth, mvuRight, grid cell id or -1)."""
        images = np.ascontiguousarray(images); depths = np.ascontiguousarray(depths, np.float32)
        b, h, w = images.shape
        cap = self.capacity
        kps = np.zeros((b, cap), KEYPOINT_DTYPE); desc = np.zeros((b, cap, 32), np.uint8)
        un = np.zeros((b, cap, 2), np.float32); dep = np.zeros((b, cap), np.float32); ur = np.zeros((b, cap), np.float32)
        cell = np.zeros((b, cap), np.int32); n = np.zeros(b, np.int32)
        check(lib.msl_orb_extract_frame_batch(self._h, ptr(images), ptr(depths), b, w, h, w, w * h, 4 * w, 4 * w * h, MSL_MEM_HOST, ptr(params),
                                              ptr(kps), ptr(desc), ptr(un), ptr(dep), ptr(ur), ptr(cell), cap, ptr(n), MSL_MEM_HOST),
              "msl_orb_extract_frame_batch")
        return [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy(), un[f, :n[f]].copy(), dep[f, :n[f]].copy(), ur[f, :n[f]].copy(),
                 cell[f, :n[f]].copy()) for f in range(b)]

    def extract_batch_device(self, d_images, d_kps, d_desc, d_n, n_frames, width, height):
        """Asynchronous, everything resident in HBM (torch tensors or raw pointers)."""
        check(lib.msl_orb_extract_batch(self._h, ptr(d_images), n_frames, width, height, width, width * height,
                                        MSL_MEM_DEVICE, ptr(d_kps), ptr(d_desc), self.capacity, ptr(d_n),
                                        MSL_MEM_DEVICE), "msl_orb_extract_batch")

    def extract_batch_host(self, images, kps, desc, n_out, n_frames, width, height):
        """Host buffers in and out (numpy views, ideally of pinned memory), no copies on the Python side; synchronous."""
        check(lib.msl_orb_extract_batch(self._h, ptr(images), n_frames, width, height, width, width * height,
                                        MSL_MEM_HOST, ptr(kps), ptr(desc), self.capacity, ptr(n_out),
                                        MSL_MEM_HOST), "msl_orb_extract_batch")

    def wait_event(self, hip_event):
        """The extractor's stream waits for a hipEvent_t (e.g. the one SurfelFusion.staged_gray returns) before the calls that follow."""
        check(lib.msl_orb_wait_event(self._h, C.c_void_p(hip_event)), "msl_orb_wait_event")

    def extract_batch_shared(self, gray_dev, row_stride, frame_stride, kps, desc, n_out, n_frames, width, height):
        """Device images in (an address, e.g. the surfel handle's staged gray images), host buffers out (numpy views); synchronous."""
        check(lib.msl_orb_extract_batch(self._h, C.c_void_p(gray_dev), n_frames, width, height, row_stride, frame_stride,
                                        MSL_MEM_DEVICE, ptr(kps), ptr(desc), self.capacity, ptr(n_out),
                                        MSL_MEM_HOST), "msl_orb_extract_batch")

    def sync(self):
        check(lib.msl_orb_sync(self._h), "msl_orb_sync")

    def set_stream(self, hip_stream):
        check(lib.msl_orb_set_stream(self._h, C.c_void_p(hip_stream)))

    # ---- debug / profiling ----
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        check(lib.msl_orb_debug_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def debug_level(self, frame, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(lib.msl_orb_debug_level(self._h, frame, level, int(blurred), ptr(out)), "msl_orb_debug_level")
        return out

    def debug_candidates(self, frame, level, cap=1 << 20):
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int()
        check(lib.msl_orb_debug_candidates(self._h, frame, level, ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def profile_enable(self, mode=-1):
        check(lib.msl_orb_profile_enable(self._h, int(mode)))

    @staticmethod
    def kernel_names():
        return [lib.msl_orb_kernel_name(k).decode() for k in range(MSL_ORB_NKERNELS)]

    def profile_read(self):
        ms = np.zeros(MSL_ORB_NKERNELS, np.float32)
        cnt = np.zeros(MSL_ORB_NKERNELS, np.int32)
        check(lib.msl_orb_profile_read(self._h, ptr(ms), ptr(cnt)))
        return {lib.msl_orb_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(MSL_ORB_NKERNELS)}
