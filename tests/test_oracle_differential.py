"""Whole-image / whole-problem DIFFERENTIAL checks of the oracle's restatements of third-party code against independently written models
(round 5).  The reference needs OpenCV and Eigen, which this image lacks, so oracle/ restates their arithmetic; every GPU
parity test compares the HIP library with that restatement.  A misreading shared by the restatement and its own known-answer tests would
pass everything -- these tests shrink that common-mode risk: each model below is written from the published algorithm alone, in a different
style (vectorised numpy / scipy / LAPACK, exact integers or float64) from the C++ it checks, and is compared on random inputs of every
geometry the front end uses.  They do NOT turn parity green: the statement remains "oracle-relative" (README).

  cv::resize, INTER_LINEAR, 8-bit       11-bit fixed-point bilinear, src/ORBextractor.cc:870-871          -> exact integer numpy model, bit-identical
  cv::GaussianBlur 7x7, sigma 2, 8-bit  fixed-point separable kernel, BORDER_REFLECT_101, :1058-1060    -> scipy.ndimage.correlate1d on int64, bit-identical
  Eigen Matrix4d::inverse + the Gauss-Newton update of getHuberNorm, src/SurfelFusion.cpp:91-165          -> numpy.linalg (LAPACK), relative 1e-9 / 2e-6
  cv::undistortPoints (5 fixed-point iterations), src/Frame.cc:441-477                                    -> float64 iteration to convergence, 2e-3 px
"""
import ctypes as C

import numpy as np
import pytest

from manhattanslam_amd import synth

LEVELS = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]   # 1.2^-l, round() per level (:420-431)


def _resize_model(src, dw, dh):
    """cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) for CV_8UC1, written from the published algorithm (imgproc/src/resize.cpp:
    HResizeLinear / VResizeLinear with INTER_RESIZE_COEF_BITS = 11): per destination column / row a source index and an 11-bit weight pair from
    the float fraction, horizontal pass to int32 in 11-bit fixed point, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2."""
    sh, sw = src.shape

    def taps(dn, sn):
        scale = np.float64(sn) / np.float64(dn)                                  # inv_scale computed in double
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)                         # fx = (float)((dx + 0.5) * scale_x - 0.5)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo] = 0; f[lo] = 0
        hi = s >= sn - 1
        s[hi] = sn - 1; f[hi] = 0
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)                      # saturate_cast<short>(fx * INTER_RESIZE_COEF_SCALE): round half to even
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, sn - 1), w0, w1

    x0, x1, a0, a1 = taps(dw, sw)
    y0, y1, b0, b1 = taps(dh, sh)
    S = src.astype(np.int64)
    H = S[:, x0] * a0[None, :] + S[:, x1] * a1[None, :]                          # rows in 11-bit fixed point
    r0, r1 = H[y0, :], H[y1, :]
    out = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("level", range(1, 8))
def test_resize_equals_an_exact_integer_model_on_every_pyramid_geometry(oracle, level):
    (sw, sh), (dw, dh) = LEVELS[level - 1], LEVELS[level]
    rng = np.random.default_rng(100 + level)
    for kind in range(3):
        if kind == 0:
            src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)                 # white noise: every tap pair matters
        elif kind == 1:
            src = synth.orb_frame(synth.ORB_SEED + level, sw, sh) if (sw, sh) == (640, 480) else np.ascontiguousarray(synth.orb_frame(synth.ORB_SEED + level)[:sh, :sw])
        else:
            src = np.where(rng.random((sh, sw)) < 0.5, 0, 255).astype(np.uint8)  # saturating extremes
        got = oracle.resize(src, dw, dh)
        exp = _resize_model(src, dw, dh)
        assert got.shape == exp.shape == (dh, dw)
        assert np.array_equal(got, exp), (level, kind, np.argwhere(got != exp)[:5], np.abs(got.astype(int) - exp.astype(int)).max())


def test_resize_model_on_odd_geometries(oracle):
    rng = np.random.default_rng(7)
    for (sw, sh, dw, dh) in ((37, 29, 31, 24), (64, 48, 53, 40), (401, 305, 334, 254), (1280, 960, 1067, 800), (16, 16, 13, 13)):
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        assert np.array_equal(oracle.resize(src, dw, dh), _resize_model(src, dw, dh)), (sw, sh, dw, dh)


def _blur_model(img):
    """cv::GaussianBlur(img, out, Size(7, 7), 2, 2, BORDER_REFLECT_101) for CV_8UC1 as OpenCV >= 4.1 computes it: the 16-bit fixed-point kernel
    [18, 34, 49, 55, 49, 34, 18] / 256 (sum 257 / 256), rows then columns, one rounding at the end, saturated."""
    from scipy import ndimage
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    r = ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")      # mirror = reflect-101 (the edge pixel is not repeated)
    c = ndimage.correlate1d(r, k, axis=0, mode="mirror")
    return np.minimum((c + 32768) >> 16, 255).astype(np.uint8)


@pytest.mark.parametrize("level", range(8))
def test_blur_equals_scipy_correlate_on_every_pyramid_geometry(oracle, level):
    w, h = LEVELS[level]
    rng = np.random.default_rng(200 + level)
    assert oracle.gaussian_kernel().tolist() == [18, 34, 49, 55, 49, 34, 18]
    for kind in range(2):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8) if kind == 0 else np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8)
        got, exp = oracle.blur(img), _blur_model(img)
        assert np.array_equal(got, exp), (level, kind, np.argwhere(got != exp)[:5])
    tiny = rng.integers(0, 256, (7, 9), dtype=np.uint8)                           # an image barely larger than the kernel: the border on every side
    assert np.array_equal(oracle.blur(tiny), _blur_model(tiny))


def test_inverse4_double_and_gauss_newton_update_against_lapack():
    from tests.oracle_lib import load
    dll = load().dll
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(300):
        # the matrices the plane fit inverts: 2 * sum([p; 1] [p; 1]^T) + 5 I over 16 .. 256 points of a small noisy patch (src/SurfelFusion.cpp:104-151)
        n = int(rng.integers(16, 257))
        P = np.concatenate([rng.normal(0, rng.uniform(0.005, 0.2), (n, 3)), np.ones((n, 1))], 1)
        Hm = 2 * P.T @ P + 5 * np.eye(4)
        inv = np.zeros(16)
        dll.mslo_inverse4d(np.ascontiguousarray(Hm.T).ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p))     # column-major in and out
        got = inv.reshape(4, 4).T
        ref = np.linalg.inv(Hm)
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        worst = max(worst, rel)
        J = rng.normal(0, 1, 4)
        assert np.allclose(got @ J, np.linalg.solve(Hm, J), rtol=1e-9, atol=1e-12)
    assert worst < 1e-9, worst      # adjugate / determinant on these well-conditioned 4x4s: a few hundred ulp at most
    # a general (pose-like and random) 4x4
    for trial in range(100):
        A = rng.normal(0, 1, (4, 4)) + 3 * np.eye(4)
        inv = np.zeros(16)
        dll.mslo_inverse4d(np.ascontiguousarray(A.T).ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p))
        assert np.allclose(inv.reshape(4, 4).T @ A, np.eye(4), atol=1e-9)


def _huber_model(pts, n0, huber=0.4):
    """getHuberNorm (src/SurfelFusion.cpp:91-165) in float64 with numpy.linalg.solve: five damped Gauss-Newton steps on (n, b) of the Huber loss of
    the point-plane residuals about the centroid; then the offset back to the original frame and the normalisation."""
    p = pts.astype(np.float64)
    c = p.mean(0)
    q = p - c
    x = np.array([n0[0], n0[1], n0[2], 0.0])
    A = np.concatenate([q, np.ones((len(q), 1))], 1)
    for _ in range(5):
        r = A @ x
        inl = np.abs(r) < huber
        J = 2 * (A[inl] * r[inl, None]).sum(0) + huber * A[r >= huber].sum(0) - huber * A[r <= -huber].sum(0)
        Hm = 2 * A[inl].T @ A[inl] + 5 * np.eye(4)
        x = x - np.linalg.solve(Hm, J)
    nb = x[3] - x[:3] @ c
    L = np.linalg.norm(x[:3])
    return np.array([x[0] / L, x[1] / L, x[2] / L, nb / L])


def test_huber_plane_fit_against_a_float64_lapack_model():
    from tests.oracle_lib import load
    dll = load().dll
    rng = np.random.default_rng(11)
    worst = 0.0
    for trial in range(400):
        n = int(rng.integers(16, 200))
        nrm = rng.normal(0, 1, 3); nrm /= np.linalg.norm(nrm)
        d = rng.uniform(0.6, 6.0)
        t1 = np.cross(nrm, [1.0, 0.0, 0.0]); t1 /= np.linalg.norm(t1); t2 = np.cross(nrm, t1)
        uv = rng.uniform(-0.15, 0.15, (n, 2)) * d
        pts = nrm * d + uv[:, :1] * t1 + uv[:, 1:] * t2 + nrm * rng.normal(0, 0.003 * d * d, (n, 1))
        if trial % 4 == 0:                                     # a few gross outliers beyond the Huber band
            pts[: max(1, n // 12)] += nrm * rng.choice([-1.0, 1.0]) * rng.uniform(0.5, 1.5)
        n0 = nrm + rng.normal(0, 0.08, 3)                      # the mean of the pixel normals: near the truth
        pts32 = np.ascontiguousarray(pts.astype(np.float32))
        out = np.array([n0[0], n0[1], n0[2], 0.0], np.float32)
        dll.mslo_huber_norm(pts32.ctypes.data_as(C.c_void_p), C.c_int(n), out.ctypes.data_as(C.c_void_p))
        ref = _huber_model(pts32, out.copy() * 0 + np.array([n0[0], n0[1], n0[2], 0.0], np.float32))
        err = np.abs(out.astype(np.float64) - ref).max()
        worst = max(worst, err)
    # float positions / residuals and a float state against float64 throughout: a few float ulp of values of order 1 .. 6
    assert worst < 2e-5, worst


def _undistort_model(xy, fx, fy, cx, cy, k1, k2, p1, p2, k3, iters=200):
    """cv::undistortPoints(P = K): normalise, invert the Brown-Conrady model by the fixed-point iteration x <- (x_d - tangential(x)) / radial(x), here
    in float64 until it no longer moves; back to pixels."""
    xd = (xy[:, 0].astype(np.float64) - cx) / fx; yd = (xy[:, 1].astype(np.float64) - cy) / fy
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x); dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (xd - dx) * icd, (yd - dy) * icd
    return np.stack([x * fx + cx, y * fy + cy], 1)


def test_undistort_against_a_converged_float64_iteration(oracle):
    from tests import oracle_lib
    I = synth.TUM1
    # TUM1.yaml's coefficients (Example/TUM1.yaml: k1, k2, p1, p2, k3) and a milder set
    for (k1, k2, p1, p2, k3) in ((0.262383, -0.953104, -0.005358, 0.002628, 1.163314), (-0.05, 0.02, 0.001, -0.0007, 0.0)):
        pd = oracle_lib.frame_params(I["fx"], I["fy"], I["cx"], I["cy"], 40.0, 640, 480, k1=k1, k2=k2, p1=p1, p2=p2, k3=k3)
        rng = np.random.default_rng(5)
        kps = np.zeros(500, oracle_lib.KEYPOINT_DTYPE)
        kps["x"] = rng.uniform(19, 620, 500); kps["y"] = rng.uniform(19, 460, 500)
        depth = np.full((480, 640), 2.0, np.float32)
        un, _, _, _ = oracle_lib.frame_epilogue(pd, kps, depth)
        ref = _undistort_model(np.stack([kps["x"], kps["y"]], 1), I["fx"], I["fy"], I["cx"], I["cy"], k1, k2, p1, p2, k3)
        # OpenCV stops after 5 iterations in its own precision; at these distortion levels the fixed point is reached to a few 1e-4 px
        assert np.abs(un.astype(np.float64) - ref).max() < 2e-3, np.abs(un.astype(np.float64) - ref).max()
