"""SurfelMapping's host bookkeeping, EXECUTED: manhattanslam_amd/adapter/SurfelMapping.cpp + SurfelFusion.cpp are
compiled against tests/stubs/, linked with libmsl.so (tests/mapping_host.cpp) and driven through InsertKeyFrame / ProcessNewKeyFrame over
random pose graphs whose revisits force detachPose -> unparkPoses -> msl_sf_map_append; after EVERY keyframe mvInactiveSurfels,
pointcloudPoseIndex, localSurfelsIndexs, every PoseElement and the downloaded resident map are compared with the oracle's statement-level
restatement of src/SurfelMapping.cpp:148-392 (oracle/surfel_oracle.cpp, mslo_mapping_*)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_surfel_gpu import assert_surfels_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 320, 240


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("mapping_host") / "libmapping_host.so"
    ad = os.path.join(ROOT, "manhattanslam_amd", "adapter")
    lib = os.path.join(ROOT, "manhattanslam_amd", "libmsl.so")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "tests", "stubs"), "-I",
                        os.path.join(ROOT, "include"), "-I", ad, os.path.join(ROOT, "tests", "mapping_host.cpp"), os.path.join(ad, "SurfelMapping.cpp"),
                        os.path.join(ad, "SurfelFusion.cpp"), lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import manhattanslam_amd  # noqa: F401  (loads torch's HIP runtime first, then libmsl.so: one runtime per process)
    d = C.CDLL(str(out))
    d.mh_create.restype = C.c_void_p
    d.mh_create.argtypes = [C.c_int, C.c_int] + [C.c_float] * 6
    d.mh_error.restype = C.c_char_p
    d.mh_error.argtypes = [C.c_void_p]
    d.mh_destroy.argtypes = [C.c_void_p]
    d.mh_keyframe.argtypes = [C.c_void_p] * 5 + [C.c_int]
    for f in ("mh_local", "mh_inactive", "mh_cloud_index", "mh_local_indexs", "mh_stop"):
        getattr(d, f).restype = C.c_size_t
        getattr(d, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    d.mh_poses.argtypes = [C.c_void_p]
    d.mh_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    d.mh_pose_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return d


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleMapping:
    def __init__(self, o, intr):
        from tests.oracle_lib import OracleSurfel
        self.sf = OracleSurfel(W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5)
        self.d = o.dll
        d = self.d
        d.mslo_mapping_create.restype = C.c_void_p
        d.mslo_mapping_create.argtypes = [C.c_void_p]
        d.mslo_mapping_destroy.argtypes = [C.c_void_p]
        d.mslo_mapping_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        for f in ("mslo_mapping_inactive", "mslo_mapping_cloud_index", "mslo_mapping_local_indexs"):
            getattr(d, f).restype = C.c_size_t
            getattr(d, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        d.mslo_mapping_poses.argtypes = [C.c_void_p]
        d.mslo_mapping_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        d.mslo_mapping_pose_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.h = d.mslo_mapping_create(self.sf.hd)

    def close(self):
        self.d.mslo_mapping_destroy(self.h)


def _state(fn_local, fn_inactive, fn_cloud, fn_lidx, fn_poses, fn_pose, fn_pose_data, h):
    """The complete bookkeeping of one side as numpy arrays (the accessor names differ, the layout does not)."""
    from manhattanslam_amd import SURFEL_DTYPE
    def vec(fn, dtype):
        n = fn(h, None, 0)
        out = np.zeros(n, dtype)
        if n:
            assert fn(h, _p(out), n) == n
        return out
    st = {"inactive": vec(fn_inactive, SURFEL_DTYPE), "cloud": vec(fn_cloud, np.int32), "lidx": vec(fn_lidx, np.int32), "local": fn_local()}
    poses = []
    for i in range(fn_poses(h)):
        info = np.zeros(4, np.int32)
        fn_pose(h, i, _p(info))
        att = np.zeros(info[2], SURFEL_DTYPE)
        links = np.zeros(info[3], np.int32)
        fn_pose_data(h, i, _p(att) if info[2] else None, _p(links) if info[3] else None)
        poses.append((info.copy(), att, links))
    st["poses"] = poses
    return st


def _graph(seed, n):
    """Reference index and camera view of each keyframe: chains long enough that poses leave the 10-hop window, jumps back to old poses
    (loop revisits: their parked surfels re-enter, several poses at once -> adjacent and non-adjacent range erases), branches."""
    rng = np.random.default_rng(seed)
    ref, view = [0], [int(rng.integers(0, 720))]
    for k in range(1, n):
        u = rng.random()
        if u < 0.70:
            r = k - 1
        elif u < 0.85:
            r = int(rng.integers(max(0, k - 4), k))
        else:
            r = int(rng.integers(0, k))           # revisit of an arbitrary old pose
        ref.append(r)
        view.append(view[r] + int(rng.integers(-3, 4)))
    return ref, view


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_surfel_mapping_bookkeeping_matches_reference_restatement(oracle, host_lib, seed):
    from manhattanslam_amd import synth, SURFEL_DTYPE
    intr = synth.scaled_intrinsics(synth.TUM1, W)
    d = host_lib
    hh = d.mh_create(W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5)
    assert not d.mh_error(hh), d.mh_error(hh)
    om = OracleMapping(oracle, intr)
    n_kf = 200
    ref, view = _graph(seed, n_kf)
    detached = unparked = adjacent_runs = 0

    def host_local():
        n = d.mh_local(hh, None, 0)
        out = np.zeros(n, SURFEL_DTYPE)
        if n:
            assert d.mh_local(hh, _p(out), n) == n
        return out

    prev_cloud = np.zeros(0, np.int32)
    for k in range(n_kf):
        gray, depth, member, pose_cm = synth.surfel_frame(view[k] % 720, w=W, h=H, intr=intr, seed=7 + seed)
        pose_rm = np.ascontiguousarray(pose_cm.reshape(4, 4).T)     # the cv::Mat (row-major) Tracking hands over
        assert d.mh_keyframe(hh, _p(gray), _p(depth), _p(member), _p(pose_rm), ref[k]) == 0, d.mh_error(hh)
        om.d.mslo_mapping_keyframe(om.h, _p(gray), gray.strides[0], _p(depth), depth.strides[0], _p(member), member.strides[0], _p(pose_rm), ref[k])
        a = _state(host_local, d.mh_inactive, d.mh_cloud_index, d.mh_local_indexs, d.mh_poses, d.mh_pose, d.mh_pose_data, hh)
        b = _state(om.sf.map_get, om.d.mslo_mapping_inactive, om.d.mslo_mapping_cloud_index, om.d.mslo_mapping_local_indexs, om.d.mslo_mapping_poses,
                   om.d.mslo_mapping_pose, om.d.mslo_mapping_pose_data, om.h)
        assert np.array_equal(a["cloud"], b["cloud"]), (k, a["cloud"], b["cloud"])
        assert np.array_equal(a["lidx"], b["lidx"]), k
        assert len(a["poses"]) == len(b["poses"]) == k + 1
        for i, ((ia, sa, la), (ib, sb, lb)) in enumerate(zip(a["poses"], b["poses"])):
            assert np.array_equal(ia, ib), (k, i, ia, ib)
            assert np.array_equal(la, lb), (k, i)
            assert_surfels_close(sa, sb, f"keyframe {k}: attachedSurfels of pose {i}")
        assert_surfels_close(a["inactive"], b["inactive"], f"keyframe {k}: mvInactiveSurfels")
        assert_surfels_close(a["local"], b["local"], f"keyframe {k}: local map")
        # what this keyframe exercised
        cur = b["cloud"]
        gone = [p for p in prev_cloud if p not in set(cur.tolist())]
        if len(cur) > len(prev_cloud) - len(gone):
            detached += 1
        if gone:
            unparked += 1
            pos = sorted(int(np.flatnonzero(prev_cloud == p)[0]) for p in gone)
            adjacent_runs += int(any(q - p == 1 for p, q in zip(pos, pos[1:])))
        prev_cloud = cur.copy()
    assert detached > 20 and unparked > 5 and adjacent_runs > 0, (detached, unparked, adjacent_runs)
    # SurfelMapping::Stop (:62-104): local surfels seen >= 5 times in map order, then every inactive surfel
    loc, ina = om.sf.map_get(), np.zeros(om.d.mslo_mapping_inactive(om.h, None, 0), SURFEL_DTYPE)
    if len(ina):
        om.d.mslo_mapping_inactive(om.h, _p(ina), len(ina))
    want = np.concatenate([loc[loc["updateTimes"] >= 5], ina])
    n = d.mh_stop(hh, None, 0)
    assert n == len(want)
    cloud = np.zeros((n, 11), np.float32)
    assert d.mh_stop(hh, _p(cloud), n) == n
    for col, f in enumerate(("px", "py", "pz", "nx", "ny", "nz")):
        assert np.allclose(cloud[:, col], want[f], atol=1e-4, rtol=0, equal_nan=True), f
    for col, f in zip((6, 7, 8), ("r", "g", "b")):
        assert np.array_equal(cloud[:, col].astype(np.int64), want[f] & 255), f
    assert np.allclose(cloud[:, 9], want["size"] * np.float32(1000), atol=1e-1, rtol=1e-6) and np.allclose(cloud[:, 10], want["weight"], atol=1e-4, rtol=0)
    om.close()
    d.mh_destroy(hh)
