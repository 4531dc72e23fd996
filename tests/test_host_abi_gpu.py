"""The C ABI driven by a plain C++ host (tests/host_abi.cpp, built with g++ against include/msl.h and libmsl.so) instead of
ctypes: ORB output bit-exact, surfels within 1e-4 of the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_matches_oracle(oracle, tmp_path):
    from manhattanslam_amd import synth, SURFEL_DTYPE, KEYPOINT_DTYPE
    from tests.oracle_lib import OracleSurfel
    from tests.test_surfel_gpu import assert_surfels_close
    lib = os.path.join(ROOT, "manhattanslam_amd", "libmsl.so")
    exe = tmp_path / "host_abi"
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_abi.cpp"), lib,
                        "-o", str(exe), "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    I = synth.TUM1
    _, depth, member, pose = synth.surfel_frame(2, variant="B")
    gray = synth.orb_frame(synth.ORB_SEED + 3)
    local = synth.surfel_map(30000, ref=2).astype(SURFEL_DTYPE)
    for name, a in (("gray", gray), ("depth", depth), ("member", member), ("pose", pose), ("local", local)):
        np.ascontiguousarray(a).tofile(tmp_path / f"{name}.bin")
    r = subprocess.run([str(exe), str(tmp_path), "640", "480", repr(I["fx"]), repr(I["fy"]), repr(I["cx"]), repr(I["cy"]), str(len(local)), "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ex = oracle.orb_create()
    ko, do = ex.extract(gray)
    kg = np.fromfile(tmp_path / "kps.bin", KEYPOINT_DTYPE)
    dg = np.fromfile(tmp_path / "desc.bin", np.uint8).reshape(-1, 32)
    assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do)
    tabs = np.fromfile(tmp_path / "tables.bin", np.float32).reshape(4, 8)
    for got, want in zip(tabs, ex.tables()[:4]):
        assert np.array_equal(got.view(np.int32), want.view(np.int32))
    osf = OracleSurfel(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    lo, no = osf.fuse(2, gray, depth, member, pose, local)
    assert_surfels_close(np.fromfile(tmp_path / "local_out.bin", SURFEL_DTYPE), lo, "local (C++ host)")
    assert_surfels_close(np.fromfile(tmp_path / "new.bin", SURFEL_DTYPE), no, "new (C++ host)")
