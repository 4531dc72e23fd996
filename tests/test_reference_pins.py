"""Constants, tables and record layouts pinned against the reference's OWN source text (read as data: numbers and field names only).

The reference has no tests or golden vectors, but the numbers it hard-codes are part of its behaviour: the learned rBRIEF pattern, the patch and
border constants, the surfel-fusion macros, the field order of the records that cross the drop-in boundary, the matcher thresholds and the PEAC
defaults.  These tests run only where /root/reference exists (this container; the GPU box has no copy and they are not `-m gpu` tests)."""
import math
import os
import re

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _ref(path):
    return open(os.path.join(REF, path), errors="replace").read()


def _const(text, name):
    m = re.search(r"\b" + re.escape(name) + r"\s*=\s*([-+0-9.eE]+)", text) or re.search(r"#define\s+" + re.escape(name) + r"\s+([-+0-9.eE]+)", text)
    assert m, name
    return float(m.group(1))


def test_rbrief_pattern_is_the_references_table():
    src = _ref("src/ORBextractor.cc")
    body = src[src.index("bit_pattern_31_[256 * 4]"):]
    body = body[body.index("{") + 1:body.index("};")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ref = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int32)
    ours = np.array([int(v) for v in re.findall(r"-?\d+", re.sub(r"//.*", "", open(os.path.join(ROOT, "include", "msl_orb_pattern.inc")).read()))], np.int32)
    assert ref.size == 1024 and np.array_equal(ours[-1024:], ref)


def test_orb_constants():
    src = _ref("src/ORBextractor.cc")
    assert (_const(src, "PATCH_SIZE"), _const(src, "HALF_PATCH_SIZE"), _const(src, "EDGE_THRESHOLD")) == (31, 15, 19)
    assert _const(src, "const float W") == 30                      # FAST cell size, :726
    hip = open(os.path.join(ROOT, "manhattanslam_amd", "csrc", "msl_orb.hip")).read()
    assert "const float Wc = 30;" in hip and "(int)(31 * h->scale[l])" in hip
    fr = _ref("include/Frame.h")
    hdr = open(os.path.join(ROOT, "include", "msl.h")).read()
    assert int(_const(fr, "FRAME_GRID_ROWS")) == int(_const(hdr, "MSL_FRAME_GRID_ROWS")) == 48
    assert int(_const(fr, "FRAME_GRID_COLS")) == int(_const(hdr, "MSL_FRAME_GRID_COLS")) == 64


def test_surfel_fusion_macros():
    h = _ref("include/SurfelFusion.h")
    got = {k: _const(h, k) for k in ("ITERATION_NUM", "THREAD_NUM", "SP_SIZE", "MAX_ANGLE_COS", "HUBER_RANGE", "BASELINE", "DISPARITY_ERROR", "MIN_TOLERATE_DIFF")}
    assert got == {"ITERATION_NUM": 3, "THREAD_NUM": 10, "SP_SIZE": 8, "MAX_ANGLE_COS": 0.1, "HUBER_RANGE": 0.4, "BASELINE": 0.5, "DISPARITY_ERROR": 4.0,
                   "MIN_TOLERATE_DIFF": 0.1}
    csrc = os.path.join(ROOT, "manhattanslam_amd", "csrc")
    hip = "".join(open(os.path.join(csrc, f)).read() for f in ("msl_sf.h", "msl_sf_superpixel.hip", "msl_sf_map.hip", "msl_surfel.hip"))
    assert "constexpr int SP = 8;" in hip and "constexpr int NCHUNK = 10;" in hip
    assert "MAX_ANGLE_COS = 0.1, HUBER_RANGE = 0.4, MIN_TOLERATE_DIFF = 0.1" in hip
    assert "halfF = 0.5f * cameraF" in hip and "/ halfF * 4.0f" in hip          # BASELINE and DISPARITY_ERROR as exact float factors in k_fuse
    assert hip.count("for (int it = 0; it < 3; it++)") >= 1                      # ITERATION_NUM passes of (pixels, seeds)
    sm = _ref("src/SurfelMapping.cpp")
    assert "driftFreePoses(10)" in sm
    assert "driftFreePoses(10)" in open(os.path.join(ROOT, "manhattanslam_amd", "adapter", "SurfelMapping.cpp")).read()


def _fields(struct_text):
    out = []
    for decl in re.findall(r"\b(float|int|bool)\s+([^;]+);", struct_text):
        for name in decl[1].split(","):
            out.append((decl[0], name.split("=")[0].strip()))
    return out


def test_record_layouts_follow_the_references_structs():
    from manhattanslam_amd._lib import SEED_DTYPE, SURFEL_DTYPE
    s = _ref("include/Surfel.h")
    ref = _fields(s[s.index("struct Surfel"):s.index("};")])
    assert [n for _, n in ref] == list(SURFEL_DTYPE.names)
    assert [("f" if t == "float" else "i") for t, _ in ref] == [SURFEL_DTYPE[n].kind for n in SURFEL_DTYPE.names]
    h = _ref("include/SurfelFusion.h")
    body = h[h.index("struct SuperpixelSeed"):]
    ref = _fields(body[:body.index("};")])
    ours = [n for n in SEED_DTYPE.names if n != "_pad"]
    assert [n for _, n in ref] == ours
    assert [{"float": "f", "int": "i", "bool": "u"}[t] for t, _ in ref] == [SEED_DTYPE[n].kind for n in ours]
    assert SURFEL_DTYPE.itemsize == 56 and SEED_DTYPE.itemsize == 64


def test_matcher_constants():
    src = _ref("src/ORBmatcher.cc")
    assert (_const(src, "ORBmatcher::TH_HIGH"), _const(src, "ORBmatcher::TH_LOW"), _const(src, "ORBmatcher::HISTO_LENGTH")) == (100, 50, 30)
    assert "constexpr int TH_HIGH = 100, HISTO_LENGTH = 30;" in open(os.path.join(ROOT, "manhattanslam_amd", "csrc", "msl_match.hip")).read()


def test_peac_defaults_are_the_references():
    from manhattanslam_amd import peac
    p = peac.default_params()[0]
    ps = _ref("include/peac/AHCParamSet.hpp")
    ctor = ps[ps.index("ParamSet() :"):ps.index("initType(INIT_STRICT)")]
    num = lambda name: float(re.search(name + r"\(([-+0-9.eE]+)\)", ctor).group(1))
    assert p["depth_sigma"] == num("depthSigma") and p["std_tol_init"] == num("stdTol_init") and p["std_tol_merge"] == num("stdTol_merge")
    assert p["z_near"] == num("z_near") and p["z_far"] == num("z_far") and p["depth_alpha"] == num("depthAlpha") and p["depth_change_tol"] == num("depthChangeTol")
    deg = lambda name: float(re.search(name + r"\((?:std::cos\()?MACRO_DEG2RAD\(([0-9.]+)\)", ctor).group(1))
    assert p["angle_near"] == math.radians(deg("angle_near")) and p["angle_far"] == math.radians(deg("angle_far"))
    assert p["similarity_th_merge"] == math.cos(math.radians(deg("similarityTh_merge"))) and p["similarity_th_refine"] == math.cos(math.radians(deg("similarityTh_refine")))
    assert p["init_loose"] == 0 and "INIT_STRICT" in ps
    pf = _ref("include/peac/AHCPlaneFitter.hpp")
    ctor = pf[pf.index("maxStep(100000)") - 200:pf.index("erodeType(ERODE_ALL_BORDER)") + 40]
    num = lambda name: float(re.search(name + r"\(([-+0-9.eE]+)\)", ctor).group(1))
    assert (p["max_step"], p["min_support"], p["window_w"], p["window_h"]) == (num("maxStep"), num("minSupport"), num("windowWidth"), num("windowHeight"))
    assert p["do_refine"] == 1 and "doRefine(true)" in ctor and p["erode_type"] == 2      # ERODE_ALL_BORDER is the third enumerator
    assert int(re.search(r"ERODE_ALL_BORDER\s*=\s*(\d+)", pf).group(1)) == p["erode_type"]
