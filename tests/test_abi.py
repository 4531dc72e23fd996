"""The C-ABI library loads and exports every symbol include/msl.h (the drop-in boundary) and include/msl_debug.h (test / measurement
accessors) declare; no compute calls (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="msl.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"MSL_API[^;(]*?\b(msl_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from manhattanslam_amd import _lib
    api, dbg = _declared("msl.h"), _declared("msl_debug.h")
    assert len(api) >= 40 and len(dbg) >= 14
    # the drop-in header carries no debug / profiling / scratch-peek entry points, the debug header nothing else
    assert not [n for n in api if "debug" in n or "profile" in n or "kernel_name" in n], api
    assert all("debug" in n or "profile" in n or "kernel_name" in n for n in dbg), dbg
    names = sorted(api + dbg)
    dll = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/ but not exported by libmsl.so"
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)


def test_struct_layouts_match_reference_types():
    from manhattanslam_amd import KEYPOINT_DTYPE, SURFEL_DTYPE, SEED_DTYPE
    assert KEYPOINT_DTYPE.itemsize == 28      # cv::KeyPoint
    assert SURFEL_DTYPE.itemsize == 56        # struct Surfel, reference include/Surfel.h:28-37
    assert SEED_DTYPE.itemsize == 64          # SurfelFusion::SuperpixelSeed, include/SurfelFusion.h:46-58
    assert [SURFEL_DTYPE.fields[n][1] for n in ("px", "size", "r", "weight", "updateTimes", "lastUpdate")] == [0, 24, 32, 44, 48, 52]


def test_version_and_error_strings():
    from manhattanslam_amd import lib
    assert b"gfx950" in lib.msl_version()
    assert isinstance(lib.msl_last_error(), bytes)
    assert lib.msl_orb_kernel_name(1) == b"k_fast" and lib.msl_sf_kernel_name(7) == b"k_fuse"


def test_no_cpu_fallback():
    """Without an MI355X the constructors fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from manhattanslam_amd import MslError, ORBextractor, SurfelFusion, device_count
    assert device_count() == 0
    with pytest.raises(MslError, match="no HIP device|no CPU fallback|not usable"):
        ORBextractor(1000, 1.2, 8, 20, 7)
    with pytest.raises(MslError):
        SurfelFusion(640, 480, 500.0, 500.0, 320.0, 240.0, 30.0, 0.5)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "manhattanslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in txt.replace("oracle/ is test infrastructure", "") or f.endswith(".md"), os.path.join(dirpath, f)
                assert "libmsl_oracle" not in txt and "oracle_lib" not in txt, os.path.join(dirpath, f)


def test_synth_is_deterministic():
    from manhattanslam_amd import synth
    a, b = synth.orb_frame(123), synth.orb_frame(123)
    assert np.array_equal(a, b) and a.shape == (480, 640) and a.dtype == np.uint8
    g1 = synth.surfel_frame(3)
    g2 = synth.surfel_frame(3)
    assert all(np.array_equal(x, y) for x, y in zip(g1, g2))
    m = synth.surfel_map(1000)
    assert m.dtype.itemsize == 56 and np.array_equal(m, synth.surfel_map(1000))


def test_no_exception_crosses_the_c_boundary(tmp_path):
    """every entry point is noexcept and returns a status.  (1) every declaration of both headers carries MSL_NOEXCEPT and a C++17
    compiler agrees (noexcept is part of the function type); (2) every extern "C" definition of the library is `noexcept { try { ... } MSL_ABI_CATCH_* }`;
    (3) failures raised INSIDE the library -- std::bad_alloc, another std::exception, a non-standard one, std::bad_alloc in a worker thread of the
    plane extractor's pool -- come back as MSL_ERR_NOMEM / MSL_ERR_INTERNAL with a message; the process lives on."""
    import subprocess
    from manhattanslam_amd import _lib
    names = []
    for hname in ("msl.h", "msl_debug.h"):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", hname)).read(), flags=re.S)
        decls = re.findall(r"^MSL_API[^;]*;", src, flags=re.M)
        assert decls and all(d.rstrip(";").rstrip().endswith("MSL_NOEXCEPT") for d in decls), [d for d in decls if "MSL_NOEXCEPT" not in d]
        names += re.findall(r"MSL_API[^;(]*?\b(msl_\w+)\s*\(", src)
    cpp = tmp_path / "noexcept_check.cpp"
    cpp.write_text('#include "msl.h"\n#include "msl_debug.h"\n'
                   "template <class R, class... A> constexpr bool ne(R (*)(A...) noexcept) { return true; }\n"
                   "template <class R, class... A> constexpr bool ne(R (*)(A...)) { return false; }\n" +
                   "".join(f'static_assert(ne(&{n}), "{n} is not noexcept");\n' for n in sorted(set(names))))
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(cpp)], check=True)
    subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-x", "c", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "include", "msl_debug.h")], check=True)
    csrc = os.path.join(ROOT, "manhattanslam_amd", "csrc")
    allsrc = ""
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            src = open(os.path.join(csrc, f)).read()
            assert src.count(" noexcept {") == src.count("MSL_ABI_CATCH_"), f      # every noexcept entry has its try / catch
            allsrc += src
    for n in sorted(set(names)):
        assert re.search(r"\b%s\s*\([^;{}]*\)\s*noexcept\s*\{" % n, allsrc), f"{n}: no noexcept definition behind the barrier"
    lib = _lib.lib
    want = {0: -6, 1: -7, 2: -7, 3: -6}     # MSL_ERR_NOMEM, MSL_ERR_INTERNAL
    for kind, rc in want.items():
        assert lib.msl_debug_throw(kind) == rc, kind
        assert lib.msl_last_error()
    assert lib.msl_debug_throw(99) == 0
