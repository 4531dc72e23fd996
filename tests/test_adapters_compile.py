"""The drop-in boundary meets a compiler: the C++ adapters that keep the reference's class
declarations are syntax-checked against declaration-only stubs of the host project's headers (tests/stubs/, see its README),
and include/msl.h is compiled as plain C11 -- the language a cgo / JNI / ctypes binding sees."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = os.path.join(ROOT, "manhattanslam_amd", "adapter")
INC = ["-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "include"), "-I", ADAPTER]


@pytest.mark.parametrize("tu", ["ORBextractor.cc", "SurfelFusion.cpp", "SurfelMapping.cpp", "PlaneExtractor.cpp"])
def test_adapter_translation_unit_compiles(tu):
    assert shutil.which("g++")
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", *INC, os.path.join(ADAPTER, tu)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_msl_h_is_valid_c11():
    for hdr in ("msl.h", "msl_debug.h"):
        r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                            os.path.join(ROOT, "include", hdr)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_c_host_links_against_libmsl(tmp_path):
    """A C translation unit that takes the address of every msl.h entry point links against libmsl.so (no GPU needed)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "msl.h")).read() + open(os.path.join(ROOT, "include", "msl_debug.h")).read()
    names = sorted(set(re.findall(r"MSL_API[^;(]*?\b(msl_\w+)\s*\(", hdr)))
    assert len(names) > 55
    src = tmp_path / "link_all.c"
    src.write_text('#include "msl_debug.h"\n#include <stdio.h>\nint main(void) {\n  const void *p[] = {' +
                   ", ".join(f"(const void *)&{n}" for n in names) +
                   '};\n  printf("%zu\\n", sizeof(p) / sizeof(p[0]));\n  return p[0] == 0;\n}\n')
    exe = tmp_path / "link_all"
    lib = os.path.join(ROOT, "manhattanslam_amd", "libmsl.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), lib, "-o", str(exe),
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
