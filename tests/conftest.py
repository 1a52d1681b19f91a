import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, as __graft_entry__.build() does
    lib = os.path.join(ROOT, "climateparameterizations.jl_amd", "libcolnde.so")
    ref = os.path.join(ROOT, "oracle", "_build", "libcolnde_ref.so")
    if not (os.path.exists(lib) and os.path.exists(ref)):
        import __graft_entry__
        __graft_entry__.build()


# `make -C climateparameterizations.jl_amd/csrc asan_test`: the same suite against the host-sanitizer build of the library (COLNDE_LIB=libcolnde_asan.so)
if os.environ.get("COLNDE_LIB"):
    from colnde import _lib as _colnde_lib
    _colnde_lib.LIB_PATH = os.path.join(ROOT, "climateparameterizations.jl_amd", os.environ["COLNDE_LIB"])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
