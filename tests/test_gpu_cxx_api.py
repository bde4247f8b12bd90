"""-m gpu: builds and runs the C++ drop-in API test (tests/cxx/test_cuhe_api.cpp): CuHE.h classes and
gates on top of the C ABI, checked against host ZZX arithmetic (NTL if installed, else mini_ntl)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_api_program():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_cuhe_api")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout
