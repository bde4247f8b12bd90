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


@pytest.mark.parametrize("params", [(3, 2, 8, 40, 20, 1155), (5, 2, 1, 61, 20, 8191)], ids=["toy1155", "dhs_simple"])
def test_dhs_scheme_flow(params):
    """keygen / encrypt / XOR / NOT / AND + relin + modSwitch (two levels) / decrypt through CuHE.h -- the checks of
    examples/DHS/simple_DHS.cu:49-170, with the reference example's own parameter set as the second case"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_dhs_flow")
    r = subprocess.run([exe] + [str(v) for v in params], capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout and "wrong" not in r.stdout
