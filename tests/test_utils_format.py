"""CPU: builds and runs tests/cxx/test_utils_format.cpp -- the key-file text format of cuhe/Utils.h
(Picklable / PicklableMap) as provided by cuhe_amd/cxx/Utils.h.  Host only, no HIP library involved."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_key_file_format_program():
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_utils_format")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-1000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout
