"""CPU: the plain PRINCE implementation and the S-box normal forms used by the homomorphic PRINCE clients
(tests/cxx/prince_common.hpp) against the known answers the reference's example holds (tests/golden/prince_kat.json:
examples/Prince/Prince.cu:96,108-145) and the five test vectors of the PRINCE paper."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "prince_plain_cli")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cxx", "prince_plain_cli.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unused-function", "-I" + os.path.join(ROOT, "tests", "cxx"), src, "-o", exe])
    return exe


def test_plain_prince_against_reference_known_answers():
    exe = _build()
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "prince_kat.json")))
    out = subprocess.run([exe, kat["plaintext"], kat["k0"], kat["k1"]], capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == kat["ciphertext"] == "%016x" % int(kat["ciphertext_bits"], 2)
    assert [int(s, 16) for s in out[1:13]] == [int(b, 2) for b in kat["round_states_bits"]]
    assert out[13:] == ["anf", "ok"]
    for pt, k0, k1, ct in kat["paper_vectors"]:
        got = subprocess.run([exe, pt, k0, k1], capture_output=True, text=True, check=True).stdout.split()[0]
        assert got == ct
