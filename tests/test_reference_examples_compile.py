"""CPU, only where /root/reference is mounted (this container; skipped on the GPU box): the reference's example
sources -- examples/DHS/{DHS,simple_DHS}.cu and examples/Prince/{DHS,Prince,Timer,test_Prince}.cu -- are run through
`g++ -fsyntax-only` UNCHANGED, from where they lie, against this repository's headers (cuhe_amd/cxx/CuHE.h, Utils.h).
They include "../../cuhe/CuHE.h" relative to their own directory, so a scratch tree of symlinks puts cuhe_amd/cxx in
that place.  NTL is not installed: its types come from this repository's fallback headers (cuhe_amd/cxx/mini_ntl).
Nothing is built or run -- this is a check that every cuHE symbol the examples use exists here with a compatible
signature; what the calls DO is covered by the clients in tests/cxx."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/examples"
SOURCES = ["DHS/DHS.cu", "DHS/simple_DHS.cu", "Prince/DHS.cu", "Prince/Prince.cu", "Prince/Timer.cu", "Prince/test_Prince.cu"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not mounted here")
def test_reference_examples_pass_syntax_check_against_our_headers(tmp_path):
    (tmp_path / "examples").mkdir()
    os.symlink(os.path.join(ROOT, "cuhe_amd", "cxx"), tmp_path / "cuhe")
    for d in ("DHS", "Prince"):
        (tmp_path / "examples" / d).mkdir()
        for f in os.listdir(os.path.join(REF, d)):
            if f.endswith((".cu", ".h")):
                os.symlink(os.path.join(REF, d, f), tmp_path / "examples" / d / f)
    inc = ["-I" + os.path.join(ROOT, "cuhe_amd", "cxx", "mini_ntl")]
    for src in SOURCES:
        r = subprocess.run(["g++", "-std=c++17", "-x", "c++", "-fsyntax-only", "-fopenmp"] + inc + [os.path.join("examples", src)],
                           cwd=tmp_path, capture_output=True, text=True, timeout=300)
        errors = [l for l in r.stderr.splitlines() if "error" in l]
        assert r.returncode == 0 and not errors, src + "\n" + "\n".join(errors[:20])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not mounted here")
def test_reference_examples_compile_and_link_against_our_library(tmp_path):
    """One step further than the syntax check: the six sources are COMPILED, from where they lie and
    unchanged, and the two programs (simple_DHS, test_Prince) are LINKED against libcuHE.so + libcuhe_hip.so, so that an
    undefined or mis-mangled cuHE symbol is caught.  Objects and programs stay in tmp_path (never committed, never run:
    there is no GPU here, and the NTL types are this repository's mini_ntl stand-ins)."""
    import __graft_entry__ as ge
    lib = os.path.join(ROOT, "cuhe_amd", "lib")
    if not (os.path.exists(os.path.join(lib, "libcuHE.so")) and os.path.exists(os.path.join(lib, "libcuhe_hip.so"))):
        ge.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s"])
    (tmp_path / "examples").mkdir()
    os.symlink(os.path.join(ROOT, "cuhe_amd", "cxx"), tmp_path / "cuhe")
    for d in ("DHS", "Prince"):
        (tmp_path / "examples" / d).mkdir()
        for f in os.listdir(os.path.join(REF, d)):
            if f.endswith((".cu", ".h")):
                os.symlink(os.path.join(REF, d, f), tmp_path / "examples" / d / f)
    inc = ["-I" + os.path.join(ROOT, "cuhe_amd", "cxx", "mini_ntl")]
    programs = {"simple_DHS": ["DHS/DHS.cu", "DHS/simple_DHS.cu"], "test_Prince": ["Prince/DHS.cu", "Prince/Prince.cu", "Prince/Timer.cu", "Prince/test_Prince.cu"]}
    for prog, srcs in programs.items():
        objs = []
        for src in srcs:
            obj = str(tmp_path / (src.replace("/", "_") + ".o"))
            r = subprocess.run(["g++", "-std=c++17", "-O1", "-fno-lifetime-dse", "-x", "c++", "-c", "-fopenmp"] + inc + [os.path.join("examples", src), "-o", obj],
                               cwd=tmp_path, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, src + "\n" + r.stderr[-3000:]
            objs.append(obj)
        r = subprocess.run(["g++", "-fopenmp", "-o", str(tmp_path / prog)] + objs + ["-L" + lib, "-lcuHE", "-lcuhe_hip", "-Wl,-rpath," + lib, "-lpthread"],
                           cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, prog + " does not link:\n" + r.stderr[-3000:]
        undefined = subprocess.run(["nm", "-u", "-C", str(tmp_path / prog)], capture_output=True, text=True).stdout
        assert "cuHE::" in undefined, "the program is expected to import its cuHE symbols from libcuHE.so"
