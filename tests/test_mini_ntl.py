"""CPU: the header-only NTL subset the C++ layer falls back to when NTL is absent (cuhe_amd/cxx/mini_ntl: ZZ with
Knuth division, ZZX, ZZ_p, ZZ_pX, ZZ_pE).  tests/cxx/test_mini_ntl.cpp prints ~1400 operations on seeded random
operands (word-boundary sizes 31..65, 96, 127/128, up to 1000 bits); every line is recomputed here with Python ints."""
import math
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _polys(text):
    return [[int(v) for v in part.split()] for part in text.split(";")[:-1]]


def _trim(p):
    while p and p[-1] == 0:
        p = p[:-1]
    return p


def _pmul(a, b):
    r = [0] * (len(a) + len(b) - 1) if a and b else []
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            r[i + j] += x * y
    return _trim(r)


def _pmod(a, m, q=None):
    a = list(a)
    while len(a) >= len(m):
        c = a[-1]
        if c:
            for i in range(len(m)):
                a[len(a) - len(m) + i] -= c * m[i]
        a.pop()
    if q:
        a = [v % q for v in a]
    return _trim(a)


def test_mini_ntl_against_python_integers(tmp_path):
    exe = str(tmp_path / "test_mini_ntl")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DCUHE_MINI_NTL", "-I" + os.path.join(ROOT, "cuhe_amd", "cxx", "mini_ntl"),
                           "-o", exe, os.path.join(ROOT, "tests", "cxx", "test_mini_ntl.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, check=True).stdout.splitlines()
    assert len(out) > 1300
    seen = set()
    for line in out:
        op, rest = line.split(" ", 1)
        lhs, rhs = rest.split(" = ") if " = " in rest else rest.split(" =")
        seen.add(op)
        if op in ("zzxmul", "zzxmod", "pEinv"):
            if op == "pEinv":
                q, lhs = lhs.split(" ;", 1)
                q = int(q)
                (f,), (g, one) = _polys(lhs), _polys(rhs)
                P = [1, 1, 0, 1, 1, 0, 0, 0, 1]
                assert one == [1] and _pmod(_pmul(f, g), P, q) == [1], line
                assert all(0 <= c < q for c in g)
            else:
                (a, b), (r,) = _polys(lhs), _polys(rhs)
                assert r == (_pmul(a, b) if op == "zzxmul" else _pmod(a, b)), line
            continue
        args, got = [int(v) for v in lhs.split()], int(rhs)
        a = args[0]; b = args[1] if len(args) > 1 else None
        want = {"add": lambda: a + b, "sub": lambda: a - b, "mul": lambda: a * b, "div": lambda: a // b, "mod": lambda: a % b,
                "gcd": lambda: math.gcd(a, b), "cmp": lambda: (a > b) - (a < b), "shl": lambda: a << b, "shr": lambda: a >> b,
                "bits": lambda: a.bit_length(), "bytes": lambda: a, "low40": lambda: a & ((1 << 40) - 1), "pow2": lambda: 1 << a,
                "power": lambda: a ** b,
                "invmod": lambda: pow(a, -1, b) if math.gcd(a, b) == 1 else -1}[op]()
        assert got == want, line
    assert {"add", "sub", "mul", "div", "mod", "gcd", "cmp", "shl", "shr", "bits", "bytes", "low40", "pow2", "power", "invmod", "zzxmul", "zzxmod",
            "pEinv"} <= seen
