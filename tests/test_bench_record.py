"""CPU: the bookkeeping between bench.py and the committed counter record (profiles/traffic_rNN.json).

bench.py reports `roofline.traffic` (HBM-side bytes of the dominant kernel, from rocprofv3 PMC passes) only when the record was
measured on the kernel sources it is running: both carry a hash of the CODE of the three kernel headers -- comments and blank
space removed, so that editing a comment does not orphan a measurement, while any change of an instruction does."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_hash_ignores_comments_and_spacing_but_not_code():
    import bench
    a = "int f(int x) {   // adds one\n    return x + 1;  /* really */\n}\n\n"
    b = "int f(int x) {\n\treturn x + 1;\n}\n// trailing remark\n"
    c = "int f(int x) {\n    return x + 2;\n}\n"
    assert bench.code_only(a) == bench.code_only(b)
    assert bench.code_only(a) != bench.code_only(c)
    assert "/*" not in bench.code_only("x = 1; /* a\nmulti-line\nremark */ y = 2;") and "y = 2;" in bench.code_only("x = 1; /* a\nb */ y = 2;")


def test_newest_traffic_record_matches_the_kernel_sources_in_the_tree():
    """the record the round's bench line quotes was measured on THIS code (otherwise bench.py prints traffic: null and says why)"""
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    assert recs, "no counter record committed"
    rec = json.load(open(recs[-1]))
    if rec["kernel_sha16"] != bench.kernel_sha16():        # mid-round state: bench.py then reports traffic: null with the reason
        pytest.skip("the kernel headers changed since %s was measured: re-run tools/profile_final.sh before the round ends" % os.path.basename(recs[-1]))
    assert rec["transform_len"] == 65536 and rec["bytes_per_transform"] >= 10 * 65536          # at least the algorithmic bytes
    lanes = sum(rec["valu_lane_instructions_per_point"].values())
    assert 100 < lanes < 250                                                                   # vector lane-instructions per point
