"""CPU (cross-compile only): the generated gfx950 code never has a VALU instruction reading an SGPR / SGPR pair / VCC
less than two wait states after a VALU instruction wrote it (LLVM pads its own code for that on gfx940/950, nothing
pads inside or behind an inline-asm statement, and the field arithmetic produces its carries in asm).
tools/asm_hazard_check.py replays the rule over every instruction of every kernel; cuhe_amd/build.py runs it on every
build and refuses a library with findings -- this test checks both the checker (it must fire on a known-bad sequence)
and the shipped code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_checker_fires_on_known_hazards():
    import asm_hazard_check as H
    bad = """_Z1kv:
	v_sub_co_u32_e32 v0, vcc, v4, v0
	v_subb_co_u32_e32 v1, vcc, v5, v1, vcc
	v_mad_u64_u32 v[2:3], s[0:1], v4, v5, v[2:3]
	v_cndmask_b32_e64 v6, 0, 1, s[0:1]
	v_cmp_lt_u64_e64 s[2:3], v[2:3], v[4:5]
	s_nop 0
	v_cndmask_b32_e64 v6, 0, 1, s[2:3]
"""
    findings, _, _ = H.check(bad.split("\n"))
    assert len(findings) == 3, findings
    good = """_Z1kv:
	v_sub_co_u32_e32 v0, vcc, v4, v0
	s_nop 1
	v_subb_co_u32_e32 v1, vcc, v5, v1, vcc
	v_mad_u64_u32 v[2:3], s[0:1], v4, v5, v[2:3]
	s_or_b64 s[0:1], s[0:1], s[4:5]
	v_cndmask_b32_e64 v6, 0, 1, s[0:1]
	v_cmp_lt_u64_e64 s[2:3], v[2:3], v[4:5]
	v_add_u32_e32 v9, v9, v9
	v_add_u32_e32 v9, v9, v9
	v_cndmask_b32_e64 v6, 0, 1, s[2:3]
"""
    findings, _, _ = H.check(good.split("\n"))
    assert findings == [], findings


def test_checker_follows_branches_and_loop_back_edges():
    import asm_hazard_check as H
    # the carry written at the end of the loop body is read by the first instruction of the next iteration
    loop = """_Z1kv:
.LBB0_1:
	v_addc_co_u32_e64 v1, s[4:5], v1, v2, s[4:5]
	v_add_u32_e32 v9, v9, v9
	v_add_u32_e32 v9, v9, v9
	v_add_co_u32_e64 v0, s[4:5], v0, v3
	s_cbranch_scc1 .LBB0_1
"""
    findings, _, _ = H.check(loop.split("\n"))
    assert len(findings) == 1 and "v_addc_co_u32_e64" in findings[0], findings
    # a write just before a taken forward branch reaches the read at its target
    fwd = """_Z1kv:
	v_cmp_lt_u64_e64 s[2:3], v[2:3], v[4:5]
	s_cbranch_execz .LBB0_2
	v_add_u32_e32 v9, v9, v9
	v_add_u32_e32 v9, v9, v9
	v_add_u32_e32 v9, v9, v9
.LBB0_2:
	v_cndmask_b32_e64 v6, 0, 1, s[2:3]
"""
    findings, _, _ = H.check(fwd.split("\n"))
    assert len(findings) == 1, findings


def test_device_code_has_no_sgpr_hazards():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_check.py")], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "findings: 0" in r.stdout


def test_checker_rejects_a_matrix_core_result_that_overlaps_its_operands():
    """Second rule (profiles/r03_mac_tile_experiments.txt): the destination of a v_mfma must not share registers with its A / B
    source -- hipcc allows it for 128-bit results, MI355X then returned wrong values in the overlapping registers."""
    import asm_hazard_check as H
    bad = """_Z1kv:
	v_mfma_i32_16x16x32_i8 v[98:101], v[98:99], v[160:161], v[108:111]
	v_mfma_i32_16x16x64_i8 v[102:105], v[70:73], v[102:105], v[78:81]
	v_mfma_f32_32x32x8_f16 a[0:15], a[16:17], a[2:3], a[0:15]
"""
    found = H.check_mfma_overlap(bad.split("\n"))
    assert len(found) == 3 and "source A" in found[0] and "source B" in found[1], found
    good = """_Z1kv:
	v_mfma_i32_16x16x32_i8 v[98:101], v[102:103], v[160:161], v[98:101]
	v_mfma_i32_16x16x64_i8 v[66:69], v[70:73], v[2:5], v[100:103]
	v_mfma_f32_32x32x8_f16 a[0:15], v[0:1], v[2:3], a[0:15]
"""
    assert H.check_mfma_overlap(good.split("\n")) == []
