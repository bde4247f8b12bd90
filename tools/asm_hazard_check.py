#!/usr/bin/env python3
"""Static hazard check of the generated gfx950 device code: VALU writes an SGPR / SGPR pair / VCC -> VALU reads it.

gfx940/950 need two wait states between a VALU instruction that writes an SGPR (a compare result, a carry-out, the
carry of v_mad_u64_u32, v_readlane ...) and a VALU instruction that reads it as a lane mask or carry-in -- VCC included
(LLVM GCNHazardRecognizer: VALUWriteSGPRVALURead, the `s_nop 1` hipcc puts between v_add_co_u32 and v_addc_co_u32).
LLVM pads its own code; it does not look inside or behind an inline-asm statement, and the field arithmetic of
cuhe_amd/csrc/modp.cuh / ops_kernels.cuh produces its carries in asm.  This script compiles the device code to assembly
and replays the rule over EVERY instruction of every kernel (not only the asm sites): for each VALU instruction it checks
that every SGPR it reads was last written by a VALU instruction at least two wait states earlier (an instruction = one
wait state, s_nop N = N + 1; an SALU write in between clears the hazard: SALU -> VALU is interlocked).
Exit code 0 = clean.  usage: tools/asm_hazard_check.py [extra hipcc flags] | --asm file.s
cuhe_amd/build.py runs it after compiling and refuses to keep a library whose code has findings."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VCC = (106, 107)
TWO_DST = re.compile(r"^v_(add|sub|subrev|addc|subb|subbrev)_co_|^v_mad_u64_u32|^v_mad_i64_i32|^v_div_scale")


def sregs(tok):
    """SGPR indices named by one operand (empty for VGPRs, literals, exec ...)."""
    tok = tok.strip()
    if tok == "vcc": return set(VCC)
    if tok == "vcc_lo": return {106}
    if tok == "vcc_hi": return {107}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    if m: return {int(m.group(1))}
    return set()


def check(lines):
    findings, asm_sites, valu_sgpr_reads = [], 0, 0
    age = {}                     # sgpr index -> wait states since a VALU instruction wrote it (absent: no pending VALU write)
    in_asm, func = False, "?"
    for ln, raw in enumerate(lines, 1):
        t = raw.strip()
        if ";;#ASMSTART" in t: in_asm = True; asm_sites += 1; continue
        if ";;#ASMEND" in t: in_asm = False; continue
        m = re.match(r"^(_Z\w+):", raw)
        if m: func, age = m.group(1), {}; continue
        if not t or t[0] in ";." or t.endswith(":") or t.startswith("//"): continue
        t = t.split(";")[0].strip()
        ops = t.split(None, 1)
        op = ops[0]
        if not re.match(r"^[a-z][a-z0-9_]+$", op): continue
        args = [a.strip() for a in ops[1].split(",")] if len(ops) > 1 else []
        if op == "s_nop":
            n = int(args[0], 0) + 1
            age = {r: a + n for r, a in age.items()}
            continue
        if op.startswith("v_"):
            ndst = 2 if TWO_DST.match(op) else 1
            reads, writes = set(), set()
            for a in args[ndst:]: reads |= sregs(a.split(" ")[0])
            for a in args[:ndst]: writes |= sregs(a)
            if reads: valu_sgpr_reads += 1
            hot = [r for r in reads if age.get(r, 99) < 2]
            if hot:
                findings.append("%s line %d%s: '%s' reads s%s %d wait state(s) after a VALU write" %
                                (func[:50], ln, " (inside asm)" if in_asm else "", t, sorted(hot), min(age[r] for r in hot)))
            age = {r: a + 1 for r, a in age.items()}
            for r in writes: age[r] = 0
        else:
            # SALU / memory / branch: one wait state; an SALU write to a register ends the VALU-write hazard on it
            age = {r: a + 1 for r, a in age.items()}
            if op.startswith("s_") and args:
                for r in sregs(args[0]): age.pop(r, None)
                if op.startswith(("s_and", "s_or", "s_xor", "s_andn2", "s_orn2", "s_nand", "s_nor", "s_xnor", "s_not", "s_add", "s_sub", "s_cmp", "s_lshl", "s_lshr", "s_ashr", "s_bfe", "s_mul", "s_min", "s_max", "s_abs", "s_addc", "s_subb", "s_bitcmp", "s_wqm", "s_bcnt", "s_ff", "s_flbit", "s_sext", "s_absdiff", "s_cselect") ):
                    pass                                     # (SCC is not an SGPR mask: nothing to track)
    return findings, asm_sites, valu_sgpr_reads


def main(argv):
    if len(argv) >= 2 and argv[0] == "--asm":
        lines = open(argv[1]).read().split("\n")
    else:
        with tempfile.TemporaryDirectory() as d:
            s = os.path.join(d, "dev.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S"] + argv +
                                  ["-o", s, os.path.join(ROOT, "cuhe_amd/csrc/cuhe_hip.hip"), "-I" + os.path.join(ROOT, "include")],
                                  stderr=subprocess.DEVNULL)
            lines = open(s).read().split("\n")
    findings, asm_sites, reads = check(lines)
    for f in findings[:40]: print(f)
    print("asm sites: %d, VALU instructions reading an SGPR / VCC: %d, findings: %d" % (asm_sites, reads, len(findings)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
