#!/usr/bin/env python3
"""Static hazard check of the generated gfx950 device code: VALU writes an SGPR / SGPR pair / VCC -> VALU reads it.

gfx940/950 need two wait states between a VALU instruction that writes an SGPR (a compare result, a carry-out, the
carry of v_mad_u64_u32, v_readlane ...) and a VALU instruction that reads it as a lane mask or carry-in -- VCC included
(LLVM GCNHazardRecognizer: VALUWriteSGPRVALURead, the `s_nop 1` hipcc puts between v_add_co_u32 and v_addc_co_u32).
LLVM pads its own code; it does not look inside or behind an inline-asm statement, and the field arithmetic of
cuhe_amd/csrc/modp.cuh / ops_kernels.cuh produces its carries in asm.  This script compiles the device code to assembly
and replays the rule over EVERY instruction of every kernel (not only the asm sites): for each VALU instruction it checks
that every SGPR it reads was last written by a VALU instruction at least two wait states earlier (an instruction = one
wait state, s_nop N = N + 1; an SALU write in between clears the hazard: SALU -> VALU is interlocked).  Branches and
labels are followed: a read at a loop head or behind a taken branch sees the writes of every predecessor.
Second rule (found on hardware, round 3): the destination registers of a v_mfma must not overlap its A / B source registers.
hipcc lets the result of v_mfma_i32_16x16x32_i8 start in the two registers of its 64-bit A operand when that operand dies
there; on MI355X the first two result registers then came out wrong (k_relin_mac_mfma_t32, ciphertext rows 4g, 4g + 1 of every
tile).  The kernels keep their operands alive past the instruction (`keep_alive` in ops_kernels.cuh); this check keeps it so.
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


def vrange(tok):
    """(file, first, last) of a VGPR / AGPR operand, None for anything else"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok.strip())
    if m: return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.fullmatch(r"([va])(\d+)", tok.strip())
    if m: return m.group(1), int(m.group(2)), int(m.group(2))
    return None


def check_mfma_overlap(lines):
    found, func = [], "?"
    for ln, raw in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m: func = m.group(1); continue
        t = raw.split(";")[0].strip()
        if not t.startswith("v_mfma"): continue
        ops = t.split(None, 1)
        args = [a.strip() for a in ops[1].split(",")]
        d = vrange(args[0])
        for name, a in (("A", args[1]), ("B", args[2])):
            r = vrange(a)
            if d and r and d[0] == r[0] and not (d[2] < r[1] or r[2] < d[1]):
                found.append("%s line %d: '%s': destination overlaps source %s" % (func[:50], ln, t, name))
    return found


def check(lines):
    """Replays the rule over the listing.  Control flow: the state (wait states since the last VALU write, per SGPR) that
    reaches a label is the most recent over ALL its predecessors -- the fall-through and every branch that names it, loop
    back-edges included -- found by iterating each function's listing until the states at its labels stop changing."""
    # split into functions
    funcs, cur = [], None
    for ln, raw in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m: cur = (m.group(1), []); funcs.append(cur); continue
        if cur is not None: cur[1].append((ln, raw))
    findings, asm_sites, valu_sgpr_reads = [], 0, 0

    def merge(a, b):                       # most recent write wins, register by register
        if a is None: return dict(b)
        out = dict(a)
        for r, v in b.items(): out[r] = min(out.get(r, 99), v)
        return out

    for func, body in funcs:
        incoming = {}                      # label -> state arriving over branches
        # ages saturate at 2 (two wait states are enough: entries that old are dropped), so the states at the labels can only
        # move finitely often and the iteration runs until nothing changes; a function that has not settled after kMaxSweeps
        # is reported as a finding, never passed silently
        kMaxSweeps = 64
        for sweep in range(kMaxSweeps):
            last = sweep == kMaxSweeps - 1
            found, sites, reads_n = [], 0, 0
            age, in_asm, dead, changed = {}, False, False, False
            for ln, raw in body:
                t = raw.strip()
                if ";;#ASMSTART" in t: in_asm = True; sites += 1; continue
                if ";;#ASMEND" in t: in_asm = False; continue
                m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", t)
                if m and not t.startswith("//"):
                    lab = m.group(1)
                    age = merge(None if dead else age, incoming.get(lab, {})) if (lab in incoming or not dead) else {}
                    dead = False
                    continue
                if not t or t[0] in ";." or t.startswith("//"): continue
                t = t.split(";")[0].strip()
                ops = t.split(None, 1)
                op = ops[0]
                if not re.match(r"^[a-z][a-z0-9_]+$", op): continue
                args = [a.strip() for a in ops[1].split(",")] if len(ops) > 1 else []
                if op == "s_nop":
                    n = int(args[0], 0) + 1
                    age = {r: a + n for r, a in age.items() if a + n < 2}
                    continue
                if op.startswith("v_"):
                    ndst = 2 if TWO_DST.match(op) else 1
                    reads, writes = set(), set()
                    for a in args[ndst:]: reads |= sregs(a.split(" ")[0])
                    for a in args[:ndst]: writes |= sregs(a)
                    if reads: reads_n += 1
                    hot = [r for r in reads if age.get(r, 99) < 2]
                    if hot:
                        found.append("%s line %d%s: '%s' reads s%s %d wait state(s) after a VALU write" %
                                     (func[:50], ln, " (inside asm)" if in_asm else "", t, sorted(hot), min(age[r] for r in hot)))
                    age = {r: a + 1 for r, a in age.items() if a + 1 < 2}
                    for r in writes: age[r] = 0
                    continue
                # SALU / memory / branch: one wait state; an SALU write to a register ends the VALU-write hazard on it
                age = {r: a + 1 for r, a in age.items() if a + 1 < 2}
                if op.startswith("s_") and args and not op.startswith(("s_cbranch", "s_branch")):
                    for r in sregs(args[0]): age.pop(r, None)
                if op.startswith(("s_cbranch", "s_branch")) and args:
                    tgt = args[-1]
                    new = merge(incoming.get(tgt), age)
                    if new != incoming.get(tgt): incoming[tgt] = new; changed = True
                    if op == "s_branch": dead = True           # nothing falls through an unconditional branch
            if last or not changed:
                if changed: found.append("%s: the label states did not settle in %d sweeps -- result not trusted" % (func[:50], kMaxSweeps))
                findings += found; asm_sites += sites; valu_sgpr_reads += reads_n
                break
    return findings, asm_sites, valu_sgpr_reads


def main(argv):
    if len(argv) >= 2 and argv[0] == "--asm":
        listings = [open(argv[1]).read().split("\n")]
    else:                                      # every translation unit of the library (cuhe_amd/build.py: UNITS)
        sys.path.insert(0, ROOT)
        from cuhe_amd import build as B
        listings = []
        with tempfile.TemporaryDirectory() as d:
            procs = []
            for name, src, extra in B.UNITS:
                out = os.path.join(d, name + ".s")
                procs.append((out, subprocess.Popen([B.HIPCC] + B.FLAGS + extra + ["--cuda-device-only", "-S"] + argv + ["-o", out, os.path.join(B.CSRC, src)],
                                                    stderr=subprocess.DEVNULL)))
            for out, p in procs:
                if p.wait() != 0: raise RuntimeError("compilation failed: " + out)
                listings.append(open(out).read().split("\n"))
    findings, asm_sites, reads = [], 0, 0
    for lines in listings:
        f, a, r = check(lines)
        findings += f + check_mfma_overlap(lines); asm_sites += a; reads += r
    for f in findings[:40]: print(f)
    print("asm sites: %d, VALU instructions reading an SGPR / VCC: %d, findings: %d" % (asm_sites, reads, len(findings)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
