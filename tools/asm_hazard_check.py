#!/usr/bin/env python3
"""Static check of the device code around the inline-asm carry producers of modp.cuh.

gfx940/950: a VALU instruction that reads an SGPR pair written by the VALU instruction just before it needs two wait states
(LLVM: VALUWriteSGPRVALURead); LLVM pads its own code but does not look inside or behind an asm statement.  mad_eps hands
the carry-out of its v_mad_u64_u32 to the compiler as a lane mask in an SGPR pair; this script compiles the device code to
assembly and verifies that in every place the first reader of that pair within two issue slots is an SALU instruction
(s_or_b64 ...), never a VALU one, and that the multi-instruction asm blocks keep their carries in VCC.
Exit code 0 = clean.  usage: tools/asm_hazard_check.py [extra hipcc flags, e.g. -DCUHE_SUBP_VARIANT=4 to see it fire]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "dev.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S"] + sys.argv[1:] +
                              ["-o", s, os.path.join(ROOT, "cuhe_amd/csrc/cuhe_hip.hip"), "-I" + os.path.join(ROOT, "include")],
                              stderr=subprocess.DEVNULL)
        lines = open(s).read().split("\n")
    single = multi = bad = 0
    i = 0
    while i < len(lines):
        if ";;#ASMSTART" not in lines[i]:
            i += 1
            continue
        j, body = i + 1, []
        while ";;#ASMEND" not in lines[j]:
            if lines[j].strip():
                body.append(lines[j].strip())
            j += 1
        if len(body) == 1 and body[0].startswith("v_mad_u64_u32"):
            single += 1
            sreg = re.match(r"v_mad_u64_u32 v\[\d+:\d+\], (s\[\d+:\d+\]|vcc)", body[0]).group(1)
            k, slots = j + 1, 0
            while k < len(lines) and slots < 2:
                t = lines[k].strip(); k += 1
                if not t or t[0] in ";." or t.endswith(":"):
                    continue
                if t.startswith("s_nop"):
                    slots += int(t.split()[1]) + 1
                    continue
                ops = t.split(None, 1)
                args = [x.strip() for x in ops[1].split(",")] if len(ops) > 1 else []
                if sreg in args:
                    if t.startswith("v_") and (t.startswith("v_cndmask") or sreg in args[2:]):
                        bad += 1
                        print("VALU reads the carry pair %s %d slot(s) after the asm: %s" % (sreg, slots, t))
                    break
                slots += 1
        elif len(body) > 1:
            multi += 1
            for a, b in zip(body, body[1:]):            # VALU -> VALU through an explicit SGPR pair inside one string
                m = re.match(r"v_\w+ v\d+, (s\[\d+:\d+\])", a)
                if m and b.startswith("v_") and m.group(1) in b.split(None, 1)[1]:
                    bad += 1
                    print("inside an asm string: '%s' then '%s'" % (a, b))
        i = j + 1
    print("single-instruction carry producers: %d, multi-instruction asm blocks: %d, findings: %d" % (single, multi, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
