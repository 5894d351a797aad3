#!/usr/bin/env python3
"""Finds dependent matrix instructions of DIFFERENT shapes with too few wait states between them in gfx950 ISA listings.

Background (round 5, profiles/r5_kernels/a96_mixed_chain.txt): hipcc (ROCm 7.2) inserts no wait states between a
v_mfma_f32_16x16x32_f16 and a v_mfma_f32_16x16x16_f16 that reads the first one's vDst as its SrcC - it treats the pair like two
instructions of ONE opcode, which the hardware forwards - and the second product then starts from a half-written accumulator:
outputs off by up to 0.8 on unit-scale data and different from run to run.  Two instructions of the same opcode are fine
back to back; a pair of different opcodes needs the first one's passes to have drained (the CDNA3 ISA's table for "XDL write
VGPR -> XDL read SrcC, different opcode": 5 wait states behind a 4-pass instruction, 9 behind 8 passes, 17 behind 16).

    tools/isa_mfma_chain.py waifu2x-tensorrt_amd/build/*.isa.s        (the Makefile leaves one listing per kernel file)

prints every such pair with fewer than `--need` wait states (default 10) and exits 1 if there is one.  A register that a vector or
memory instruction rewrites between the two is not a dependency and is not reported."""
import re
import sys

PASSES = {"4x4x": 2, "16x16x": 4, "32x32x": 8}        # f16 / bf16 shapes of this library: passes of 4 cycles (gfx950: 16x16x32 and 32x32x16 at the passes of 16x16x16 / 32x32x8)


def reg_range(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok.strip().rstrip(","))
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r"([va])(\d+)$", tok.strip().rstrip(","))
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None


def overlap(a, b):
    return a and b and a[0] == b[0] and not (a[2] < b[1] or b[2] < a[1])


def scan(path, need):
    lines = [l.strip() for l in open(path)]
    code = [(n + 1, l) for n, l in enumerate(lines) if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    found = []
    for i, (ln, l) in enumerate(code):
        if not l.startswith("v_mfma"):
            continue
        opc = l.split()[0]
        ops = l.split(None, 1)[1].split(", ")
        srcc = reg_range(ops[3]) if len(ops) > 3 else None
        if not srcc:
            continue
        live = set(range(srcc[1], srcc[2] + 1))           # registers of SrcC not rewritten on the way back
        ws = 0
        for k in range(1, 40):
            if i - k < 0 or ws >= need or not live:
                break
            pl = code[i - k][1]
            if pl.startswith("s_nop"):
                ws += int(pl.split()[1]) + 1
                continue
            if pl.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc")):
                break
            parts = pl.split(None, 1)
            dst = reg_range(parts[1].split(", ")[0]) if len(parts) > 1 else None
            if pl.startswith("v_mfma"):
                if dst and dst[0] == srcc[0] and live & set(range(dst[1], dst[2] + 1)) and parts[0] != opc:
                    found.append((path, code[i - k][0], pl, ln, l, ws))
                    break
            if dst and dst[0] == srcc[0] and not pl.startswith(("ds_write", "buffer_store", "global_store", "s_")):
                live -= set(range(dst[1], dst[2] + 1))
            ws += 1
    return found


def main(argv):
    need = 10
    paths = []
    for a in argv:
        if a.startswith("--need="):
            need = int(a.split("=")[1])
        else:
            paths.append(a)
    bad = []
    nm = 0
    for p in paths:
        nm += sum(1 for l in open(p) if l.strip().startswith("v_mfma"))
        bad += scan(p, need)
    for path, l0, a, l1, b, ws in bad:
        print(f"{path}:{l0}-{l1}: {ws} wait states between\n    {a}\n    {b}")
    print(f"{len(paths)} listings, {nm} matrix instructions, {len(bad)} dependent pairs of different shapes closer than {need} wait states")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
