#!/usr/bin/env python3
"""Build-time check of the one hazard the compiler cannot see in match.hip.

The prefilter folds MFMA results with inline-asm VALU instructions (v_max3_f32, v_cmp + v_addc ...).
gfx950 does not interlock a VALU read of a register an MFMA is still writing: the ISA requires wait
states between the two (8-pass XDL op -> VALU read / write of its destination: 11), the compiler's
hazard recognizer inserts them for instructions it schedules itself, and it knows nothing about the
operands of an asm block.  Round 2 found a stale-accumulator read of exactly this kind in a variant.

    tools/check_mfma_hazard.py <device assembly of match.hip> [required wait states, default 12]

For every instruction between ';;#ASMSTART' and ';;#ASMEND' in the MFMA kernels, walk back through
the preceding instructions (straight-line order) and fail if a v_mfma whose destination overlaps any
VGPR the asm instruction names lies closer than the required distance.  Distance in wait states: one
per instruction, N + 1 for 's_nop N', and for an MFMA in between its number of passes -- a later MFMA
cannot enter the matrix pipe before the earlier one has left it (PMC: 32 busy cycles = 8 passes per
v_mfma_f32_32x32x16_bf16), so everything behind it is at least that far from the earlier one.  Prints the closest pair it saw.  Limits: the walk follows layout order, not
the control-flow graph (a branch INTO the middle of such a window from elsewhere is not seen); it stops
at function boundaries."""
import re
import sys

path = sys.argv[1]
need = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lines = open(path).read().split("\n")

# passes of the MFMA opcodes this file uses (an unknown one counts as the shortest, 4)
PASSES = {"v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16}

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels():
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_ZN") and l.rstrip().endswith(":") or (l.startswith("_ZN") and ": " in l and not l.startswith("\t")):
            start, name = i, l.split(":")[0]
        elif "s_endpgm" in l and start is not None:
            yield name, start, i
            start = None


worst = None
bad = []
checked = 0
for name, a, b in kernels():
    if "mfma" not in name:
        continue
    ins = []          # (line number, text, in_asm)
    in_asm = False
    for i in range(a, b):
        t = lines[i].strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or re.match(r"^\.?L\w+:", t):
            continue
        t = t.split(";")[0].strip()
        if re.match(r"^[a-z_0-9]+(\s|$)", t):
            ins.append((i + 1, t, in_asm))
    for k, (ln, t, asm) in enumerate(ins):
        if not asm or t.startswith("s_nop") or t.startswith("s_waitcnt"):
            continue
        used = regs(t)
        if not used:
            continue
        checked += 1
        states = 0
        j = k - 1
        while j >= 0 and states < max(need, 64):
            pl, pt, _ = ins[j]
            if pt.startswith("v_mfma"):
                dst = regs(pt.split(",")[0])
                if dst & used:
                    if worst is None or states < worst[0]:
                        worst = (states, name, ln, t, pl, pt)
                    if states < need:
                        bad.append((states, name, ln, t, pl, pt))
                    break
            m = re.match(r"^s_nop\s+(\d+)", pt)
            if m:
                states += int(m.group(1)) + 1
            elif pt.startswith("v_mfma"):
                states += PASSES.get(pt.split()[0], 4)
            else:
                states += 1
            j -= 1

print("checked %d inline-asm instructions in the MFMA kernels of %s; required distance %d wait states" % (checked, path, need))
if bad:
    for states, name, ln, t, pl, pt in bad[:10]:
        print("HAZARD: line %d  %s\n   reads a destination of line %d  %s\n   only %d wait states earlier (%s)"
              % (ln, t, pl, pt, states, name[:60]))
    sys.exit(1)
print("no asm instruction reads an MFMA destination within %d wait states of the MFMA" % need)
if worst:
    print("closest pair: %d wait states -- line %d  %s  after line %d  %s" % (worst[0], worst[2], worst[3], worst[4], worst[5]))
