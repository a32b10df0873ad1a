#!/usr/bin/env python3
"""dpp_hazard_check.py -- checks gfx950 assembly (hipcc -S output) for the DPP read-after-VALU-write hazard.

A DPP instruction reads its first source through the cross-lane network; gfx9 needs two wait states between a VALU
write of that VGPR and the DPP read.  The compiler inserts them for its own DPP instructions, but not around inline
assembly -- and gpim_amd/csrc/chol16lp.hpp places its v_fmac_f64_dpp / v_mov_b64_dpp by hand (gen_chol16lp.py).  This
script re-derives the distance from the compiled code: for every `*_dpp` instruction, no VALU instruction among the
two preceding instruction slots (an `s_nop N` counts N + 1) may write a register of the DPP source operand.  Straight-line
check per basic block, conservative at block entries (labels reset the window only for the number of slots seen).

usage: dpp_hazard_check.py file.s [...]     exit status 1 if a violation is found
"""
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip().lstrip("-|").rstrip("|"))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def operands(line):
    parts = line.split(None, 1)
    if len(parts) < 2:
        return []
    body = parts[1]
    # drop trailing modifiers (row_newbcast:.., quad_perm:[..], ...)
    body = re.split(r"\s+(?:row_|quad_perm|wave_|bank_mask|bound_ctrl|clamp|op_sel|neg_|dst_sel|src0_sel|src1_sel|mul:|div:|offset|glc|slc|sc0|sc1|nt)", body)[0]
    return [t for t in (x.strip() for x in body.split(",")) if t]


def check_text(text, name="<asm>"):
    bad = []
    window = []          # (written registers, text) of the last instruction slots, newest last
    kernel = None
    for ln, raw in enumerate(text.split("\n"), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        if line.endswith(":"):
            if not line.startswith(".L"):
                kernel = line[:-1]
                window = []
            continue
        mn = line.split()[0]
        ops = operands(line)
        if "_dpp" in mn or " row_" in raw or "quad_perm" in raw:
            # VOP1 dpp: dst, src0;  VOP2 dpp: dst, src0, src1 -- src0 is the DPP operand
            if len(ops) >= 2:
                src = regs(ops[1])
                for w, wtext in window[-2:]:
                    if w & src:
                        bad.append(f"{name}:{ln} [{kernel}] `{line}` reads {sorted(w & src)} written by `{wtext}` less than 2 slots earlier")
        if mn == "s_nop":
            n = int(ops[0], 0) + 1 if ops else 1
            window += [(set(), "s_nop")] * n
        else:
            written = set()
            if mn.startswith("v_") and not mn.startswith("v_cmp") and not mn.startswith("v_readlane") and not mn.startswith("v_readfirstlane") and ops:
                written = regs(ops[0])
            window.append((written, line))
        window = window[-4:]
    return bad


def main(argv):
    bad = []
    n_dpp = 0
    for path in argv:
        text = open(path).read()
        n_dpp += len(re.findall(r"^\s+v_\w+_dpp\s", text, re.M))
        bad += check_text(text, path)
    for b in bad:
        print(b)
    print(f"{n_dpp} DPP instructions checked, {len(bad)} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
