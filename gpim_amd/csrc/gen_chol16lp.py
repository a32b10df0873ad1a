#!/usr/bin/env python3
"""gen_chol16lp.py -- writes chol16lp_steps.inc: the 16 elimination steps of chol16_lp (chol16lp.hpp) as one
hand-ordered stream of statements.

Why a generator: a lone wave issues one instruction every ~5 cycles whatever it is, and a dependent fp64 operation can
issue 10 cycles (v_rcp_f64: 20) after its producer (tools/r5_lat_probe.hip).  The 16x16 factorisation is ~290
instructions against a dependent chain of 7 per pivot, so its duration is decided by the ORDER of the stream: the chain
of step j is interleaved with the independent row updates of the steps before it ("fillers").  Every statement is
followed by __builtin_amdgcn_sched_barrier(0), so the compiler keeps program order and only allocates registers.

The stream per pivot j (O = ordinary instruction, A = inline assembly; F = the oldest pending update):

    d_j   A  v_mov_b64_dpp      pivot = DG of lane j            (two instructions after dg_{j-1}: R1)
    mk_j  O  v_mov_b32_dpp      lane mask of the rows below the pivot (row_shr:1 of the previous one)
    y_j   O  v_rcp_f64
    F
    amlo_j, amhi_j  O           a_ij on lanes i > j, 0 elsewhere (bit mask: the upper triangle may hold anything, also NaN)
    e_j   O  v_fma_f64          (four slots after y_j: the reciprocal's latency; an ordinary instruction between: R3)
    F, p_j, F, ninv_j, F, negc_j, F
    dg_j  O  v_fma_f64          own diagonal entry
    a_j_{j+1}  A                column j+1, the next pivot's column, first
    F

Only the DPP instructions that the compiler cannot produce are inline assembly: the fused v_fmac_f64_dpp updates and the
64-bit pivot broadcast (as a builtin the compiler would want two ORDINARY instructions between dg and d: it does not
count inline assembly as wait states).  Instruction distances the order respects (check() below; the compiled code
object is re-checked by tests/test_dpp_hazards.py):
  R1 (hardware)  a DPP read of a VGPR needs two other instructions between it and the VALU write of that VGPR; the
                 hazard recogniser cannot see this inside inline assembly (an `s_nop 1` is emitted where no filler is left).
  R2 (compiler)  DPP builtin: two ORDINARY instructions between the write and the read, or the compiler adds an s_nop.
  R3 (compiler)  one ordinary instruction between v_rcp_f64 and the first reader of its result (trans forwarding).
  R4 (compiler)  a reader of a register written by inline assembly costs an `s_nop 0` unless an ordinary instruction
                 sits between them (the recogniser assumes the assembly may carry a dst_sel forwarding hazard).
R2-R4 only cost issue slots.

Run:  python3 gpim_amd/csrc/gen_chol16lp.py   (rewrites gpim_amd/csrc/chol16lp_steps.inc; tests compare the two)
"""
import os
import sys
from collections import deque

SB = " __builtin_amdgcn_sched_barrier(0);"


class Op:
    def __init__(self, name, kind, text, outs, ins, dpp_src=None):
        self.name, self.kind, self.text = name, kind, text + SB
        self.outs, self.ins, self.dpp_src = outs, ins, dpp_src
        self.is_asm = kind in ("fmacdpp", "dppmov64")


def upd_a(j, k):
    op = Op(f"a{j}_{k}", "fmacdpp",
              f'asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:{k} row_mask:0xf bank_mask:0xf" : "+v"(A{k}) : "v"(A{j}), "v"(NC_{j}));',
              [f"A{k}"], [f"A{k}", f"A{j}", f"NC_{j}"], dpp_src=f"A{j}")
    op.urgency = k
    return op


def upd_m(j, r):
    op = Op(f"m{j}_{r}", "fmacdpp",
              f'asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:{j} row_mask:0xf bank_mask:0xf" : "+v"(M{r}) : "v"(NC_{j}));',
              [f"M{r}"], [f"M{r}", f"NC_{j}"], dpp_src=f"M{r}")
    op.urgency = j + 6
    return op


def build_stream():
    out = []
    pending = deque()

    def filler(must=False):
        """The most urgent pending update (the column that becomes the pivot column first; the identity part of step j
        ranks with column j + 6 so that its four accumulation chains keep moving) whose DPP source was not written by
        the last two instructions and whose accumulator was not written by inline assembly since the last ordinary
        instruction (R4)."""
        recent = [v for op in out[-2:] for v in op.outs]
        asm_tail = []
        for op in reversed(out):
            if not op.is_asm:
                break
            asm_tail += op.outs
        best = None
        for avoid_r4 in (True, False):
            for idx, op in enumerate(pending):
                if op.dpp_src in recent or (avoid_r4 and op.outs[0] in asm_tail):
                    continue
                if any(q.outs[0] == op.outs[0] for q in list(pending)[:idx]):
                    continue                   # keep the order of the updates of one register
                if best is None or op.urgency < pending[best].urgency:
                    best = idx
            if best is not None or not must:
                break
        if best is not None:
            out.append(pending[best])
            del pending[best]
        elif must:
            out.append(Op("nop", "snop", "__builtin_amdgcn_s_nop(1);", [], []))

    for j in range(15):
        # two instructions between dg_{j-1} and the DPP read of DG (R1)
        while j > 0 and sum(1 for op in out[out.index(last_dg) + 1:]) < 2:
            filler(must=True)
        out.append(Op(f"d{j}", "dppmov64",
                      f'double D_{j}; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:{j} row_mask:0xf bank_mask:0xf" : "=v"(D_{j}) : "v"(DG));',
                      [f"D_{j}"], ["DG"], dpp_src="DG"))
        out.append(Op(f"mk{j}", "mask", "MSK = __builtin_amdgcn_update_dpp(0, MSK, 0x111, 0xF, 0xF, true);", ["MSK"], ["MSK"], dpp_src="MSK"))
        out.append(Op(f"y{j}", "rcp", f"const double Y_{j} = __builtin_amdgcn_rcp(D_{j});", [f"Y_{j}"], [f"D_{j}"]))
        filler()
        out.append(Op(f"amlo{j}", "and", f"const int AL_{j} = __double2loint(A{j}) & MSK;", [f"AL_{j}"], [f"A{j}", "MSK"]))
        out.append(Op(f"amhi{j}", "and", f"const int AH_{j} = __double2hiint(A{j}) & MSK;", [f"AH_{j}"], [f"A{j}", "MSK"]))
        out.append(Op(f"e{j}", "fma", f"const double E_{j} = fma(-D_{j}, Y_{j}, 1.0);", [f"E_{j}"], [f"D_{j}", f"Y_{j}"]))
        filler()
        out.append(Op(f"p{j}", "fma", f"const double P_{j} = fma(E_{j}, E_{j}, E_{j});", [f"P_{j}"], [f"E_{j}"]))
        filler()
        out.append(Op(f"ninv{j}", "fma", f"const double NI_{j} = fma(-Y_{j}, P_{j}, -Y_{j});", [f"NI_{j}"], [f"Y_{j}", f"P_{j}"]))
        filler()
        out.append(Op(f"negc{j}", "mul", f"const double NC_{j} = __hiloint2double(AH_{j}, AL_{j}) * NI_{j};", [f"NC_{j}"], [f"AH_{j}", f"AL_{j}", f"NI_{j}"]))
        filler()
        # (the masked copy of a_ij: on the lanes above the pivot A_j is not this matrix' data, and 0 * NaN would reach DG)
        last_dg = Op(f"dg{j}", "fmac", f"DG = fma(NC_{j}, __hiloint2double(AH_{j}, AL_{j}), DG);", ["DG"], ["DG", f"NC_{j}", f"AH_{j}", f"AL_{j}"])
        out.append(last_dg)
        # updates with row j: the next pivot's column at once, the others join the queue (columns in the order they
        # are needed, then the identity part)
        if j + 1 <= 14:
            out.append(upd_a(j, j + 1))
        for k in range(j + 2, 15):
            pending.append(upd_a(j, k))
        for r in range(j // 4 + 1):
            pending.append(upd_m(j, r))
        # in the issue-bound first steps the queue is longer than the six filler slots of a step: drain the excess here
        # so that a column is up to date when it becomes the pivot column
        filler()
        while len(pending) > 14 - j + 6:
            filler()
    while pending:
        filler(must=True)
    return out


def check(stream):
    """R1 must hold (returns the violations); R2-R4 are counted as the nops the compiler is expected to add."""
    hard, soft = [], 0
    for pos, op in enumerate(stream):          # a column is complete when it becomes the pivot column
        if op.name.startswith("amlo"):
            j = int(op.name[4:])
            late = [q.name for q in stream[pos:] if f"A{j}" in q.outs]
            if late:
                hard.append((op.name, late))

    def last_write(pos, var):
        for q in range(pos - 1, -1, -1):
            if var in stream[q].outs:
                return q
        return None

    for pos, op in enumerate(stream):
        if op.dpp_src is not None:
            q = last_write(pos, op.dpp_src)
            if q is not None:
                between = stream[q + 1:pos]
                if op.is_asm and len(between) < 2:
                    hard.append((op.name, stream[q].name))                                   # R1
                if not op.is_asm and sum(1 for b in between if not b.is_asm) < 2:
                    soft += 1                                                                # R2
        for v in op.ins:
            q = last_write(pos, v)
            if q is None:
                continue
            w = stream[q]
            if (w.is_asm or w.kind == "rcp") and not any(not b.is_asm for b in stream[q + 1:pos]):
                soft += 1                                                                    # R3 / R4
                break
    return hard, soft


def emit(path):
    stream = build_stream()
    hard, soft = check(stream)
    if hard:
        raise SystemExit(f"DPP distance violated: {hard}")
    n = len(stream)
    lines = [
        "// chol16lp_steps.inc -- GENERATED by gen_chol16lp.py, do not edit: the elimination steps of chol16_lp in issue order.",
        f"// {n} instructions ({sum(1 for s in stream if s.kind == 'snop')} nops of its own, {soft} expected from the compiler).",
        "// Variables (chol16lp.hpp): A0..A14 columns of this lane's row, DG its diagonal entry, M0..M3 its share of the identity part,",
        "// MSK the lane mask of the rows below the pivot; temporaries D_j Y_j E_j P_j NI_j AL_j AH_j NC_j of step j.",
    ]
    for s in stream:
        lines.append(s.text + f"    // {s.name}")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return n, soft


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "chol16lp_steps.inc")
    n, soft = emit(out)
    print(f"{out}: {n} instructions, {soft} compiler nops expected")
