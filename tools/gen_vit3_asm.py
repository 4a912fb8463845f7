#!/usr/bin/env python
"""Generates nrsc5_amd/csrc/viterbi_v3_asm.h: the gfx950 instruction streams of 8 consecutive trellis steps of the
third-generation K=7 Viterbi forward pass (viterbi_v3.h), one inline-asm string per (start phase, opens-a-history-word).

Why generated: LLVM's hazard recogniser does not look into inline asm, so the wait states gfx950 wants between
dependent instructions are the author's job.  This script schedules the steps (software-pipelined branch metrics, the
history push as filler) and then CHECKS every hazard rule on the emitted stream, inserting s_nop only where the
schedule leaves a gap:

  R1  VALU writes a VGPR            -> DPP read of it                     2 wait states
  R2  VALU writes a VGPR            -> v_permlane16/32_swap reads it      2 wait states
  R3  DOT (v_dot4) writes a VGPR    -> other VALU reads it                3 wait states
  R4  DOT writes a VGPR             -> other VALU overwrites it           2 wait states

(an instruction in between counts as one wait state, s_nop N as N + 1).

One step, phase r (see viterbi_v3.h for the algebra):
  DPP phases (r < 4)   x = u + D ; D = dpp(u) - D ; ns = max(x, D) ; u = (ns & ~1) | z[r+1]
  swap phases (r >= 4) x = u + Dp ; Dq = u + Dq ; swap(x, Dq) ; ns = max(x, Dq) ; u = (ns & ~1) | z[r+1]
with D / Dp / Dq = v_dot4 of the step's soft word with the lane's weights, issued during the PREVIOUS step, and
h = alignbit(ns_prev, h, 1) (decision history) placed wherever a filler is needed.
"""
import os

DPP_CTRL = {0: "quad_perm:[1,0,3,2]", 1: "quad_perm:[2,3,0,1]", 2: "row_half_mirror", 3: "row_ror:8"}
SWAP_INSN = {4: "v_permlane16_swap_b32", 5: "v_permlane32_swap_b32"}


class Stream:
    def __init__(self):
        self.ins = []           # (text, kind, reads, dpp_reads, swap_reads, writes)
        self.last_write = {}    # reg -> (index, kind)

    def _need(self, reg, reader_kind, how):
        """wait states required before an instruction that reads/writes `reg` can issue"""
        if reg not in self.last_write:
            # written by the previous asm statement: assume it was its very last instruction (u) -- distance 0
            idx, kind = -1, "valu" if reg in ("u", "ns", "h") else None
            if kind is None:
                return 0
        else:
            idx, kind = self.last_write[reg]
        dist = len(self.ins) - 1 - idx      # instructions in between (s_nop N was expanded to N + 1 entries)
        need = 0
        if how == "dpp" or how == "swap":
            need = 2 if kind == "valu" else 3           # R1 / R2 (DOT results additionally fall under R3)
        elif how == "read" and kind == "dot" and reader_kind != "dot_acc":
            need = 3                                     # R3
        elif how == "write" and kind == "dot":
            need = 2                                     # R4
        return max(0, need - dist)

    def emit(self, text, kind, reads=(), dpp_reads=(), swap_reads=(), writes=()):
        need = 0
        for r in reads:
            need = max(need, self._need(r, kind, "read"))
        for r in dpp_reads:
            need = max(need, self._need(r, kind, "dpp"))
        for r in swap_reads:
            need = max(need, self._need(r, kind, "swap"))
        for r in writes:
            need = max(need, self._need(r, kind, "write"))
        if need:
            self.ins.append((f"s_nop {need - 1}", "nop", (), (), (), ()))
            for _ in range(need - 1):
                self.ins.append((None, "nop", (), (), (), ()))       # bookkeeping only: s_nop N = N + 1 wait states
        self.ins.append((text, kind, reads, dpp_reads, swap_reads, writes))
        for r in writes:
            self.last_write[r] = (len(self.ins) - 1, "dot" if kind == "dot" else "valu")

    def text(self):
        return [t for t, *_ in self.ins if t is not None]


def dots_for(step, phase, regs, S):
    """emit-closures for the branch metrics of `step` (soft word a<step>) into `regs`"""
    a = f"a{step}"
    if phase < 4:
        return [lambda: S.emit(f"v_dot4_i32_i8 %[{regs[0]}], %[{a}], %[w{phase}], 0", "dot", writes=(regs[0],))]
    return [lambda: S.emit(f"v_dot4_i32_i8 %[{regs[0]}], %[{a}], %[p{phase}], 0", "dot", writes=(regs[0],)),
            lambda: S.emit(f"v_dot4_i32_i8 %[{regs[1]}], %[{a}], %[q{phase}], 0", "dot", writes=(regs[1],))]


def gen_block(ph0, opens_word):
    S = Stream()
    phases = [(ph0 + i) % 6 for i in range(8)]
    # dot registers: consecutive steps use disjoint registers out of d0..d3
    regs, nxt = [], 0
    for r in phases:
        n = 1 if r < 4 else 2
        regs.append([f"d{(nxt + k) % 4}" for k in range(n)])
        nxt = (nxt + n) % 4
    push_pending = not opens_word      # the previous block's last maximum (ns) still has to enter the history word

    def push():
        S.emit("v_alignbit_b32 %[h], %[ns], %[h], 1", "valu", reads=("ns", "h"), writes=("h",))

    # prologue: this block's first branch metrics (every later step's are issued one step ahead)
    for f in dots_for(0, phases[0], regs[0], S):
        f()
    if push_pending:
        push()
        push_pending = False
    for i, r in enumerate(phases):
        nd = dots_for(i + 1, phases[i + 1], regs[i + 1], S) if i < 7 else []
        z = f"z{(r + 1) % 6}"
        # the first of the next step's branch metrics goes FIRST: it separates the and_or that produced u from the add that
        # reads it (a dependent instruction issued back to back costs an extra cycle on this in-order pipe)
        if nd and r < 4:               # (swap steps need their fillers between the adds and the swap instead)
            nd.pop(0)()
        if r < 4:
            d = regs[i][0]
            S.emit(f"v_add_u32 %[x], %[u], %[{d}]", "valu", reads=("u", d), writes=("x",))
            if nd:
                nd.pop(0)()
            S.emit(f"v_sub_u32_dpp %[{d}], %[u], %[{d}] {DPP_CTRL[r]} row_mask:0xf bank_mask:0xf", "valu",
                   reads=(d,), dpp_reads=("u",), writes=(d,))
            if push_pending:
                push(); push_pending = False
            S.emit(f"v_max_i32 %[ns], %[x], %[{d}]", "valu", reads=("x", d), writes=("ns",))
        else:
            dp, dq = regs[i]
            S.emit(f"v_add_u32 %[x], %[u], %[{dp}]", "valu", reads=("u", dp), writes=("x",))
            S.emit(f"v_add_u32 %[{dq}], %[u], %[{dq}]", "valu", reads=("u", dq), writes=(dq,))
            while nd:
                nd.pop(0)()
            if push_pending:
                push(); push_pending = False
            S.emit(f"{SWAP_INSN[r]} %[x], %[{dq}]", "valu", swap_reads=("x", dq), writes=("x", dq))
            S.emit(f"v_max_i32 %[ns], %[x], %[{dq}]", "valu", reads=("x", dq), writes=("ns",))
        S.emit(f"v_and_or_b32 %[u], %[ns], -2, %[{z}]", "valu", reads=("ns", z), writes=("u",))
        push_pending = True            # this step's maximum is pushed during the next step (or by the next block / the flush)
    return S.text()


def main():
    out = ["// GENERATED by tools/gen_vit3_asm.py -- do not edit.  8 trellis steps per asm statement; see the generator for the",
           "// hazard rules its scheduler enforces and viterbi_v3.h for the operands.", "#pragma once", ""]
    stats = []
    for ph0 in (0, 2, 4):
        for opens in (True, False):
            lines = gen_block(ph0, opens)
            name = f"VIT3_ASM_PH{ph0}_{'OPEN' if opens else 'CONT'}"
            out.append(f"#define {name} \\")
            for k, ln in enumerate(lines):
                out.append(f'    "{ln}' + ('\\n\\t" \\' if k < len(lines) - 1 else '"'))
            out.append("")
            nops = sum(int(l.split()[1]) + 1 for l in lines if l.startswith("s_nop"))
            stats.append((name, len([l for l in lines if not l.startswith("s_nop")]), nops))
    out.append("// instructions (without s_nop) / s_nop wait states per 8-step block:")
    for name, n, nops in stats:
        out.append(f"//   {name}: {n} / {nops}")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nrsc5_amd", "csrc", "viterbi_v3_asm.h")
    open(path, "w").write("\n".join(out) + "\n")
    for s in stats:
        print(s)


if __name__ == "__main__":
    main()
