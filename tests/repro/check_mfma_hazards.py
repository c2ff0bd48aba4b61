#!/usr/bin/env python3
"""Static check of gfx950 ISA listings for the MFMA -> VALU read hazard (tests/repro/README.md section 2).

gfx950 does NOT interlock a vector (non-MFMA) instruction that reads or overwrites the LAST registers of the
destination of an MFMA still in flight: it sees the OLD value unless `passes + 2` wait states separate the two
(measured with tests/repro/mfma_wait_states.hip: v_mfma_f32_32x32x2_f32 registers 14 and 15 of 16 until 18 wait
states, v_mfma_f32_16x16x4_f32 registers 2 and 3 of 4 until 10, v_mfma_f32_4x4x1_16b_f32 register 3 of 4 until 4;
the registers before them read fresh at any distance).  hipcc (ROCm 7.2) pads for this along the LAYOUT order of the blocks only: where a
conditional branch skips a block that lies between the MFMA and the reader, the taken path comes out short
(k_linearize_regs<2, 32> with the scheduling hints: 12 wait states where 18 are needed -- a 3.8e-2 error).

This script walks every path of every kernel's control-flow graph in `hipcc -S --cuda-device-only` output and
reports each MFMA whose destination is touched too early.  An MFMA between the two counts as its own number of
passes (the matrix pipe is in-order: it cannot start before its predecessor has gone through), everything else
as one wait state, `s_nop N` as N + 1.

    python tests/repro/check_mfma_hazards.py file.s [file.s ...]        exit code 1 when anything is flagged
"""
import re
import sys
from collections import deque

# passes of the matrix instructions this library issues (wait states needed = passes + 2)
PASSES = {
    "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16,
    "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_16x16x4f32": 8,
    "v_mfma_f32_4x4x1_16b_f32": 2, "v_mfma_f32_4x4x1f32": 2,
    "v_mfma_f32_32x32x1_2b_f32": 16, "v_mfma_f32_16x16x1_4b_f32": 8,
}
TAIL = {16: 2, 8: 2, 2: 1}      # unprotected registers at the end of the destination, by passes
REG = re.compile(r"\b([av])(?:(\d+)\b|\[(\d+):(\d+)\])")
LABEL = re.compile(r"^([.\w$]+):")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+([.\w$]+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Ins:
    __slots__ = ("line", "op", "text", "regs", "dst", "srcc", "ab", "passes", "ws", "tail")

    def __init__(self, line, text):
        self.line, self.text = line, text
        body = text.split(";")[0].strip()
        self.op = body.split()[0] if body else ""
        self.regs = regs_of(body[len(self.op):])
        self.passes = PASSES.get(self.op, 0)
        self.dst = self.srcc = self.tail = self.ab = set()
        if self.op.startswith("v_mfma") or self.op.startswith("v_smfma"):
            ops = [o.strip() for o in body[len(self.op):].split(",")]
            self.dst = regs_of(ops[0])
            self.srcc = regs_of(ops[3]) if len(ops) > 3 else set()
            self.ab = (regs_of(ops[1]) if len(ops) > 1 else set()) | (regs_of(ops[2]) if len(ops) > 2 else set())
            if not self.passes:
                self.passes = 16          # unknown matrix instruction: assume the longest
            hi = max(r for _, r in self.dst)
            self.tail = {(f, r) for f, r in self.dst if r > hi - TAIL.get(self.passes, 2)}
        if self.op == "s_nop":
            self.ws = int(body.split()[1], 0) + 1
        else:
            self.ws = 1


def kernels(path):
    """yield (name, [Ins], {label: index}) per function of the listing"""
    name, ins, labels = None, [], {}
    with open(path) as fp:
        for no, raw in enumerate(fp, 1):
            s = raw.rstrip("\n")
            m = LABEL.match(s)
            if m and not s.startswith("\t"):
                lab = m.group(1)
                if lab.startswith(".L") and name:
                    labels[lab] = len(ins)
                elif not lab.startswith("."):
                    if name and ins:
                        yield name, ins, labels
                    name, ins, labels = lab, [], {}
                continue
            if not s.startswith("\t") or name is None:
                continue
            t = s.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            ins.append(Ins(no, t))
    if name and ins:
        yield name, ins, labels


def successors(ins, labels, i):
    t = ins[i].text
    m = BRANCH.match(t)
    if ins[i].op == "s_endpgm" or ins[i].op.startswith("s_setpc"):
        return []
    if m:
        tgt = labels.get(m.group(2))
        out = [] if tgt is None else [tgt]
        if m.group(1) != "s_branch" and i + 1 < len(ins):
            out.append(i + 1)
        return out
    return [i + 1] if i + 1 < len(ins) else []


def check(path, strict=False):
    bad = []
    for name, ins, labels in kernels(path):
        for i, mf in enumerate(ins):
            if not mf.dst:
                continue
            need = mf.passes + 2
            # shortest-path walk (wait states) from the MFMA to the first toucher of its destination
            best = {}
            dq = deque((j, 0) for j in successors(ins, labels, i))
            while dq:
                j, ws = dq.popleft()
                if ws >= need or best.get(j, 1 << 30) <= ws:
                    continue
                best[j] = ws
                c = ins[j]
                if c.dst and c.dst == mf.dst and c.srcc == mf.dst:
                    continue         # accumulate chain on the same registers: handled by the matrix pipe, and from
                                     # here on this younger instruction is the one in flight
                if c.dst and not (c.regs & mf.tail) & c.ab:
                    pass             # another matrix instruction takes the registers as its addend and / or as its
                                     # own destination only (an accumulation chain hipcc renamed: d1 = mfma(a, b, d0);
                                     # d0 = mfma(a', b', d1)): the matrix pipe is in order, it orders the addend read
                                     # and the later write behind the earlier write (hipcc's own straight-line code
                                     # does this every k-step).  A / B operand reads are NOT exempt.
                elif c.regs & mf.tail:
                    bad.append((path, name, mf.line, mf.op, c.line, c.text.split(";")[0].strip(), ws, need))
                    continue
                step = c.ws if (strict or not c.dst) else max(1, c.passes)
                for k in successors(ins, labels, j):
                    dq.append((k, ws + step))
    return bad


def main(argv):
    strict = "--strict" in argv
    files = [a for a in argv if not a.startswith("--")]
    total = 0
    for f in files:
        seen = set()
        for path, name, l0, op, l1, text, ws, need in check(f, strict):
            key = (name, l0, l1)
            if key in seen:
                continue
            seen.add(key)
            total += 1
            print(f"{path}:{l1}: {name}: `{text}` touches the destination of {op} (line {l0}) after {ws} wait "
                  f"states, {need} needed")
    print(f"{total} hazard(s) in {len(files)} listing(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
