#!/usr/bin/env python3
"""Static instruction breakdown of one kernel in a hipcc -S / -save-temps assembly file.

  tools/isa_breakdown.py <file.s> <kernel-name-substring> [--blocks]

Splits the kernel into basic blocks (labels / branches), classifies every instruction and prices
it in VALU issue cycles per wave on gfx950 (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost':
plain VALU 4 cycles per wave-instruction on one wave's stream = 2 on the SIMD with other waves
filling in; packed f32 ops take two passes; transcendentals four).  The loop structure is not
interpreted: --blocks prints the per-block table so that a block (one Jacobi sweep, the epilogue
loop ...) can be weighted by how often it runs.
"""
import collections
import re
import sys

SIMD_CYCLES = {"valu": 2, "valu_pk": 4, "valu_trans": 8, "salu": 0, "smem": 0, "vmem": 0, "branch": 0, "wait": 0,
               "other": 0, "valu_f64": 4}


def classify(op: str) -> str:
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "valu_pk"
    if re.match(r"v_(rsq|rcp|sqrt|exp|log|sin|cos)_", op):
        return "valu_trans"
    if op.endswith("_f64") and op.startswith("v_"):
        return "valu_f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    want_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []          # (label, Counter of class, Counter of opcode)
    cur = ("entry", collections.Counter(), collections.Counter())
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\S*:", s):
            if not re.match(r"^\.LBB\S*:", s):
                continue
        m = re.match(r"^(\.LBB\S*):", s)
        if m:
            blocks.append(cur)
            cur = (m.group(1), collections.Counter(), collections.Counter())
            continue
        op = s.split()[0]
        if op.startswith(";"):
            continue
        c = classify(op)
        cur[1][c] += 1
        cur[2][op] += 1
    blocks.append(cur)
    tot_c, tot_o = collections.Counter(), collections.Counter()
    for _, c, o in blocks:
        tot_c.update(c); tot_o.update(o)
    print(f"kernel matching '{pat}': lines {start + 1}-{end + 1}, {sum(tot_c.values())} instructions (static)")
    print("class            count   SIMD-cycles")
    cyc = 0
    for k, v in sorted(tot_c.items(), key=lambda kv: -kv[1]):
        cc = v * SIMD_CYCLES.get(k, 0)
        cyc += cc
        print(f"  {k:14s} {v:6d}   {cc:8d}")
    print(f"  total VALU issue cycles (static, each block once): {cyc}")
    print("top opcodes:")
    for k, v in tot_o.most_common(28):
        print(f"  {k:28s} {v:6d}")
    if want_blocks:
        print("blocks with >= 20 instructions:")
        for name, c, o in blocks:
            n = sum(c.values())
            if n < 20:
                continue
            bc = sum(v * SIMD_CYCLES.get(k, 0) for k, v in c.items())
            print(f"  {name:12s} n={n:5d} cycles={bc:6d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
            print("      " + " ".join(f"{k}={v}" for k, v in o.most_common(12)))


if __name__ == "__main__":
    main()
