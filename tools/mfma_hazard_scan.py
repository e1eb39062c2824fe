#!/usr/bin/env python3
"""Static scan of a gfx950 .s file (hipcc -save-temps) for the register hazards around MFMAs.

For every v_mfma in a kernel it reports, per operand (A, B, C, D):
  * the nearest earlier instruction that writes one of the operand's registers (RAW side) and what kind it is
    (ds_read return, VALU, another MFMA's D) with the number of instructions / s_nop wait states between, and
  * the nearest later instruction that overwrites one of the A/B/C registers (WAR side), same figures.
Straight-line distance only (the MLP layers are fully unrolled; a label ends a window).  Usage:
    tools/mfma_hazard_scan.py file.s [kernel-name-substring]
Written for VERDICT r01 item 7 (v_mfma_f32_16x16x32_f16 irreproducibility); summary in DESIGN 4.1b.
"""
import collections
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    if m:
        return {int(m.group(1))}
    return set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    args = [a.strip() for a in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    args = [a.split()[0] if a else a for a in args]           # drop modifiers after the register
    return op, args


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def writes(op, args):
    k = kind(op)
    if k in ("mfma", "lds", "vmem", "valu") and args:
        if op.startswith(("v_cmp", "v_cmpx")) or op.startswith("v_nop"):
            return set()
        w = regs(args[0])
        if op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")) and len(args) > 1:
            w |= regs(args[1])
        return w
    return set()


def states(op, args):
    """wait states an instruction contributes when it sits between a producer and a consumer"""
    if op == "s_nop":
        return int(args[0], 0) + 1
    return 1


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):.*\n", text, flags=re.M):
        name = m.group(1)
        if want not in name or not name.startswith("_Z"):
            continue
        end = text.find(".Lfunc_end", m.end())
        body = text[m.end():end].splitlines()
        ins = []
        for ln in body:
            if re.match(r"^\.L\w+:", ln):
                ins.append(("label", []))
                continue
            if not ln.startswith("\t"):
                continue
            p = parse(ln)
            if p:
                ins.append(p)
        n_mfma = sum(1 for op, _ in ins if kind(op) == "mfma")
        if not n_mfma:
            continue
        raw = collections.defaultdict(lambda: (10**9, None))   # (operand, producer kind) -> min states, example
        war = collections.defaultdict(lambda: (10**9, None))
        for i, (op, a) in enumerate(ins):
            if kind(op) != "mfma":
                continue
            operands = {"D": regs(a[0]), "A": regs(a[1]), "B": regs(a[2]), "C": regs(a[3])}
            for nm in ("A", "B", "C"):
                r = operands[nm]
                if not r:
                    continue
                st = 0
                for j in range(i - 1, max(i - 400, -1), -1):
                    o2, a2 = ins[j]
                    if o2 == "label":
                        break
                    if writes(o2, a2) & r:
                        key = (nm, kind(o2))
                        if st < raw[key][0]:
                            raw[key] = (st, f"{o2} {', '.join(a2)}  ->  {op} {', '.join(a)}")
                        break
                    st += states(o2, a2)
                st = 0
                for j in range(i + 1, min(i + 400, len(ins))):
                    o2, a2 = ins[j]
                    if o2 == "label":
                        break
                    if writes(o2, a2) & r:
                        key = (nm, kind(o2))
                        if st < war[key][0]:
                            war[key] = (st, f"{op} {', '.join(a)}  ->  {o2} {', '.join(a2)}")
                        break
                    st += states(o2, a2)
        dst = collections.defaultdict(lambda: (10**9, None))   # D -> first later reader / writer, by kind
        for i, (op, a) in enumerate(ins):
            if kind(op) != "mfma":
                continue
            r = regs(a[0])
            st = 0
            seen_read = seen_write = False
            for j in range(i + 1, min(i + 400, len(ins))):
                o2, a2 = ins[j]
                if o2 == "label":
                    break
                w = writes(o2, a2)
                rd = set()
                for t in (a2[1:] if w else a2):
                    rd |= regs(t)
                if kind(o2) == "mfma" and regs(a2[3]) == r:
                    rd -= r                                   # whole-register accumulate chain: exempt
                if not seen_read and rd & r:
                    key = ("read", kind(o2))
                    if st < dst[key][0]:
                        dst[key] = (st, f"{op} {', '.join(a)}  ->  {o2} {', '.join(a2)}")
                    seen_read = True
                if not seen_write and w & r:
                    key = ("write", kind(o2))
                    if st < dst[key][0]:
                        dst[key] = (st, f"{op} {', '.join(a)}  ->  {o2} {', '.join(a2)}")
                    seen_write = True
                if seen_read and seen_write:
                    break
                st += states(o2, a2)
        print(f"== {name}: {n_mfma} MFMAs")
        print("  MFMA result D -> first later reader / writer other than the accumulate chain (minimum wait states)")
        for key in sorted(dst):
            print(f"    D {key[0]:5s} by {key[1]:5s}: {dst[key][0]:3d}   {dst[key][1]}")
        print("  producer -> MFMA source read (minimum wait states in between)")
        for key in sorted(raw):
            print(f"    {key[0]} <- {key[1]:5s}: {raw[key][0]:3d}   {raw[key][1]}")
        print("  MFMA source read -> next writer of the same registers (minimum wait states in between)")
        for key in sorted(war):
            print(f"    {key[0]} -> {key[1]:5s}: {war[key][0]:3d}   {war[key][1]}")


if __name__ == "__main__":
    main()
