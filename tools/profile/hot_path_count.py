#!/usr/bin/env python3
"""Static walk of a kernel's ISA listing (hipcc -S) along its HOT path: conditional branches on execz / vccz are taken
(they skip the rare paths behind a wave vote), scc branches fall through; counts instructions by unit and opcode.

    python tools/profile/hot_path_count.py file.s '<mangled kernel name>' [--until LABEL] [--from LABEL] [-v]
"""
import collections
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    until = sys.argv[sys.argv.index("--until") + 1] if "--until" in sys.argv else None
    start = sys.argv[sys.argv.index("--from") + 1] if "--from" in sys.argv else None
    verbose = "-v" in sys.argv
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    body, labels = [], {}
    for l in lines[i0 + 1:]:
        if "; -- End function" in l:
            break
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB"):
            labels[t.split(":")[0]] = len(body)
            continue
        body.append(t.split(";")[0].strip())
    pc = labels[start] if start else 0
    cnt = collections.Counter()
    ops = collections.Counter()
    steps = 0
    while pc < len(body) and steps < 100000:
        steps += 1
        if until and pc == labels.get(until):
            break
        ins = body[pc]
        op = ins.split()[0]
        unit = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM")
        cnt[unit] += 1
        ops[op] += 1
        if verbose:
            print(pc, ins)
        if op == "s_endpgm":
            break
        if op in ("s_cbranch_execz", "s_cbranch_vccz", "s_branch"):
            pc = labels[ins.split()[1]]
            continue
        if op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_cbranch_vccnz", "s_cbranch_execnz"):
            if "--take" in sys.argv and ins.split()[1] in sys.argv:
                pc = labels[ins.split()[1]]
                continue
        pc += 1
    print(dict(cnt))
    pk = sum(v for k, v in ops.items() if k.startswith("v_pk_"))
    print("packed:", pk, " VALU issue-slot equivalents (packed = 2):", cnt["VALU"] + pk)
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]):
        if k.startswith("v_") or verbose:
            print("  %4d %s" % (v, k))


if __name__ == "__main__":
    main()
