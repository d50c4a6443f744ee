#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop in a device assembly listing (hipcc -S --cuda-device-only): for every kernel whose
mangled name contains <pattern>, the loop (a backward branch to a label) that holds the most matrix instructions, and the counts
of matrix / other vector / scalar / LDS / global-memory / wait instructions inside it.
usage: isa_loop_count.py <file.s> <pattern> [<pattern> ...]"""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip("\n"))
    if name:
        yield name, body


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "matrix"
    if op.startswith("v_"):
        return "vector"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "memory"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait/barrier"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\w+)|^\s+s_branch\s+(\.LBB\w+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                yield labels[tgt], i


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if not any(p in name for p in pats):
            continue
        best = None
        for a, b in loops(body):
            ins = [l.strip() for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
            mix = {}
            for x in ins:
                mix[classify(x)] = mix.get(classify(x), 0) + 1
            if best is None or mix.get("matrix", 0) > best[1].get("matrix", 0) or (mix.get("matrix", 0) == best[1].get("matrix", 0) and len(ins) < best[2]):
                best = ((a, b), mix, len(ins))
        print(name)
        if best:
            print("   hottest loop: %d instructions: %s" % (best[2], ", ".join("%s %d" % kv for kv in sorted(best[1].items()))))


if __name__ == "__main__":
    main()
