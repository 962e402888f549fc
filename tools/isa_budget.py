#!/usr/bin/env python3
"""Instruction budget of a kernel's main loop from the device assembly (blasr_amd/csrc/gba_gfx950.s):
per basic block of the loop, the instructions by class (VALU, SALU, LDS, VMEM, branch, wait).
usage: isa_budget.py <mangled-name-substring> [assembly file]"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    name = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 else "blasr_amd/csrc/gba_gfx950.s"
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    loop_blocks = set()
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            if "in Loop" in m.group(2) or "Loop Header" in m.group(2):
                loop_blocks.add(cur)
            continue
        m = re.match(r"^; %bb\.(\d+):(.*)", l)
        if m:
            cur = "bb." + m.group(1)
            blocks[cur] = []
            if "in Loop" in m.group(2) or "Loop Header" in m.group(2):
                loop_blocks.add(cur)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s.split()[0])
    tot = Counter()
    print("%-12s %5s %5s %5s %5s %5s %5s %6s" % ("block", "valu", "salu", "lds", "vmem", "br", "wait", "all"))
    for b, ops in blocks.items():
        if b not in loop_blocks:
            continue
        c = Counter(classify(o) for o in ops)
        tot.update(c)
        print("%-12s %5d %5d %5d %5d %5d %5d %6d" % (b, c["valu"], c["salu"], c["lds"], c["vmem"], c["branch"], c["wait"], len(ops)))
    print("%-12s %5d %5d %5d %5d %5d %5d %6d" % ("loop total", tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["branch"], tot["wait"], sum(tot.values())))
    ops = Counter(o for b, v in blocks.items() if b in loop_blocks for o in v)
    print("\nopcodes in the loop:")
    for o, n in ops.most_common(40):
        print("  %-28s %4d" % (o, n))


if __name__ == "__main__":
    main()
