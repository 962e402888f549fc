"""Checks on the generated gfx950 code of the lane-per-pair fill kernels that the source cannot
express.  k_fill requests the next sector of its guide stream with four global_load_dwordx4 from an
inline-asm block and waits for them only at the top of the block's NEXT execution; until then the
sixteen destination registers are in flight and the hardware has no interlock for them.  The
compiler sees them as ordinary values, so a copy, spill or re-use it inserts between the request
and the wait would silently carry stale data.  This test disassembles what the build produced and
fails if it did.  (The assembly is written by blasr_amd.build next to the library, from the same
source with the same flags.)"""
import os
import re
import subprocess

import pytest

from blasr_amd import build

KERNELS = {"k_fill<false>": "_ZN12_GLOBAL__N_16k_fillILb0EEEvNS_3DevENS_9FillConstEjj",
           "k_fill<true>": "_ZN12_GLOBAL__N_16k_fillILb1EEEvNS_3DevENS_9FillConstEjj"}


def _regs(tok):
    """v5 -> {5}; v[2:5] -> {2,3,4,5}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _function(text, symbol):
    start = text.index("\n" + symbol + ":")
    end = text.index(".Lfunc_end", start)
    return text[start:end].split("\n")


@pytest.fixture(scope="module")
def isa():
    build.build_libgba()
    return open(build.ISA).read()


@pytest.mark.parametrize("name", sorted(KERNELS))
def test_guide_sector_registers_are_left_alone_while_in_flight(isa, name):
    lines = _function(isa, KERNELS[name])
    # the request block: the inline-asm region that holds the four loads
    blocks, cur = [], None
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            cur = i
        elif "#ASMEND" in ln and cur is not None:
            blocks.append((cur, i)); cur = None
    req = [(a, b) for a, b in blocks if sum("global_load_dwordx4" in x for x in lines[a:b]) == 4]
    assert len(req) == 1, "expected exactly one request block in the loop of %s" % name
    a, b = req[0]
    body = [x.strip() for x in lines[a:b]]
    assert body[1].startswith("s_waitcnt vmcnt(0)"), "the block must begin by waiting for the previous request"
    flight = set()
    for x in body:
        if x.startswith("global_load_dwordx4"):
            flight |= _regs(x.split()[1].rstrip(","))
    assert len(flight) == 16
    # the loop: from its header label (the last label before the block that a later branch targets) to that branch
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
    back = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)]
            if m and labels.get(m.group(1), 1 << 30) < a and i > b]
    assert back, "no back edge around the request block"
    head = min(labels[t] for _, t in back)
    tail = max(i for i, _ in back)
    # Inside the loop, outside the block:
    #  - after the request until the end of the loop body nothing may touch the sixteen registers at all;
    #  - from the loop head to the block (the word pop, which by construction only takes a group whose load was
    #    waited for three pops earlier) they may only be READ by moves / selects into ONE group of four registers
    #    (the kernel's group of unread words), never written, stored or copied anywhere else.
    def insns(rng):
        for i in rng:
            ln = lines[i].split(";")[0].strip()
            if not ln or ln.endswith(":") or ln.startswith("."):
                continue
            op, _, rest = ln.partition(" ")
            toks = [t.strip().split()[0] if t.strip() else "" for t in rest.split(",")]
            yield i, ln, op, [_regs(t) for t in toks]
    bad, group = [], set()
    for i, ln, op, regs in insns(range(b + 1, tail + 1)):
        if any(r & flight for r in regs):
            bad.append((i, ln))
    for i, ln, op, regs in insns(range(head, a)):
        if not any(r & flight for r in regs):
            continue
        if regs[0] & flight or not op.startswith(("v_mov_b32", "v_mov_b64", "v_cndmask_b32")):
            bad.append((i, ln))
        else:
            group |= regs[0]
    assert not bad, "in-flight guide registers are touched outside the request block:\n" + "\n".join("%d: %s" % x for x in bad)
    assert len(group) == 4, "the guide registers are copied to more than the one group of four: %s" % sorted(group)


@pytest.mark.parametrize("name", sorted(KERNELS))
def test_fill_kernels_use_no_scratch(isa, name):
    sym = KERNELS[name]
    m = re.search(r"\.amdhsa_kernel " + re.escape(sym) + r"\n(.*?)\.end_amdhsa_kernel", isa, re.S)
    assert m
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(1)), "spills: registers would pass through memory"


def test_compiler_version_is_recorded():
    """The check above holds for the code one compiler produced; the version is part of the record."""
    v = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    assert "HIP version" in v
    print(v.splitlines()[0])
