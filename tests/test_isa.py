"""Checks on the generated gfx950 code of the lane-per-pair fill kernels that the source cannot
express: no register spills (a spilled DP word would pass through memory on every cell), the guide
stream requested a sector at a time and only at the one place in the loop where the whole wave asks
together, and the arrow pack of the linear kernel at two instructions per cell (left to itself the
compiler shifts first and masks with a 32-bit literal, three instructions for every second cell).
(The assembly is written by blasr_amd.build next to the library, from the same source with the same
flags.)"""
import os
import re
import subprocess

import pytest

from blasr_amd import build

KERNELS = {"k_fill<false>": "_ZN12_GLOBAL__N_16k_fillILb0ELb0ELb0EEEvNS_3DevENS_9FillConstEjj",
           "k_fill<true>": "_ZN12_GLOBAL__N_16k_fillILb1ELb0ELb0EEEvNS_3DevENS_9FillConstEjj"}


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


def _loop(lines):
    """The instructions between the first loop header label and the last branch back to it."""
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
    head = min(i for i, ln in enumerate(lines) if "Loop Header" in ln)
    back = [i for i, ln in enumerate(lines) for m in [re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)]
            if m and labels.get(m.group(1), 1 << 30) <= head and i > head]
    assert back
    out = []
    for ln in lines[head:max(back) + 1]:
        ln = ln.split(";")[0].strip()
        if ln and not ln.endswith(":") and not ln.startswith("."):
            out.append(ln.split()[0])
    return out


@pytest.mark.parametrize("name", sorted(KERNELS))
def test_guide_stream_is_requested_a_sector_at_a_time_in_one_place(isa, name):
    ops = _loop(_function(isa, KERNELS[name]))
    loads = [o for o in ops if o.startswith(("global_load", "flat_load", "buffer_load"))]
    # four steps per turn of the loop; the sector behind the one in use is asked for in one place: four 16-byte loads
    assert loads == ["global_load_dwordx4"] * 4, loads


def test_arrow_pack_of_the_linear_kernel_is_two_instructions_per_cell(isa):
    ops = _loop(_function(isa, KERNELS["k_fill<false>"]))
    cells = ops.count("v_min3_i32")
    assert cells == 4 * 32                                  # four steps of 32 cell slots
    assert ops.count("v_bfe_u32") >= cells                  # the arrow of every cell
    assert ops.count("v_lshl_or_b32") >= cells - 4 * 2      # shifted into its record dword (the first cell of a dword is a move)
    assert ops.count("v_or3_b32") == 0


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
