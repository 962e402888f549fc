"""Host mirror of StoreMapQVs (blasr_amd/mapqv.py, gba::StoreMapQVs in gba_refine.hpp)."""
import math
import os
import subprocess

import numpy as np
import pytest

from blasr_amd import mapqv, refine
from blasr_amd.synth import BLOCK_DT


def _cand(q0, q1, qpos=0):
    b = np.zeros(1, dtype=BLOCK_DT)
    b[0] = (0, 0, q1 - q0)
    return refine.AlignmentCandidate(tAlignedSeq=b"", qAlignedSeqPos=q0 - qpos, qPos=qpos, blocks=b)


def test_overlap_and_partitions():
    a, b, c, d = _cand(0, 1000), _cand(100, 900), _cand(950, 2000), _cand(5000, 6000)
    assert mapqv.alignments_overlap(a, b, 0.75)              # one holds the other
    assert not mapqv.alignments_overlap(a, c, 0.75)          # 50 bases of 1000
    assert mapqv.alignments_overlap(_cand(0, 1000), _cand(200, 1100), 0.75)    # 800 of 900
    parts = mapqv.partition_overlapping_alignments([a, b, c, d], 0.75)
    assert parts == [[0, 1], [2], [3]]
    assert mapqv.q_interval(_cand(120, 480, qpos=20)) == (120, 480)


def test_store_map_qvs_values():
    # one partition of three: scores -2000, -1990 (in tenths of log10) and a short one that pays for its missing ends
    a, b, c = _cand(0, 1000), _cand(0, 1000), _cand(0, 900)
    mapqv.store_map_qvs([a, b, c], [-2000, -1990, -1995])
    assert (a.probScore, b.probScore) == (200.0, 199.0)
    assert c.probScore == 199.5 - 0.5 * 15 * 100             # 100 bases short of the widest stretch, 15 a base, halved
    denom = mapqv.log_sum_of_two(mapqv.log_sum_of_two(200.0, 199.0), c.probScore)
    assert a.mapQV == int(-10 * math.log10(1 - 10 ** (200.0 - denom)))
    assert b.mapQV == int(-10 * math.log10(1 - 10 ** (199.0 - denom)))
    assert c.mapQV == 0                                       # more than 20 orders of magnitude below
    lone = _cand(3000, 4000)
    mapqv.store_map_qvs([lone], [-500])
    assert lone.mapQV == mapqv.MAX_PHRED_SCORE
    same = [_cand(0, 1000), _cand(0, 1000)]
    mapqv.store_map_qvs(same, [-100, -100])
    assert same[0].mapQV == same[1].mapQV == int(-10 * math.log10(0.5))


def test_cpp_store_map_qvs_equals_python():
    from blasr_amd import build
    exe = build.build_host_demo()
    rng = np.random.default_rng(11)
    for _ in range(20):
        n = int(rng.integers(1, 9))
        cands, raws, lines = [], [], ["%d" % n]
        for _i in range(n):
            q0 = int(rng.integers(0, 3000)); q1 = q0 + int(rng.integers(50, 4000))
            raw = int(rng.integers(-30000, -100))
            cands.append(_cand(q0, q1)); raws.append(raw)
            lines.append("%d %d %d" % (q0, q1, raw))
        mapqv.store_map_qvs(cands, raws)
        out = subprocess.run([exe, "mapqv"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        got = [int(x) for x in out.stdout.split()]
        assert got == [c.mapQV for c in cands]
