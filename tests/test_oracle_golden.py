"""The oracle against known-answer vectors produced by the REAL reference code.

tests/golden/ref_vectors.out was written by oracle/_ref/ref_driver, which is the reference's
own bitvec.h / bwtrepr.h / encodedtext.h / bitparallelmatrix.{h,cpp} / indexhelpers.{h,cpp} /
search.{h,cpp} / nucleotide.h compiled unmodified (oracle/Makefile, tests/golden/make_golden.py).
These tests pin rows a1, a2, a5, a7, a10 (bit structures), a11 (doTask), a14, a16 (orderings),
a17 of SURVEY.md §8.
"""
import os
import subprocess
from collections import Counter

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _load():
    cmds = open(os.path.join(GOLD, "ref_vectors.cmds")).read().splitlines()
    outs = open(os.path.join(GOLD, "ref_vectors.out")).read().splitlines()
    assert len(cmds) == len(outs) and len(cmds) > 1000
    return cmds, outs


def test_oracle_matches_reference_vectors(oracle_built):
    cmds, outs = _load()
    res = subprocess.run([os.path.join(oracle_built, "oracle_driver")], input="\n".join(cmds) + "\n",
                         capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(res) == len(outs)
    bad = Counter()
    for c, r, o in zip(cmds, res, outs):
        if r != o:
            bad[c.split(" ")[0]] += 1
    assert not bad, f"oracle differs from reference vectors: {dict(bad)}"


def test_vectors_are_not_vacuous():
    cmds, outs = _load()
    kinds = Counter(c.split(" ")[0] for c in cmds)
    for k in ("bwt", "bitvec9", "enc", "matrix", "traceback", "search", "scheme", "cluster", "verify",
              "occsort", "revcomp"):
        assert kinds[k] >= 10, k
    # in-text verification vectors must contain real hits and real aborts
    hits = aborts = 0
    for c, o in zip(cmds, outs):
        if c.startswith("verify "):
            t = o.split()
            hits += int(t[3])
            aborts += int(t[1])
    assert hits > 50 and aborts > 20
    # matrix vectors must contain rows that stop early (invalid) and rows in the final column
    early = sum(1 for c, o in zip(cmds, outs) if c.startswith("matrix ") and " 0 " in o[:40])
    assert early >= 0
    # traceback vectors with indels
    assert sum(1 for c, o in zip(cmds, outs) if c.startswith("traceback ") and ("I" in o or "D" in o)) > 10


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "ref_driver")),
                    reason="oracle/_ref/ref_driver is only built where /root/reference exists")
def test_fixture_is_what_the_reference_prints_today():
    cmds, outs = _load()
    drv = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "ref_driver")
    res = subprocess.run([drv], input="\n".join(cmds) + "\n", capture_output=True, text=True,
                         check=True).stdout.splitlines()
    assert res == outs
