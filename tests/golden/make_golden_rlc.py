#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors_rlc.{cmds,out}: known-answer vectors of the move table of the run-length compressed
flavour, produced by the REAL reference code (oracle/_ref/ref_driver_rlc64 and _rlc32 = the reference's
bmove/moverepr.cpp, indexhelpers.cpp, logger.cpp compiled unmodified with -DRUN_LENGTH_COMPRESSION; `make -C oracle ref`,
build container only).  Inputs are seeded and synthetic; both files are data.
Run:  python tests/golden/make_golden_rlc.py
"""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVERS = {64: os.path.join(ROOT, "oracle", "_ref", "ref_driver_rlc64"), 32: os.path.join(ROOT, "oracle", "_ref", "ref_driver_rlc32")}

rng = random.Random(20261004)
ACGT = "ACGT"


def rseq(n):
    return "".join(rng.choice(ACGT) for _ in range(n))


def pangenome(base_len, copies, rate):
    """near-identical copies of one sequence: long BWT runs, as in a pan-genome"""
    base = rseq(base_len)
    out = []
    for _ in range(copies):
        s = list(base)
        for i in range(len(s)):
            if rng.random() < rate:
                s[i] = rng.choice(ACGT)
        out.append("".join(s))
    return "".join(out)


def texts():
    for n in (1, 2, 3, 5, 9, 14, 30, 61, 100, 254, 300, 511):
        yield rseq(n)
    for n in (6, 40, 200):
        yield "A" * n
        yield "AC" * (n // 2) + "G"
    for base_len, copies, rate in ((40, 8, 0.02), (90, 6, 0.01), (25, 20, 0.03), (150, 4, 0.005), (60, 10, 0.0)):
        yield pangenome(base_len, copies, rate)
    for _ in range(14):
        yield rseq(rng.randint(20, 600))


cmds = []
for t in texts():
    n = len(t) + 1
    if n & (n - 1) == 0:
        # a text size that is a power of two loses the terminating row's start position to the bit mask
        # (moverepr.cpp:75-77, :170-176: ceil(log2(n)) bits cannot hold n) and the reference's fast-forward runs off the
        # table: not a vector
        t += "C"
        n += 1
    for width in (64, 32):
        for rev in (0, 1):
            qs = [(0, n, rng.randint(1, 4))]
            for _ in range(11):
                b = rng.randrange(n)
                e = rng.randint(b + 1, min(n, b + rng.choice([1, 2, 5, 20, n])))
                qs.append((b, e, rng.randint(1, 4)))
            cmds.append(f"move {width} {t} {rev} {len(qs)} " + " ".join(f"{b} {e} {c}" for b, e, c in qs))


def main():
    for d in DRIVERS.values():
        if not os.path.exists(d):
            sys.exit(f"{d} missing: run `make -C oracle ref` in the build container")
    outs = [None] * len(cmds)
    for width, drv in DRIVERS.items():
        idx = [i for i, c in enumerate(cmds) if c.split()[1] == str(width)]
        res = subprocess.run([drv], input="\n".join(cmds[i] for i in idx) + "\n", capture_output=True, text=True,
                             check=True, cwd=HERE).stdout.splitlines()
        assert len(res) == len(idx)
        for i, r in zip(idx, res):
            outs[i] = r
    assert all(o is not None and o != "width" for o in outs)
    with open(os.path.join(HERE, "ref_vectors_rlc.cmds"), "w") as f:
        f.write("\n".join(cmds) + "\n")
    with open(os.path.join(HERE, "ref_vectors_rlc.out"), "w") as f:
        f.write("\n".join(outs) + "\n")
    print(f"{len(cmds)} vectors, {sum(map(len, cmds))} B in, {sum(map(len, outs))} B out")


if __name__ == "__main__":
    main()
