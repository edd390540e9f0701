#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.{cmds,out}: known-answer vectors produced by the REAL
reference code (oracle/_ref/ref_driver, built by `make -C oracle ref` from the sources under
/root/reference — possible only in the build container).

The .cmds file holds the (seeded, synthetic) inputs, the .out file what the reference's own
functions returned for them.  Both are data; no reference source text is stored.
Run:  python tests/golden/make_golden.py
"""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

rng = random.Random(20251003)
ACGT = "ACGT"


def rseq(n, alphabet=ACGT):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(s, nedits):
    s = list(s)
    for _ in range(nedits):
        if not s:
            break
        p = rng.randrange(len(s))
        u = rng.random()
        if u < 0.6:
            s[p] = rng.choice([c for c in ACGT if c != s[p]])
        elif u < 0.8:
            s.insert(p, rng.choice(ACGT))
        else:
            del s[p]
    return "".join(s)


def init_ed_vector(max_ed):
    """first-column vectors as getClusterCentra produces them: neighbours differ by <= 1,
    first and last <= maxED (bitparallelmatrix.cpp:84-86)."""
    n = rng.randint(1, min(2 * max_ed + 1, 9))
    while True:
        v = [rng.randint(0, max_ed)]
        for _ in range(n - 1):
            v.append(max(0, v[-1] + rng.choice([-1, 0, 1, 1])))
        if v[-1] <= max_ed and max(v) <= max_ed + 1:
            return v


def connected_perm(p):
    start = rng.randrange(p)
    lo = hi = start
    order = [start]
    while len(order) < p:
        opts = []
        if lo > 0:
            opts.append("l")
        if hi < p - 1:
            opts.append("h")
        if rng.choice(opts) == "l":
            lo -= 1
            order.append(lo)
        else:
            hi += 1
            order.append(hi)
    return order


def bounds(p, k):
    U = sorted(rng.randint(0, k) for _ in range(p))
    U[-1] = k
    L = sorted(rng.randint(0, u) for u in U)
    L = [min(l, u) for l, u in zip(L, U)]
    for i in range(1, p):
        L[i] = max(L[i], L[i - 1])
        L[i] = min(L[i], U[i])
    return L, U


cmds = ["consts"]

# a1/a2: BWTRepresentation<5>
for n in [1, 2, 5, 63, 64, 65, 127, 128, 129, 511, 512, 513, 700] + [rng.randint(3, 600) for _ in range(12)]:
    s = list(rseq(n))
    s[rng.randrange(n)] = "$"
    cmds.append("bwt " + "".join(s))
cmds.append("bwt " + "A" * 100 + "$" + "T" * 100)  # runs
cmds.append("bwt $")

# a10: rank9 Bitvec
for n in [1, 64, 65, 511, 512, 513, 1024, 1500, 2500] + [rng.randint(2, 2000) for _ in range(8)]:
    dens = rng.choice([0.05, 0.25, 0.5, 0.9])
    cmds.append("bitvec9 " + "".join("1" if rng.random() < dens else "0" for _ in range(n)))

# a10: EncodedText<5>
for n in [1, 20, 21, 22, 42, 43, 64, 65, 200] + [rng.randint(2, 400) for _ in range(6)]:
    s = list(rseq(n))
    s[rng.randrange(n)] = "$"
    cmds.append("enc " + "".join(s))

# a5: matrix rows (in-index use: parts of 5..60 chars, both directions, initED vectors)
for _ in range(260):
    max_ed = rng.randint(0, 6)
    xl = rng.randint(max(2, max_ed), 60)
    X = rseq(xl, "ACGTN" if rng.random() < 0.1 else ACGT)
    d = rng.randint(0, 1)
    base = X if d == 0 else X[::-1]
    base = base.replace("N", "A")
    Y = mutate(base, rng.randint(0, max_ed + 2)) + rseq(12)
    if rng.random() < 0.2:
        Y = rseq(len(Y))
    init = [] if rng.random() < 0.4 else init_ed_vector(max_ed)
    if init and (init[0] > max_ed or init[-1] > max_ed):
        init = []
    cmds.append(f"matrix {X} {d} {Y} {max_ed} {len(init)} " + " ".join(map(str, init)))
# long patterns crossing several 32-row blocks (full-read matrices)
for _ in range(40):
    max_ed = rng.randint(1, 6)
    X = rseq(rng.choice([100, 150, 151, 250]))
    Y = rseq(rng.randint(0, 8)) + mutate(X, rng.randint(0, max_ed + 1)) + rseq(10)
    nz = rng.choice([1, 2 * max_ed + 1])
    cmds.append(f"matrix {X} 0 {Y} {max_ed} {nz} " + " ".join(["0"] * nz))

# a5/a11: cluster centres + traceback on full-read matrices
for _ in range(160):
    max_ed = rng.randint(1, 6)
    X = rseq(rng.choice([30, 64, 100, 150]))
    nz = rng.choice([1, 2 * max_ed + 1])
    lead = 0 if nz == 1 else rng.randint(0, 2 * max_ed)
    Y = rseq(lead) + mutate(X, rng.randint(0, max_ed)) + rseq(rng.randint(0, 12))
    if rng.random() < 0.15:
        Y = Y[:len(X) - rng.randint(1, 3)]
    cmds.append(f"traceback {X} {Y} {max_ed} {rng.randint(0, 1)} {nz}")

# a14: Search / SearchScheme
for _ in range(120):
    p = rng.randint(2, 8)
    k = rng.randint(1, 7)
    pi = connected_perm(p)
    L, U = bounds(p, k)
    cmds.append(f"search {p} " + " ".join(map(str, pi + L + U)))
SCHEMES = {
    2: ["012 011 022", "102 000 012", "210 002 012"],
    4: ["01234 00222 02244", "12034 00000 01244", "21034 01111 01244", "34210 00003 01444",
        "43210 01114 01444"],
    41: ["01234 01114 01444", "10234 00003 01444", "23410 01111 02244", "32410 00000 01244",
         "43210 00222 01244"],
    42: ["43210 00222 02244", "32410 00000 01244", "23410 01111 01244", "10234 00003 01444",
         "01234 01114 01444"],
    43: ["01234 00000 02244", "43210 00000 01344", "10234 00133 01334", "01234 00133 01334",
         "32410 00011 01244", "21034 00013 01244", "10234 00124 01244", "01234 00034 00444"],
    44: ["01234 00000 04444", "12340 00000 04444", "23410 00000 04444", "34210 00000 04444",
         "43210 00000 04444"],
}
for k, rows in SCHEMES.items():
    kk = k if k < 10 else 4
    np_ = len(rows[0].split()[0])
    flat = []
    for r in rows:
        for part in r.split():
            flat += list(part)
    cmds.append(f"scheme {kk} {len(rows)} {np_} " + " ".join(flat))
for _ in range(30):  # random schemes incl. invalid ones (error path)
    p = rng.randint(2, 6)
    k = rng.randint(1, 5)
    ns = rng.randint(1, 5)
    flat = []
    for _ in range(ns):
        pi = connected_perm(p) if rng.random() < 0.85 else rng.sample(range(p), p)
        L, U = bounds(p, k)
        flat += pi + L + U
    cmds.append(f"scheme {k} {ns} {p} " + " ".join(map(str, flat)))

# a7: MatrixMetaInfo
for _ in range(300):
    max_ed = rng.randint(1, 6)
    size = rng.randint(1, 4 * max_ed + 1)
    start_depth = rng.randint(0, 120)
    shift = rng.randint(0, 5)
    nset = rng.randint(1, size)
    eds = [rng.randint(0, max_ed + 1)]
    for _ in range(nset - 1):
        eds.append(max(0, min(max_ed + 2, eds[-1] + rng.choice([-1, -1, 0, 1, 1]))))
    toks = []
    for i in range(nset):
        a = rng.randint(0, 1000)
        w = rng.randint(1, 50)
        c = rng.randint(0, 1000)
        toks += [i, eds[i], i + rng.randint(0, 30), a, a + w, c, c + w, rng.choice(ACGT)]
    op = rng.choice(["centra", "centra", "centers", "deepest"])
    arg = rng.randint(0, max_ed) if op == "centra" else rng.randint(0, 1)
    cmds.append(f"cluster {size} {max_ed} {start_depth} {shift} {nset} " + " ".join(map(str, toks)) + f" {op} {arg}")

# a11: InTextVerificationTask::doTask
for _ in range(160):
    max_ed = rng.randint(1, 6)
    tl = rng.randint(200, 700)
    text = rseq(tl)
    pl = rng.choice([20, 36, 50, 100, 150])
    pl = min(pl, tl - 30)
    pos = rng.randint(0, tl - pl)
    pattern = mutate(text[pos:pos + pl], rng.randint(0, max_ed + 1))
    if rng.random() < 0.1:
        p = rng.randrange(len(pattern))
        pattern = pattern[:p] + "N" + pattern[p + 1:]
    nz = rng.choice([1, 2 * max_ed + 1])
    starts = []
    for _ in range(rng.randint(1, 5)):
        u = rng.random()
        if u < 0.6:
            starts.append(max(0, pos - (0 if nz == 1 else rng.randint(0, 2 * max_ed))))
        elif u < 0.8:
            starts.append(rng.randint(0, tl))
        else:
            starts.append(max(0, tl - rng.randint(0, pl + 10)))
    min_ed = rng.choice([0, 0, 0, 1, 2])
    cmds.append(f"verify {text}$ {pattern} {max_ed} {min_ed} {nz} {rng.randint(0, 1)} {len(starts)} "
                + " ".join(map(str, starts)))

# a16: TextOcc ordering + dedup, FMOcc dedup
for _ in range(60):
    n = rng.randint(0, 25)
    toks = []
    for _ in range(n):
        b = rng.randint(0, 30)
        toks += [b, b + rng.randint(95, 105), rng.randint(0, 4), rng.randint(0, 1)]
    cmds.append(f"occsort {n} " + " ".join(map(str, toks)))
for _ in range(40):
    base = []
    for _ in range(rng.randint(0, 12)):
        a = rng.randint(0, 40)
        base.append((a, a + rng.randint(1, 9), rng.randint(0, 4), rng.randint(90, 110), rng.randint(0, 3),
                     rng.randint(0, 1)))
    # ties under operator< must be exact duplicates (std::sort is unstable in Release builds)
    seen = {}
    for t in base:
        seen.setdefault((t[0], t[2], t[1] - t[0], t[4]), t)
    uniq = list(seen.values())
    items = uniq + [rng.choice(uniq) for _ in range(rng.randint(0, 5)) if uniq]
    rng.shuffle(items)
    cmds.append(f"fmoccsort {len(items)} " + " ".join(str(x) for t in items for x in t))

for _ in range(20):
    cmds.append("revcomp " + rseq(rng.randint(1, 60), "ACGTN"))

# ---- round 2: more reference units that compile without parallel_hashmap -----------------------------------
# (appended with their own generator so that the vectors above stay what they were)
rng2 = random.Random(20261004)


def rseq2(n, alphabet=ACGT):
    return "".join(rng2.choice(alphabet) for _ in range(n))


# a10: SparseSuffixArray — written by the reference's writer, read back by its mmap reader (suffixArray.h)
for n in [1, 2, 63, 64, 65, 127, 128, 129, 511, 512, 513, 520, 1023, 1025, 1600] + [rng2.randint(3, 900) for _ in range(10)]:
    t = rseq2(n - 1) + "$" if n > 1 else "$"
    sa = sorted(range(n), key=lambda i: t[i:])
    for sp in ([1, 4, 32] if n in (1, 64, 513) else [rng2.choice([1, 2, 4, 8, 16, 32, 64, 128])]):
        cmds.append(f"ssa {sp} {n} " + " ".join(map(str, sa)))

# a17 / App. A 15: Read + ReadBundle clean-up (reads.h)
IUPAC = "ACGTNacgtnRYKMSWBDHVryu.-*"
for _ in range(60):
    rid = rng2.choice("@>") + "".join(rng2.choice("abcXYZ019:/#") for _ in range(rng2.randint(1, 12)))
    for _ in range(rng2.randint(0, 2)):
        rid += "_" + "".join(rng2.choice("descr=12") for _ in range(rng2.randint(0, 6)))  # '_' stands for a space
    ln = rng2.randint(1, 80)
    rd = rseq2(ln, IUPAC if rng2.random() < 0.6 else "ACGTacgt")
    ql = "".join(chr(rng2.randint(33, 73)) for _ in range(ln))
    cmds.append(f"read {rid} {rd} {ql}")

# a15: Kmer keys of the k-mer table (tkmer.h) + Substring::containsN
for _ in range(80):
    ws = rng2.choice([1, 3, 4, 5, 8, 10, 10, 10, 11, 12, 15])
    a = rseq2(ws + rng2.randint(0, 6), "ACGTN" if rng2.random() < 0.3 else ACGT)
    oa = rng2.randint(0, len(a) - ws)
    u = rng2.random()
    if u < 0.4:
        b, ob = a, oa
    elif u < 0.7:
        pos = rng2.randrange(ws)
        b = list(a[oa:oa + ws])
        b[pos] = rng2.choice([c for c in "ACGTN" if c != b[pos]])
        b, ob = rseq2(2) + "".join(b), 2
    else:
        b = rseq2(ws + 3, "ACGTN")
        ob = rng2.randint(0, 3)
    cmds.append(f"kmer {ws} {a} {oa} {b} {ob}")

# a17: Substring accessors in both directions
for _ in range(60):
    t = rseq2(rng2.randint(1, 40), "ACGTN")
    b = rng2.randint(0, len(t))
    e = rng2.randint(0, len(t) + 5)
    cmds.append(f"substr {t} {b} {e} {rng2.randint(0, 1)}")

# a14: SearchScheme::readScheme on every scheme file of the reference's search_schemes/ (copied verbatim — data —
# to tests/golden/search_schemes/; the driver runs with tests/golden/ as working directory) and on malformed files
SCHEME_DIR = os.path.join(HERE, "search_schemes")
for root, _dirs, files in sorted(os.walk(SCHEME_DIR)):
    for fn in sorted(files):
        if fn == "searches.txt" or (fn.startswith("scheme") and fn.endswith(".txt")):
            rel = os.path.relpath(os.path.join(root, fn), HERE)
            k = int(os.path.basename(root))
            cmds.append(f"readscheme {rel} {k}")
for fn in sorted(os.listdir(os.path.join(HERE, "bad_schemes"))):
    cmds.append(f"readscheme bad_schemes/{fn} 2")
cmds.append("readscheme bad_schemes/does_not_exist.txt 2")

# f1: findCIGAR (the CIGAR of in-index occurrences and of every occurrence without one) on alignments of 0..6 edits,
# incl. indels at both ends; SAM records of single-end reads
rng3 = random.Random(20261005)


def mutate3(s, nedits):
    s = list(s)
    for _ in range(nedits):
        p = rng3.randrange(len(s))
        u = rng3.random()
        if u < 0.5:
            s[p] = rng3.choice([c for c in ACGT if c != s[p]])
        elif u < 0.75:
            s.insert(p, rng3.choice(ACGT))
        elif len(s) > 2:
            del s[p]
    return "".join(s)


for _ in range(200):
    X = "".join(rng3.choice(ACGT) for _ in range(rng3.choice([20, 36, 50, 100, 150, 250])))
    k = rng3.randint(0, 6)
    Y = mutate3(X, k)
    if rng3.random() < 0.2 and k:   # edits packed at an end
        cut = rng3.randint(1, k)
        Y = (Y[cut:] if rng3.random() < 0.5 else Y[:-cut]) if len(Y) > cut + 5 else Y
    # the score the occurrence carries: its true distance or (in-index occurrences, see findCIGAR's comment) a bit more
    import itertools

    def ed(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]
    d = ed(X, Y)
    if d > 6:
        continue
    score = d + (rng3.randint(0, 2) if rng3.random() < 0.2 else 0)
    if "N" not in X and rng3.random() < 0.1:
        p = rng3.randrange(len(X))
        X = X[:p] + "N" + X[p + 1:]
        score = min(score + 1, 8)
    cmds.append(f"findcigar {X} {Y} {min(score, 8)}")

CIGS = ["100M", "57M1I42M", "3M1D97M", "1I99M", "99M1D", "20M2I30M1D48M", "150M", "36M"]
for _ in range(60):
    rid = rng3.choice("@>") + "".join(rng3.choice("readXY019:/") for _ in range(rng3.randint(1, 10)))
    ln = rng3.randint(4, 40)
    rd = "".join(rng3.choice("ACGTNacgt") for _ in range(ln))
    ql = "".join(chr(rng3.randint(33, 73)) for _ in range(ln)) if rng3.random() < 0.85 else "-"

    def occ():
        b = rng3.randint(0, 100000)
        return f"{b} {b + ln} {rng3.randint(0, 6)} {rng3.choice(CIGS)} {rng3.randint(0, 1)} {rng3.randint(0, 2)}"
    u = rng3.random()
    if u < 0.5:
        d = rng3.randint(0, 6)
        cmds.append(f"sam1 {rid} {rd} {ql} {rng3.randint(1, 40)} {rng3.choice([d, d, max(0, d - 1)])} {rng3.randint(0, 1)} "
                    + occ().replace(f" {d} ", f" {d} ", 1))
    elif u < 0.9:
        n = rng3.randint(1, 6)
        cmds.append(f"samxa {rid} {rd} {ql} {rng3.randint(1, n)} {n} " + " ".join(occ() for _ in range(n)))
    else:
        cmds.append(f"samun {rid} {rd} {ql if ql != '-' else '*'}")

# a5/a11 beyond the 64-bit matrix (fmindex.h:240-246: in-text verification with 2k+1 zeros needs BitParallelED128 from
# k = 7): band cells of every row, cluster centres and tracebacks of the reference's 128-bit matrix
rng4 = random.Random(20260207)
for _ in range(60):
    max_ed = rng4.randint(7, 12)
    X = "".join(rng4.choice(ACGT) for _ in range(rng4.choice([64, 100, 150, 151, 250])))
    nz = rng4.choice([1, 2 * max_ed + 1, 2 * max_ed + 1])
    lead = 0 if nz == 1 else rng4.randint(0, 2 * max_ed)
    def mut4(seq, n):
        seq = list(seq)
        for _ in range(n):
            pos = rng4.randrange(len(seq))
            op = rng4.randint(0, 2)
            if op == 0:
                seq[pos] = rng4.choice(ACGT)
            elif op == 1:
                seq.insert(pos, rng4.choice(ACGT))
            elif len(seq) > 1:
                del seq[pos]
        return "".join(seq)
    Y = "".join(rng4.choice(ACGT) for _ in range(lead)) + mut4(X, rng4.randint(0, max_ed + 1)) + \
        "".join(rng4.choice(ACGT) for _ in range(rng4.randint(0, 12)))
    if rng4.random() < 0.15:
        Y = Y[:len(X) - rng4.randint(1, 3)]
    cmds.append(f"traceback128 {X} {Y} {max_ed} {rng4.randint(0, 2)} {nz}")
    if rng4.random() < 0.5:
        cmds.append(f"matrix128 {X} {Y} {max_ed} {nz}")

# f4: SAM records of paired-end reads (generateSAMPairedEnd, generateSAMUnpaired, createUnmappedSAMOccurrencePE)
rng5 = random.Random(20261005)
for _ in range(90):
    rid = rng5.choice("@>") + "".join(rng5.choice("pairAB27:/") for _ in range(rng5.randint(1, 10)))
    ln = rng5.randint(4, 40)
    rd = "".join(rng5.choice("ACGTNacgt") for _ in range(ln))
    ql = "".join(chr(rng5.randint(33, 73)) for _ in range(ln)) if rng5.random() < 0.85 else "-"

    def occ5():
        b = rng5.randint(0, 100000)
        return f"{b} {b + ln} {rng5.randint(0, 6)} {rng5.choice(CIGS)} {rng5.randint(0, 1)} {rng5.randint(0, 2)}"
    u = rng5.random()
    first = rng5.randint(0, 1)
    if u < 0.6:
        mate_mapped = 1 if rng5.random() < 0.8 else 0
        cmds.append(f"sampe {rid} {rd} {ql} {first} {rng5.randint(1, 30)} {rng5.randint(0, 12)} {rng5.randint(0, 900)} "
                    f"{rng5.randint(0, 1)} {rng5.randint(0, 1)} {mate_mapped} {occ5()} " + (occ5() if mate_mapped else str(rng5.randint(0, 1))))
    elif u < 0.85:
        cmds.append(f"samunpaired {rid} {rd} {ql} {first} {rng5.randint(1, 30)} {rng5.randint(0, 6)} {rng5.randint(0, 1)} {occ5()}")
    else:
        cmds.append(f"samunpe {rid} {rd} {ql if ql != '-' else '*'} {first} {rng5.randint(0, 1)} {rng5.randint(0, 1)}")


def main():
    if not os.path.exists(DRIVER):
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle ref` in the build container")
    inp = "\n".join(cmds) + "\n"
    out = subprocess.run([DRIVER], input=inp, capture_output=True, text=True, check=True, cwd=HERE).stdout
    assert out.count("\n") == len(cmds)
    with open(os.path.join(HERE, "ref_vectors.cmds"), "w") as f:
        f.write(inp)
    with open(os.path.join(HERE, "ref_vectors.out"), "w") as f:
        f.write(out)
    print(f"{len(cmds)} vectors, {len(inp)} B in, {len(out)} B out")


if __name__ == "__main__":
    main()
