"""The C++ host adapter (include/columba_amd.hpp) compiles against the C-ABI library; on a GPU box the
example driver built from it returns what the Python binding returns."""
import os
import subprocess

import numpy as np
import pytest

import columba_amd as ca

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_example(tmp):
    ca.build_library()
    exe = os.path.join(tmp, "columba_chunk")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "columba_chunk.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "columba_amd"), "-lcolumba_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "columba_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adapter_compiles_and_reports_missing_index(tmp_path):
    exe = _build_example(str(tmp_path))
    r = subprocess.run([exe, str(tmp_path / "nope"), str(tmp_path / "reads.txt"), "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot open file" in r.stderr  # same error text as the reference loader


@pytest.mark.gpu
def test_adapter_example_matches_python_binding(tmp_path):
    from columba_amd import indexbuild as ib, synth
    exe = _build_example(str(tmp_path))
    g, starts = synth.genome_rep(seed=2, n=300_000, scale=2.0)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    ib.save_index(ix, str(tmp_path / "idx"))
    reads = synth.sample_reads(g, 300, 120, seed=9)
    (tmp_path / "reads.txt").write_bytes(b"\n".join(reads) + b"\n")
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "reads.txt"), "4"], capture_output=True, text=True,
                       check=True)
    got = sorted(tuple(int(x) for x in line.split()) for line in r.stdout.splitlines())
    occ, offs, _ = ca.match_batch(ca.Index(ix), ca.SearchStrategy("multiple_opt"), 4, reads)
    exp = sorted((i, int(o["begin"]), int(o["end"]), int(o["distance"]), int(o["strand"]))
                 for i in range(len(reads)) for o in occ[int(offs[i]):int(offs[i + 1])])
    assert got == exp and len(exp) > 0
    # a chunk with reads of 3 and 5 characters (matched by naive backtracking, as in the reference): the short reads are named,
    # every other read keeps its list
    mixed = reads[:50] + [b"ACG", b"ACGTA"] + reads[50:100]
    (tmp_path / "mixed.txt").write_bytes(b"\n".join(mixed) + b"\n")
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "mixed.txt"), "4"], capture_output=True, text=True, check=True)
    got = sorted(tuple(int(x) for x in line.split()) for line in r.stdout.splitlines())
    shift = lambda i: i if i < 50 else i + 2
    assert [t for t in got if t[0] not in (50, 51)] == sorted((shift(t[0]),) + t[1:] for t in exp if t[0] < 100)
    assert sum(t[0] == 50 for t in got) > 1000 and sum(t[0] == 51 for t in got) > 1000   # (they match all over the text)
    assert "read 50 (3 characters) was matched by naive backtracking" in r.stderr
    assert "read 51 (5 characters) was matched by naive backtracking" in r.stderr


def _build_align(tmp):
    ca.build_library()
    exe = os.path.join(tmp, "columba_align")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "columba_align.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "columba_amd"), "-lcolumba_amd", "-lz",
                           "-Wl,-rpath," + os.path.join(ROOT, "columba_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_align_driver_compiles_and_checks_its_arguments(tmp_path):
    exe = _build_align(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    (tmp_path / "r.fq").write_text("@r1\nACGT\n+\nIIII\n")
    r = subprocess.run([exe, "-r", str(tmp_path / "nope"), "-f", str(tmp_path / "r.fq"), "-o", str(tmp_path / "o.sam")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot open file" in r.stderr


@pytest.mark.gpu
def test_align_driver_fastq_to_sam(tmp_path, oracle_built):
    """FASTQ in, SAM out through the C++ host layer (Reader / OutputWriter of include/columba_amd_io.hpp, chunks of 400
    reads): the records of ALL mode equal the oracle's SAM text for the same reads; BEST mode (the default) gives
    one primary record per mapped read whose position the oracle's BEST mode agrees with."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    import schemes_py as sp
    from columba_amd import indexbuild as ib, synth
    exe = _build_align(str(tmp_path))
    g, starts = synth.genome_rep(seed=2, n=300_000, scale=2.0)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    ib.save_index(ix, str(tmp_path / "idx"))
    reads = synth.sample_reads(g, 1000, 120, seed=9, n_frac=0.01)
    rng = np.random.default_rng(3)
    ids = [f"@read{i} len=120" for i in range(len(reads))]
    quals = ["".join(chr(33 + int(q)) for q in rng.integers(0, 41, 120)) for _ in reads]
    with open(tmp_path / "reads.fq", "w") as f:
        for i, r in enumerate(reads):
            f.write(f"{ids[i]}\n{r.decode()}\n+\n{quals[i]}\n")
    names = ix.seq_names
    out = tmp_path / "all.sam"
    subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "reads.fq"), "-o", str(out), "-a", "all", "-e", "4",
                    "-S", "columba", "-b", "400"], check=True, capture_output=True, text=True)
    lines = out.read_text().splitlines()
    header = [x for x in lines if x.startswith("@")]
    assert header[0] == "@HD\tVN:1.6\tSO:queryname" and sum(x.startswith("@SQ") for x in header) == len(names)
    want = op.match_batch_sam(op.OracleIndex(ix), op.OracleStrategy(sp.COLUMBA, "edit", "dynamic"), 4, reads, ids, quals, names,
                              unmapped=True, xa=False).splitlines()
    assert [x for x in lines if not x.startswith("@")] == want and len(want) > 1000
    # the same reads gz-compressed (SeqFile of the reference picks zlib by the .gz extension, seqfile.cpp): the same records
    import gzip
    with gzip.open(tmp_path / "reads.fq.gz", "wb") as f:
        f.write((tmp_path / "reads.fq").read_bytes())
    outz = tmp_path / "all_gz.sam"
    subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "reads.fq.gz"), "-o", str(outz), "-a", "all", "-e", "4",
                    "-S", "columba", "-b", "400"], check=True, capture_output=True, text=True)
    assert [x for x in outz.read_text().splitlines() if not x.startswith("@")] == want
    # ... and a gz-compressed SAM file out (OutputWriter: ".sam.gz", fastq.cpp:482)
    outgz = tmp_path / "all_out.sam.gz"
    subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "reads.fq"), "-o", str(outgz), "-a", "all", "-e", "4",
                    "-S", "columba", "-b", "400"], check=True, capture_output=True, text=True)
    with gzip.open(outgz, "rt") as f:
        assert [x for x in f.read().splitlines() if not x.startswith("@")] == want
    # BEST mode (the reference's default): one primary record per read, in input order
    outb = tmp_path / "best.sam"
    subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "reads.fq"), "-o", str(outb), "-I", "95"], check=True,
                   capture_output=True, text=True)
    recs = [x.split("\t") for x in outb.read_text().splitlines() if not x.startswith("@")]
    prim = [f for f in recs if not int(f[1]) & 256]
    assert [f[0] for f in prim] == [i[1:].split(" ")[0] for i in ids]
    o_occ, o_sid, o_sb, o_cig, o_off, o_best, o_hits, _ = op.match_best(op.OracleIndex(ix), op.OracleStrategy(sp.COLUMBA, "edit", "dynamic"),
                                                                        reads, x=0, min_identity=95, max_supported=6, threads=8)
    for i, f in enumerate(prim):
        if o_best[i] == 0xFFFFFFFF:
            assert f[1] == "4"
        else:
            j = int(o_off[i])
            assert (f[2], int(f[3]), f[5], f[11]) == (names[int(o_sid[j])], int(o_sb[j]) + 1, o_cig[j], f"AS:i:{int(o_best[i])}")


def _build_bmove(tmp):
    ca.build_library()
    exe = os.path.join(tmp, "bmove_exact")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bmove_exact.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "columba_amd"), "-lcolumba_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "columba_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_bmove_adapter_compiles_and_reports_missing_index(tmp_path):
    exe = _build_bmove(str(tmp_path))
    r = subprocess.run([exe, str(tmp_path / "nope"), str(tmp_path / "reads.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot open file" in r.stderr and ".LFBP" in r.stderr


@pytest.mark.gpu
def test_bmove_adapter_example_matches_python_binding(tmp_path):
    """BMove of include/columba_amd_bmove.hpp on files written by movebuild.save_move: exactMatchesOutput of a chunk =
    the Python binding's occurrences; a bidirectional walk with the reference's method names ends on the same interval"""
    from columba_amd import movebuild, synth
    exe = _build_bmove(str(tmp_path))
    rng = np.random.default_rng(3)
    base = rng.integers(0, 4, 5000)
    parts = []
    for _ in range(12):
        s = base.copy()
        m = rng.random(base.shape[0]) < 0.01
        s[m] = rng.integers(0, 4, int(m.sum()))
        parts.append(s)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)]
    mv = movebuild.build_move(text.tobytes(), device="cuda")
    movebuild.save_move(mv, str(tmp_path / "idx"))
    t = mv.text.tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for i in range(200):
        L = int(rng.choice([8, 20, 60, 150]))
        p0 = int(rng.integers(0, len(t) - 1 - L))
        r = t[p0:p0 + L]
        reads.append(r.translate(comp)[::-1] if i % 3 == 1 else r)
    (tmp_path / "reads.txt").write_bytes(b"\n".join(reads) + b"\n")
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "reads.txt")], capture_output=True, text=True, check=True)
    got = [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()]
    dev = ca.MoveIndex(mv)
    occ, offs, cnt = dev.match_exact(reads)
    exp = [(i, int(o["begin"]), int(o["end"]), int(o["distance"]), int(o["strand"]))
           for i in range(len(reads)) for o in occ[int(offs[i]):int(offs[i + 1])]]
    assert got == exp and len(exp) > 1000
    # the same index built by the C++ tool from a FASTA file (columba_build --rlc: include/columba_amd_build.hpp) loads and answers alike
    build_exe = str(tmp_path / "columba_build")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "columba_build.cpp"),
                           "-o", build_exe, "-lz"])
    (tmp_path / "ref.fa").write_bytes(b">seq0\n" + t[:-1] + b"\n")
    subprocess.check_call([build_exe, "--rlc", "-r", str(tmp_path / "idx_cpp"), "-f", str(tmp_path / "ref.fa")])
    r2 = subprocess.run([exe, str(tmp_path / "idx_cpp"), str(tmp_path / "reads.txt")], capture_output=True, text=True, check=True)
    assert r2.stdout == r.stdout and r2.stderr == r.stderr
    err = r.stderr.split("\n")
    assert err[0] == f"nodes {cnt['NODE_COUNTER']}"
    walk = err[1].split()
    fw = sorted(int(o["begin"]) for o in occ[int(offs[0]):int(offs[1])] if o["strand"] == 0)
    assert walk[0] == "walk" and int(walk[1]) == len(reads[0]) and [int(x) for x in walk[4:]] == fw
    # the approximate search through the adapter (rlc::SearchStrategy::matchApproxBatch): the Python binding's lists and node count
    areads = [bytes(r) for r in synth.sample_reads(np.frombuffer(t[:-1], dtype=np.uint8), 150, 100, seed=8, edit_choices=(0, 1, 2, 3))]
    (tmp_path / "areads.txt").write_bytes(b"\n".join(areads) + b"\n")
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "areads.txt"), "3", "kuch1", "6"], capture_output=True, text=True, check=True)
    got = [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()]
    occ, offs, cnt = dev.match_batch(ca.SearchStrategy("kuch1", "edit", "dynamic"), 3, areads, kmer_size=6)
    exp = [(i, int(o["begin"]), int(o["end"]), int(o["distance"]), int(o["strand"]))
           for i in range(len(areads)) for o in occ[int(offs[i]):int(offs[i + 1])]]
    assert got == exp and len(exp) > 100
    assert r.stderr.split("\n")[0] == f"nodes {cnt['NODE_COUNTER']}"
    # SAM records through the adapter (BMove::attachText + rlc::SearchStrategy::samOfChunk) = the Python binding's
    (tmp_path / "text.txt").write_bytes(t[:-1])
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "areads.txt"), "3", "kuch1", "6", str(tmp_path / "text.txt")],
                       capture_output=True, text=True, check=True)
    dev.attach_text(t)
    mb = ca.MoveBatch(dev, ca.SearchStrategy("kuch1", "edit", "dynamic"), 3, reads=areads, kmer_size=6)
    mb.want_alignments()
    mb.run()
    sam = mb.sam([f"r{i}" for i in range(len(areads))], ["I" * len(x) for x in areads], ["seq0"])
    assert r.stdout == sam and sam.count("\n") >= len(areads) and "\t100M\t" in sam
    # BEST mode through the adapter (rlc::SearchStrategy::samOfChunkBest = cmb_move_match_best + the record builders): one primary
    # record per read carrying the binding's best alignment
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "areads.txt"), "3", "columba", "6", str(tmp_path / "text.txt"), "best", "0", "96"],
                       capture_output=True, text=True, check=True)
    b_occ, b_aln, b_ops, b_off, b_best, b_hits, _ = ca.match_best(dev, ca.SearchStrategy("columba", "edit", "dynamic"), areads, x=0,
                                                                  min_identity=96, kmer_size=6)
    prim = [f.split("\t") for f in r.stdout.splitlines() if not int(f.split("\t")[1]) & 256]
    assert len(prim) == len(areads) and r.stderr.split("\n")[0] == f"mapped {int((b_best != 0xFFFFFFFF).sum())}"
    for i, f in enumerate(prim):
        if b_best[i] == 0xFFFFFFFF:
            assert f[1] == "4"
        else:
            a = b_aln[int(b_off[i])]
            cig = ca.cigar_string(b_ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])])
            assert (f[2], int(f[3]), f[5], f[11]) == ("seq0", int(a["seq_begin"]) + 1, cig, f"AS:i:{int(b_best[i])}")
    assert (b_best != 0xFFFFFFFF).sum() > 100
    # read pairs in BEST mode through the adapter (rlc::SearchStrategy::samOfChunkPairedBest: cmb_pair_best_* over b-move batches) = the
    # Python host layer's records (ca.pair_chunk_sam_best on the same index)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    m1, m2 = [], []
    for i in range(120):
        p0 = int(rng.integers(0, len(t) - 700))
        frag = int(rng.integers(220, 400))
        a, b = bytearray(t[p0:p0 + 100]), bytearray(t[p0 + frag - 100:p0 + frag].translate(comp)[::-1])
        if i % 4 == 0:
            a[17] = b"ACGT"[(b"ACGT".index(bytes([a[17]])) + 1) % 4]
        m1.append(bytes(a if i % 2 == 0 else b))
        m2.append(bytes(b if i % 2 == 0 else a))
    (tmp_path / "m1.txt").write_bytes(b"\n".join(m1) + b"\n")
    (tmp_path / "m2.txt").write_bytes(b"\n".join(m2) + b"\n")
    r = subprocess.run([exe, str(tmp_path / "idx"), str(tmp_path / "m1.txt"), "3", "columba", "6", str(tmp_path / "text.txt"), "pairs",
                        str(tmp_path / "m2.txt"), "0", "95", "600"], capture_output=True, text=True, check=True)
    want, mapped, _ = ca.pair_chunk_sam_best(dev, ca.SearchStrategy("columba", "edit", "dynamic"), m1, m2, [f"r{i}/1" for i in range(120)],
                                             [f"r{i}/2" for i in range(120)], ["I" * 100] * 120, ["I" * 100] * 120, ["seq0"], x=0, min_identity=95,
                                             orientation=ca.ORIENTATION_FR, max_frag=600, min_frag=0, kmer_size=6)
    assert r.stdout == want and r.stderr.split("\n")[0] == f"mapped pairs {mapped}" and mapped > 100


@pytest.mark.gpu
def test_align_driver_paired_end(tmp_path):
    """two FASTQ files in, SAM out through the C++ host layer in paired mode (samOfChunkPairedAll: two GPU batches + cmb_pair_sam
    per pair, chunks of 120 pairs): the records equal what the Python host layer makes of the same reads"""
    from columba_amd import indexbuild as ib, synth
    exe = _build_align(str(tmp_path))
    g, starts = synth.genome_rep(seed=2, n=300_000, scale=2.0)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    names = [f"chr{i}" for i in range(len(starts) - 1)]
    ix.seq_names = names
    ib.save_index(ix, str(tmp_path / "idx"))
    rng = np.random.default_rng(4)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n, L = 300, 80
    r1, r2 = [], []
    for i in range(n):
        frag = int(rng.integers(200, 380))
        p0 = int(rng.integers(500, len(g) - 900))
        f = g[p0:p0 + frag].tobytes()
        a, b = bytearray(f[:L]), bytearray(f[-L:].translate(comp)[::-1])
        if i % 7 == 0:
            b = bytearray(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))  # a mate from nowhere
        if i % 5 == 0:
            a[10] = ord("N")
        if i % 2:
            a, b = b, a
        r1.append(bytes(a))
        r2.append(bytes(b))
    q = "I" * L
    (tmp_path / "r1.fq").write_text("".join(f"@p{i}/1 x\n{r1[i].decode()}\n+\n{q}\n" for i in range(n)))
    (tmp_path / "r2.fq").write_text("".join(f"@p{i}/2 x\n{r2[i].decode()}\n+\n{q}\n" for i in range(n)))
    out = tmp_path / "o.sam"
    run = subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "r1.fq"), "-F", str(tmp_path / "r2.fq"), "-o", str(out),
                          "-a", "all", "-e", "2", "-S", "multiple_opt", "-b", "120", "-X", "500", "-N", "100"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    body = "".join(ln for ln in out.read_text().splitlines(keepends=True) if not ln.startswith("@"))
    dev = ca.Index(ix)
    want, mapped = ca.pair_chunk_sam(dev, ca.SearchStrategy("multiple_opt", "edit", "dynamic"), 2, r1, r2, [f"@p{i}/1 x" for i in range(n)],
                                     [f"@p{i}/2 x" for i in range(n)], [q] * n, [q] * n, names, ca.ORIENTATION_FR, 500, 100, True, True)
    assert body == want and mapped > 0.6 * n
    flags = [int(ln.split("\t")[1]) for ln in body.splitlines()]
    assert sum(1 for f in flags if f & 2) > n and any(f & 8 for f in flags)  # proper pairs, and pairs with an unmapped mate
    # the same pairs in BEST mode, the CLI's default (samOfChunkPairedBest: the chunk walks through its strata together over
    # cmb_pair_best_*, a device batch per mate and distance asked for): the records of the Python host layer over the same C-ABI,
    # chunk by chunk (pairs are independent of their chunk)
    out2 = tmp_path / "o2.sam"
    run = subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "r1.fq"), "-F", str(tmp_path / "r2.fq"), "-o", str(out2),
                          "-I", "95", "-x", "0", "-S", "columba", "-b", "120", "-X", "500", "-N", "100"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    body2 = "".join(ln for ln in out2.read_text().splitlines(keepends=True) if not ln.startswith("@"))
    want2, mapped2, _ = ca.pair_chunk_sam_best(dev, ca.SearchStrategy("columba", "edit", "dynamic"), r1, r2, [f"@p{i}/1 x" for i in range(n)],
                                               [f"@p{i}/2 x" for i in range(n)], [q] * n, [q] * n, names, x=0, min_identity=95,
                                               orientation=ca.ORIENTATION_FR, max_frag=500, min_frag=100)
    assert body2 == want2 and mapped2 > 0.6 * n
    flags2 = [int(ln.split("\t")[1]) for ln in body2.splitlines()]
    assert sum(1 for f in flags2 if f & 2) > n and any(f & 4 for f in flags2)
    # without -O / -X / -N the parameters are inferred from the first chunk (single-end, the pairs whose mates both map unambiguously:
    # parallel.cpp:236-312, :402-466), which is then paired from its single-end results (cmb_pair_best_seed); the other chunks as before
    out3 = tmp_path / "o3.sam"
    run = subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "r1.fq"), "-F", str(tmp_path / "r2.fq"), "-o", str(out3),
                          "-I", "95", "-x", "0", "-S", "columba", "-b", "120"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    import re
    m = re.search(r"Found (\d+) unambiguous pairs while processing 240 reads", run.stderr)
    assert m and int(m.group(1)) > 50, run.stderr
    m = re.search(r"orientation FR, insert size ([0-9.]+) \+- ([0-9.]+), bounds \[(\d+), (\d+)\]", run.stderr)
    assert m and 250 < float(m.group(1)) < 330 and 30 < float(m.group(2)) < 80 and int(m.group(4)) > 400, run.stderr
    body3 = [ln for ln in out3.read_text().splitlines() if not ln.startswith("@")]
    by = {}
    for ln in body3:
        by.setdefault(ln.split("\t")[0].split("/")[0], []).append(int(ln.split("\t")[1]))
    assert len(by) == n
    assert sum(1 for fl in by.values() if any(f & 2 for f in fl)) > 0.6 * n
    # the first chunk, record by record: the Python host layer's inference phase and seeded walk over the same C-ABI
    inf = ca.infer_paired_end_best(dev, ca.SearchStrategy("columba", "edit", "dynamic"), r1[:120], r2[:120], min_identity=95)
    got_inf = inf["inferred"]
    assert (int(got_inf.min_insert), int(got_inf.max_insert)) == (int(m.group(3)), int(m.group(4))) and got_inf.orientation == ca.ORIENTATION_FR
    want3, _, _ = ca.pair_chunk_sam_best(dev, ca.SearchStrategy("columba", "edit", "dynamic"), r1[:120], r2[:120], [f"@p{i}/1 x" for i in range(120)],
                                         [f"@p{i}/2 x" for i in range(120)], [q] * 120, [q] * 120, names, x=0, min_identity=95,
                                         orientation=int(got_inf.orientation), max_frag=int(got_inf.max_insert), min_frag=int(got_inf.min_insert),
                                         start_from=inf)
    first3 = "".join(ln + "\n" for ln in body3 if int(ln.split("\t")[0][1:].split("/")[0]) < 120)
    assert first3 == want3 and f"Found {inf['unambiguous_pairs']} unambiguous pairs" in run.stderr
    # beyond the first chunk the records are those of the run with the same bounds given
    lo, hi = int(m.group(3)), int(m.group(4))
    out4 = tmp_path / "o4.sam"
    run = subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "r1.fq"), "-F", str(tmp_path / "r2.fq"), "-o", str(out4),
                          "-I", "95", "-x", "0", "-S", "columba", "-b", "120", "-X", str(hi), "-N", str(lo)], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    later = lambda path: [ln for ln in path.read_text().splitlines() if not ln.startswith("@") and int(ln.split("\t")[0][1:].split("/")[0]) >= 120]
    assert later(out3) == later(out4) and len(later(out3)) > 300
    # the same in ALL mode: single-end phase with the strands of a read filtered together, the chunk paired from those lists
    # (pairSingleEndedMatchesAll; a read 2 that was not matched yet is matched strand by strand), the later chunks as with the bounds given
    out5, out6 = tmp_path / "o5.sam", tmp_path / "o6.sam"
    run = subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "r1.fq"), "-F", str(tmp_path / "r2.fq"), "-o", str(out5),
                          "-a", "all", "-e", "2", "-S", "multiple_opt", "-b", "120"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    m = re.search(r"orientation FR, insert size ([0-9.]+) \+- ([0-9.]+), bounds \[(\d+), (\d+)\]", run.stderr)
    assert m and 250 < float(m.group(1)) < 330 and "unambiguous pairs while processing 240 reads" in run.stderr, run.stderr
    run = subprocess.run([exe, "-r", str(tmp_path / "idx"), "-f", str(tmp_path / "r1.fq"), "-F", str(tmp_path / "r2.fq"), "-o", str(out6),
                          "-a", "all", "-e", "2", "-S", "multiple_opt", "-b", "120", "-X", m.group(4), "-N", m.group(3)], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    assert later(out5) == later(out6) and len(later(out5)) > 300
    first = lambda path: [ln.split("\t") for ln in path.read_text().splitlines() if not ln.startswith("@") and int(ln.split("\t")[0][1:].split("/")[0]) < 120]
    proper5 = {(f[0], f[2], f[3]) for f in first(out5) if int(f[1]) & 2}
    proper6 = {(f[0], f[2], f[3]) for f in first(out6) if int(f[1]) & 2}
    assert len(proper5) > 100 and len(proper5 ^ proper6) <= 0.05 * len(proper6)   # (joint against per-strand filtering: the same proper pairs but for a few)
