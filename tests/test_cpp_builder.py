"""The C++ index builder (include/columba_amd_build.hpp, examples/columba_build.cpp: SURVEY.md section 8f rank 4) against the harness
builder (columba_amd/indexbuild.py, whose files the reference's own readers load: tests/test_index_files.py and the golden vectors):
the suffix array of a text is unique, so every index file must be byte-identical — and the text itself follows the reference's
preprocessing (buildindex.cpp:150-262, :614-683): upper case, one '$', non-ACGT characters replaced from std::minstd_rand(42)."""
import os
import subprocess

import numpy as np
import pytest

from columba_amd import indexbuild as ib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def builder(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("cpp_builder"))
    exe = os.path.join(tmp, "columba_build")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "columba_build.cpp"), "-o", exe, "-lz"])
    return exe, tmp


def _write_fasta(path, records, width=70, lower_every=0):
    with open(path, "w") as f:
        for k, (name, seq) in enumerate(records):
            f.write(f">{name} description {k}\n")
            s = seq.decode()
            if lower_every and k % lower_every == 0:
                s = s.lower()
            for o in range(0, len(s), width):
                f.write(s[o:o + width] + "\n")
            f.write("\n")


FILES = ["meta", "cct", "txt.bin", "bwt", "brt", "rev.brt", "pos", "sna", "fsid", "headerSN.bin"]


@pytest.mark.parametrize("n,sparseness,n_seqs", [(20_000, 4, 3), (5_003, 8, 1), (64 * 512 - 1, 1, 2), (64 * 512, 16, 5), (777, 2, 1)])
def test_index_files_equal_the_harness_builder(builder, n, sparseness, n_seqs):
    exe, tmp = builder
    g, _ = synth.genome_rep(seed=100 + n, n=n, scale=2.0)
    g = g[:n]   # ACGT only
    cuts = [0] + sorted(np.random.default_rng(n).choice(np.arange(1, n), n_seqs - 1, replace=False).tolist()) + [n]
    records = [(f"chr{j}", g[cuts[j]:cuts[j + 1]].tobytes()) for j in range(n_seqs)]
    fa = os.path.join(tmp, f"g{n}.fa")
    _write_fasta(fa, records, lower_every=2)
    base_c = os.path.join(tmp, f"c{n}")
    subprocess.check_call([exe, "-s", str(sparseness), "-r", base_c, "-f", fa])
    ix = ib.build_index(g.tobytes(), sparseness=sparseness, seq_starts=np.array(cuts, dtype=np.uint32), seq_names=[f"chr{j}" for j in range(n_seqs)],
                        device="cpu")
    base_p = os.path.join(tmp, f"p{n}")
    ib.save_index(ix, base_p)
    for ext in FILES + [f"sa.{sparseness}", f"sa.bv.{sparseness}"]:
        a, b = open(f"{base_c}.{ext}", "rb").read(), open(f"{base_p}.{ext}", "rb").read()
        assert a == b, (ext, len(a), len(b))
    # ... and they load (the harness loader reads the reference's formats)
    back = ib.load_index(base_c, sparseness=sparseness)
    assert back.text.tobytes() == g.tobytes() + b"$" and np.array_equal(back.sa_samples, ix.sa_samples)


def test_all_sparseness_factors(builder):
    """-a: the sparse suffix array for every factor 1 ... 128 (buildindex.cpp:1914-1918), each equal to the build with that factor alone"""
    exe, tmp = builder
    g, _ = synth.genome_rep(seed=9, n=6_000, scale=2.0)
    fa = os.path.join(tmp, "all.fa")
    _write_fasta(fa, [("s", g[:6000].tobytes())])
    subprocess.check_call([exe, "-a", "-r", os.path.join(tmp, "alls"), "-f", fa])
    for sf in (1, 2, 4, 8, 16, 32, 64, 128):
        subprocess.check_call([exe, "-s", str(sf), "-r", os.path.join(tmp, f"one{sf}"), "-f", fa])
        for ext in (f"sa.{sf}", f"sa.bv.{sf}"):
            assert open(os.path.join(tmp, f"alls.{ext}"), "rb").read() == open(os.path.join(tmp, f"one{sf}.{ext}"), "rb").read(), ext


def test_text_preprocessing_follows_the_reference(builder):
    """several files, several sequences per file, lower case, a file without a header line, runs of N and other IUPAC codes: positions, names,
    first sequence per file; replaced characters are ACGT, identical between two runs, a seeded pattern with -l"""
    exe, tmp = builder
    f1, f2, f3 = (os.path.join(tmp, n) for n in ("a.fa", "b.fasta", "c.fna"))
    _write_fasta(f1, [("s1", b"ACGTNNNNACGTRYACGT"), ("s2", b"TTTTGGGG")])
    _write_fasta(f2, [("s3", b"NNNNNNNNNNCCCC")])
    with open(f3, "w") as f:
        f.write("acgtacgtnnacgt\nACGT\n")
    outs = []
    for run, extra in enumerate(([], [], ["-l", "4"])):
        base = os.path.join(tmp, f"pre{run}")
        subprocess.check_call([exe, "-r", base, "-f", f1, f2, f3] + extra)
        raw = open(base + ".txt.bin", "rb").read()
        n = int(np.frombuffer(raw[:4], dtype=np.uint32)[0])
        outs.append(raw[4:])
        assert n == len(raw) - 4 == 18 + 8 + 14 + 18 + 1 and raw.endswith(b"$") and set(raw[4:-1]) <= set(b"ACGT")
        assert np.fromfile(base + ".pos", dtype=np.uint32).tolist() == [0, 18, 26, 40, 58]
        assert np.fromfile(base + ".fsid", dtype=np.uint32).tolist() == [0, 2, 3]
        assert open(base + ".headerSN.bin").read() == f"@SQ\tSN:s1\tLN:18\n@SQ\tSN:s2\tLN:8\n@SQ\tSN:s3\tLN:14\n@SQ\tSN:{f3}\tLN:18\n"
        t = raw[4:]
        assert t[:4] == b"ACGT" and t[8:12] == b"ACGT" and t[14:18] == b"ACGT" and t[18:26] == b"TTTTGGGG" and t[36:40] == b"CCCC"
        assert t[40:48] == b"ACGTACGT" and t[50:58] == b"ACGTACGT"
    assert outs[0] == outs[1]                      # the generator is seeded: the same text every time
    seeded = outs[2]
    assert seeded[4:8] == seeded[26:30] == seeded[30:34] and seeded[34:36] == seeded[4:6] == seeded[12:14] == seeded[48:50]   # runs restart the 4-character seed


def test_suffix_array_against_plain_sorting(builder):
    """the tool's own SA-IS on texts with long repeats: the sampled suffix array at sparseness 1 IS the suffix array"""
    exe, tmp = builder
    rng = np.random.default_rng(5)
    for trial, text in enumerate([b"A" * 300, b"ACGT" * 100, b"AC" * 50 + b"G" + b"AC" * 50, rng.choice(np.frombuffer(b"AC", dtype=np.uint8), 400).tobytes(), b"T", b"GATTACA" * 30 + b"GATTA"]):
        fa = os.path.join(tmp, f"sa{trial}.fa")
        _write_fasta(fa, [("x", text)])
        base = os.path.join(tmp, f"sa{trial}")
        subprocess.check_call([exe, "-s", "1", "-r", base, "-f", fa])
        t = text + b"$"
        want = sorted(range(len(t)), key=lambda i: t[i:])
        assert np.fromfile(base + ".sa.1", dtype=np.uint32).tolist() == want


MOVE_FILES = ["LFBP", "rev.LFBP"] + [e + ".u64" for e in ("smpf", "smpl", "rev.smpf", "rev.smpl", "prdf", "ftr", "prdl", "ltr", "plcp.pos", "plcp.sum")]


@pytest.mark.parametrize("base_len,copies,snp,n_seqs", [(3_000, 8, 0.01, 8), (701, 3, 0.0, 1), (20_000, 4, 0.002, 2), (97, 1, 0.0, 1)])
def test_move_index_files_equal_the_harness_builder(builder, base_len, copies, snp, n_seqs):
    """--rlc: move tables in the reference's .LFBP format, samples, predecessors and the run-length PLCP — byte for byte what
    columba_amd/movebuild.py writes for the same text (whose tables the oracle's move_driver and the golden vectors of the b-move tests read)"""
    from columba_amd import movebuild
    exe, tmp = builder
    text = movebuild.pangenome(base_len, copies, snp, seed=base_len)
    n = int(text.shape[0])
    cuts = np.linspace(0, n, n_seqs + 1).astype(np.int64).tolist()
    records = [(f"hap{j}", text[cuts[j]:cuts[j + 1]].tobytes()) for j in range(n_seqs)]
    fa = os.path.join(tmp, f"pg{base_len}.fa")
    _write_fasta(fa, records, lower_every=3)
    base_c = os.path.join(tmp, f"mc{base_len}")
    subprocess.check_call([exe, "--rlc", "--keep-text", "-r", base_c, "-f", fa])
    mv = movebuild.build_move(text, device="cpu")
    base_p = os.path.join(tmp, f"mp{base_len}")
    movebuild.save_move(mv, base_p)
    for ext in MOVE_FILES:
        a, b = open(f"{base_c}.{ext}", "rb").read(), open(f"{base_p}.{ext}", "rb").read()
        assert a == b, (ext, len(a), len(b))
    # the files every flavour has, in the 64-bit length_t of the reference's RLC build
    assert np.fromfile(base_c + ".pos", dtype=np.uint64).tolist() == cuts
    raw = open(base_c + ".txt.bin", "rb").read()
    assert int(np.frombuffer(raw[:8], dtype=np.uint64)[0]) == n + 1 and raw[8:] == text.tobytes() + b"$"
    cct = np.fromfile(base_c + ".cct", dtype=np.uint64)
    assert cct.shape[0] == 256 and int(cct.sum()) == n + 1 and cct[ord("$")] == 1
    assert open(base_c + ".meta").read().split() == ["21", "8", "RLC"]
    assert not os.path.exists(base_c + ".bwt")


def test_move_builder_defaults_and_refusals(builder):
    """seed length 100 by default in this flavour (definitions.h:41): a run of N becomes the same 100-character pattern, restarted after
    every ACGT character; no text file unless asked; a text of 2^k characters is refused as the reference's table does (moverepr.cpp:75-77)"""
    exe, tmp = builder
    fa = os.path.join(tmp, "n.fa")
    _write_fasta(fa, [("a", b"ACGT" + b"N" * 230 + b"ACGTT" + b"N" * 30 + b"GG")])
    base = os.path.join(tmp, "mdef")
    subprocess.check_call([exe, "--rlc", "-r", base, "-f", fa])
    assert not os.path.exists(base + ".txt.bin")
    subprocess.check_call([exe, "--rlc", "--keep-text", "-r", base, "-f", fa])
    t = open(base + ".txt.bin", "rb").read()[8:]
    fill = t[4:234]
    assert set(fill) <= set(b"ACGT") and fill[:100] == fill[100:200] and fill[:30] == fill[200:230] == t[239:269]
    fa2 = os.path.join(tmp, "p2.fa")
    _write_fasta(fa2, [("a", (b"ACGTTGCA" * 8)[:63])])   # 63 characters + '$'
    r = subprocess.run([exe, "--rlc", "-r", os.path.join(tmp, "p2"), "-f", fa2], capture_output=True, text=True)
    assert r.returncode != 0 and "power of two" in r.stderr
