"""Output records of the product against the reference's own answers (tests/golden/ref_vectors.*):
read clean-up, SAM lines of single-end reads (host-only C-ABI calls, no GPU), and the claim the device's CIGAR kernel
rests on — IBitParallelED::findCIGAR of an in-text occurrence IS the CIGAR of the traceBack that found it."""
import os
import subprocess

import pytest

import columba_amd as ca

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
SEQ_NAMES = ["chr1", "chr2_alt", "seqC"]   # (the names oracle/ref_driver.cpp uses for the sam* vectors)


def _vectors(kind):
    cmds = open(os.path.join(GOLD, "ref_vectors.cmds")).read().splitlines()
    outs = open(os.path.join(GOLD, "ref_vectors.out")).read().splitlines()
    return [(c.split(" ")[1:], o) for c, o in zip(cmds, outs) if c.split(" ")[0] == kind]


def test_read_cleanup_like_the_reference():
    n = 0
    for args, out in _vectors("read"):
        rid, seq, qual = args
        want = out.split(" ")   # seqID read revComp revQuality size fw rc
        got = ca.read_prepare(rid.replace("_", " "), seq, qual)
        assert list(got) == want[:4], (args, got, want)
        n += 1
    assert n >= 50


def _hit(tok):
    b, e, d, cig, strand, sq = tok
    return (SEQ_NAMES[int(sq)], int(b), int(d), bool(int(strand)), ca.parse_cigar(cig))


def test_sam_lines_like_the_reference():
    seen = {"sam1": 0, "samxa": 0, "samun": 0}
    for kind in seen:
        for args, out in _vectors(kind):
            want = out.replace("|", "\t").replace("~", "\n")
            rid, seq, qual = args[:3]
            qual = "" if qual == "-" else qual
            sid, read, rc, rq = ca.read_prepare(rid, seq, qual)
            if kind == "samun":
                got = ca.sam_unmapped_se(sid, read, qual)
            elif kind == "sam1":
                n_hits, min_score, primary = map(int, args[3:6])
                h = _hit(args[6:12])
                if primary:   # generateSAMSingleEndFirst: the read as it aligns (indexhelpers.h:625-636)
                    got = ca.sam_se(sid, h, True, n_hits, min_score, rc if h[3] else read, rq if h[3] else qual)
                else:         # generateSAMSingleEndNotFirst: "*" for sequence and quality (:673-680)
                    got = ca.sam_se(sid, h, False, n_hits, min_score, "*", "*")
            else:
                n_hits, n = int(args[3]), int(args[4])
                hits = [_hit(args[5 + 6 * i:11 + 6 * i]) for i in range(n)]
                got = ca.sam_se_xa(sid, hits, n_hits, rc if hits[0][3] else read, rq if hits[0][3] else qual)
            assert got == want, (kind, args)
            seen[kind] += 1
    assert seen["sam1"] >= 20 and seen["samxa"] >= 15 and seen["samun"] >= 2


def test_paired_end_sam_lines_like_the_reference():
    """generateSAMPairedEnd / generateSAMUnpaired / createUnmappedSAMOccurrencePE (90 vectors of the reference's own code)"""
    seen = {"sampe": 0, "samunpaired": 0, "samunpe": 0}
    flags = set()
    for kind in seen:
        for args, out in _vectors(kind):
            want = out.replace("|", "\t").replace("~", "\n")
            rid, seq, qual = args[:3]
            qual = "" if qual == "-" else qual
            sid, read, rc, rq = ca.read_prepare(rid, seq, qual)
            if kind == "samunpe":
                first, mate_mapped, mate_rev = map(int, args[3:6])
                got = ca.sam_unmapped_pe(sid, read, qual, bool(first), bool(mate_mapped), bool(mate_rev))
            elif kind == "samunpaired":
                first, n_hits, min_score, primary = map(int, args[3:7])
                h = _hit(args[7:13])
                got = ca.sam_unpaired(sid, h, bool(first), n_hits, min_score, bool(primary), rc if h[3] else read, rq if h[3] else qual)
            else:
                first, n_pairs, min_score, frag, disc, primary, mate_mapped = map(int, args[3:10])
                h = _hit(args[10:16])
                mate = _hit(args[16:22]) if mate_mapped else None
                got = ca.sam_pe(sid, h, bool(first), mate, n_pairs, min_score, frag, bool(disc), bool(primary),
                                rc if h[3] else read, rq if h[3] else qual)
            assert got == want, (kind, args)
            flags.add(int(got.split("\t")[1]))
            seen[kind] += 1
    assert seen["sampe"] >= 40 and seen["samunpaired"] >= 12 and seen["samunpe"] >= 6
    assert len(flags) >= 25  # proper / discordant pairs, both strands of read and mate, first / second, secondary, unmapped mates


def test_cigar_string_round_trip():
    for cig in ("100M", "57M1I42M", "3M1D97M", "1I99M", "20M2I30M1D48M"):
        assert ca.cigar_string(ca.parse_cigar(cig)) == cig


def test_findcigar_of_a_traced_occurrence_is_its_traceback_cigar(oracle_built):
    """k_cigar computes the CIGAR of EVERY final occurrence as findCIGAR would (fresh matrix over text[begin, end),
    maxED = the occurrence's distance).  For occurrences found by in-text verification the reference keeps the CIGAR of
    the verification's own traceBack instead: on the reference's traceback vectors (160 cases, the CIGARs were
    printed by the reference) both are the same string."""
    cmds, want = [], []
    for args, out in _vectors("traceback"):
        X, Y, max_ed, min_ed, nz = args
        t = out.split(" ")
        n_ends = int(t[1])
        for i in range(n_ends):
            e, b, ed, cig = t[2 + 4 * i:6 + 4 * i]
            if int(e) == int(b):
                continue
            cmds.append(f"findcigar {X} {Y[int(b):int(e)]} {ed}")
            want.append(cig)
    # and the `verify` vectors (InTextVerificationTask::doTask with CIGARs: text pattern maxED minED nZeros noCIGAR n starts)
    for args, out in _vectors("verify"):
        text, pattern = args[0], args[1]
        t = out.split(" ")
        for i in range(int(t[3])):
            b, e, d, cig = t[4 + 4 * i:8 + 4 * i]
            if cig == "*" or cig == "":
                continue
            cmds.append(f"findcigar {pattern} {text[int(b):int(e)]} {d}")
            want.append(cig)
    assert len(cmds) > 150
    res = subprocess.run([os.path.join(oracle_built, "oracle_driver")], input="\n".join(cmds) + "\n", capture_output=True,
                         text=True, check=True).stdout.splitlines()
    bad = [(c, r, w) for c, r, w in zip(cmds, res, want) if r != w]
    assert not bad, bad[:3]
