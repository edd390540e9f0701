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
