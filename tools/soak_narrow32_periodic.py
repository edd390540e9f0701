"""CPU soak of the 32-bit narrow-block experiment on PERIODIC and low-complexity texts (tandem repeats of period 1 ... 12, mosaics of them):
the texts on which a search carries the widest first columns and the most alternative alignments — where a matrix whose window has no
slack beside the band shows.  Found in round 4: at 7 errors (Wv up to 14 = DIAG, no slack) the 32-bit matrix computes a few rows more or
fewer than the reference's in replays on such texts (tools/soak_tiny_texts.py on the device, this tool on the CPU); up to 6 errors none.
   python3 tools/soak_narrow32_periodic.py [max k [reads per configuration]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp

op.build()
max_k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(11)
units = [b"A", b"AC", b"ACG", b"ACGT", b"AACGT", b"ACGTTGCA", b"AAAAAAAC", b"ACGTACGTTT", b"ACGTTGCAAGCT", b"AAT", b"ACACG"]
def mosaic(n):
    out = b""
    while len(out) < n:
        u = units[int(rng.integers(len(units)))]
        out += u * int(rng.integers(3, 40))
        if rng.random() < 0.3:
            out += bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 12))).tolist())
    return out[:n]
bad = 0
cfgs = 0
t0 = time.time()
texts = [(u * (n // len(u) + 1))[:n] for u in units for n in (200, 1500)] + [mosaic(n) for n in (300, 2000, 6000) for _ in range(4)]
for ti, t in enumerate(texts):
    n = len(t)
    g = np.frombuffer(t, np.uint8)
    for sparse in (1, 4):
        ix = ib.build_index(t, sparseness=sparse, seq_starts=np.array([0, n // 3, n], np.uint32), device="cpu")
        for sw in (0, 4):
            orc = op.OracleIndex(ix, kmer_size=4 if n < 1000 else 8, switch_point=sw)
            for spec, part in (("columba", "dynamic"), ("columba", "uniform"), ("multiple_opt", "dynamic"), ("minU", "static"), ("pigeon", "uniform")):
                for k in range(2, max_k + 1):
                    if spec == "multiple_opt" and k > 4: continue
                    try:
                        st = op.OracleStrategy(sp.BY_NAME[spec], "edit", part)
                        reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).tolist()) for ln in (36, 100) for _ in range(5)]
                        for ln in (36, 60, 150):
                            if n > ln + 8: reads += synth.sample_reads(g, n_reads, ln, seed=int(rng.integers(1 << 30)), edit_choices=(0, 1, 2, k - 1, k, k + 1))
                            else: reads += [(t * (ln // n + 2))[j:j + ln] for j in range(0, 20)]
                        os.environ.pop("ORC_NARROW_BLOCKS", None)
                        a = op.match_batch(orc, st, k, reads, threads=8)
                        os.environ["ORC_NARROW_BLOCKS"] = "32"
                        b = op.match_batch(orc, st, k, reads, threads=8)
                    except Exception as e:
                        if "not supported" in str(e): continue
                        raise
                    finally:
                        os.environ.pop("ORC_NARROW_BLOCKS", None)
                    cfgs += 1
                    same = np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
                    diff = {c: (a[2][c], b[2][c]) for c in a[2] if a[2][c] != b[2][c]}
                    if not same or diff:
                        bad += 1
                        print(f"text {ti} ({n} characters, {t[:16]!r}...) sparse {sparse} switch {sw} {spec} {part} k={k}: occurrences {'identical' if same else 'DIFFER'}, counters {diff}", flush=True)
print(f"periodic texts, up to {max_k} errors: {'OK' if not bad else str(bad) + ' problems'} ({cfgs} configurations, {time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
