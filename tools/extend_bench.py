"""rank/extend microbenchmark (cmb_extend_bench): random parents, all four children per parent."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 256_000_000
N = 1 << 24
g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
del g
dev = ca.Index(ix)
rng = np.random.default_rng(1)
for name in ("wide", "narrow"):
    b_ = rng.integers(0, ix.n, N)
    if name == "wide":   # log-uniform widths: begin and end in different rank blocks
        w = np.minimum((2.0 ** rng.uniform(0, np.log2(ix.n), N)).astype(np.int64), ix.n - b_)
    else:                # deep nodes of a search: widths 1..64
        w = np.minimum(rng.integers(1, 65, N), ix.n - b_)
    b2 = rng.integers(0, ix.n - 64, N)
    r = np.stack([b_, b_ + w, b2, np.minimum(b2 + w, ix.n)], axis=1).astype(np.uint32)
    din = torch.from_numpy(r.view(np.int32)).cuda()
    dout = torch.empty((N, 16), dtype=torch.int32, device="cuda")
    dok = torch.empty((N, 4), dtype=torch.uint8, device="cuda")
    ms = C.c_float()
    ref = None
    for mode in (0, 1, 2):
        rc = ca.lib().cmb_extend_bench(dev.h, mode, din.data_ptr(), N, dout.data_ptr(), dok.data_ptr(), 10, C.byref(ms))
        assert rc == 0, ca.lib().cmb_last_error()
        torch.cuda.synchronize()
        print(name, "mode", mode, "ms", round(ms.value, 3), "parents/s %.3g" % (N / ms.value * 1e3), flush=True)
