"""Development helper: S concurrent sub-batches (one host thread + HIP stream each) over the same index."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
n = int(sys.argv[1]); nreads = int(sys.argv[2])
g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
del g
torch.cuda.empty_cache()
dev = ca.Index(ix)
buf, offs = synth.sample_reads_fast(ix.text[:-1], nreads, 150, seed=3, device="cuda")
torch.cuda.empty_cache()
st = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
for S in [int(x) for x in sys.argv[3:]]:
    per = nreads // S
    bs = []
    for j in range(S):
        o = np.arange(per + 1, dtype=np.uint64) * np.uint64(150)
        bs.append(ca.Batch(dev, st, 4, packed=(buf[j * per * 150:(j + 1) * per * 150], o)))
    def step():
        th = [threading.Thread(target=b.run) for b in bs]
        [t.start() for t in th]; [t.join() for t in th]
    step()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print("streams", S, "reads/s", round(per * S / dt), "ms/step", round(dt * 1e3, 2), flush=True)
    for b in bs: b.close()
