#!/usr/bin/env python3
"""Headline benchmark: reads/s of the search-scheme FM-index hot path (150 bp, k = 4 edit
distance, ALL mode, multiple_opt schemes with dynamic selection and dynamic partitioning —
BASELINE.json configs[2]) on a human-like synthetic reference, one process per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N ranks by itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path (k_prep -> k_parts/k_exact -> frontier search -> k_verify ->
k_traceback -> k_fmocc -> k_filter, results copied back to the host)
over this rank's read shard, reads already resident in HBM (cmb_batch_run).  Weak scaling: every
rank matches `--reads` reads against a full replica of the index; no collective on the data path
(rank 0 builds the index and broadcasts its DEVICE layout over RCCL, read shards are scattered once, both
before the timed region; after it the occurrence lists are gathered on rank 0 and the counters all-reduced,
timed apart as `result_gather_ms`).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch   # (importing torch does not touch the GPU; nothing below does before launch_ranks has had its say)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import columba_amd as ca  # noqa: E402
from columba_amd import indexbuild as ib  # noqa: E402
from columba_amd import synth  # noqa: E402
from columba_amd.dist import (allreduce_counters, broadcast_device_index, gather_occurrences,  # noqa: E402
                              scatter_reads)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
# MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues over 2 cycles on its SIMD-32; 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2
# profiles/r03_fetch_calibration.txt: scattered 16 / 32-byte loads sustain 43 - 45 G distinct 128-byte lines per second
RANDOM_LINE_RATE = 45.0e9

def log(*a):
    print(*a, file=sys.stderr, flush=True)


# kernels behind each timed group of cmb_batch_timings (rocPRIM sorts / scans between them are not attributed)
SERIAL_TABLE_STEPS = 3  # extra steps behind the timed region that the kernel table is the median of (tools/pmc_traffic.py divides by 1 + this)
GROUP_KERNELS = {"k_prep": ["k_prep", "k_match_words"], "k_partition": ["k_parts", "k_exact"],
                 "k_dfs": ["k_bfs_start", "k_bfs_pass", "k_bfs_finish", "k_hbfs"],
                 "k_verify": ["k_verify"], "k_verify_edit": ["k_verify_stage"], "k_traceback": ["k_traceback"],
                 "k_fmocc": ["k_fm_keys", "k_fm_unique", "k_fmocc"], "k_filter": ["k_pack_keys", "k_filter_segments", "k_filter_mark", "k_filter_write"]}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a launcher: start N ranks (one process per GPU, RCCL rendezvous on
    127.0.0.1) as a CHILD torch.distributed.run of this same command line and wait for it.  Called before
    anything in this process has touched the GPU; the parent only waits and passes the exit code on (rank 0 of
    the child prints the JSON line to the shared stdout)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    log(f"[bench] starting {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def pin_to_gpu_numa_node(local: int):
    """Keep this rank's host threads (the Python thread and the library's sub-batch workers, which inherit the mask) on the
    CPUs of the NUMA node its GPU hangs on: eight ranks x three workers otherwise wander across both sockets.  Best effort:
    returns a description for the JSON line, or None when the topology cannot be read."""
    try:
        prop = torch.cuda.get_device_properties(local)
        bus, dom, devn = getattr(prop, "pci_bus_id", None), getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_device_id", 0)
        if bus is None:
            return None
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{devn:02x}.0/numa_node"
        node = int(open(path).read().strip())
        if node < 0:
            return None
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        allowed = sorted(set(cpus) & os.sched_getaffinity(0))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {"numa_node": node, "cpus": len(allowed)}
    except Exception as e:  # (no sysfs, no permission: run unpinned)
        return {"error": str(e)[:80]}


def confine_cpus(n: int):
    """--cpus-per-rank N: this rank may use N CPUs (those numbered [rank N, rank N + N) of the CPUs it was given), set before the first
    GPU call; returns the list, or None without the option.  With it the NUMA pinning is skipped (CMB_BENCH_NO_PIN)."""
    if not n or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    r = int(os.environ.get("RANK", 0))
    mine = [allowed[(r * n + j) % len(allowed)] for j in range(min(n, len(allowed)))]
    os.sched_setaffinity(0, mine)
    os.environ["CMB_BENCH_NO_PIN"] = "1"
    return mine


def host_cpu_seconds() -> float:
    """user + system CPU time of this process, all threads (the library's sub-batch workers included)"""
    t = os.times()
    return t.user + t.system


def effective_cpus() -> int:
    """CPUs this process can really use: the affinity mask AND the cgroup CPU quota (the GPU boxes of this pool give a
    16-CPU share of a 256-thread host to a one-GPU job: 256 oracle threads then time-slice on 16 CPUs)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def source_digest() -> str:
    """what a committed PMC profile must have been measured on: the kernel sources of this tree"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "columba_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.startswith(("move_", "pair_")):
            continue  # translation units of their own (b-move backend, paired-end records): none of the kernels timed here
        with open(os.path.join(d, fn), "rb") as f:
            h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def move_source_digest() -> str:
    """the same for the b-move backend's translation unit: everything it is compiled from (its own files and the headers it shares with the matcher)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "columba_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.startswith("pair_") or fn == "columba_amd.hip":
            continue
        with open(os.path.join(d, fn), "rb") as f:
            h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def serial_move_table(batch, n: int = 1):
    """kernel table of a b-move batch: extra steps with the concurrent halves of the chunk run one after the other
    (CMB_MOVE_SERIAL_SUBBATCHES), so that every kernel has the device to itself; the median per kernel group"""
    os.environ["CMB_MOVE_SERIAL_SUBBATCHES"] = "1"
    runs = []
    try:
        for _ in range(n):
            batch.run()
            torch.cuda.synchronize()
            runs.append(dict(batch.timings()))
    finally:
        del os.environ["CMB_MOVE_SERIAL_SUBBATCHES"]
    return {kn: float(np.median([r.get(kn, 0.0) for r in runs])) for kn in runs[0]}


def load_rlc_traffic(reads, read_len, k):
    """HBM bytes per step of the b-move frontier kernels (k_mvs_start / k_mvs_pass / k_mvs_finish) from tools/profile_rlc.sh's PMC passes
    (profiles/*_rlc_pmc_traffic.json: the difference of a two-step and a one-step run, so that the pool-sizing warm-up cancels out), or
    (None, None) when no profile was measured on these sources and this workload"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rlc_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("reads"), w.get("read_len"), w.get("k")) != (reads, read_len, k) or d.get("kernel_src_sha") != move_source_digest():
            continue
        tot = 0.0
        for kn, e in d.get("kernels", {}).items():
            if kn.startswith(("k_mvs_pass", "k_mvs_start", "k_mvs_finish")):
                tot += 2.0 * e.get("FETCH_SIZE_KiB", 0.0) * 1024 + e.get("WRITE_SIZE_KiB", 0.0) * 1024
        if tot > 0:
            return round(tot / 1e9, 3), os.path.relpath(f, ROOT)
    return None, None


def load_counters(args, genome_bp, reads, group, names):
    """sums of SQ counters per step over the kernels of a group, from the same profile file as load_traffic (or None)"""
    import glob
    cands = [args.traffic_from] if args.traffic_from else sorted(
        glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True)
    for f in cands:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("genome_bp") != genome_bp or w.get("reads_per_gpu") != reads or w.get("k") != args.k:
            continue
        if d.get("kernel_src_sha") != source_digest() and not args.traffic_from:
            continue
        out = {}
        for kn in GROUP_KERNELS.get(group, [group]):
            e = d.get("kernels", {}).get(kn, {})
            for nm in names:
                if nm in e:
                    out[nm] = out.get(nm, 0.0) + e[nm]
        if out:
            return out
    return None


def load_traffic(args, genome_bp, reads, group):
    """HBM bytes per step of one kernel group from the PMC passes of tools/profile_round.sh (FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same command with --steps 1 --warmup 0).
    MI355X_MICROARCH.md: both are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled."""
    import glob
    cands = [args.traffic_from] if args.traffic_from else sorted(
        glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True)
    for f in cands:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("genome_bp") != genome_bp or w.get("reads_per_gpu") != reads or w.get("k") != args.k:
            continue
        if d.get("kernel_src_sha") != source_digest() and not args.traffic_from:
            continue  # measured on other kernels than the ones being timed
        tot = 0.0
        for kn in GROUP_KERNELS.get(group, [group]):
            e = d.get("kernels", {}).get(kn)
            if e:
                tot += 2.0 * e.get("FETCH_SIZE_KiB", 0.0) * 1024 + e.get("WRITE_SIZE_KiB", 0.0) * 1024
        if tot > 0:
            return round(tot / 1e9, 3), os.path.relpath(f, ROOT), (
                f"GB per step, 2 x FETCH_SIZE + WRITE_SIZE of {os.path.basename(f)} (separate rocprofv3 --pmc passes "
                "of this command on these kernel sources; gfx950 factor for FETCH_SIZE, see profiles/README.md for its "
                "calibration on scattered 16-byte loads)")
    return None, None, ("no PMC pass on record for this workload AND these kernel sources "
                        "(tools/profile_round.sh writes profiles/*_pmc_traffic.json with kernel_src_sha)")


def setup_ranks(args):
    """one process per GPU: rank / world from the launcher's environment, this rank's device, host affinity and — for N > 1 —
    the process group (RCCL, or gloo for the one-GPU rehearsal), proven to span all ranks by an all-reduce of ones"""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    # CMB_BENCH_SHARE_GPU=1 + CMB_DIST_BACKEND=gloo: all ranks on cuda:0 — a rehearsal of the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); numbers from such a run mean nothing
    if os.environ.get("CMB_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    affinity = pin_to_gpu_numa_node(local) if not os.environ.get("CMB_BENCH_NO_PIN") else None
    dist = None
    rccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CMB_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # proof that the collective library really spans all ranks: a sum of ones over device tensors (RCCL for `nccl`)
        ones = torch.ones(1, dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)
    return rank, world, local, dev, affinity, dist, rccl_ranks


def main_rlc(args):
    """BASELINE.json configs[4] on ONE GPU: a pan-genome-like text under the run-length compressed b-move index, 250 bp reads,
    k = 6 edit distance, multiple_opt schemes.  `python bench.py --config rlc [--haplotypes 64 --base-mbp 4 --snp 0.005 --reads N]`.
    A step = cmb_move_batch_run on the resident reads (prologue, frontier search, de-duplication, locate, filter, results on the
    host).  The 64-haplotype HUMAN collection of the config (200 Gbp, tables of ~130 GB) cannot be built offline; the stand-in keeps
    what the b-move kernels react to — n / r of the BWT and tables far beyond the 256 MB Infinity Cache."""
    from columba_amd import movebuild
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # (before any GPU call in this process)
    rank, world, local, dev, affinity, dist, rccl_ranks = setup_ranks(args)
    ca.lib()
    t0 = time.time()
    mv = index = None
    if rank == 0:
        text = movebuild.pangenome(int(args.base_mbp * 1e6), args.haplotypes, args.snp, seed=1)
        mv = movebuild.build_move(text, device=dev, with_locate=False)
        mv.plcp = movebuild.plcp_gpu(mv)
        torch.cuda.empty_cache()
        index = ca.MoveIndex(mv, device=local)
        log(f"[bench] b-move index of {mv.n / 1e6:.1f} Mbp ({args.haplotypes} haplotypes of {args.base_mbp} Mbp, {args.snp} SNPs): "
            f"{mv.runs_fwd} / {mv.runs_rev} runs (n/r = {mv.n / mv.runs_fwd:.1f}), {index.device_bytes() / 1e6:.0f} MB in HBM, built in "
            f"{time.time() - t0:.0f} s")
    R, L, k = args.reads, args.read_len, args.k
    broadcast_ms = None
    if world > 1:
        # replicated index (its DEVICE layout travels, one collective per array), sharded reads — as for the FM-index (§5)
        from columba_amd.dist import broadcast_device_move_index, scatter_reads
        dist.barrier()
        tb = time.perf_counter()
        index = broadcast_device_move_index(index, rank, local)
        torch.cuda.synchronize()
        dist.barrier()
        broadcast_ms = (time.perf_counter() - tb) * 1e3
        allr = None
        if rank == 0:
            log(f"[bench] device index ({index.device_bytes() / 1e9:.2f} GB) broadcast to {world - 1} peers in {broadcast_ms:.0f} ms")
            buf, _ = synth.sample_reads_fast(torch.from_numpy(mv.text[:-1]).to(dev), R * world, L, seed=3, device=dev,
                                             edit_choices=(0, 1, 2, 3, 4, 5, 6))
            allr = torch.from_numpy(buf).to(dev).reshape(world, R * L)
        buf = scatter_reads(allr, R * L, rank, world, dev)
        del allr
    else:
        buf, _ = synth.sample_reads_fast(torch.from_numpy(mv.text[:-1]).to(dev), R, L, seed=3, device=dev,
                                         edit_choices=(0, 1, 2, 3, 4, 5, 6))
    offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    torch.cuda.empty_cache()
    strategy = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    batch = ca.MoveBatch(index, strategy, k, packed=(buf, offs), kmer_size=args.kmer_size)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    sync()
    kern = {}
    tstart = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
        for kname, ms in batch.timings().items():
            kern[kname] = kern.get(kname, 0.0) + ms
    sync()
    elapsed = time.perf_counter() - tstart
    steps = max(args.steps, 1)
    per_rank_ms = [round(elapsed / steps * 1e3, 3)]
    concurrent = {kn: round(v / steps, 3) for kn, v in kern.items()}  # busy ms per step, summed over the chunk's concurrent halves
    serial = serial_move_table(batch, 1)
    occ, occ_offs, cnt = batch.results()
    total_occ = len(occ)
    if dist is not None:
        from columba_amd.dist import allreduce_counters
        cdev = "cpu" if dist.get_backend() == "gloo" else dev
        mine = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        every = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(t.item()) / steps * 1e3, 3) for t in every]
        elapsed = max(float(t.item()) for t in every)
        tt = torch.tensor([total_occ], dtype=torch.int64, device=cdev)
        dist.all_reduce(tt)
        total_occ = int(tt.item())
        cnt = allreduce_counters(cnt, dev)   # (counters of the whole job; the occurrence lists stay on their ranks)
    if rank != 0:
        batch.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    avg = serial
    dominant = max(avg, key=avg.get)
    # algorithmic bytes: one move-table row fetched = 16 B (an aligned 16-byte row here; the reference reads the
    # same 16 bytes with one unaligned 128-bit load per row access, moverepr.cpp:36-47).  Rows are counted on the device where
    # they are loaded: run scans, LF rows, fast-forward steps, run-index searches.
    alg = {"k_partition": 16.0 * (cnt["TABLE_ROWS"] - cnt["DFS_TABLE_ROWS"]), "k_dfs": 16.0 * cnt["DFS_TABLE_ROWS"]}
    per_kernel = {}
    for kname, ms in avg.items():
        gbs = alg[kname] / (ms * 1e-3) / 1e9 if ms > 0 and kname in alg else None
        per_kernel[kname] = {"ms": round(ms, 3), "algorithmic_GBps": None if gbs is None else round(gbs, 1),
                             "frac_of_hbm_peak": None if gbs is None else round(gbs / HBM_PEAK_GBS, 4)}
    achieved = per_kernel[dominant]["algorithmic_GBps"] or 0.0
    passes = None
    traffic, traffic_source = load_rlc_traffic(R, L, k) if dominant == "k_dfs" else (None, None)
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source, "kernel_src_sha": move_source_digest(),
                "traffic_note": "GB per step of the frontier kernels, 2 x FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes of this command, two-step "
                                "minus one-step run: tools/profile_rlc.sh -> profiles/*_rlc_pmc_traffic.json)",
                "avg_launch_ms": round(avg[dominant], 3),
                "timing_note": "kernel times (HIP events on the batch's streams) are those of one extra step with the chunk's halves run one after "
                               "the other; in the timed steps the halves overlap (busy ms per step there: see concurrent_ms)",
                "concurrent_ms": concurrent,
                "overlap": {"serial_sum_ms": round(sum(avg.values()), 3), "step_over_serial_sum": round(elapsed / steps * 1e3 / max(sum(avg.values()), 1e-9), 4)},
                "unit_note": "16 B per move-table row fetched (TABLE_ROWS counted on the device); "
                             f"{cnt['DFS_TABLE_ROWS'] / max(cnt['DFS_EXPANSIONS'], 1):.1f} rows per node expansion of the search",
                "per_kernel": per_kernel}
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as op
        import schemes_py as sp
        ns = min(args.cpu_sample if args.cpu_sample != 1_000_000 else 20_000, R)
        cores = effective_cpus()
        oidx = op.OracleMoveIndex(mv)
        oidx.prepare(args.kmer_size)   # (the k-mer table: index loading, not matching)
        ost = op.OracleStrategy(sp.MULTIPLE_OPT, "edit", "dynamic")
        packed = (np.ascontiguousarray(buf[:ns * L]), offs[:ns + 1].copy())
        tc = time.perf_counter()
        o_occ, o_off, o_cnt = oidx.match_batch(ost, k, threads=cores, word_size=args.kmer_size, packed=packed)
        dt = time.perf_counter() - tc
        m = int(occ_offs[ns])
        same = (len(o_occ) == m and np.array_equal(o_occ["begin"].astype(np.uint64), occ["begin"][:m]) and
                np.array_equal(o_occ["end"].astype(np.uint64), occ["end"][:m]) and
                np.array_equal(o_occ["distance"], occ["distance"][:m]))
        cpu = {"value": round(ns / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"first {ns} reads of the GPU batch, oracle/ (C++ restatement of the RUN_LENGTH_COMPRESSION flavour) with {cores} "
                         f"threads, {dt:.1f} s (reads packed and k-mer table built before the clock); occurrences identical to the GPU's: "
                         f"{bool(same)}; table rows stepped over by the reference's walks on the sample: {o_cnt['ROW_STEPS']}"}
    line = {
        "metric": "reads/sec (250bp, k=6 edit, pan-genome RLC b-move index)",
        "value": round(world * R * steps / elapsed, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "rccl_ranks": rccl_ranks, "per_rank_ms_per_step": per_rank_ms,
        "index_broadcast_ms": None if broadcast_ms is None else round(broadcast_ms, 1), "host_affinity": affinity,
        "config": {"workload": f"BASELINE configs[4] stand-in: {args.haplotypes} haplotypes x {args.base_mbp} Mbp with {args.snp} SNPs "
                               f"({mv.n / 1e6:.0f} Mbp, r = {mv.runs_fwd}, n/r = {mv.n / mv.runs_fwd:.1f}) under the b-move index, {R} x {L} bp "
                               f"reads, k={k} edit distance, ALL mode, multiple_opt schemes with dynamic selection, dynamic partitioning, "
                               f"k-mer size {args.kmer_size}",
                   "reads_per_gpu": R, "read_len": L, "k": k, "text_bp": int(mv.n), "runs": [int(mv.runs_fwd), int(mv.runs_rev)],
                   "index_bytes_hbm": index.device_bytes(), "occurrences": int(total_occ), "parallelism": f"read-shard x{world}",
                   "counters": {kk: int(cnt[kk]) for kk in ("NODE_COUNTER", "EXPANSIONS", "DFS_EXPANSIONS", "TABLE_ROWS",
                                                            "DFS_TABLE_ROWS", "TOTAL_REPORTED_POSITIONS", "SEARCH_STARTED", "MATRIX_ROWS")}},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def streaming_leg(index, strategy, k, buf, R, L, steps, rank, dev, dist, world):
    """What a host that feeds chunks sees (never `value`): every step takes FRESH reads from page-locked host memory and hands its
    results to the caller.  Two batches of R / 2 reads each, driven by two host threads: while one thread registers its next chunk
    and copies its results out (34 ms per 10^7 reads), the other one's chunk is being matched; every run
    uploads the chunk of its batch's NEXT run on a copy stream of its own (cmb_batch_stage_reads)."""
    import threading
    # chunk size: the batch's own R reads where HBM holds two such batches (each keeps its queues: ~10 KB per read), else halves
    free0 = torch.cuda.mem_get_info(dev)[0]
    H = max(R, 1)
    offs = np.arange(H + 1, dtype=np.uint64) * np.uint64(L)
    first = ca.Batch(index, strategy, k, packed=(np.ascontiguousarray(buf[:H * L]), offs))
    first.run()
    torch.cuda.synchronize()
    per_batch = free0 - torch.cuda.mem_get_info(dev)[0]
    first.close()
    if os.environ.get("CMB_BENCH_STREAM_HALVES") or 2.3 * per_batch > free0:
        H = max(R // 2, 1)
        offs = np.arange(H + 1, dtype=np.uint64) * np.uint64(L)
    # four chunks in page-locked memory (batch j alternates chunks j and 2 + j): this rank's reads, whole or in halves, and copies
    parts = [np.ascontiguousarray(buf[:H * L]), np.ascontiguousarray(buf[(R - H) * L:R * L])]
    chunks = [torch.from_numpy(parts[j % 2 if j < 2 else (j + 1) % 2].copy()).pin_memory().numpy() for j in range(4)]
    batches = [ca.Batch(index, strategy, k, packed=(chunks[j], offs)) for j in (0, 1)]
    for j, b in enumerate(batches):   # warm-up: sizes the pools, uploads the first staged chunk
        b.run()
        b.stage((chunks[2 + j], offs))
        b.run()
        b.results(reuse=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    host = [dict(stage=0.0, run=0.0, res=0.0) for _ in batches]
    errs = []
    # one chunk is matched at a time (two batches matching side by side share the device: 0.82 of the resident rate, measured);
    # what overlaps with a chunk's matching is the OTHER batch's host work.  CMB_BENCH_STREAM_OVERLAP=1: no such turn-taking.
    turn = threading.Lock() if not os.environ.get("CMB_BENCH_STREAM_OVERLAP") else None

    def worker(j):
        try:
            b = batches[j]
            for i in range(steps):
                t0 = time.perf_counter()
                b.stage((chunks[2 * (i % 2) + j], offs))   # registered; travels while this step's chunk is matched
                t1 = time.perf_counter()
                if turn is not None:
                    with turn:
                        b.run()
                else:
                    b.run()
                t2 = time.perf_counter()
                b.results(reuse=True)
                t3 = time.perf_counter()
                host[j]["stage"] += t1 - t0
                host[j]["run"] += t2 - t1
                host[j]["res"] += t3 - t2
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = time.perf_counter()
    th = [threading.Thread(target=worker, args=(j,)) for j in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - ts
    for b in batches:
        b.close()
    if errs:
        raise errs[0]
    if dist is not None:
        te = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt = float(te.item())
    n = max(steps, 1)
    return {"value": round(world * 2 * H * n / dt, 1), "unit": "reads/s", "ms_per_chunk": round(dt / (2 * n) * 1e3, 3),
            "reads_per_chunk": H, "chunks": 2 * steps, "hbm_per_batch_GB": round(per_batch / 1e9, 1),
            "host_ms_per_chunk": {"register next chunk": round(sum(h["stage"] for h in host) / (2 * n) * 1e3, 1),
                                  "run (waiting for the turn + matching + upload of the next chunk)": round(sum(h["run"] for h in host) / (2 * n) * 1e3, 1),
                                  "copy results out": round(sum(h["res"] for h in host) / (2 * n) * 1e3, 1)},
            "note": "every step matches fresh reads taken from page-locked host memory (1.5 GB per 10 M reads over PCIe, uploaded by "
                    "cmb_batch_stage_reads while the previous chunk is matched) and copies its results to the host; two batches, two "
                    "host threads taking turns on the device: one batch's host work runs beside the other's matching"}


def rlc_leg(args, dev, local):
    """BASELINE configs[4] (b-move index, 250 bp, k = 6) as a bounded leg of the DEFAULT line, after the headline's timed region: the
    stand-in of `--config rlc` (64 haplotypes x 4 Mbp, 0.5 % SNPs) with 10^6 reads, one warm-up and two timed steps, the oracle's RLC
    flavour on the first 10 000 reads beside it.  `python bench.py --config rlc` is the full-length version of this leg."""
    from columba_amd import movebuild
    t0 = time.time()
    text = movebuild.pangenome(int(args.base_mbp * 1e6), args.haplotypes, args.snp, seed=1)
    mv = movebuild.build_move(text, device=dev, with_locate=False)
    mv.plcp = movebuild.plcp_gpu(mv)
    torch.cuda.empty_cache()
    index = ca.MoveIndex(mv, device=local)
    built = time.time() - t0
    R, L, k, steps = 1_000_000, 250, 6, 2
    buf, _ = synth.sample_reads_fast(torch.from_numpy(mv.text[:-1]).to(dev), R, L, seed=3, device=dev, edit_choices=(0, 1, 2, 3, 4, 5, 6))
    offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    torch.cuda.empty_cache()
    strategy = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    batch = ca.MoveBatch(index, strategy, k, packed=(buf, offs), kmer_size=10)
    batch.run()
    torch.cuda.synchronize()
    kern = {}
    ts = time.perf_counter()
    for _ in range(steps):
        batch.run()
        for kname, ms in batch.timings().items():
            kern[kname] = kern.get(kname, 0.0) + ms
    torch.cuda.synchronize()
    dt = time.perf_counter() - ts
    concurrent = {kn: round(v / steps, 3) for kn, v in kern.items()}
    avg = serial_move_table(batch, 1)  # (the chunk's halves one after the other: every kernel has the device to itself)
    occ, occ_offs, cnt = batch.results()
    dominant = max(avg, key=avg.get)
    alg = {"k_partition": 16.0 * (cnt["TABLE_ROWS"] - cnt["DFS_TABLE_ROWS"]), "k_dfs": 16.0 * cnt["DFS_TABLE_ROWS"]}
    achieved = alg.get(dominant, 0.0) / (avg[dominant] * 1e-3) / 1e9 if avg[dominant] > 0 else 0.0
    cpu = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as op
        import schemes_py as sp
        ns, cores = 10_000, effective_cpus()
        oidx = op.OracleMoveIndex(mv)
        oidx.prepare(10)
        ost = op.OracleStrategy(sp.MULTIPLE_OPT, "edit", "dynamic")
        packed = (np.ascontiguousarray(buf[:ns * L]), offs[:ns + 1].copy())
        tc = time.perf_counter()
        o_occ, o_off, o_cnt = oidx.match_batch(ost, k, threads=cores, word_size=10, packed=packed)
        dc = time.perf_counter() - tc
        m = int(occ_offs[ns])
        same = (len(o_occ) == m and np.array_equal(o_occ["begin"].astype(np.uint64), occ["begin"][:m]) and
                np.array_equal(o_occ["end"].astype(np.uint64), occ["end"][:m]) and np.array_equal(o_occ["distance"], occ["distance"][:m]))
        cpu = {"value": round(ns / dc, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"first {ns} reads of the leg's batch, oracle/ (RLC flavour) with {cores} threads, {dc:.1f} s; occurrences identical to the GPU's: {bool(same)}"}
    batch.close()
    return {"metric": "reads/sec (250bp, k=6 edit, pan-genome RLC b-move index)", "value": round(R * steps / dt, 1), "unit": "reads/s",
            "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": 1, "reads": R,
            "config": {"workload": f"BASELINE configs[4] stand-in: {args.haplotypes} haplotypes x {args.base_mbp} Mbp with {args.snp} SNPs "
                                   f"({mv.n / 1e6:.0f} Mbp, r = {mv.runs_fwd}, n/r = {mv.n / mv.runs_fwd:.1f}; index built in {built:.0f} s), "
                                   f"{R} x {L} bp reads, k={k} edit distance, ALL mode, multiple_opt, dynamic partitioning",
                       "index_bytes_hbm": index.device_bytes(), "occurrences": int(len(occ))},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_rlc_traffic(R, L, k)[0] if dominant == "k_dfs" else None,
                         "traffic_source": load_rlc_traffic(R, L, k)[1] if dominant == "k_dfs" else None, "kernel_src_sha": move_source_digest(),
                         "avg_launch_ms": round(avg[dominant], 3),
                         "unit_note": f"16 B per move-table row fetched; {cnt['DFS_TABLE_ROWS'] / max(cnt['DFS_EXPANSIONS'], 1):.1f} rows per node expansion",
                         "per_kernel_ms": {kn: round(v, 3) for kn, v in avg.items()}, "concurrent_ms": concurrent,
                         "overlap": {"serial_sum_ms": round(sum(avg.values()), 3),
                                     "step_over_serial_sum": round(dt / steps * 1e3 / max(sum(avg.values()), 1e-9), 4)}},
            "cpu_baseline": cpu}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["fm", "rlc"], default="fm",
                    help="fm: BASELINE configs[2] on the FM-index (the headline line, default); rlc: configs[4] on the b-move index")
    ap.add_argument("--haplotypes", type=int, default=64)
    ap.add_argument("--base-mbp", type=float, default=4.0)
    ap.add_argument("--snp", type=float, default=0.005)
    ap.add_argument("--kmer-size", type=int, default=10)
    ap.add_argument("--sparseness", type=int, default=4, help="suffix-array sparseness of the FM-index (the reference's -s option; default 4)")
    ap.add_argument("--in-text-switch", type=int, default=4, help="range width at which the search switches to in-text verification (the reference's -i option; default 4)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("CMB_BENCH_GENOME_MBP", 3000)),
                    help="length of the synthetic human-like reference (GRCh38 scale by default)")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("CMB_BENCH_READS", 10_000_000)),
                    help="reads per GPU and step (weak scaling); BASELINE.json configs[2]: 10 M x 150 bp")
    ap.add_argument("--traffic-from", default=None,
                    help="JSON written by tools/profile_round.sh (PMC passes of this same command); default: the "
                         "newest profiles/*_pmc_traffic.json measured on the same workload")
    ap.add_argument("--total-reads", type=int, default=0,
                    help="STRONG scaling: this many reads for the whole job, split evenly over the ranks (BASELINE.json configs[3]: "
                         "100 M reads over 8 GPUs = 12.5 M per GPU); default 0 = weak scaling with --reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads timed on the CPU oracle (bounded sample)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the (untimed) gather of the results on rank 0")
    ap.add_argument("--include-upload", action="store_true", help="kept for older command lines: the streaming leg is part of the default line")
    ap.add_argument("--no-streaming", action="store_true",
                    help="skip the `streaming` leg (steps that take FRESH reads from page-locked host memory and hand their results to the "
                         "host; reported beside `value`, never as `value`)")
    ap.add_argument("--cpus-per-rank", type=int, default=0,
                    help="confine every rank to this many CPUs (set before anything touches the GPU): a rehearsal of a CPU-starved 8-rank "
                         "node on a one-GPU box; the line's host_cpu field shows what a step costs the host")
    ap.add_argument("--no-rlc", action="store_true",
                    help="skip the bounded BASELINE configs[4] leg (`rlc`: b-move index, 250 bp, k = 6) of the default one-GPU line")
    args = ap.parse_args()
    if args.config == "rlc":
        if "--reads" not in " ".join(sys.argv):
            args.reads = 1_000_000
        if "--read-len" not in " ".join(sys.argv):
            args.read_len = 250
        if "--k" not in sys.argv:
            args.k = 6
        if "--steps" not in " ".join(sys.argv):
            args.steps = 3
        return main_rlc(args)

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # (before any GPU call in this process)
    confined = confine_cpus(args.cpus_per_rank)  # (before any GPU call as well: the library's worker threads inherit the mask)
    rank, world, local, dev, affinity, dist, rccl_ranks = setup_ranks(args)
    if confined is not None:
        affinity = {"confined_to": confined}
    if args.total_reads:
        args.reads = args.total_reads // world

    ca.lib()  # fail loudly if the HIP extension is missing
    n = int(args.genome_mbp * 1e6)
    t0 = time.time()
    ix = None
    if rank == 0:
        g, starts = synth.genome_human_like(n, seed=2025, device=dev)
        ix = ib.build_index(g, sparseness=args.sparseness, seq_starts=starts, device=dev, with_bwt=(world == 1 and not args.no_cpu_baseline))
        del g
        torch.cuda.empty_cache()
        log(f"[bench] index for {n / 1e6:.0f} Mbp built in {time.time() - t0:.1f} s "
            f"({ix.nbytes() / 1e9:.2f} GB host arrays)")
    fm_kmer = args.kmer_size if "--kmer-size" in " ".join(sys.argv) else 10   # (the reference's default -K 10)
    index = ca.Index(ix, in_text_switch=args.in_text_switch, kmer_size=fm_kmer, device=local) if rank == 0 else None
    broadcast_ms = None
    if world > 1:
        dist.barrier()
        tb = time.time()
        index = broadcast_device_index(index, rank, local)  # the device layout itself, one collective per array
        torch.cuda.synchronize()
        dist.barrier()
        broadcast_ms = (time.time() - tb) * 1e3
        if rank == 0:
            log(f"[bench] device index ({index.device_bytes() / 1e9:.2f} GB) broadcast to {world - 1} peers in "
                f"{broadcast_ms / 1e3:.2f} s")
    strategy = ca.SearchStrategy("multiple_opt", "edit", "dynamic")

    # reads: rank 0 samples the global batch on its GPU and scatters equal shards
    R, L = args.reads, args.read_len
    t1 = time.time()
    if world > 1:
        allr = None
        if rank == 0:
            buf, _ = synth.sample_reads_fast(torch.from_numpy(ix.text[:-1]).to(dev), R * world, L, seed=3, device=dev)
            allr = torch.from_numpy(buf).to(dev).reshape(world, R * L)
        buf = scatter_reads(allr, R * L, rank, world, dev)
        del allr
    else:
        buf, _ = synth.sample_reads_fast(torch.from_numpy(ix.text[:-1]).to(dev), R, L, seed=3, device=dev)
    offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    torch.cuda.empty_cache()
    if rank == 0:
        log(f"[bench] {R} reads/GPU sampled in {time.time() - t1:.1f} s")
    batch = ca.Batch(index, strategy, args.k, packed=(buf, offs))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    sync()
    kern = {}
    cpu0 = host_cpu_seconds()
    tstart = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
        for kname, ms in batch.timings().items():
            kern[kname] = kern.get(kname, 0.0) + ms
    sync()
    elapsed = time.perf_counter() - tstart
    host_cpu_ms = [round((host_cpu_seconds() - cpu0) / max(args.steps, 1) * 1e3, 2)] # CPU time the host spent per step, this rank
    per_rank_ms = [round(elapsed / max(args.steps, 1) * 1e3, 3)]
    if dist is not None:
        cdev = "cpu" if dist.get_backend() == "gloo" else dev
        mine = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        every = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(t.item()) / max(args.steps, 1) * 1e3, 3) for t in every]
        mine = torch.tensor([host_cpu_ms[0]], dtype=torch.float64, device=cdev)
        dist.all_gather(every, mine)
        host_cpu_ms = [round(float(t.item()), 2) for t in every]
        te = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    # kernel table: one extra step with the sub-batches of the batch run one after the other, so that every kernel
    # has the device to itself (in the timed steps above they overlap, which is where 15 % of the throughput
    # comes from, but a kernel's duration then depends on what happens to run beside it)
    os.environ["CMB_SERIAL_SUBBATCHES"] = "1"
    serial_runs = []
    try:
        for _ in range(SERIAL_TABLE_STEPS):  # (three such steps, the median per kernel group: one sample moved by 3 ms between otherwise identical runs)
            batch.run()
            torch.cuda.synchronize()
            serial_runs.append(dict(batch.timings()))
    finally:
        del os.environ["CMB_SERIAL_SUBBATCHES"]
    kern_serial = {kn: float(np.median([r.get(kn, 0.0) for r in serial_runs])) for kn in serial_runs[0]}
    occ, occ_offs, cnt = batch.results()
    total_occ = len(occ)
    gather_ms = None
    if dist is not None:
        tt = torch.tensor([total_occ], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt)
        total_occ = int(tt.item())
        if not args.no_gather:
            # result path of the sharded job: gatherv of the occurrence lists on rank 0 + all-reduce of the counters
            # (SURVEY.md §8e).  Not part of a step: one rank's host link would carry every rank's results.
            sync()
            tg = time.perf_counter()
            g_occ, g_offs = gather_occurrences(occ, occ_offs, rank, world, dev)
            cnt_all = allreduce_counters(cnt, dev)
            sync()
            gather_ms = (time.perf_counter() - tg) * 1e3
            if rank == 0:
                assert len(g_occ) == total_occ and len(g_offs) == world * R + 1 and int(g_offs[-1]) == total_occ
                assert np.array_equal(g_occ[:len(occ)], occ)
                log(f"[bench] gathered {len(g_occ)} occurrences of {world * R} reads on rank 0 in {gather_ms:.1f} ms; "
                    f"NODE_COUNTER of the job {cnt_all['NODE_COUNTER']}")
                del g_occ, g_offs

    streaming = None
    if rank == 0:
        steps = max(args.steps, 1)
        value = world * R * steps / elapsed
        avg_concurrent = {k: v / steps for k, v in kern.items()}
        avg = kern_serial
        dominant = max(avg, key=avg.get)
        # algorithmic bytes per step of every kernel group:
        #   k_partition (k_parts + k_exact) / k_dfs (frontier search): 192 B per node expansion
        #       (2 positions x (64 B counts line + 32 B bit group) in the reference layout)
        #   k_verify (locate, key sort): 112 B per LF step + 28 B per located row
        #   k_verify_edit (matrix stages) / k_traceback: 1 B per text character
        alg = {"k_partition": 192.0 * (cnt["EXPANSIONS"] - cnt["DFS_EXPANSIONS"]),
               "k_dfs": 192.0 * cnt["DFS_EXPANSIONS"],
               "k_verify": 112.0 * cnt["LF_STEPS"] + 28.0 * cnt["LOCATED_ROWS"] +
                           (0.0 if "k_verify_edit" in avg else 1.0 * cnt["TEXT_BYTES"]),
               "k_verify_edit": 1.0 * cnt["TEXT_BYTES"],
               "k_traceback": 1.0 * cnt["TEXT_BYTES"]}
        per_kernel = {}
        for kname, ms in avg.items():
            gbs = alg.get(kname, 0.0) / (ms * 1e-3) / 1e9 if ms > 0 and kname in alg else None
            per_kernel[kname] = {"ms": round(ms, 3),
                                 "algorithmic_GBps": None if gbs is None else round(gbs, 1),
                                 "frac_of_hbm_peak": None if gbs is None else round(gbs / HBM_PEAK_GBS, 4)}
        # the matrix kernels are bound by VALU issue, not by bytes: their fraction of THAT peak, where a PMC pass
        # of these kernel sources is on record (SQ_INSTS_VALU = wave-instructions)
        valu = None
        for kname in ("k_verify_edit", "k_traceback"):
            c = load_counters(args, n, R, kname, ["SQ_INSTS_VALU"])
            if c and kname in avg and avg[kname] > 0:
                rate = c["SQ_INSTS_VALU"] / (avg[kname] * 1e-3)
                per_kernel[kname]["valu_wave_insts_per_step"] = c["SQ_INSTS_VALU"]
                per_kernel[kname]["frac_of_valu_peak"] = round(rate / VALU_PEAK_WAVE_INSTS, 4)
                if valu is None:
                    valu = {"bound": "valu", "kernel": kname, "achieved": round(rate / 1e9, 1), "peak": round(VALU_PEAK_WAVE_INSTS / 1e9, 1),
                            "unit": "G wave-instructions/s", "frac": round(rate / VALU_PEAK_WAVE_INSTS, 4),
                            "note": "SQ_INSTS_VALU per step (rocprofv3 --pmc pass of these kernel sources) / HIP-event time; peak = 256 CUs x 4 SIMD-32 "
                                    "x 2.4 GHz / 2 cycles per wave64 instruction"}
        achieved = per_kernel[dominant]["algorithmic_GBps"] or 0.0
        traffic, traffic_source, traffic_note = load_traffic(args, n, R, dominant)
        # lines, not bytes, are what scattered rank fetches cost (profiles/r03_fetch_calibration.txt): HBM lines moved per second
        # by the dominant kernel against the rate at which the chip serves random 128-byte lines
        line_rate = None
        if traffic:
            lr = traffic * 1e9 / 128.0 / (avg[dominant] * 1e-3)
            line_rate = {"achieved_Glines_s": round(lr / 1e9, 2), "ceiling_Glines_s": RANDOM_LINE_RATE / 1e9, "frac": round(lr / RANDOM_LINE_RATE, 4),
                         "lines_per_expansion": round(traffic * 1e9 / 128.0 / max(cnt["DFS_EXPANSIONS"] if dominant == "k_dfs" else
                                                                                   cnt["EXPANSIONS"] - cnt["DFS_EXPANSIONS"], 1), 3)
                         if dominant in ("k_dfs", "k_partition") else None}
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_source, "kernel_src_sha": source_digest(),
                    "traffic_note": traffic_note, "avg_launch_ms": round(avg[dominant], 3),
                    "timing_note": "kernel times (HIP events on the batch's streams) are the medians of three extra steps with the "
                                   "batch's sub-batches run one after the other; in the timed steps the sub-batches "
                                   "overlap (busy ms per step there: see concurrent_ms)",
                    "concurrent_ms": {k: round(v, 3) for k, v in avg_concurrent.items()},
                    # the step against the sum of its kernel groups run one after the other (what the concurrent sub-batches hide)
                    "overlap": {"serial_sum_ms": round(sum(avg.values()), 3), "step_over_serial_sum": round(elapsed / steps * 1e3 / max(sum(avg.values()), 1e-9), 4)},
                    "line_rate": line_rate, "valu": valu,
                    "per_kernel": per_kernel}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as op
            import schemes_py as sp
            ns = min(args.cpu_sample, R)
            oidx = op.OracleIndex(ix, kmer_size=fm_kmer, switch_point=args.in_text_switch)
            ost = op.OracleStrategy(sp.MULTIPLE_OPT, "edit", "dynamic")
            cores = effective_cpus()
            packed = (np.ascontiguousarray(buf[:ns * L]), offs[:ns + 1].copy())  # (packed before the clock starts)
            tc = time.perf_counter()
            o_occ, o_off, _ = op.match_batch(oidx, ost, args.k, threads=cores, packed=packed)
            dt = time.perf_counter() - tc
            same = (len(o_occ) == int(occ_offs[ns]) and
                    np.array_equal(o_occ["begin"], occ["begin"][:len(o_occ)]) and
                    np.array_equal(o_occ["end"], occ["end"][:len(o_occ)]) and
                    np.array_equal(o_occ["distance"], occ["distance"][:len(o_occ)]))
            cpu = {"value": round(ns / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
                   "sample": f"first {ns} reads of the GPU batch, oracle/ (C++ restatement) with {cores} threads = the CPUs this job may "
                             f"use ({os.cpu_count()} hardware threads on the host), {dt:.1f} s, reads packed before the clock; "
                             f"{ns / dt / cores:.0f} reads/s per thread (BASELINE.md: Columba itself 21 k reads/s per thread on a cache-resident "
                             f"16 Mbp index, probe of the survey); occurrences identical to the GPU's: {bool(same)}"}
    # ---- legs after the timed region (they leave `value` untouched): the resident batch gives its memory back first
    batch.close()
    del occ, occ_offs
    torch.cuda.empty_cache()
    rlc = None
    if not args.no_streaming:
        streaming = streaming_leg(index, strategy, args.k, buf, R, L, min(max(args.steps, 1), 8), rank, dev, dist, world)
        if streaming is not None and rank == 0:
            streaming["frac_of_resident"] = round(streaming["value"] / (world * R * max(args.steps, 1) / elapsed), 4)
    if rank == 0 and world == 1 and not args.no_rlc:
        torch.cuda.empty_cache()
        try:
            rlc = rlc_leg(args, dev, local)
        except Exception as e:  # noqa: BLE001  (the headline line must not depend on the leg)
            rlc = {"error": str(e)[:300]}
    if rank == 0:
        line = {
            "metric": "reads/sec (150bp, k=4 edit, human ref)",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.total_reads else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "rccl_ranks": rccl_ranks, "per_rank_ms_per_step": per_rank_ms, "index_broadcast_ms": None if broadcast_ms is None else round(broadcast_ms, 1),
            "host_affinity": affinity,
            # what a step costs the HOST: CPU time (all threads of the rank: three sub-batch workers + the Python thread) per step and
            # its share of the step's wall time — the budget eight ranks must find on one node (the GPU boxes give a one-GPU job 16 CPUs)
            "host_cpu": {"cpu_ms_per_step": host_cpu_ms, "cpus_busy": [round(c / max(elapsed / steps * 1e3, 1e-9), 2) for c in host_cpu_ms],
                         "cpus_available": effective_cpus()},
            "config": {"workload": f"synthetic human-like reference {n / 1e6:.0f} Mbp (GRCh38 is not available "
                                   f"offline), {R} x {L} bp reads per GPU, k={args.k} edit distance, ALL mode, "
                                   "multiple_opt schemes with dynamic selection, dynamic partitioning, "
                                   f"in-text switch {args.in_text_switch}, SA sparseness {args.sparseness}" + ("" if fm_kmer == 10 else f", k-mer table of {fm_kmer}-mers (-K {fm_kmer}; the reference's default is 10)"),
                       "reads_per_gpu": R, "read_len": L, "k": args.k, "genome_bp": n,
                       "index_bytes_hbm": index.device_bytes(), "parallelism": f"read-shard x{world}",
                       "occurrences": total_occ, "result_gather_ms": None if gather_ms is None else round(gather_ms, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "streaming": streaming, "rlc": rlc,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
