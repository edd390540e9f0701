"""Pairing of two mates' single-end occurrences in ALL mode (cmb_pair_sam) against a restatement of the
reference's logic in Python — SearchStrategy::pairSingleEndedMatchesAll (searchstrategy.cpp:1345-1399), processComb*All
(searchstrategy.h:753-861), pairOccurrences (:1281-1344), pairDiscordantly / addDiscPairs / addUnpairedMatches /
addOneUnmapped / addBothUnmapped (:1401-1646, searchstrategy.h:1178-1247), generateSAMPairedEnd (:1904-1970) and the order of
OutputWriter::writeChunks (fastq.cpp:662-702) — with the single records taken from the pinned functions (cmb_sam_pe,
cmb_sam_unpaired, cmb_sam_unmapped_pe: 90 vectors of the reference's own code, test_output_records.py) — and against brute force
for the concordant pairs.  searchstrategy.cpp itself cannot be built here (parallel_hashmap): this layer is parity-unpinned.
Host code only: runs without a GPU."""
import numpy as np
import pytest

import columba_amd as ca

SEQ_NAMES = ["chr1", "chr2_alt", "seqC"]
SEQ_START = [0, 40_000, 90_000]
CIGS = ["50M", "20M1I29M", "10M1D40M", "49M1I"]


def _occ_sort_key(o):  # TextOcc::operator< (indexhelpers.h:776-792)
    return (o[3], o[4], o[2] - o[1])


def _reference_pairing(r1, r2, orientation, max_frag, min_frag, disc, unmapped):
    R = [r1, r2]

    def hit(o):
        return (SEQ_NAMES[o[0]], o[1], o[4], bool(o[5]), o[6])

    def seq_of(r, o):
        return R[r][2] if o[5] else R[r][1]

    def qual_of(r, o):
        return R[r][4] if o[5] else R[r][3]

    st = [[sorted([o for o in R[r][5] if o[5] == s], key=_occ_sort_key) for s in (0, 1)] for r in (0, 1)]
    fw1, rc1, fw2, rc2 = [(0, o) for o in st[0][0]], [(0, o) for o in st[0][1]], [(1, o) for o in st[1][0]], [(1, o) for o in st[1][1]]
    pairs = []  # [up (r, o) or None, down or None, frag, distance, discordant, up line, down line]

    def pair_occurrences(U, D):  # searchstrategy.cpp:1281-1344
        if not U or not D:
            return
        for ur, u in U:
            k = 0
            while k < len(D) and D[k][1][3] < u[3]:
                k += 1
            for dr, d in D[k:]:
                frag = d[3] + (d[2] - d[1]) - u[3]
                if min_frag <= frag <= max_frag:
                    if u[0] is None:
                        break
                    if d[0] is None or d[0] != u[0]:
                        continue
                    pairs.append([(ur, u), (dr, d), d[2] - u[1], u[4] + d[4], False, "", ""])
                elif frag > max_frag:
                    break

    combos = {ca.ORIENTATION_FR: ((fw1, rc2), (fw2, rc1)), ca.ORIENTATION_FF: ((fw1, fw2), (rc2, rc1)),
              ca.ORIENTATION_RF: ((rc1, fw2), (rc2, fw1))}[orientation]
    for U, D in combos:
        pair_occurrences(U, D)
    unpaired = []

    def unmapped_line(r, mate_mapped, mate_rev):
        return ca.sam_unmapped_pe(R[r][0], R[r][1], R[r][3], r == 0, mate_mapped, mate_rev)

    def add_unpaired():  # searchstrategy.h:1216-1230 + searchstrategy.cpp:1401-1462 (stable order among equal distances)
        unpaired.clear()
        for r, (fw, rc) in enumerate(((fw1, rc1), (fw2, rc2))):
            temp = [o for _, o in fw + rc if o[0] is not None]
            fw.clear()
            rc.clear()
            if not temp:
                if unmapped:
                    unpaired.append(unmapped_line(r, False, False))
                continue
            temp.sort(key=lambda o: o[4])
            best = temp[0][4]
            cnt = sum(1 for o in temp if o[4] == best)
            for i, o in enumerate(temp):
                unpaired.append(ca.sam_unpaired(R[r][0], hit(o), r == 0, cnt, best, i == 0, seq_of(r, o), qual_of(r, o)))

    if not pairs:  # pairDiscordantly (searchstrategy.cpp:1586-1646)
        m1, m2 = len(fw1) + len(rc1), len(fw2) + len(rc2)
        done = False
        if disc and m1 and m2:
            if m1 * m2 > 10000:
                add_unpaired()
            else:
                for A in (fw1, rc1):
                    for ar, a in A:
                        for B in (fw2, rc2):
                            for br, b in B:
                                if a[0] is None or b[0] is None:
                                    continue
                                a_up = a[1] < b[1]
                                frag = (b[2] - a[1] if a_up else a[2] - b[1]) if a[0] == b[0] else 0
                                pairs.append([(ar, a) if a_up else (br, b), (br, b) if a_up else (ar, a), frag, a[4] + b[4], True, "", ""])
                done = bool(pairs)
        if not done:
            if m1 and m2:
                add_unpaired()
            elif not m1 and not m2:
                if unmapped:
                    pairs.append([None, None, 0, 0, False, unmapped_line(0, False, False), unmapped_line(1, False, False)])
            else:
                mr = 0 if m1 else 1
                for _, o in (fw1 + rc1) if m1 else (fw2 + rc2):
                    if o[0] is None:
                        continue
                    pairs.append([(mr, o), None, 0, o[4], False, "", unmapped_line(1 - mr, True, bool(o[5]))])
                if not pairs and unmapped:
                    pairs.append([None, None, 0, 0, False, unmapped_line(0, False, False), unmapped_line(1, False, False)])
    n_pairs = 0
    if pairs:  # generateSAMPairedEnd (searchstrategy.cpp:1904-1970)
        mi = min(range(len(pairs)), key=lambda i: (pairs[i][3], i))
        best = pairs[mi][3]
        n_pairs = sum(1 for p in pairs if p[3] == best)
        pairs[0], pairs[mi] = pairs[mi], pairs[0]
        for i, p in enumerate(pairs):
            for side in (0, 1):
                me, mate = p[side], p[1 - side]
                if me is None:
                    continue
                r, o = me
                p[5 + side] = ca.sam_pe(R[r][0], hit(o), r == 0, hit(mate[1]) if mate is not None else None, n_pairs, best, p[2], p[4],
                                        i == 0, seq_of(r, o), qual_of(r, o))
    mapped = bool(pairs) and pairs[0][0] is not None and pairs[0][1] is not None
    half = not mapped and bool(pairs) and (pairs[0][0] is not None or pairs[0][1] is not None)
    text = ""
    for i, p in enumerate(pairs):
        text += p[5]
        if not half or i == 0:
            text += p[6]
    return text + "".join(unpaired), (len(pairs) if mapped else 0), pairs


def _random_read(rng, name, n_occ, assigned_frac=0.95, cluster=None):
    ln = 50
    seq = "".join(rng.choice(list("ACGT"), ln))
    _, cs, rc, rq = ca.read_prepare("@" + name, seq, "I" * ln)
    occs = []
    for _ in range(n_occ):
        sid = int(rng.integers(0, 3))
        pos = int(rng.integers(0, 3000)) if cluster is None else int(cluster + rng.integers(-400, 400))
        pos = max(0, pos)
        cig = CIGS[int(rng.integers(0, len(CIGS)))]
        ops = ca.parse_cigar(cig)
        width = sum(int(op) >> 2 for op in ops if (int(op) & 3) != 1)  # reference characters: M and D
        occs.append((sid if rng.random() < assigned_frac else None, pos, pos + width, SEQ_START[sid] + pos, int(rng.integers(0, 5)),
                     int(rng.integers(0, 2)), ops))
    return (name, cs, rc, "I" * ln, "I" * ln, occs)


@pytest.mark.parametrize("orientation", [ca.ORIENTATION_FR, ca.ORIENTATION_RF, ca.ORIENTATION_FF])
def test_pairing_equals_the_restated_reference_logic(orientation):
    rng = np.random.default_rng(100 + orientation)
    kinds = {"pairs": 0, "disc": 0, "unpaired": 0, "half": 0, "none": 0}
    for trial in range(300):
        n1 = int(rng.choice([0, 0, 1, 2, 5, 12]))
        n2 = int(rng.choice([0, 1, 1, 3, 8]))
        centre = int(rng.integers(500, 2500)) if rng.random() < 0.7 else None
        r1 = _random_read(rng, f"p{trial}/1", n1, cluster=centre)
        r2 = _random_read(rng, f"p{trial}/2", n2, cluster=centre)
        disc = bool(rng.integers(0, 2))
        unmapped = bool(rng.integers(0, 4))
        max_frag, min_frag = int(rng.choice([300, 600, 1000])), int(rng.choice([0, 0, 80]))
        want, want_n, pairs = _reference_pairing(r1, r2, orientation, max_frag, min_frag, disc, unmapped)
        got, got_n = ca.pair_sam(r1, r2, SEQ_NAMES, orientation, max_frag, min_frag, disc, unmapped)
        assert got == want, (trial, got, want)
        assert got_n == want_n
        if pairs and pairs[0][0] is not None and pairs[0][1] is not None:
            kinds["disc" if pairs[0][4] else "pairs"] += 1
        elif pairs and (pairs[0][0] is not None or pairs[0][1] is not None):
            kinds["half"] += 1
        elif pairs:
            kinds["none"] += 1
        elif got:
            kinds["unpaired"] += 1
    assert all(v >= 5 for v in kinds.values()), kinds


def test_concordant_pairs_by_brute_force():
    """FR: every (forward occurrence of one mate, reverse-complement occurrence of the other) on one sequence whose fragment
    lies in [min, max] and whose downstream begin is not before the upstream begin is reported exactly once as a proper pair"""
    rng = np.random.default_rng(5)
    for trial in range(60):
        r1 = _random_read(rng, f"b{trial}/1", 10, assigned_frac=1.0, cluster=1500)
        r2 = _random_read(rng, f"b{trial}/2", 10, assigned_frac=1.0, cluster=1500)
        text, n = ca.pair_sam(r1, r2, SEQ_NAMES, ca.ORIENTATION_FR, 700, 100, False, True)
        want = set()
        for U, D in ((r1[5], r2[5]), (r2[5], r1[5])):
            for u in U:
                for d in D:
                    if u[5] == 0 and d[5] == 1 and u[0] == d[0] and d[3] >= u[3] and 100 <= d[3] + (d[2] - d[1]) - u[3] <= 700:
                        want.add((SEQ_NAMES[u[0]], u[1] + 1, d[1] + 1, d[2] - u[1]))
        lines = [ln.split("\t") for ln in text.splitlines()]
        got = set()
        for f in lines:
            flag = int(f[1])
            if flag & 2 and not flag & 16:  # the forward mate of a proper pair: upstream
                got.add((f[2], int(f[3]), int(f[7]), abs(int(f[8]))))
        if want:
            assert got == want and n == len([f for f in lines if int(f[1]) & 2]) // 2
        else:
            assert not any(int(f[1]) & 2 for f in lines) and n == 0


def test_too_many_discordant_candidates_follow_the_reference_to_the_letter():
    """more than 10 000 discordant candidates: the reference collects the unpaired records, then collects them again from the
    emptied lists (searchstrategy.cpp:1619-1638): what is left are two unmapped records"""
    rng = np.random.default_rng(9)
    r1 = _random_read(rng, "q/1", 120, assigned_frac=1.0)
    r2 = _random_read(rng, "q/2", 120, assigned_frac=1.0)
    # no concordant pair: all of read 1 on chr1, all of read 2 on seqC
    r1 = r1[:5] + ([(0,) + o[1:3] + (SEQ_START[0] + o[1],) + o[4:] for o in r1[5]],)
    r2 = r2[:5] + ([(2,) + o[1:3] + (SEQ_START[2] + o[1],) + o[4:] for o in r2[5]],)
    text, n = ca.pair_sam(r1, r2, SEQ_NAMES, ca.ORIENTATION_FR, 500, 0, True, True)
    lines = text.splitlines()
    assert n == 0 and len(lines) == 2 and [int(ln.split("\t")[1]) for ln in lines] == [77, 141]
    want, _, _ = _reference_pairing(r1, r2, ca.ORIENTATION_FR, 500, 0, True, True)
    assert text == want


def _infer_reference(samples):
    """parallel.cpp:329-360 + :402-466 in numpy with the reference's types (length_t fragment sizes, float statistics)"""
    frag, ori = [], []
    for b1, e1, s1, b2, e2, s2 in samples:
        first = b1 < b2
        frag.append(e2 - b1 if first else e1 - b2)
        ori.append(ca.ORIENTATION_FF if bool(s1) == bool(s2) else (ca.ORIENTATION_RF if first == bool(s1) else ca.ORIENTATION_FR))

    def median(v):
        v = sorted(v)
        return (v[len(v) // 2 - 1] + v[len(v) // 2]) // 2 if len(v) % 2 == 0 else v[len(v) // 2]

    def avg(v):
        return np.float32(np.float64(sum(v)) / len(v))

    def sd(v, m):
        acc = np.float32(0)
        for x in v:
            d = np.float32(x) - m
            acc = np.float32(acc + d * d)
        return np.float32(np.sqrt(acc / np.float32(len(v) - 1)))

    med = median(frag)
    mad = median([abs(f - med) for f in frag])
    kept = [f for f in frag if abs(f - med) < 6 * mad]
    with np.errstate(all="ignore"):
        mean = avg(kept) if kept else np.float32("nan")
        dev = sd(kept, mean) if kept else np.float32("nan")
        if mean == 0 or dev == 0 or np.isnan(mean) or np.isnan(dev):
            mean = avg(frag)
            dev = sd(frag, mean)
    max_dev = int(np.float32(6) * dev)
    cnt = [ori.count(o) for o in (ca.ORIENTATION_FR, ca.ORIENTATION_RF, ca.ORIENTATION_FF)]
    o = ca.ORIENTATION_FR if cnt[0] >= cnt[1] and cnt[0] >= cnt[2] else (ca.ORIENTATION_RF if cnt[1] >= cnt[2] else ca.ORIENTATION_FF)
    return o, int(mean + np.float32(max_dev)), int(mean - np.float32(max_dev)) if mean > max_dev else 0, float(mean), float(dev)


def test_inference_of_the_paired_end_parameters():
    rng = np.random.default_rng(12)
    for trial in range(60):
        n = int(rng.choice([2, 3, 10, 200, 750]))
        mode = trial % 3
        samples = []
        for _ in range(n):
            frag = max(60, int(rng.normal(320, 35)))
            if rng.random() < 0.03:
                frag = int(rng.integers(2000, 50000))  # a chimeric pair: an outlier
            p0 = int(rng.integers(0, 10 ** 6))
            up, down = (p0, p0 + 50), (p0 + frag - 50, p0 + frag)
            first_is_up = bool(rng.integers(0, 2))
            if mode == 0:      # FR: the upstream mate forward, the downstream one reverse complement
                su, sd_ = 0, 1
            elif mode == 1:    # RF
                su, sd_ = 1, 0
            else:              # FF
                su = sd_ = int(rng.integers(0, 2))
            if rng.random() < 0.1:
                su, sd_ = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            a, b = ((up, su), (down, sd_)) if first_is_up else ((down, sd_), (up, su))
            samples.append((a[0][0], a[0][1], a[1], b[0][0], b[0][1], b[1]))
        got = ca.pair_infer(samples)
        o, mx, mn, mean, dev = _infer_reference(samples)
        assert got.inferred == 1 and got.n_pairs == n
        assert (got.orientation, got.max_insert, got.min_insert) == (o, mx, mn), (trial, got.max_insert, mx)
        assert abs(got.mean_insert - mean) < 1e-3 * max(1.0, mean) and abs(got.stddev_insert - dev) < 1e-3 * max(1.0, dev)
        if n >= 200:
            assert got.orientation == (ca.ORIENTATION_FR, ca.ORIENTATION_RF, ca.ORIENTATION_FF)[mode]
            assert 250 < got.mean_insert < 400 and got.max_insert < 1500
    assert ca.pair_infer(np.zeros((0, 6), np.uint32)).inferred == 0
