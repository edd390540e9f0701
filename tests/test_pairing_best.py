"""Pairing of read pairs in BEST (+x strata) mode (cmb_pair_best_*) against a restatement of the reference's logic in
Python — SearchStrategy::matchApproxPairedEndBestPlusX (searchstrategy.cpp:1091-1179), processCombFF / RF / FR (:936-1062), processComb
(:834-912), processSeq (:778-812), pairOccurrencesForBestMapping (:1743-1815), handleTrimmedOccs (:814-832), mergeOrMovePairs (:914-934),
pairDiscordantlyBest (:1664-1741), mapStratum (searchstrategy.h:1354-1361), addDiscPairs (:1518-1585), findBestMapping / findBestAlignments /
checkAlignments / combineOccVectors (:536-712, :1648-1662), addUnpairedMatches / addOneUnmapped / addBothUnmapped, generateSAMPairedEnd
(:1904-1970) and the order of OutputWriter::writeChunks (fastq.cpp:662-702) — the single records from the pinned functions (cmb_sam_pe,
cmb_sam_unpaired, cmb_sam_unmapped_pe) — and, for x = 0, against brute force over everything the mates have (the best concordant pairs).

The restatement walks ONE pair straight through, as the reference does, calling `map_read` whenever it wants a stratum; the library walks a
chunk of pairs and asks for the lists (cmb_pair_best_advance / _supply): the test answers every request from the same table the
restatement reads, and the table is made to differ between distances (an occurrence can be absent from the list of one distance), so a
walk that asked for another distance than the reference would end elsewhere.  searchstrategy.cpp itself cannot be built here
(parallel_hashmap): this layer is parity-unpinned.  Host code only: runs without a GPU."""
import numpy as np
import pytest

import columba_amd as ca

SEQ_NAMES = ["chr1", "chr2_alt", "seqC"]
SEQ_START = [0, 40_000, 90_000, 140_000]
UNCHECKED, FOUND, TRIMMED, NOT_FOUND = 0, 1, 2, 3


def _seq_of(pos):
    return max(i for i in range(3) if SEQ_START[i] <= pos)


def synthetic_trim(stratum, begin, end, distance):
    """the test's stand-in for findSeqName on an occurrence that runs over the end of its sequence (indexinterface.cpp:833-899): the two
    options of the reference, with a made-up but deterministic verification"""
    idx = _seq_of(begin)
    nxt = SEQ_START[idx + 1]
    if nxt - begin <= stratum:
        if idx + 2 >= len(SEQ_START):
            return None
        b, e, sid = nxt, min(end, SEQ_START[idx + 2]), idx + 1
        d = distance + (nxt - begin)
    elif end - nxt <= stratum:
        b, e, sid = begin, nxt, idx
        d = distance + (end - nxt)
    else:
        return None
    if d > stratum or e <= b:
        return None
    return b, e, d, sid, b - SEQ_START[sid], [(e - b) << 2]


class Occ:
    __slots__ = ("ib", "w", "d", "s", "m", "sid", "sb", "state", "spans", "ops")

    def __init__(self, ib, w, d, s, m, ops):
        self.ib, self.w, self.d, self.s, self.m, self.ops = ib, w, d, s, m, ops
        self.spans = _seq_of(ib) != _seq_of(ib + w - 1)
        self.sid = _seq_of(ib)
        self.sb = ib - SEQ_START[self.sid]
        self.state = UNCHECKED

    def copy(self):
        o = Occ.__new__(Occ)
        for k in Occ.__slots__:
            setattr(o, k, getattr(self, k))
        return o

    def assigned(self):
        return self.state in (FOUND, TRIMMED)

    def rb(self):  # begin of the range: relative to the sequence once assigned
        return self.sb if self.assigned() else self.ib

    def key(self):  # TextOcc::operator<
        return (self.rb(), self.d, self.w)


class RefWalk:
    """one pair, straight through"""

    def __init__(self, table, reads, cutoffs, x, orientation, max_frag, min_frag, disc, unmapped, seeds=None, hamming=False):
        self.table, self.R, self.cut, self.hamming = table, reads, cutoffs, hamming
        self.x, self.ori, self.max_frag, self.min_frag, self.disc, self.unmapped = x, orientation, max_frag, min_frag, disc, unmapped
        self.ov = [[[[False, []] for _ in range(cutoffs[m] + 1)] for _s in (0, 1)] for m in (0, 1)]
        self.pairs = []     # [up, down, frag, distance, discordant]; up / down = Occ or ("unmapped", mate)
        self.unpaired = []  # ("occ", Occ, best_count, best, first) or ("unmapped", mate)
        self.asked = []
        if seeds is not None:  # addSingleEndedForBest (searchstrategy.cpp:1064-1089)
            se1, se2, read2done = seeds
            for o in se1 + se2:
                o = o.copy()
                o.state = FOUND
                self.ov[o.m][o.s][o.d][1].append(o)
            for s in (0, 1):
                for st in self.ov[0][s]:
                    st[0] = True
                for st in self.ov[1][s]:
                    st[0] = read2done

    def map_read(self, m, s, k, min_d=0):
        self.asked.append((m, s, k))
        lst = [o.copy() for o in self.table[(m, s, k)] if o.d >= min_d]
        if k == 0:
            lst.sort(key=Occ.key)
        return lst

    def process_seq(self, m, s, max_dist):
        v = self.ov[m][s]
        if not v[max_dist][0]:
            min_d = next((i for i, st in enumerate(v) if not st[0]), len(v))
            min_d = min(min_d, max_dist)
            for o in self.map_read(m, s, max_dist, min_d):
                v[o.d][1].append(o)
            for i in range(min_d, max_dist + 1):
                v[i][0] = True
        return any(v[i][1] for i in range(max_dist + 1))

    def assign(self, o, max_ed):
        if o.state == UNCHECKED:
            if not o.spans:
                o.state = FOUND
            else:
                t = None if self.hamming else synthetic_trim(max_ed, o.ib, o.ib + o.w, o.d)  # (no trimming under Hamming distance, indexinterface.cpp:828-832)
                if t is None:
                    o.state = NOT_FOUND
                else:
                    b, e, d, sid, sb, ops = t
                    o.ib, o.w, o.d, o.sid, o.sb, o.ops, o.state = b, e - b, d, sid, sb, ops, TRIMMED
        return o.state

    def pair_for_best(self, U, D, out, u_max, d_max, u_trim, d_trim):
        if not U or not D:
            return
        D.sort(key=lambda o: o.ib)
        for i, u in enumerate(U):
            upos = u.ib
            j = 0
            while j < len(D) and D[j].ib < upos:
                j += 1
            while j < len(D):
                d = D[j]
                frag = d.ib + d.w - upos
                if self.min_frag <= frag <= self.max_frag:
                    uf = self.assign(u, u_max)
                    if uf != FOUND:
                        if uf == TRIMMED:
                            u_trim.add(i)
                        break
                    df = self.assign(d, d_max)
                    if df != FOUND:
                        if df == TRIMMED:
                            d_trim.add(j)
                        j += 1
                        continue
                    if u.sid == d.sid:
                        out.append([u.copy(), d.copy(), d.rb() + d.w - u.rb(), u.d + d.d, False])
                elif frag > self.max_frag:
                    break
                j += 1

    @staticmethod
    def handle_trimmed(ids, o_dist, v):
        for i in sorted(ids, reverse=True):
            occ = v[o_dist][1].pop(i)
            occ.state = FOUND
            t = v[occ.d][1]
            k = 0
            while k < len(t) and t[k].key() < occ.key():
                k += 1
            t.insert(k, occ)

    @staticmethod
    def first_pos_dist(v):
        return next((i for i, st in enumerate(v) if st[1] or not st[0]), len(v))

    def process_comb(self, um, us, dm, ds, out, tot):
        U, D = self.ov[um][us], self.ov[dm][ds]
        M = 1 << 32
        min_d, min_u = self.first_pos_dist(D), self.first_pos_dist(U)
        mx = {"u": min((tot - min_d) % M, len(U) - 1), "d": min((tot - min_u) % M, len(D) - 1)}

        def process_read(m, s, v, me, other):
            if not self.process_seq(m, s, mx[me]):
                return False
            mx[other] = min((tot - self.first_pos_dist(v)) % M, mx[other])
            return True

        if mx["u"] <= mx["d"]:
            if not (process_read(um, us, U, "u", "d") and process_read(dm, ds, D, "d", "u")):
                return
        elif not (process_read(dm, ds, D, "d", "u") and process_read(um, us, U, "u", "d")):
            return
        for dist in range(min_u + min_d, tot + 1):
            for u_dist in range(min_u, min(mx["u"], dist) + 1):
                d_dist = dist - u_dist
                if d_dist > mx["d"] or d_dist < min_d:
                    continue
                ut, dt = set(), set()
                self.pair_for_best(U[u_dist][1], D[d_dist][1], out, mx["u"], mx["d"], ut, dt)
                self.handle_trimmed(ut, u_dist, U)
                self.handle_trimmed(dt, d_dist, D)
            if out:
                return

    def process_ori(self, tot, min_tot):
        ov = self.ov

        def any_below(v, n):
            return any(st[1] for st in v[:n])

        if self.ori == ca.ORIENTATION_FF:
            A, B = (0, 0, 1, 0), (1, 1, 0, 1)
            a_first = any_below(ov[0][0], 99) or any_below(ov[1][0], 99)
        elif self.ori == ca.ORIENTATION_RF:
            A, B = (0, 1, 1, 0), (1, 1, 0, 0)
            a_first = any_below(ov[0][1], min_tot) or any_below(ov[1][0], min_tot)
        else:
            A, B = (0, 0, 1, 1), (1, 0, 0, 1)
            a_first = any_below(ov[0][0], min_tot) or any_below(ov[1][1], min_tot)
        pa, pb = [], []
        if a_first:
            self.process_comb(*A, pa, tot)
            tot = pa[0][3] if pa else tot
            self.process_comb(*B, pb, tot)
        else:
            self.process_comb(*B, pb, tot)
            tot = pb[0][3] if pb else tot
            self.process_comb(*A, pa, tot)
        if not pa or not pb:  # mergeOrMovePairs
            return pa or pb
        if pa[0][3] <= pb[0][3]:
            return pa + (pb if pa[0][3] == pb[0][3] else [])
        return pb

    # ---- without a concordant pair
    def map_stratum(self, m, s, k):
        st = self.ov[m][s][k]
        if not st[0]:
            st[1] = self.map_read(m, s, k, k)
            st[0] = True

    def add_disc_pairs(self, fw1, rc1, fw2, rc2, max_ed):
        if not (fw1 or rc1) or not (fw2 or rc2):
            return
        for A in (fw1, rc1):
            for a in A:
                for B in (fw2, rc2):
                    for b in B:
                        if self.assign(a, max_ed) == NOT_FOUND or self.assign(b, max_ed) == NOT_FOUND:
                            continue
                        a_up = a.rb() < b.rb()
                        frag = (b.rb() + b.w - a.rb() if a_up else a.rb() + a.w - b.rb()) if a.sid == b.sid else 0
                        up, down = (a, b) if a_up else (b, a)
                        self.pairs.append([up.copy(), down.copy(), frag, a.d + b.d, True])

    def check_alignments(self, m, s, best, l, cutoff):
        v = self.ov[m][s]
        trimmed, kept = [], []
        for o in v[l][1]:
            f = self.assign(o, cutoff)
            if f != FOUND:
                if f == TRIMMED and o.d > l:
                    trimmed.append(o)
            else:
                kept.append(o)
                best = min(best, l)
        v[l][1] = kept
        for o in trimmed:
            o.state = FOUND
            v[o.d][1].append(o)
        return best

    def find_best_alignments(self, m, x):
        fw, rc = self.ov[m]
        cutoff = len(fw) - 1
        best, found = cutoff + 1, False
        if x == 0:
            for s in (0, 1):
                if not self.ov[m][s][0][0]:
                    self.ov[m][s][0] = [True, self.map_read(m, s, 0, 0)]
            if fw[0][1] or rc[0][1]:
                best = self.check_alignments(m, 0, best, 0, cutoff)
                best = self.check_alignments(m, 1, best, 0, cutoff)
                found = best == 0
        max_ed = x if best == 0 else cutoff
        prev_k, k = 0, max(x, 1)

        def has_update(s, k):
            st = self.ov[m][s][k]
            return bool(st[1]) if st[0] else self.process_seq(m, s, k)

        while k <= max_ed:
            u0 = has_update(0, k)
            u1 = has_update(1, k)
            update = u0 or u1
            if update:
                for l in range(prev_k + 1, min(k, best + x) + 1):
                    best = self.check_alignments(m, 0, best, l, max_ed)
                    best = self.check_alignments(m, 1, best, l, max_ed)
            if found:
                break
            if update and best < cutoff + 1:
                found = True
                if x == 0:
                    break
                prev_k, k = k, min(best + x, max_ed)
            else:
                if k == max_ed:
                    break
                prev_k, k = k, min(k + x + (2 if k < 5 else 4), max_ed)
        return found, best

    def find_best_mapping(self, m, x):
        found, best = self.find_best_alignments(m, x)
        if not found:
            return []
        out = []
        for i in range(best, min(best + x, len(self.ov[m][0]) - 1) + 1):
            for s in (0, 1):
                v = sorted(self.ov[m][s][i][1], key=lambda o: (o.sid, o.rb()))
                uniq = []
                for o in v:
                    if not uniq or (uniq[-1].sid, uniq[-1].rb()) != (o.sid, o.rb()):
                        uniq.append(o)
                self.ov[m][s][i][1] = uniq
                out += uniq
        return out

    def add_both_unmapped(self):
        if self.unmapped:
            self.pairs.append([("unmapped", 0), ("unmapped", 1), 0, 0, False])

    def add_one_unmapped(self, m1, m2, max_ed):
        first = bool(m1)
        for o in (m1 if first else m2):
            if self.assign(o, max_ed) == NOT_FOUND:
                continue
            self.pairs.append([o.copy(), ("unmapped", 1 if first else 0), 0, o.d, False])
        if not self.pairs:
            self.add_both_unmapped()

    def add_unpaired(self, allm, m, max_ed):
        temp = [o for o in allm if self.assign(o, max_ed) != NOT_FOUND]
        if not temp:
            if self.unmapped:
                self.unpaired.append(("unmapped", m))
            return
        temp.sort(key=lambda o: o.d)
        best = temp[0].d
        cnt = sum(1 for o in temp if o.d == best)
        for i, o in enumerate(temp):
            self.unpaired.append(("occ", o, cnt, best, i == 0))

    def pair_discordantly_best(self, x):
        ov = self.ov
        max1, max2 = len(ov[0][0]) - 1, len(ov[1][0]) - 1
        if self.disc:
            total = len(ov[0][0]) + len(ov[1][0])
            best_stratum, found = total + 1, False
            for i in range(total):
                if i <= max1:
                    self.map_stratum(0, 0, i)
                    self.map_stratum(0, 1, i)
                if i <= max2:
                    self.map_stratum(1, 0, i)
                    self.map_stratum(1, 1, i)
                for e1 in range(i - max2 if i > max2 else 0, min(i, max1) + 1):
                    e2 = i - e1
                    self.add_disc_pairs(ov[0][0][e1][1], ov[0][1][e1][1], ov[1][0][e2][1], ov[1][1][e2][1], i)
                if self.pairs:
                    if not found:
                        best_stratum, found = i, True
                    if i == best_stratum + x:
                        return
        b1, b2 = self.find_best_mapping(0, x), self.find_best_mapping(1, x)
        if not b1 and not b2:
            self.add_both_unmapped()
        elif not b1:
            self.add_one_unmapped(b1, b2, max2)
        elif not b2:
            self.add_one_unmapped(b1, b2, max1)
        else:
            self.add_unpaired(b1, 0, max1)
            self.add_unpaired(b2, 1, max2)

    def run(self):
        x, c1, c2 = self.x, self.cut[0], self.cut[1]
        best, not_explored = c1 + c2 + 1, 0
        if x == 0:
            self.pairs = self.process_ori(0, 0)
            not_explored = 1
        found = bool(self.pairs)
        if found:
            best = 0
        max_stratum = x if best == 0 else c1 + c2
        k = max(x, 1)
        while k <= max_stratum:
            self.pairs = self.process_ori(k, not_explored)
            if not found:
                if self.pairs:
                    best, found = k, True
                    max_stratum = min(best + x, c1 + c2)
                    not_explored = k + 1
                    if x == 0:
                        break
                    k = max_stratum
                else:
                    if k == max_stratum:
                        break
                    k = min(max_stratum, k + x + (2 if k < 6 else 4))
            else:
                break
        if not self.pairs:
            self.pair_discordantly_best(x)
        return self

    # ---- records
    def sam(self):
        R = self.R

        def hit(o):
            return (SEQ_NAMES[o.sid], o.sb, o.d, bool(o.s), np.asarray(o.ops, dtype=np.uint16))

        def seq_of(o):
            return R[o.m][2] if o.s else R[o.m][1]

        def qual_of(o):
            return R[o.m][4] if o.s else R[o.m][3]

        def unmapped_line(m, mate_mapped, mate_rev):
            return ca.sam_unmapped_pe(R[m][0], R[m][1], R[m][3], m == 0, mate_mapped, mate_rev)

        pairs = [list(p) + ["", ""] for p in self.pairs]
        if pairs:
            mi = min(range(len(pairs)), key=lambda i: (pairs[i][3], i))
            best = pairs[mi][3]
            n_pairs = sum(1 for p in pairs if p[3] == best)
            pairs[0], pairs[mi] = pairs[mi], pairs[0]
            for i, p in enumerate(pairs):
                for side in (0, 1):
                    me, mate = p[side], p[1 - side]
                    mate_valid = isinstance(mate, Occ)
                    if not isinstance(me, Occ):
                        p[5 + side] = unmapped_line(me[1], mate_valid, mate_valid and bool(mate.s))
                        continue
                    p[5 + side] = ca.sam_pe(R[me.m][0], hit(me), me.m == 0, hit(mate) if mate_valid else None, n_pairs, best, p[2], p[4], i == 0,
                                            seq_of(me), qual_of(me))
        mapped = bool(pairs) and isinstance(pairs[0][0], Occ) and isinstance(pairs[0][1], Occ)
        half = not mapped and bool(pairs) and (isinstance(pairs[0][0], Occ) or isinstance(pairs[0][1], Occ))
        text = ""
        for i, p in enumerate(pairs):
            text += p[5]
            if not half or i == 0:
                text += p[6]
        for u in self.unpaired:
            if u[0] == "unmapped":
                text += unmapped_line(u[1], False, False)
            else:
                _, o, cnt, best, first = u
                text += ca.sam_unpaired(R[o.m][0], hit(o), o.m == 0, cnt, best, first, seq_of(o), qual_of(o))
        return text, (len(pairs) if mapped else 0)


# ---------------------------------------------------------------------------------------------------------------- synthetic pairs
def _make_pair(rng, i, len1, len2, min_identity, max_supported, kind, ori=None):
    """reads, cut-offs and the table of mapRead results: (mate, strand, k) -> [Occ], every k up to the mate's cut-off"""
    reads, cut = [], []
    for m, ln in enumerate((len1, len2)):
        seq = "".join(rng.choice(list("ACGT"), ln))
        _, cs, rc, rq = ca.read_prepare(f"@p{i}/{m + 1}", seq, "".join(rng.choice(list("FGHI"), ln)))
        reads.append((f"p{i}/{m + 1}", cs, rc, "".join(rng.choice(list("FGHI"), ln)), None))
        reads[-1] = reads[-1][:4] + (reads[-1][3][::-1],)
        cut.append(min(max_supported, ln * (100 - min_identity) // 100))
    ground = {(m, s): [] for m in (0, 1) for s in (0, 1)}

    def put(m, s, begin, d, hide=None, ln=None):
        ln = ln if ln is not None else (len1, len2)[m] + int(rng.integers(-2, 3))
        ground[(m, s)].append((begin, ln, d, hide))

    n_frag = {"none": 0, "one_sided": 0, "sparse": 1, "dense": int(rng.integers(2, 6)), "boundary": 2, "far": 0}[kind]
    ori = int(rng.integers(0, 3)) if ori is None else ori
    for _ in range(n_frag):  # a fragment: both mates near each other, orientation as drawn
        base = int(rng.integers(100, 135_000))
        if kind == "boundary" and rng.random() < 0.7:  # just over a sequence end on either side (trimming succeeds) or far over it (it does not)
            edge = SEQ_START[int(rng.integers(1, 3))]
            base = edge - int(rng.choice([int(rng.integers(1, 4)), (len1, len2)[0] - int(rng.integers(1, 4)), int(rng.integers(0, 60))]))
        gap = int(rng.integers(0, 300))
        up_s, down_s, up_m = {ca.ORIENTATION_FR: (0, 1), ca.ORIENTATION_RF: (1, 0), ca.ORIENTATION_FF: (0, 0)}[ori] + (int(rng.integers(0, 2)),)
        if ori == ca.ORIENTATION_FF and up_m == 1:
            up_s = down_s = 1  # (the reverse complements: read 2 upstream of read 1)
        put(up_m, up_s, base, int(rng.integers(0, cut[up_m] + 1)))
        put(1 - up_m, down_s, base + gap + 40, int(rng.integers(0, cut[1 - up_m] + 1)))
    extra = {"none": (0, 0), "one_sided": (3, 0), "sparse": (2, 2), "dense": (4, 4), "boundary": (2, 2), "far": (3, 3)}[kind]
    for m in (0, 1):
        for _ in range(int(rng.integers(0, extra[m] + 1)) if kind != "one_sided" else extra[m]):
            begin = int(rng.integers(0, 139_000))
            if kind == "boundary" and rng.random() < 0.5:
                edge = SEQ_START[int(rng.integers(1, 3))]
                begin = edge - int(rng.choice([int(rng.integers(1, 4)), (len1, len2)[m] - int(rng.integers(1, 4)), int(rng.integers(1, 50))]))
            d = int(rng.integers(0, cut[m] + 1))
            hide = int(rng.integers(d, cut[m] + 1)) if rng.random() < 0.25 else None  # absent from the list of ONE distance
            put(m, int(rng.integers(0, 2)), begin, d, hide)
    table = {}
    for (m, s), occs in ground.items():
        for k in range(cut[m] + 1):
            lst = [Occ(b, ln, d, s, m, [ln << 2]) for (b, ln, d, hide) in occs if d <= k and hide != k and b + ln <= SEQ_START[-1]]
            lst.sort(key=Occ.key)
            table[(m, s, k)] = lst
    return reads, cut, table, ori


def _as_arrays(lst):
    occ = np.zeros(len(lst), dtype=ca.OCC_DTYPE)
    aln = np.zeros(len(lst), dtype=ca.ALN_DTYPE)
    ops = []
    for j, o in enumerate(lst):
        occ[j] = (o.ib, o.ib + o.w, o.d, o.s)
        aln[j] = (o.sid, o.sb, len(ops), len(o.ops), 1 if o.spans else 0, 0)
        ops += o.ops
    return occ, aln, np.asarray(ops, dtype=np.uint16)


def _run_library(pairs, x, min_identity, max_supported, orientation, max_frag, min_frag, disc, unmapped, seeds=None, metric="edit"):
    """drive cmb_pair_best_* over a chunk: answer every request from the pair's table"""
    def trim(pair, mate, strand, stratum, occ):
        assert metric == "edit", "an occurrence over a sequence end is never trimmed under Hamming distance"
        return synthetic_trim(stratum, *occ)

    pb = ca.PairBest([p[0][0] for p in pairs], [p[0][1] for p in pairs], x, min_identity, max_supported, orientation, max_frag, min_frag, disc,
                     unmapped, metric=metric, trim=trim)
    asked = [[] for _ in pairs]
    for i, sd in enumerate(seeds or []):
        if sd is not None:
            pb.seed(i, _as_arrays(sd[0]), _as_arrays(sd[1]), sd[2])
    for rounds in range(400):
        req = pb.advance()
        if req.shape[0] == 0:
            break
        for r in req:
            i, m, s, k = int(r["pair"]), int(r["mate"]), int(r["strand"]), int(r["max_distance"])
            asked[i].append((m, s, k))
            pb.supply(i, m, s, k, *_as_arrays(pairs[i][2][(m, s, k)]))
    else:
        raise AssertionError("the walk does not end")
    out = [pb.sam(i, SEQ_NAMES) for i in range(len(pairs))]
    cut = [(pb.cutoff(i, 0), pb.cutoff(i, 1)) for i in range(len(pairs))]
    pb.close()
    return out, asked, cut


KINDS = ["none", "one_sided", "sparse", "dense", "boundary", "far"]


@pytest.mark.parametrize("x", [0, 1, 2])
@pytest.mark.parametrize("orientation", [ca.ORIENTATION_FR, ca.ORIENTATION_RF, ca.ORIENTATION_FF])
def test_best_pairing_equals_the_restatement(x, orientation):
    rng = np.random.default_rng(1000 + 10 * x + orientation)
    for disc, unmapped in ((True, True), (False, True), (True, False)):
        pairs = []
        for i in range(120):
            kind = KINDS[i % len(KINDS)]
            reads, cut, table, ori = _make_pair(rng, i, int(rng.choice([50, 100, 151])), int(rng.choice([50, 100, 151])), 95, 7, kind,
                                                ori=orientation if i % 2 else None)
            pairs.append((reads, cut, table))
        got, asked, cuts = _run_library(pairs, x, 95, 7, orientation, 600, 0, disc, unmapped)
        n_mapped = n_lines = 0
        for i, (reads, cut, table) in enumerate(pairs):
            assert cuts[i] == tuple(cut)
            ref = RefWalk(table, reads, cut, x, orientation, 600, 0, disc, unmapped).run()
            text, n = ref.sam()
            assert got[i] == (text, n), (i, KINDS[i % len(KINDS)], got[i], text)
            # the library asked for exactly the lists the straight walk reads, in the order it first reads them
            first_asked = list(dict.fromkeys(ref.asked))
            assert asked[i] == first_asked, (i, asked[i], first_asked)
            n_mapped += n > 0
            n_lines += text.count("\n")
        assert n_lines > 150 and (n_mapped > 10 or not disc)


@pytest.mark.parametrize("orientation", [ca.ORIENTATION_FR, ca.ORIENTATION_RF, ca.ORIENTATION_FF])
def test_pairs_that_start_from_single_end_results(orientation):
    """pairSingleEndedMatchesBest (searchstrategy.h:1454-1462): the walk starts from the mates' single-end BEST results — the best stratum of
    each mate, both strands — with every stratum of read 1 (and of read 2 if it was processed) counting as looked at"""
    rng = np.random.default_rng(300 + orientation)
    pairs, seeds = [], []
    for i in range(240):
        reads, cut, table, _ = _make_pair(rng, i, 100, int(rng.choice([80, 100])), 95, 5, KINDS[i % len(KINDS)], ori=orientation if i % 3 else None)
        pairs.append((reads, cut, table))
        se = []
        for m in (0, 1):
            every = [o for s in (0, 1) for o in table[(m, s, cut[m])] if not o.spans]
            best = min((o.d for o in every), default=None)
            se.append([o for o in every if o.d == best])
        read2done = bool(i % 4)
        seeds.append((se[0], se[1] if read2done else [], read2done))
    got, asked, _ = _run_library(pairs, 0, 95, 5, orientation, 600, 0, True, True, seeds=seeds)
    n_mapped = 0
    for i, (reads, cut, table) in enumerate(pairs):
        ref = RefWalk(table, reads, cut, 0, orientation, 600, 0, True, True, seeds=seeds[i]).run()
        text, n = ref.sam()
        assert got[i] == (text, n), (i, got[i], text)
        assert asked[i] == list(dict.fromkeys(ref.asked)) and all(m == 1 for m, _s, _k in asked[i])  # read 1 is never searched again
        n_mapped += n > 0
    assert n_mapped > 60
    pb = ca.PairBest([pairs[0][0][0]], [pairs[0][0][1]], 1, 95, 5)
    with pytest.raises(ca.CmbError, match="x = 0"):
        pb.seed(0, _as_arrays([]), _as_arrays([]), True)
    pb.close()


@pytest.mark.parametrize("x,orientation", [(0, ca.ORIENTATION_FR), (1, ca.ORIENTATION_RF), (3, ca.ORIENTATION_FF), (2, ca.ORIENTATION_FR)])
def test_deep_strata_and_fragment_bounds(x, orientation):
    """cut-offs up to 13 errors (90 % identity on 100 - 151 bp: the strata 1, 3, 5, 9, 13 of the walk and its steps of 4), a lower fragment
    bound and a tight upper one, mates of different lengths"""
    rng = np.random.default_rng(4000 + 10 * x + orientation)
    pairs = []
    for i in range(150):
        reads, cut, table, _ = _make_pair(rng, i, int(rng.choice([100, 130, 151])), int(rng.choice([60, 100, 151])), 90, 13, KINDS[i % len(KINDS)],
                                          ori=orientation if i % 3 else None)
        pairs.append((reads, cut, table))
    got, asked, cuts = _run_library(pairs, x, 90, 13, orientation, 420, 150, True, True)
    deepest = 0
    for i, (reads, cut, table) in enumerate(pairs):
        assert cuts[i] == tuple(cut) and max(cut) >= 10
        ref = RefWalk(table, reads, cut, x, orientation, 420, 150, True, True).run()
        assert got[i] == ref.sam(), (i, KINDS[i % len(KINDS)])
        assert asked[i] == list(dict.fromkeys(ref.asked)), (i, asked[i])
        deepest = max([deepest] + [k for _m, _s, k in asked[i]])
    assert deepest >= 9


def test_hamming_distance_never_trims():
    rng = np.random.default_rng(9)
    pairs = []
    for i in range(150):
        reads, cut, table, _ = _make_pair(rng, i, 100, 100, 95, 5, ["boundary", "sparse", "dense"][i % 3], ori=ca.ORIENTATION_FR)
        pairs.append((reads, cut, table))
    got, asked, _ = _run_library(pairs, 1, 95, 5, ca.ORIENTATION_FR, 600, 0, True, True, metric="hamming")
    n = 0
    for i, (reads, cut, table) in enumerate(pairs):
        ref = RefWalk(table, reads, cut, 1, ca.ORIENTATION_FR, 600, 0, True, True, hamming=True).run()
        assert got[i] == ref.sam() and asked[i] == list(dict.fromkeys(ref.asked)), i
        n += got[i][1] > 0
    assert n > 60


def test_best_pairs_are_the_best_concordant_pairs():
    """x = 0, FR: whenever the mates have a concordant pair at all (inside one sequence, within the fragment bounds, right strands) the
    records are exactly the concordant pairs of the smallest total distance — brute force over everything the mates have"""
    rng = np.random.default_rng(77)
    pairs, truth = [], []
    for i in range(200):
        reads, cut, table, _ = _make_pair(rng, i, 100, 100, 94, 6, ["sparse", "dense", "far"][i % 3], ori=ca.ORIENTATION_FR)
        for key in table:  # lists that grow with the distance, nothing hidden, nothing across a sequence end
            m, s, k = key
            table[key] = [o for o in table[(m, s, cut[m])] if o.d <= k and not o.spans]
        pairs.append((reads, cut, table))
        best = None
        for um, us, dm, ds in ((0, 0, 1, 1), (1, 0, 0, 1)):
            for u in table[(um, us, cut[um])]:
                for d in table[(dm, ds, cut[dm])]:
                    frag = d.ib + d.w - u.ib
                    if d.ib >= u.ib and 0 <= frag <= 600 and u.sid == d.sid:
                        tot = u.d + d.d
                        if best is None or tot < best[0]:
                            best = [tot, set()]
                        if tot == best[0]:
                            best[1].add((um, u.ib, d.ib))
        truth.append(best)
    got, _, _ = _run_library(pairs, 0, 94, 6, ca.ORIENTATION_FR, 600, 0, True, True)
    n = 0
    for i, best in enumerate(truth):
        text, n_pairs = got[i]
        if best is None:
            continue
        n += 1
        lines = [l.split("\t") for l in text.splitlines()]
        assert n_pairs == len(best[1]), (i, n_pairs, best)
        assert all(int(l[1]) & 2 for l in lines) and len(lines) == 2 * n_pairs
        assert {int(l[11].split(":")[2]) for l in lines} <= set(range(best[0] + 1))
        assert sum(int(l[11].split(":")[2]) for l in lines) == best[0] * n_pairs
    assert n > 80


def test_requests_and_errors():
    rng = np.random.default_rng(5)
    reads, cut, table, _ = _make_pair(rng, 0, 100, 100, 95, 5, "sparse")
    pb = ca.PairBest([reads[0]], [reads[1]], 0, 95, 5)
    with pytest.raises(ca.CmbError, match="still waits"):
        pb.sam(0, SEQ_NAMES)
    req = pb.advance()
    assert req.shape[0] == 1 and int(req[0]["max_distance"]) == 0
    with pytest.raises(ca.CmbError, match="beyond the read's cut-off"):
        pb.supply(0, 0, 0, 9, np.zeros(0, ca.OCC_DTYPE), np.zeros(0, ca.ALN_DTYPE), np.zeros(0, np.uint16))
    again = pb.advance()   # nothing supplied: the same request, nothing lost
    assert again.tolist() == req.tolist()
    pb.close()
    with pytest.raises(ca.CmbError, match="identity"):
        ca.PairBest([reads[0]], [reads[1]], 0, 30, 5)
    # an occurrence across a sequence end with neither an index nor a hook to trim it with is an error, not a silent drop
    t2 = {key: [] for key in table}
    for k in range(cut[0] + 1):
        t2[(0, 0, k)] = [Occ(SEQ_START[1] - 30, 100, 0, 0, 0, [400])]
        t2[(1, 1, k)] = [Occ(SEQ_START[1] + 100, 100, 0, 1, 1, [400])]
    pb = ca.PairBest([reads[0]], [reads[1]], 0, 95, 5)
    with pytest.raises(ca.CmbError, match="neither an index nor a hook"):
        for _ in range(20):
            req = pb.advance()
            if req.shape[0] == 0:
                break
            for r in req:
                m, s, k = int(r["mate"]), int(r["strand"]), int(r["max_distance"])
                lst = t2[(m, s, k)]
                occ = np.array([(o.ib, o.ib + o.w, o.d, o.s) for o in lst], dtype=ca.OCC_DTYPE)
                aln = np.array([(o.sid, o.sb, 0, 1, 1 if o.spans else 0, 0) for o in lst], dtype=ca.ALN_DTYPE)
                pb.supply(0, m, s, k, occ, aln, np.array([400], np.uint16))
    pb.close()


def _table_from(cut, occs):
    """(mate, strand, k) -> list for every k: occs = [(mate, strand, begin, width, distance)], lists that grow with the distance"""
    t = {}
    for m in (0, 1):
        for s in (0, 1):
            for k in range(cut[m] + 1):
                t[(m, s, k)] = sorted((Occ(b, w, d, s, m, [w << 2]) for (mm, ss, b, w, d) in occs if mm == m and ss == s and d <= k), key=Occ.key)
    return t


def _fixed_reads():
    out = []
    for m, seq in enumerate(("ACGTTGCAAC" * 10, "TTGACCAGTA" * 10)):
        _, cs, rc, rq = ca.read_prepare(f"@pair/{m + 1}", seq, "I" * 100)
        out.append((f"pair/{m + 1}", cs, rc, "I" * 100, "I" * 100))
    return out


def test_hand_checked_cases():
    """a few pairs whose outcome follows from the reference's text by hand (100 bp mates, 95 % identity: cut-off 5, FR, fragments up to 600)"""
    reads, cut = _fixed_reads(), [5, 5]

    def run(occs, x=0, disc=True):
        got, asked, _ = _run_library([(reads, cut, _table_from(cut, occs))], x, 95, 5, ca.ORIENTATION_FR, 600, 0, disc, True)
        return [ln.split("\t") for ln in got[0][0].splitlines()], got[0][1], asked[0]

    # 1. both mates match exactly, read 1 forward upstream of read 2 reverse: found in the very first stratum — only the two lists of
    #    distance 0 that this combination needs are ever asked for, then (nothing found for the 2-1 combination) its two lists
    lines, n_pairs, asked = run([(0, 0, 1000, 100, 0), (1, 1, 1300, 100, 0)])
    assert n_pairs == 1 and [int(l[1]) for l in lines] == [99, 147] and lines[0][3] == "1001" and lines[0][8] == "400" and lines[1][8] == "-400"
    assert lines[0][4] == lines[1][4] == "60" and all(k == 0 for _m, _s, k in asked) and len(asked) <= 4
    # 2. the best pair costs 1 + 2 errors; a second placement of read 2 with 4 errors is not reported at x = 0
    lines, n_pairs, asked = run([(0, 0, 1000, 100, 1), (1, 1, 1300, 100, 2), (1, 1, 1350, 100, 4)])
    assert n_pairs == 1 and [l[11] for l in lines] == ["AS:i:1", "AS:i:2"] and max(k for _m, _s, k in asked) <= 3
    # 3. two placements of the same total: both pairs reported, the first one primary, MAPQ of two candidates
    lines, n_pairs, _ = run([(0, 0, 1000, 100, 1), (1, 1, 1300, 100, 1), (0, 0, 50_000, 100, 0), (1, 1, 50_300, 100, 2)])
    assert n_pairs == 2 and [int(l[1]) & 256 for l in lines] == [0, 0, 256, 256] and {l[4] for l in lines} == {"3"}
    # 4. the mates lie on different sequences: no concordant pair; discordant pairs are allowed -> one pair without the "proper" bit and a
    #    template length of 0; not allowed -> each mate's best alignments as unpaired records
    occs = [(0, 0, 1000, 100, 0), (1, 1, 60_000, 100, 1)]
    lines, n_pairs, _ = run(occs)
    assert n_pairs == 1 and [int(l[1]) & 2 for l in lines] == [0, 0] and lines[0][8] == "0" and lines[0][6] == "chr2_alt"
    lines, n_pairs, _ = run(occs, disc=False)
    assert n_pairs == 0 and [int(l[1]) for l in lines] == [65, 129]
    # 5. read 2 maps nowhere, read 1 exactly: read 1's record says "mate unmapped", read 2's is the unmapped record with the mate's data
    lines, n_pairs, _ = run([(0, 0, 1000, 100, 0)])
    assert n_pairs == 0 and [int(l[1]) for l in lines] == [73, 133]
    # 6. ... but with read 1's only hit at 2 errors BOTH come out unmapped: pairDiscordantlyBest has looked at every stratum by itself
    #    (mapStratum), and findBestAlignments then only asks the strata 1, 3, 5 whether THEY hold something (hasUpdate, :674-681)
    lines, n_pairs, _ = run([(0, 0, 1000, 100, 2)])
    assert n_pairs == 0 and [int(l[1]) for l in lines] == [77, 141]
    lines, n_pairs, _ = run([(0, 0, 1000, 100, 3)])
    assert n_pairs == 0 and [int(l[1]) for l in lines] == [73, 133] and lines[0][11] == "AS:i:3"
    # 7. the mates are too far apart for the fragment bound: discordant, with the real template length
    lines, n_pairs, _ = run([(0, 0, 1000, 100, 0), (1, 1, 5000, 100, 0)])
    assert n_pairs == 1 and [int(l[1]) for l in lines] == [97, 145] and lines[0][8] == "4100"
