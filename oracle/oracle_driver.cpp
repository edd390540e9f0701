// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.
// Line-protocol twin of ref_driver.cpp, answering the same commands from the
// oracle's restatement (oracle_core.hpp / oracle_search.hpp).  tests/ diff its
// output with the fixtures that ref_driver (the real reference code) produced.
// ============================================================================
#include "move_driver_common.hpp"
#include "oracle_move.hpp"
#include "oracle_search.hpp"
#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>

using namespace orc;
using namespace std;

static void words(ostream& os, const vector<uint64_t>& w) {
    os << w.size();
    for (auto v : w) os << ' ' << hex << v << dec;
}
static string cigarStr(const vector<pair<char, uint32_t>>& c) {
    string s;
    for (auto& p : c) s += to_string(p.second) + p.first;
    return s;
}

// twin of ref_driver_rlc.cpp's "move" command on the restated move table
template <typename L> static void moveCommand(istringstream& in, ostream& os) {
    string text;
    int reversed, nq;
    in >> text >> reversed >> nq;
    const movedrv::Prepared p = movedrv::prepare(text, reversed != 0);
    const L cum[5] = {p.cum[0], p.cum[1], p.cum[2], p.cum[3], p.cum[4]};
    MoveLFT<L> built;
    buildMoveRows<L>(p.bwt, cum, built);
    const vector<uint8_t> file = built.fileBytes();
    movedrv::hexBytes(os, file);
    MoveLFT<L> rows;
    if (!rows.loadBytes(file.data(), file.size())) os << " LOADFAILED";
    const L r = rows.size();
    os << " | " << r;
    for (L i = 0; i <= r; i++)
        os << ' ' << (int)rows.getRunHead(i) << ':' << rows.getInputStartPos(i) << ':' << rows.getOutputStartPos(i) << ':'
           << rows.getOutputStartRun(i);
    for (int q = 0; q < nq; q++) {
        L b, e, c;
        in >> b >> e >> c;
        MoveRangeT<L> range(b, e, 0, r - 1, false), child;
        rows.computeRunIndices(range);
        rows.addChar(range, child, c);
        os << " | " << range.beginRun << ' ' << range.endRun << ' ' << range.runIndicesValid << ' ' << child.begin << ' ' << child.end
           << ' ' << child.beginRun << ' ' << child.endRun << ' ' << child.runIndicesValid << ' ' << rows.countChar(range, c) << ' '
           << rows.getCumulativeCounts(range, c);
        L pos = b, run = range.beginRun;
        rows.findLF(pos, run);
        L pos2 = e - 1;
        rows.findLFWithoutFastForward(pos2, range.endRun);
        os << ' ' << pos << ' ' << run << ' ' << pos2;
    }
}

int main() {
    string line;
    while (getline(cin, line)) {
        istringstream in(line);
        string cmd;
        in >> cmd;
        ostringstream os;
        if (cmd == "move") {
            int width;
            in >> width;
            if (width == 64) moveCommand<uint64_t>(in, os);
            else moveCommand<uint32_t>(in, os);
        } else if (cmd == "bwt") {
            string bwt;
            in >> bwt;
            vector<uint8_t> codes(bwt.size());
            for (size_t i = 0; i < bwt.size(); i++) codes[i] = (uint8_t)Index::c2i(bwt[i]);
            uint64_t N = bwt.size() + 1;
            vector<uint64_t> bv(BitvecIntl4::bvWords(N)), cnt(BitvecIntl4::cntWords(N));
            uint64_t dp;
            buildBitvecIntl4(codes.data(), bwt.size(), bv.data(), cnt.data(), dp);
            vector<uint64_t> file = {dp, N};
            file.insert(file.end(), bv.begin(), bv.end());
            file.insert(file.end(), cnt.begin(), cnt.end());
            words(os, file);
            BWTRepr r;
            r.bv = BitvecIntl4{N, bv.data(), cnt.data()};
            r.dollarPos = dp;
            for (int c = 0; c < 5; c++)
                for (size_t k = 0; k <= bwt.size(); k++) os << ' ' << r.occ(c, k) << ' ' << r.cumOcc(c, k);
        } else if (cmd == "bitvec9") {
            string bits;
            in >> bits;
            uint64_t N = bits.size();
            vector<uint64_t> bv(Bitvec9::bvWords(N), 0), cnt(Bitvec9::cntWords(N), 0);
            for (size_t i = 0; i < N; i++)
                if (bits[i] == '1') bv[i / 64] |= 1ull << (i % 64);
            buildBitvec9Counts(bv.data(), bv.size(), cnt.data());
            vector<uint64_t> file = {N};
            file.insert(file.end(), bv.begin(), bv.end());
            file.insert(file.end(), cnt.begin(), cnt.end());
            words(os, file);
            Bitvec9 b{N, bv.data(), cnt.data()};
            for (size_t p = 0; p < N; p++) os << ' ' << b.rank(p);
        } else if (cmd == "enc") {
            string txt;
            in >> txt;
            vector<uint8_t> codes(txt.size());
            for (size_t i = 0; i < txt.size(); i++) codes[i] = (uint8_t)Index::c2i(txt[i]);
            vector<uint64_t> w(EncodedBWT::nWords(txt.size()) + 1, 0);
            EncodedBWT::encode(codes.data(), txt.size(), w.data());
            vector<uint64_t> file = {txt.size(), EncodedBWT::nWords(txt.size())};
            file.insert(file.end(), w.begin(), w.end() - 1);
            words(os, file);
            EncodedBWT e;
            e.words = w.data();
            e.tSize = txt.size();
            for (size_t i = 0; i < txt.size(); i++) os << ' ' << e.at(i);
        } else if (cmd == "matrix") {
            string X, Y;
            int dir;
            uint32_t maxED, nInit;
            in >> X >> dir >> Y >> maxED >> nInit;
            vector<uint32_t> init(nInit);
            for (auto& v : init) in >> v;
            BitParallelED64 M;
            Substring sx(X.data(), (len_t)X.size(), 0, (len_t)X.size(), dir == 0 ? FORWARD : BACKWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, init);
            os << M.getNumberOfRows() << ' ' << M.getNumberOfCols() << ' ' << M.getSizeOfFinalColumn();
            os << ' ' << M.inFinalColumn(0);
            uint32_t i = 0;
            for (; i < Y.size() && i + 1 < M.getNumberOfRows(); i++) {
                bool v = M.computeRow(i + 1, Y[i]);
                uint32_t r = i + 1;
                os << ' ' << v << ' ' << M.getFirstColumn(r) << ' ' << M.inFinalColumn(r) << ' '
                   << M.onlyVerticalGapsLeft(r);
                uint32_t fc = M.getFirstColumn(r);
                uint32_t lc = std::min(M.getNumberOfCols() - 1, r + (maxED - (init.empty() ? 0 : init[0])));
                os << ' ' << (lc - fc + 1);
                for (uint32_t j = fc; j <= lc; j++) os << ' ' << M.at(r, j);
                if (!v) break;
            }
        } else if (cmd == "traceback") {
            string X, Y;
            uint32_t maxED, minED, nZeros;
            in >> X >> Y >> maxED >> minED >> nZeros;
            BitParallelED64 M;
            Substring sx(X.data(), (len_t)X.size(), 0, (len_t)X.size(), FORWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, vector<uint32_t>(nZeros, 0));
            Substring ref(Y.data(), (len_t)Y.size(), 0, (len_t)Y.size(), FORWARD);
            uint32_t i = 0;
            for (; i < Y.size() && i + 1 < M.getNumberOfRows(); i++)
                if (!M.computeRow(i + 1, Y[i])) break;
            vector<len_t> ends;
            if (M.inFinalColumn(i)) M.findClusterCenters(i, ends, maxED, minED);
            os << i << ' ' << ends.size();
            for (auto e : ends) {
                len_t b, ed;
                vector<pair<char, uint32_t>> cig;
                M.traceBack(ref, e, b, ed, &cig);
                os << ' ' << e << ' ' << b << ' ' << ed << ' ' << cigarStr(cig);
            }
        } else if (cmd == "traceback128") {
            string X, Y;
            uint32_t maxED, minED, nZeros;
            in >> X >> Y >> maxED >> minED >> nZeros;
            BitParallelED128 M;
            Substring sx(X.data(), (len_t)X.size(), 0, (len_t)X.size(), FORWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, vector<uint32_t>(nZeros, 0));
            Substring ref(Y.data(), (len_t)Y.size(), 0, (len_t)Y.size(), FORWARD);
            uint32_t i = 0;
            for (; i < Y.size() && i + 1 < M.getNumberOfRows(); i++)
                if (!M.computeRow(i + 1, Y[i])) break;
            vector<len_t> ends;
            if (M.inFinalColumn(i)) M.findClusterCenters(i, ends, maxED, minED);
            os << i << ' ' << ends.size();
            for (auto e : ends) {
                len_t b, ed;
                vector<pair<char, uint32_t>> cig;
                M.traceBack(ref, e, b, ed, &cig);
                os << ' ' << e << ' ' << b << ' ' << ed << ' ' << cigarStr(cig);
            }
        } else if (cmd == "matrix128") {
            string X, Y;
            uint32_t maxED, nZeros;
            in >> X >> Y >> maxED >> nZeros;
            BitParallelED128 M;
            Substring sx(X.data(), (len_t)X.size(), 0, (len_t)X.size(), FORWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, vector<uint32_t>(nZeros, 0));
            os << M.getNumberOfRows() << ' ' << M.getNumberOfCols() << ' ' << M.getSizeOfFinalColumn();
            for (uint32_t i = 0; i < Y.size() && i + 1 < M.getNumberOfRows(); i++) {
                bool v = M.computeRow(i + 1, Y[i]);
                uint32_t r = i + 1;
                os << ' ' << v << ' ' << M.getFirstColumn(r) << ' ' << M.inFinalColumn(r);
                uint32_t fc = M.getFirstColumn(r);
                uint32_t lc = std::min(M.getNumberOfCols() - 1, r + maxED);
                os << ' ' << (lc - fc + 1);
                for (uint32_t j = fc; j <= lc; j++) os << ' ' << M.at(r, j);
                if (!v) break;
            }
        } else if (cmd == "search") {
            uint32_t n;
            in >> n;
            vector<len_t> pi(n), L(n), U(n);
            for (auto& v : pi) in >> v;
            for (auto& v : L) in >> v;
            for (auto& v : U) in >> v;
            Search s = Search::makeSearch(pi, L, U, 0);
            for (uint32_t i = 0; i < n; i++) os << (i ? " " : "") << s.getDirection(i);
            for (uint32_t i = 0; i < n; i++) os << ' ' << s.getDirectionSwitch(i);
            os << " |";
            for (uint32_t i = 1; i < n; i++)
                os << ' ' << s.getLowestPartProcessedBefore(i) << ' ' << s.getHighestPartProcessedBefore(i);
            os << " |";
            for (uint32_t i = 0; i < n; i++) os << ' ' << s.isUnidirectionalBackwards(i);
            os << " | " << s.connectivitySatisfied() << ' ' << s.validBounds() << ' ' << s.zeroBased();
        } else if (cmd == "scheme") {
            uint32_t k, ns, np;
            in >> k >> ns >> np;
            vector<Search> ss;
            for (uint32_t i = 0; i < ns; i++) {
                vector<len_t> pi(np), L(np), U(np);
                for (auto& v : pi) in >> v;
                for (auto& v : L) in >> v;
                for (auto& v : U) in >> v;
                ss.push_back(Search::makeSearch(pi, L, U, i));
            }
            try {
                SearchScheme sch(ss, k);
                os << "ok " << sch.criticalPartIndex << ' ' << sch.getNumParts();
                SearchScheme mir = sch.mirrorPiStrings();
                os << ' ' << mir.criticalPartIndex;
            } catch (const std::exception& e) {
                os << "error " << e.what();
            }
        } else if (cmd == "cluster") {
            uint32_t size, maxED, startDepth, shift, nset;
            in >> size >> maxED >> startDepth >> shift >> nset;
            Cluster cl(size, maxED, startDepth, shift);
            for (uint32_t i = 0; i < nset; i++) {
                uint32_t idx, ed, depth, a, b, c, d;
                char ch;
                in >> idx >> ed >> depth >> a >> b >> c >> d >> ch;
                cl.setValue(idx, FMPosExt(ch, RangePair(Range(a, b), Range(c, d)), depth), ed);
            }
            string op;
            uint32_t arg;
            in >> op >> arg;
            auto pr = [&](const FMOcc& m) {
                os << ' ' << m.isValid() << ' ' << m.getRanges().sa.b << ' ' << m.getRanges().sa.e << ' '
                   << m.getRanges().rev.b << ' ' << m.getRanges().rev.e << ' ' << m.distance << ' '
                   << m.getDepth() << ' ' << m.shift;
            };
            if (op == "centra") {
                vector<FMPosExt> desc;
                vector<uint16_t> ie;
                FMOcc m = cl.getClusterCentra((uint16_t)arg, desc, ie);
                os << "centra";
                pr(m);
                os << ' ' << desc.size();
                for (auto& dn : desc) os << ' ' << dn.depth << ' ' << dn.ranges.sa.b << ' ' << dn.c;
                os << ' ' << ie.size();
                for (auto v : ie) os << ' ' << v;
            } else if (op == "centers") {
                auto v = cl.reportCentersAtEnd();
                os << "centers " << v.size();
                for (auto& m : v) pr(m);
                auto v2 = cl.reportCentersAtEnd();
                os << " again " << v2.size();
                for (auto& m : v2) os << ' ' << m.isValid();
            } else {
                FMOcc m = cl.reportDeepestMinimum(arg == 0 ? FORWARD : BACKWARD);
                os << "deepest";
                pr(m);
                FMOcc m2 = cl.reportDeepestMinimum(arg == 0 ? FORWARD : BACKWARD);
                os << " again " << m2.isValid();
            }
        } else if (cmd == "verify") {
            string text, pattern;
            uint32_t maxED, minED, nZeros, noCigar, ns;
            in >> text >> pattern >> maxED >> minED >> nZeros >> noCigar >> ns;
            vector<len_t> starts(ns);
            for (auto& v : starts) in >> v;
            Index idx;
            idx.textLength = (len_t)text.size();
            idx.text = (const uint8_t*)text.data();
            Strategy st;
            Matcher m(idx, st);
            m.noCIGAR = noCigar != 0;
            Substring pat(pattern.data(), (len_t)pattern.size(), 0, (len_t)pattern.size(), FORWARD);
            Occurrences occ;
            // nZeros == 1 <=> fixedStartPos (fmindex.cpp:276)
            if (!(nZeros == 1 || nZeros == 2 * maxED + 1)) {
                os << "unsupported";
            } else {
                m.inTextVerification(starts, maxED, minED, occ, pat, nZeros == 1);
                os << m.counters.c[IN_TEXT_STARTED] << ' ' << m.counters.c[ABORTED_IN_TEXT_VERIF] << ' '
                   << m.counters.c[CIGARS_IN_TEXT_VERIFICATION] << ' ' << occ.inTextOcc.size();
                for (const auto& t : occ.inTextOcc)
                    os << ' ' << t.range.b << ' ' << t.range.e << ' ' << t.distance << ' '
                       << (t.hasCigar() ? cigarStr(t.cigar) : string("*"));
            }
        } else if (cmd == "occsort") {
            uint32_t n;
            in >> n;
            Occurrences occ;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t b, e, d, hc;
                in >> b >> e >> d >> hc;
                TextOcc t(Range(b, e), d, FORWARD_STRAND);
                if (hc) t.cigar = {{'M', 1}};
                occ.inTextOcc.push_back(t);
            }
            occ.eraseDoublesAndSortText();
            os << occ.inTextOcc.size();
            for (const auto& t : occ.inTextOcc)
                os << ' ' << t.range.b << ' ' << t.range.e << ' ' << t.distance << ' ' << t.hasCigar();
        } else if (cmd == "fmoccsort") {
            uint32_t n;
            in >> n;
            Occurrences occ;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t a, b, d, dep, sh, st;
                in >> a >> b >> d >> dep >> sh >> st;
                occ.inFMOcc.push_back(FMOcc(RangePair(Range(a, b), Range(a, b)), d, dep,
                                            st ? REVERSE_C_STRAND : FORWARD_STRAND, sh));
            }
            occ.eraseDoublesFM();
            os << occ.inFMOcc.size();
            for (const auto& f : occ.inFMOcc)
                os << ' ' << f.getRanges().sa.b << ' ' << f.getRanges().sa.e << ' ' << f.distance << ' '
                   << f.getDepth() << ' ' << f.shift << ' ' << (f.strand == REVERSE_C_STRAND);
        } else if (cmd == "revcomp") {
            string s;
            in >> s;
            os << Matcher::revCompl(s);
        } else if (cmd == "ssa") {
            uint32_t sp, n;
            in >> sp >> n;
            vector<len_t> sa(n);
            for (auto& v : sa) in >> v;
            const SparseSAFiles f = buildSparseSA(sa, sp);
            words(os, f.bvFile);
            os << " | " << f.samples.size();
            for (auto v : f.samples) os << ' ' << v;
            os << " |";
            // read back: the Bitvec of the file + the samples (suffixArray.h:131-148)
            const uint64_t N = f.bvFile[0];
            Bitvec9 b{N, f.bvFile.data() + 1, f.bvFile.data() + 1 + Bitvec9::bvWords(N)};
            for (uint32_t i = 0; i < n; i++) {
                os << ' ' << b.get(i);
                if (b.get(i)) os << ':' << f.samples[b.rank(i)];
            }
            os << " | " << sp;
        } else if (cmd == "read") {
            string id, rd, ql;
            in >> id >> rd >> ql;
            for (auto& c : id)
                if (c == '_') c = ' ';
            const string r = cleanReadSeq(rd), rc = Matcher::revCompl(r);
            string rq = ql;
            std::reverse(rq.begin(), rq.end());
            os << cleanSeqID(id) << ' ' << r << ' ' << rc << ' ' << rq << ' ' << r.size() << ' ' << r << ' ' << rc;
        } else if (cmd == "kmer") {
            size_t ws, oa, ob;
            string a, b;
            in >> ws >> a >> oa >> b >> ob;
            const string ka = kmerKeyString(a, oa, ws), kb = kmerKeyString(b, ob, ws);
            os << ka << ' ' << kb << ' ' << (ka == kb) << ' ' << 1 << ' '
               << Substring(a.data(), (len_t)a.size(), (len_t)oa, (len_t)(oa + ws)).containsN() << ' '
               << Substring(b.data(), (len_t)b.size(), (len_t)ob, (len_t)(ob + ws)).containsN();
        } else if (cmd == "substr") {
            string t;
            uint32_t b, e;
            int d;
            in >> t >> b >> e >> d;
            Substring sub(t.data(), (len_t)t.size(), b, e, d == 0 ? FORWARD : BACKWARD);
            os << sub.size() << ' ' << sub.begin() << ' ' << sub.end() << ' ' << sub.empty() << ' ' << sub.containsN() << ' '
               << '[' << sub.tostring() << "] [";
            for (len_t i = 0; i < sub.size(); i++) os << sub[i];
            os << ']';
            sub.setDirection(d == 0 ? BACKWARD : FORWARD);
            os << " [";
            for (len_t i = 0; i < sub.size(); i++) os << sub[i];
            os << ']';
        } else if (cmd == "readscheme") {
            string path;
            unsigned k;
            in >> path >> k;
            try {
                ifstream ifs(path);
                if (!ifs) throw std::runtime_error("cannot open");
                SearchScheme sch = SearchScheme::readScheme(ifs, path, k);
                SearchScheme mir = sch.mirrorPiStrings();
                os << "ok " << sch.searches.size() << ' ' << sch.getNumParts() << ' ' << sch.criticalPartIndex << ' '
                   << mir.criticalPartIndex;
                for (const auto* sc : {&sch, &mir})
                    for (const Search& se : sc->searches) {
                        os << " |";
                        for (len_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getPart(i);
                        for (len_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getLowerBound(i);
                        for (len_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getUpperBound(i);
                        for (len_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getDirection(i);
                        for (len_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getDirectionSwitch(i);
                        for (len_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.isUnidirectionalBackwards(i);
                    }
            } catch (const std::exception& e) {
                string m = e.what();
                for (auto& c : m)
                    if (c == '\n') c = '~';
                os << "error " << m;
            }
        } else if (cmd == "findcigar") {
            string X, Y;
            uint32_t score;
            in >> X >> Y >> score;
            Substring ref(Y.data(), (len_t)Y.size(), 0, (len_t)Y.size(), FORWARD);
            vector<pair<char, uint32_t>> cig;
            if (score <= BitParallelED64::MATRIX_MAX_ED) { // (IndexInterface::generateCIGAR picks the matrix by the score, indexinterface.h:976-982)
                BitParallelED64 M;
                M.setSequence(Substring(X.data(), (len_t)X.size(), 0, (len_t)X.size(), FORWARD));
                M.findCIGAR(ref, score, cig);
            } else {
                BitParallelED128 M;
                M.setSequence(Substring(X.data(), (len_t)X.size(), 0, (len_t)X.size(), FORWARD));
                M.findCIGAR(ref, score, cig);
            }
            os << (cig.empty() ? string("*") : cigarStr(cig));
        } else if (cmd == "sam1" || cmd == "samxa" || cmd == "samun") {
            const vector<string> seqNames = {"chr1", "chr2_alt", "seqC"};
            string id, rd, ql;
            in >> id >> rd >> ql;
            if (ql == "-") ql = "";
            const string sid = cleanSeqID(id), read = cleanReadSeq(rd), rc = Matcher::revCompl(read);
            string rq = ql;
            std::reverse(rq.begin(), rq.end());
            auto readOcc = [&]() {
                uint32_t b, e, d, st, sq;
                string cg;
                in >> b >> e >> d >> cg >> st >> sq;
                SamOcc t;
                t.seqName = seqNames[sq];
                t.cigar = cg;
                t.begin = b;
                t.distance = d;
                t.revCompl = st != 0;
                return t;
            };
            string line;
            if (cmd == "samun") {
                line = samUnmappedSE(sid, read, ql);
            } else if (cmd == "sam1") {
                uint32_t nHits, minScore, primary;
                in >> nHits >> minScore >> primary;
                SamOcc t = readOcc();
                line = primary ? samSingleEnd(sid, t, t.revCompl ? rc : read, t.revCompl ? rq : ql, nHits, minScore, true)
                               : samSingleEnd(sid, t, "*", "*", nHits, minScore, false);
            } else {
                uint32_t nHits, n;
                in >> nHits >> n;
                vector<SamOcc> occs;
                for (uint32_t i = 0; i < n; i++) occs.push_back(readOcc());
                line = samSingleEndXA(sid, occs, occs[0].revCompl ? rc : read, occs[0].revCompl ? rq : ql, nHits);
            }
            for (auto& c : line)
                if (c == '\t') c = '|';
                else if (c == '\n') c = '~';
            os << line;
        } else if (cmd == "sampe" || cmd == "samunpaired" || cmd == "samunpe") {
            const vector<string> seqNames = {"chr1", "chr2_alt", "seqC"};
            string id, rd, ql;
            in >> id >> rd >> ql;
            if (ql == "-") ql = "";
            const string sid = cleanSeqID(id), read = cleanReadSeq(rd), rc = Matcher::revCompl(read);
            string rq = ql;
            std::reverse(rq.begin(), rq.end());
            auto readOcc = [&]() {
                uint32_t b, e, d, st, sq;
                string cg;
                in >> b >> e >> d >> cg >> st >> sq;
                SamOcc t;
                t.seqName = seqNames[sq];
                t.cigar = cg;
                t.begin = b;
                t.distance = d;
                t.revCompl = st != 0;
                return t;
            };
            string line;
            if (cmd == "samunpe") {
                uint32_t first, mateMapped, mateRev;
                in >> first >> mateMapped >> mateRev;
                line = samUnmappedPE(sid, read, ql, first != 0, mateMapped != 0, mateRev != 0);
            } else if (cmd == "samunpaired") {
                uint32_t first, nHits, minScore, primary;
                in >> first >> nHits >> minScore >> primary;
                const SamOcc t = readOcc();
                line = samUnpaired(sid, t, first != 0, nHits, minScore, primary != 0, t.revCompl ? rc : read, t.revCompl ? rq : ql);
            } else {
                uint32_t first, nPairs, minScore, fragSize, discordant, primary, mateMapped;
                in >> first >> nPairs >> minScore >> fragSize >> discordant >> primary >> mateMapped;
                const SamOcc t = readOcc();
                SamMate m;
                m.firstInPair = first == 0;
                if (mateMapped) {
                    const SamOcc mo = readOcc();
                    m.valid = true, m.revCompl = mo.revCompl, m.seqName = mo.seqName, m.begin = mo.begin, m.distance = mo.distance;
                } else {
                    uint32_t mateRev;
                    in >> mateRev; // (the unmapped mate lies on the forward strand with distance 0, indexhelpers.cpp:190-193)
                }
                line = samPairedEnd(sid, t, first != 0, m, nPairs, minScore, fragSize, discordant != 0, primary != 0, t.revCompl ? rc : read,
                                    t.revCompl ? rq : ql);
            }
            for (auto& c : line)
                if (c == '\t') c = '|';
                else if (c == '\n') c = '~';
            os << line;
        } else if (cmd == "consts") {
            os << BitParallelED64::MATRIX_MAX_ED << ' ' << BitParallelED64::LEFT << ' ' << 13 << ' ' << 10
               << ' ' << 4 << ' ' << sizeof(len_t);
        } else {
            os << "unknown";
        }
        cout << os.str() << "\n";
    }
    return 0;
}
