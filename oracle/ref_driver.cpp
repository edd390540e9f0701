// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Function-level driver around the REAL reference code: it #includes the
// reference's headers where they lie under /root/reference/src and is linked
// against the reference's own bitparallelmatrix.cpp, indexhelpers.cpp,
// search.cpp, logger.cpp and nucleotide.cpp, unmodified (see oracle/Makefile).
// No stand-in header, library or generated file is used: every translation
// unit listed compiles with the image's g++ and the {fmt} headers shipped in
// the image's torch wheel.  The reference units that need parallel_hashmap
// (indexinterface.cpp, searchstrategy.cpp, fmindex/fmindex.cpp) are NOT
// buildable here and are therefore not driven.
//
// Only this file is ours; it contains no reference code, only calls into it.
// It runs in the build container only (the GPU box has no /root/reference);
// its outputs are committed as fixtures by tests/golden/make_golden.py.
//
// Protocol: one command per stdin line, one result line on stdout.
// ============================================================================
#include "bitparallelmatrix.h"
#include "bitvec.h"
#include "fmindex/bwtrepr.h"
#include "fmindex/encodedtext.h"
#include "fmindex/suffixArray.h"
#include "indexhelpers.h"
#include "nucleotide.h"
#include "reads.h"
#include "search.h"
#include "substring.h"
#include "tkmer.h"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unistd.h>

using namespace std;

static string tmpName() {
    char buf[] = "/tmp/refdrvXXXXXX";
    int fd = mkstemp(buf);
    if (fd >= 0) close(fd);
    return string(buf);
}
static void dumpFileWords32(const string& fn, ostream& os) {
    ifstream ifs(fn, ios::binary);
    vector<char> data((istreambuf_iterator<char>(ifs)), istreambuf_iterator<char>());
    os << data.size() / 4;
    for (size_t i = 0; i + 4 <= data.size(); i += 4) {
        uint32_t w;
        memcpy(&w, &data[i], 4);
        os << ' ' << w;
    }
}
static void dumpFileWords(const string& fn, ostream& os) {
    ifstream ifs(fn, ios::binary);
    vector<char> data((istreambuf_iterator<char>(ifs)), istreambuf_iterator<char>());
    os << data.size() / 8;
    for (size_t i = 0; i + 8 <= data.size(); i += 8) {
        uint64_t w;
        memcpy(&w, &data[i], 8);
        os << ' ' << hex << w << dec;
    }
}

int main() {
    logger.setVerbose(false);
    logger.setLogFile("/dev/null"); // the file loaders log to the logger's stream (std::cout by default)
    string line;
    while (getline(cin, line)) {
        istringstream in(line);
        string cmd;
        in >> cmd;
        ostringstream os;
        if (cmd == "bwt") { // BWTRepresentation<5>: file words, then occ/cumOcc for all c,k
            string bwt;
            in >> bwt;
            vector<length_t> cc(256, 0);
            cc['$'] = cc['A'] = cc['C'] = cc['G'] = cc['T'] = 1;
            Alphabet<5> sigma(cc);
            BWTRepresentation<5> r(sigma, bwt);
            string fn = tmpName();
            r.write(fn);
            dumpFileWords(fn, os);
            unlink(fn.c_str());
            for (int c = 0; c < 5; c++)
                for (size_t k = 0; k <= bwt.size(); k++) os << ' ' << r.occ(c, k) << ' ' << r.cumOcc(c, k);
        } else if (cmd == "bitvec9") { // Bitvec: file words then rank(p) for all p
            string bits;
            in >> bits;
            Bitvec bv(bits.size());
            for (size_t i = 0; i < bits.size(); i++)
                if (bits[i] == '1') bv[i] = true;
            bv.index();
            string fn = tmpName();
            {
                ofstream ofs(fn, ios::binary);
                bv.write(ofs);
            }
            dumpFileWords(fn, os);
            unlink(fn.c_str());
            for (size_t p = 0; p < bits.size(); p++) os << ' ' << bv.rank(p);
        } else if (cmd == "enc") { // EncodedText<5>
            string txt;
            in >> txt;
            vector<length_t> cc(256, 0);
            cc['$'] = cc['A'] = cc['C'] = cc['G'] = cc['T'] = 1;
            Alphabet<5> sigma(cc);
            EncodedText<5> e(sigma, txt);
            string fn = tmpName();
            e.write(fn);
            dumpFileWords(fn, os);
            unlink(fn.c_str());
            for (size_t i = 0; i < txt.size(); i++) os << ' ' << e[i];
        } else if (cmd == "matrix") { // X dir Y maxED nInit init...
            string X, Y;
            int dir;
            uint32_t maxED, nInit;
            in >> X >> dir >> Y >> maxED >> nInit;
            vector<uint32_t> init(nInit);
            for (auto& v : init) in >> v;
            BitParallelED64 M;
            Substring sx(X, dir == 0 ? FORWARD : BACKWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, init);
            os << M.getNumberOfRows() << ' ' << M.getNumberOfCols() << ' ' << M.getSizeOfFinalColumn();
            // row 0 cells of the band
            os << ' ' << M.inFinalColumn(0);
            uint32_t i = 0;
            for (; i < Y.size() && i + 1 < M.getNumberOfRows(); i++) {
                bool v = M.computeRow(i + 1, Y[i]);
                uint32_t r = i + 1;
                os << ' ' << v << ' ' << M.getFirstColumn(r) << ' ' << M.inFinalColumn(r) << ' '
                   << M.onlyVerticalGapsLeft(r);
                // all band cells of this row
                uint32_t fc = M.getFirstColumn(r);
                uint32_t lc = std::min(M.getNumberOfCols() - 1, r + (maxED - (init.empty() ? 0 : init[0])));
                os << ' ' << (lc - fc + 1);
                for (uint32_t j = fc; j <= lc; j++) os << ' ' << M.at(r, j);
                if (!v) break;
            }
        } else if (cmd == "traceback") { // X Y maxED nZeros -> clusters centres + traceback (+CIGAR)
            string X, Y;
            uint32_t maxED, minED, nZeros;
            in >> X >> Y >> maxED >> minED >> nZeros;
            BitParallelED64 M;
            Substring sx(X, FORWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, vector<uint32_t>(nZeros, 0));
            Substring ref(Y, FORWARD);
            uint32_t i = 0;
            for (; i < Y.size() && i + 1 < M.getNumberOfRows(); i++)
                if (!M.computeRow(i + 1, Y[i])) break;
            vector<length_t> ends;
            if (M.inFinalColumn(i)) M.findClusterCenters(i, ends, maxED, minED);
            os << i << ' ' << ends.size();
            for (auto e : ends) {
                length_t b, ed;
                string cig;
                M.traceBack(ref, e, b, ed, cig);
                os << ' ' << e << ' ' << b << ' ' << ed << ' ' << cig;
            }
        } else if (cmd == "traceback128") { // as "traceback", on BitParallelED128 (in-text verification beyond 64 bits)
            string X, Y;
            uint32_t maxED, minED, nZeros;
            in >> X >> Y >> maxED >> minED >> nZeros;
            BitParallelED128 M;
            Substring sx(X, FORWARD);
            M.setSequence(sx);
            M.initializeMatrix(maxED, vector<uint32_t>(nZeros, 0));
            Substring ref(Y, FORWARD);
            uint32_t i = 0;
            for (; i < Y.size() && i + 1 < M.getNumberOfRows(); i++)
                if (!M.computeRow(i + 1, Y[i])) break;
            vector<length_t> ends;
            if (M.inFinalColumn(i)) M.findClusterCenters(i, ends, maxED, minED);
            os << i << ' ' << ends.size();
            for (auto e : ends) {
                length_t b, ed;
                string cig;
                M.traceBack(ref, e, b, ed, cig);
                os << ' ' << e << ' ' << b << ' ' << ed << ' ' << cig;
            }
        } else if (cmd == "matrix128") { // X Y maxED nZeros: rows of a full-read BitParallelED128 (band cells of every row)
            string X, Y;
            uint32_t maxED, nZeros;
            in >> X >> Y >> maxED >> nZeros;
            BitParallelED128 M;
            Substring sx(X, FORWARD);
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
                for (uint32_t j = fc; j <= lc; j++) os << ' ' << M(r, j);
                if (!v) break;
            }
        } else if (cmd == "search") { // n pi.. L.. U..
            uint32_t n;
            in >> n;
            vector<length_t> pi(n), L(n), U(n);
            for (auto& v : pi) in >> v;
            for (auto& v : L) in >> v;
            for (auto& v : U) in >> v;
            Search s = Search::makeSearch(pi, L, U, 0);
            for (uint32_t i = 0; i < n; i++) os << (i ? " " : "") << s.getDirection(i);
            for (uint32_t i = 0; i < n; i++) os << ' ' << s.getDirectionSwitch(i);
            // lowest/highest part processed before phase i (i >= 1)
            os << " |";
            for (uint32_t i = 1; i < n; i++)
                os << ' ' << s.getLowestPartProcessedBefore(i) << ' ' << s.getHighestPartProcessedBefore(i);
            os << " |";
            for (uint32_t i = 0; i < n; i++) os << ' ' << s.isUnidirectionalBackwards(i);
            os << " | " << s.connectivitySatisfied() << ' ' << s.validBounds() << ' ' << s.zeroBased();
        } else if (cmd == "scheme") { // k nSearches nParts rows(pi L U)...
            uint32_t k, ns, np;
            in >> k >> ns >> np;
            vector<Search> ss;
            for (uint32_t i = 0; i < ns; i++) {
                vector<length_t> pi(np), L(np), U(np);
                for (auto& v : pi) in >> v;
                for (auto& v : L) in >> v;
                for (auto& v : U) in >> v;
                ss.push_back(Search::makeSearch(pi, L, U, i));
            }
            try {
                SearchScheme sch(ss, k);
                os << "ok " << sch.getCriticalPartIndex() << ' ' << sch.getNumParts();
                SearchScheme mir = sch.mirrorPiStrings();
                os << ' ' << mir.getCriticalPartIndex();
            } catch (const std::exception& e) {
                os << "error " << e.what();
            }
        } else if (cmd == "cluster") {
            // size maxED startDepth shift nset {idx ed depth sab sae rvb rve char}* op arg
            uint32_t size, maxED, startDepth, shift, nset;
            in >> size >> maxED >> startDepth >> shift >> nset;
            MatrixMetaInfo cl(size, maxED, startDepth, shift);
            for (uint32_t i = 0; i < nset; i++) {
                uint32_t idx, ed, depth, a, b, c, d;
                char ch;
                in >> idx >> ed >> depth >> a >> b >> c >> d >> ch;
                cl.setValue(idx, FMPosExt(ch, SARangePair(SARange(a, b), SARange(c, d)), depth), ed);
            }
            string op;
            uint32_t arg;
            in >> op >> arg;
            auto pr = [&](const FMOcc& m) {
                os << ' ' << m.isValid() << ' ' << m.getRanges().getRangeSA().getBegin() << ' '
                   << m.getRanges().getRangeSA().getEnd() << ' ' << m.getRanges().getRangeSARev().getBegin()
                   << ' ' << m.getRanges().getRangeSARev().getEnd() << ' ' << m.getDistance() << ' '
                   << m.getDepth() << ' ' << m.getShift();
            };
            if (op == "centra") {
                vector<FMPosExt> desc;
                vector<uint16_t> ie;
                FMOcc m = cl.getClusterCentra(arg, desc, ie);
                os << "centra";
                pr(m);
                os << ' ' << desc.size();
                for (auto& dn : desc)
                    os << ' ' << dn.getDepth() << ' ' << dn.getRanges().getRangeSA().getBegin() << ' '
                       << dn.getCharacter();
                os << ' ' << ie.size();
                for (auto v : ie) os << ' ' << v;
            } else if (op == "centers") {
                auto v = cl.reportCentersAtEnd();
                os << "centers " << v.size();
                for (auto& m : v) pr(m);
                // second call: already-reported nodes give invalid occurrences
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
            // text pattern maxED minED nZeros noCIGAR nStarts starts...
            // replays FMIndex::inTextVerification (fmindex.cpp:267-310) around the real
            // InTextVerificationTask<uint64_t>::doTask (indexhelpers.cpp:518)
            string text, pattern;
            uint32_t maxED, minED, nZeros, noCigar, ns;
            in >> text >> pattern >> maxED >> minED >> nZeros >> noCigar >> ns;
            vector<length_t> starts(ns);
            for (auto& v : starts) in >> v;
            BitParallelED64 M;
            Substring pat(pattern, FORWARD);
            M.setSequence(pat);
            M.initializeMatrix(maxED, vector<uint32_t>(nZeros, 0));
            length_t nRows = M.getNumberOfRows();
            length_t textLength = text.size();
            vector<Substring> refs;
            for (auto start : starts) {
                length_t maxEnd = textLength - 1;
                length_t hEnd = std::min(maxEnd, start + nRows - 1);
                refs.emplace_back(Substring(text, start, hEnd));
            }
            InTextVerificationTask<uint64_t> task(refs, &M, maxED, minED, FORWARD_STRAND, FIRST_IN_PAIR,
                                                  noCigar != 0);
            Counters counters;
            Occurrences occ;
            task.doTask(counters, occ);
            os << counters.get(Counters::IN_TEXT_STARTED) << ' ' << counters.get(Counters::ABORTED_IN_TEXT_VERIF)
               << ' ' << counters.get(Counters::CIGARS_IN_TEXT_VERIFICATION) << ' ' << occ.textOccSize();
            for (auto& t : occ.getTextOccurrencesMutable())
                os << ' ' << t.getRange().getBegin() << ' ' << t.getRange().getEnd() << ' ' << t.getDistance()
                   << ' ' << (t.hasCigar() ? t.getCigar() : string("*"));
        } else if (cmd == "occsort") { // n {begin end dist hasCigar}* -> sort + unique (indexhelpers.h:2148)
            uint32_t n;
            in >> n;
            Occurrences occ;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t b, e, d, hc;
                in >> b >> e >> d >> hc;
                occ.addTextOcc(Range(b, e), d, hc ? string("1M") : string(""), FORWARD_STRAND, FIRST_IN_PAIR);
            }
            occ.eraseDoublesAndSortText();
            os << occ.textOccSize();
            for (const auto& t : occ.getTextOccurrences())
                os << ' ' << t.getRange().getBegin() << ' ' << t.getRange().getEnd() << ' ' << t.getDistance()
                   << ' ' << t.hasCigar();
        } else if (cmd == "fmoccsort") { // n {sab sae dist depth shift strand}* (indexhelpers.h:2135)
            uint32_t n;
            in >> n;
            Occurrences occ;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t a, b, d, dep, sh, st;
                in >> a >> b >> d >> dep >> sh >> st;
                occ.addFMOcc(FMOcc(SARangePair(SARange(a, b), SARange(a, b)), d, dep,
                                   st ? REVERSE_C_STRAND : FORWARD_STRAND, FIRST_IN_PAIR, sh));
            }
            occ.eraseDoublesFM();
            os << occ.getFMOccurrences().size();
            for (const auto& f : occ.getFMOccurrences())
                os << ' ' << f.getRanges().getRangeSA().getBegin() << ' ' << f.getRanges().getRangeSA().getEnd()
                   << ' ' << f.getDistance() << ' ' << f.getDepth() << ' ' << f.getShift() << ' '
                   << f.isRevCompl();
        } else if (cmd == "revcomp") {
            string s;
            in >> s;
            os << Nucleotide::getRevComplWithN(s);
        } else if (cmd == "ssa") { // s n sa[0..n): SparseSuffixArray written by its own writer, read back by the
                                    // file constructor (mmap) — suffixArray.h:160-243, :131-148
            uint32_t sp, n;
            in >> sp >> n;
            vector<length_t> sa(n);
            for (auto& v : sa) in >> v;
            string base = tmpName();
            {
                SparseSuffixArray w(sa, sp);
                w.write(base);
            }
            const string fbv = base + ".sa.bv." + to_string(sp), fsa = base + ".sa." + to_string(sp);
            dumpFileWords(fbv, os);
            os << " |";
            os << ' ';
            dumpFileWords32(fsa, os);
            os << " |";
            {
                SparseSuffixArray r(base, sp);
                for (uint32_t i = 0; i < n; i++) {
                    os << ' ' << r[i];
                    if (r[i]) os << ':' << r.get(i);
                }
                os << " | " << r.getFactor();
            }
            unlink(fbv.c_str());
            unlink(fsa.c_str());
            unlink(base.c_str());
        } else if (cmd == "read") { // seqID(with @ or >, '_' for spaces) read qual: Read + ReadBundle (reads.h:43-58, :97-160)
            string id, rd, ql;
            in >> id >> rd >> ql;
            for (auto& c : id)
                if (c == '_') c = ' ';
            Read r(id, rd, ql);
            ReadBundle b(r);
            os << r.getSeqID() << ' ' << r.getRead() << ' ' << b.getRevComp() << ' ' << b.getRevQuality() << ' ' << b.size()
               << ' ' << b.getSequence(FORWARD_STRAND) << ' ' << b.getSequence(REVERSE_C_STRAND);
        } else if (cmd == "kmer") { // wordSize strA offA strB offB: Kmer keys of the k-mer table (tkmer.h, indexinterface.h:590)
            size_t ws, oa, ob;
            string a, b;
            in >> ws >> a >> oa >> b >> ob;
            Kmer::setWordSize(ws);
            Kmer ka(a, oa), kb(b, ob);
            os << ka.str() << ' ' << kb.str() << ' ' << (ka == kb) << ' ' << (!(ka == kb) || KmerHash()(ka) == KmerHash()(kb)) << ' '
               << Substring(&a, (length_t)oa, (length_t)(oa + ws)).containsN() << ' '
               << Substring(&b, (length_t)ob, (length_t)(ob + ws)).containsN();
        } else if (cmd == "substr") { // text begin end dir: Substring accessors (substring.h)
            string t;
            uint32_t b, e;
            int d;
            in >> t >> b >> e >> d;
            Substring sub(&t, b, e, d == 0 ? FORWARD : BACKWARD);
            os << sub.size() << ' ' << sub.begin() << ' ' << sub.end() << ' ' << sub.empty() << ' ' << sub.containsN() << ' '
               << '[' << sub.tostring() << "] [";
            for (size_t i = 0; i < sub.size(); i++) os << sub[i];
            os << ']';
            sub.setDirection(d == 0 ? BACKWARD : FORWARD);
            os << " [";
            for (size_t i = 0; i < sub.size(); i++) os << sub[i];
            os << ']';
        } else if (cmd == "readscheme") { // path k: SearchScheme::readScheme (search.h:684-711) on a scheme file
            string path;
            unsigned k;
            in >> path >> k;
            try {
                ifstream ifs(path);
                if (!ifs) throw std::runtime_error("cannot open");
                SearchScheme sch = SearchScheme::readScheme(ifs, path, k);
                SearchScheme mir = sch.mirrorPiStrings();
                os << "ok " << sch.getSearches().size() << ' ' << sch.getNumParts() << ' ' << sch.getCriticalPartIndex()
                   << ' ' << mir.getCriticalPartIndex();
                for (const auto* sc : {&sch, &mir})
                    for (const Search& se : sc->getSearches()) {
                        os << " |";
                        for (length_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getPart(i);
                        for (length_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getLowerBound(i);
                        for (length_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getUpperBound(i);
                        for (length_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getDirection(i);
                        for (length_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.getDirectionSwitch(i);
                        for (length_t i = 0; i < se.getNumParts(); i++) os << ' ' << se.isUnidirectionalBackwards(i);
                    }
            } catch (const std::exception& e) {
                string m = e.what();
                for (auto& c : m)
                    if (c == '\n') c = '~';
                os << "error " << m;
            }
        } else if (cmd == "findcigar") { // X Y score: IBitParallelED::findCIGAR (bitparallelmatrix.h:460-527), Y = text[begin, end)
            string X, Y;
            uint32_t score;
            in >> X >> Y >> score;
            BitParallelED64 M;
            Substring sx(X, FORWARD);
            M.setSequence(sx);
            Substring ref(Y, FORWARD);
            string cig;
            M.findCIGAR(ref, score, cig);
            os << (cig.empty() ? "*" : cig);
        } else if (cmd == "sam1" || cmd == "samxa" || cmd == "samun") {
            // SAM records of single-end reads (indexhelpers.cpp:56-120, :177-200; indexhelpers.h:321-331, :378-388, :416-421,
            // :625-680): tabs are printed as '|'
            const vector<string> seqNames = {"chr1", "chr2_alt", "seqC"};
            string id, rd, ql;
            in >> id >> rd >> ql;
            if (ql == "-") ql = "";
            Read r(id, rd, ql);
            ReadBundle bundle(r);
            auto readOcc = [&]() {
                uint32_t b, e, d, st, sq;
                string cg;
                in >> b >> e >> d >> cg >> st >> sq;
                TextOcc t(Range(b, e), d, cg, st ? REVERSE_C_STRAND : FORWARD_STRAND, FIRST_IN_PAIR);
                t.setAssignedSequence(FOUND, sq);
                return t;
            };
            string line;
            if (cmd == "samun") {
                line = TextOcc::createUnmappedSAMOccurrenceSE(bundle).getOutputLine();
            } else if (cmd == "sam1") {
                uint32_t nHits, minScore, primary;
                in >> nHits >> minScore >> primary;
                TextOcc t = readOcc();
                if (primary) t.generateSAMSingleEndFirst(bundle, nHits, minScore, seqNames);
                else t.generateSAMSingleEndNotFirst(bundle.getSeqID(), nHits, minScore, seqNames);
                line = t.getOutputLine();
            } else {
                uint32_t nHits, n;
                in >> nHits >> n;
                vector<TextOcc> occs;
                for (uint32_t i = 0; i < n; i++) occs.emplace_back(readOcc());
                occs.front().generateSAMSingleEndXA(bundle, nHits, occs.front().getDistance(), occs.begin() + 1, occs.end(), seqNames);
                line = occs.front().getOutputLine();
            }
            for (auto& c : line)
                if (c == '\t') c = '|';
                else if (c == '\n') c = '~';
            os << line;
        } else if (cmd == "sampe" || cmd == "samunpaired" || cmd == "samunpe") {
            // SAM records of paired-end reads (indexhelpers.cpp:114-262; indexhelpers.h:340-371, :378-410): tabs printed as '|'
            const vector<string> seqNames = {"chr1", "chr2_alt", "seqC"};
            string id, rd, ql;
            in >> id >> rd >> ql;
            if (ql == "-") ql = "";
            Read r(id, rd, ql);
            ReadBundle bundle(r);
            auto readOcc = [&](PairStatus ps) {
                uint32_t b, e, d, st, sq;
                string cg;
                in >> b >> e >> d >> cg >> st >> sq;
                TextOcc t(Range(b, e), d, cg, st ? REVERSE_C_STRAND : FORWARD_STRAND, ps);
                t.setAssignedSequence(FOUND, sq);
                return t;
            };
            string line;
            if (cmd == "samunpe") { // first mateMapped mateRev
                uint32_t first, mateMapped, mateRev;
                in >> first >> mateMapped >> mateRev;
                line = TextOcc::createUnmappedSAMOccurrencePE(bundle, first ? FIRST_IN_PAIR : SECOND_IN_PAIR, mateMapped != 0,
                                                              mateRev ? REVERSE_C_STRAND : FORWARD_STRAND)
                           .getOutputLine();
            } else if (cmd == "samunpaired") { // first nHits minScore primary occ
                uint32_t first, nHits, minScore, primary;
                in >> first >> nHits >> minScore >> primary;
                TextOcc t = readOcc(first ? FIRST_IN_PAIR : SECOND_IN_PAIR);
                t.generateSAMUnpaired(bundle, nHits, minScore, primary != 0, seqNames);
                line = t.getOutputLine();
            } else { // first nPairs minScore fragSize discordant primary mateMapped occ [mateOcc | mateRev]
                uint32_t first, nPairs, minScore, fragSize, discordant, primary, mateMapped;
                in >> first >> nPairs >> minScore >> fragSize >> discordant >> primary >> mateMapped;
                TextOcc t = readOcc(first ? FIRST_IN_PAIR : SECOND_IN_PAIR);
                TextOcc mate;
                if (mateMapped) mate = readOcc(first ? SECOND_IN_PAIR : FIRST_IN_PAIR);
                else { // the unmapped mate as the reference creates it (searchstrategy.cpp:1463-1517)
                    uint32_t mateRev;
                    in >> mateRev;
                    mate = TextOcc::createUnmappedSAMOccurrencePE(bundle, first ? SECOND_IN_PAIR : FIRST_IN_PAIR, true,
                                                                  t.isRevCompl() ? REVERSE_C_STRAND : FORWARD_STRAND);
                }
                t.generateSAMPairedEnd(bundle, nPairs, minScore, mate, fragSize, discordant != 0, primary != 0, seqNames);
                line = t.getOutputLine();
            }
            for (auto& c : line)
                if (c == '\t') c = '|';
                else if (c == '\n') c = '~';
            os << line;
        } else if (cmd == "consts") {
            os << BitParallelED64::getMatrixMaxED() << ' ' << BitParallelED64::getMaxFirstColRows() << ' '
               << MAX_K << ' ' << CIGAR_THRESHOLD << ' ' << DEFAULT_SPARSENESS << ' ' << sizeof(length_t);
        } else {
            os << "unknown";
        }
        cout << os.str() << "\n";
    }
    return 0;
}
