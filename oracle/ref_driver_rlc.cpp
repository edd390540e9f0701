// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Function-level driver around the REAL reference code of the run-length compressed flavour: it #includes the
// reference's bmove/moverepr.h where it lies under /root/reference/src and is linked against the reference's own
// bmove/moverepr.cpp, indexhelpers.cpp and logger.cpp, unmodified, compiled with -DRUN_LENGTH_COMPRESSION (see
// oracle/Makefile, target ref).  No stand-in header or library.  bmove/bmove.cpp, bmove/plcp.h and
// bmove/sparsebitvec.h need sdsl-lite, which is not in the image: NOT buildable here, not driven.
//
// Only this file (and move_driver_common.hpp / the construction template in oracle_move.hpp) is ours.
// Protocol: one command per stdin line, one result line on stdout.
// ============================================================================
#include "bmove/moverepr.h"
#include "indexhelpers.h"
#include "logger.h"

#include "move_driver_common.hpp"
#include "oracle_move.hpp" // buildMoveRows<Rows> only (a template over the rows class)

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unistd.h>

using namespace std;

int main() {
    logger.setVerbose(false);
    logger.setLogFile("/dev/null");
    string line;
    while (getline(cin, line)) {
        istringstream in(line);
        string cmd;
        in >> cmd;
        ostringstream os;
        if (cmd == "move") { // width text reversed nq {begin end c}*: MoveLFReprBP built through its own setters, written by
                             // its own writer, read back by its own loader, then queried
            string text;
            int width, reversed, nq;
            in >> width >> text >> reversed >> nq;
            if (width != (int)sizeof(length_t) * 8) {
                cout << "width\n";
                continue;
            }
            const movedrv::Prepared p = movedrv::prepare(text, reversed != 0);
            const length_t cum[5] = {p.cum[0], p.cum[1], p.cum[2], p.cum[3], p.cum[4]};
            MoveLFReprBP built;
            orc::buildMoveRows<length_t>(p.bwt, cum, built);
            char buf[] = "/tmp/refmoveXXXXXX";
            int fd = mkstemp(buf);
            if (fd >= 0) close(fd);
            const string base(buf), fn = base + ".LFBP";
            built.write(fn);
            {
                ifstream ifs(fn, ios::binary);
                vector<uint8_t> data((istreambuf_iterator<char>(ifs)), istreambuf_iterator<char>());
                movedrv::hexBytes(os, data);
            }
            MoveLFReprBP rows;
            if (!rows.load(base)) os << " LOADFAILED";
            remove(fn.c_str());
            remove(base.c_str());
            const length_t r = rows.size();
            os << " | " << r;
            for (length_t i = 0; i <= r; i++)
                os << ' ' << (int)rows.getRunHead(i) << ':' << rows.getInputStartPos(i) << ':' << rows.getOutputStartPos(i) << ':'
                   << rows.getOutputStartRun(i);
            for (int q = 0; q < nq; q++) {
                length_t b, e, c;
                in >> b >> e >> c;
                SARange range(b, e, 0, r - 1, false);
                rows.computeRunIndices(range);
                SARange child;
                rows.addChar(range, child, c);
                os << " | " << range.getBeginRun() << ' ' << range.getEndRun() << ' ' << range.getRunIndicesValid() << ' '
                   << child.getBegin() << ' ' << child.getEnd() << ' ' << child.getBeginRun() << ' ' << child.getEndRun() << ' '
                   << child.getRunIndicesValid() << ' ' << rows.countChar(range, c) << ' ' << rows.getCumulativeCounts(range, c);
                // LF of both ends of the range, with and without fast-forward
                length_t pos = b, run = range.getBeginRun();
                rows.findLF(pos, run);
                length_t pos2 = e - 1;
                rows.findLFWithoutFastForward(pos2, range.getEndRun());
                os << ' ' << pos << ' ' << run << ' ' << pos2;
            }
        } else {
            os << "unknown";
        }
        cout << os.str() << "\n";
    }
    return 0;
}
