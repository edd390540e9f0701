// columba_build on the framework's side: FASTA files in, the reference's Vanilla index files out.
//   columba_build [-s <sparseness> | -a] [-l <seed length>] [--rlc [--keep-text]] -r <index base name> -f <fasta> [<fasta> ...]
// Mirrors the reference's tool (src/buildindex.cpp:main, parameters/buildparameters.cpp): -r the base name of the index files, -f the
// FASTA files (plain or .gz), -s the suffix-array sparseness (a power of two, default 4), -l the length of the seed that replaces runs
// of non-ACGT characters (default 0: random characters from std::minstd_rand(42), as the reference's Vanilla build does).
// The files load in `columba` itself and in include/columba_amd.hpp (FMIndex).
// --rlc: the run-length compressed flavour instead (the reference builds it as a separate binary, RUN_LENGTH_COMPRESSION): move tables
// (.LFBP, .rev.LFBP in the reference's format), samples, predecessors and PLCP as the .u64 files include/columba_amd_bmove.hpp (BMove)
// loads; seed length 100 unless -l is given (definitions.h:41); --keep-text also writes <base>.txt.bin for alignments (BMove::attachText).
//   g++ -std=c++17 -O2 -I include examples/columba_build.cpp -o columba_build -lz
#include "columba_amd_build.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    std::string base;
    std::vector<std::string> fasta;
    uint32_t sparseness = 4, seedLength = 0;
    bool rlc = false, keepText = false, seedGiven = false, allFactors = false; // (-a: every sparseness factor 1 ... 128, buildparameters.cpp:155-170)
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if ((a == "-r" || a == "--reference-base-name") && i + 1 < argc) base = argv[++i];
        else if ((a == "-s" || a == "--sparseness") && i + 1 < argc) sparseness = (uint32_t)std::atoi(argv[++i]);
        else if ((a == "-l" || a == "--seed-length") && i + 1 < argc) seedLength = (uint32_t)std::atoi(argv[++i]), seedGiven = true;
        else if (a == "-a" || a == "--all-sa-sparseness") allFactors = true;
        else if (a == "--rlc") rlc = true;
        else if (a == "--keep-text") keepText = true;
        else if (a == "-f" || a == "--fasta-files") {
            while (i + 1 < argc && argv[i + 1][0] != '-') fasta.push_back(argv[++i]);
        } else {
            std::fprintf(stderr, "usage: %s [-s sparseness] [-l seed length] [--rlc [--keep-text]] -r <index base name> -f <fasta> [<fasta> ...]\n", argv[0]);
            return 1;
        }
    }
    if (base.empty() || fasta.empty()) {
        std::fprintf(stderr, "usage: %s [-s sparseness] [-l seed length] [--rlc [--keep-text]] -r <index base name> -f <fasta> [<fasta> ...]\n", argv[0]);
        return 1;
    }
    try {
        if (rlc)
            columba_amd::build::buildMoveIndex(fasta, base, seedGiven ? seedLength : 100u, keepText);
        else
            columba_amd::build::buildIndex(fasta, base, sparseness, seedLength, allFactors);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "columba_build: %s\n", e.what());
        return 1;
    }
    std::printf("index files written to %s.*\n", base.c_str());
    return 0;
}
