// Minimal Columba-style driver over the C++ adapter: the counterpart of processChunk
// (reference src/parallel.cpp:67-78) for one chunk of reads.
//   usage: columba_chunk <index base> <reads file: one sequence per line> <k> [sa_sparseness]
// prints:  <read#> <begin> <end> <distance> <strand>
#include "columba_amd.hpp"

#include <iostream>

using namespace columba_amd;

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " <index base> <reads.txt> <k> [sa sparseness]\n";
        return 2;
    }
    try {
        const int sparse = argc > 4 ? atoi(argv[4]) : 4;
        FMIndex index(argv[1], 4, true, sparse, false, 10);
        MultipleSchemesStrategy* dummy = nullptr;
        (void)dummy;
        KucherovKPlus1 kuch(index, DYNAMIC, EDIT);
        NamedStrategy multiple(index, "multiple_opt", DYNAMIC, EDIT);
        const length_t k = (length_t)atoi(argv[3]);
        SearchStrategy& strategy = (k == 2 || k == 4 || k == 6) ? (SearchStrategy&)multiple : (SearchStrategy&)kuch;
        std::vector<ReadBundle> chunk;
        std::ifstream f(argv[2]);
        std::string line;
        while (std::getline(f, line))
            if (!line.empty()) chunk.emplace_back("r" + std::to_string(chunk.size()), line);
        Counters counters;
        std::vector<std::vector<TextOcc>> matches;
        strategy.matchApproxBatch(chunk, k, counters, matches);
        for (size_t i = 0; i < matches.size(); i++)
            for (const auto& o : matches[i])
                std::cout << i << ' ' << o.getBegin() << ' ' << o.getEnd() << ' ' << o.getDistance() << ' '
                          << (o.isRevCompl() ? 1 : 0) << "\n";
        // reads not longer than the number of parts of the scheme: matched by naive backtracking, as the reference does
        for (size_t i : strategy.matchedNaively)
            std::cerr << "read " << i << " (" << chunk[i].getRead().size() << " characters) was matched by naive backtracking\n";
        std::cerr << "nodes " << counters.get(Counters::NODE_COUNTER) << "\n";
    } catch (const std::exception& e) {
        std::cerr << "Fatal error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
