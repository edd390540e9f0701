// Exact matching of a chunk of reads on the run-length compressed (b-move) index through the C++ adapter — the k = 0 branch of
// SearchStrategy::matchApproxAllMap (reference src/searchstrategy.cpp:499-510) — followed by a walk that exercises the
// extension and locate calls with the reference's method names.
//   usage: bmove_exact <index base> <reads file: one sequence per line> [k [strategy [k-mer size [text file [best <x> <identity> | pairs <mates file> <x> <identity> <max fragment>]]]]]
// with k > 0: the approximate search of SearchStrategy::matchApproxAllMap (searchstrategy.cpp:495-535, RLC flavour) instead
// prints:  <read#> <begin> <end> <distance> <strand>      (stdout)
//          nodes <NODE_COUNTER>; walk <depth> <width> <positions found>   (stderr)
#include "columba_amd_bmove.hpp"

#include <algorithm>
#include <iostream>

using namespace columba_amd::rlc;

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "usage: " << argv[0] << " <index base> <reads.txt>\n";
        return 2;
    }
    try {
        BMove index(argv[1]);
        std::vector<std::string> chunk;
        std::ifstream f(argv[2]);
        std::string line;
        while (std::getline(f, line)) chunk.push_back(line);
        std::vector<std::vector<TextOcc>> matches;
        const int k = argc > 3 ? atoi(argv[3]) : 0;
        uint64_t nodes = 0;
        if (k > 0 && argc > 6) { // ... <text file>: the SAM records of the chunk (the text beside the index serves CIGARs and trimming)
            std::ifstream tf(argv[6], std::ios::binary);
            const std::string text((std::istreambuf_iterator<char>(tf)), std::istreambuf_iterator<char>());
            index.attachText(text);
            SearchStrategy strategy(index, argv[4], CMB_PARTITION_DYNAMIC, CMB_METRIC_EDIT, atoi(argv[5]));
            std::vector<std::string> ids, quals;
            for (size_t i = 0; i < chunk.size(); i++) ids.push_back("r" + std::to_string(i)), quals.push_back(std::string(chunk[i].size(), 'I'));
            if (argc > 11 && std::string(argv[7]) == "pairs") { // ... pairs <mates file> <x> <min identity> <max fragment>: read pairs in BEST mode (FR)
                std::vector<std::string> mates, ids2, quals2;
                std::ifstream mf(argv[8]);
                while (std::getline(mf, line)) mates.push_back(line);
                for (size_t i = 0; i < mates.size(); i++) ids2.push_back("r" + std::to_string(i) + "/2"), quals2.push_back(std::string(mates[i].size(), 'I'));
                for (size_t i = 0; i < ids.size(); i++) ids[i] += "/1";
                size_t mapped = 0;
                std::cout << strategy.samOfChunkPairedBest(ids, chunk, quals, ids2, mates, quals2, {"seq0"}, (uint32_t)atoi(argv[9]), (uint32_t)atoi(argv[10]),
                                                           CMB_ORIENTATION_FR, (uint32_t)atoi(argv[11]), 0, true, true, mapped);
                std::cerr << "mapped pairs " << mapped << "\n";
                return 0;
            }
            if (argc > 9 && std::string(argv[7]) == "best") { // ... best <x> <min identity>: BEST (+x strata) mode, the reference's default
                size_t nMapped = 0;
                std::cout << strategy.samOfChunkBest(ids, chunk, quals, {"seq0"}, (uint32_t)atoi(argv[8]), (uint32_t)atoi(argv[9]), true, false, nMapped);
                std::cerr << "mapped " << nMapped << "\n";
                return 0;
            }
            std::cout << strategy.samOfChunk(ids, chunk, quals, {"seq0"}, (length_t)k);
            return 0;
        }
        if (k > 0) {
            SearchStrategy strategy(index, argc > 4 ? argv[4] : "multiple_opt", CMB_PARTITION_DYNAMIC, CMB_METRIC_EDIT, argc > 5 ? atoi(argv[5]) : 10);
            std::vector<uint64_t> counters;
            strategy.matchApproxBatch(chunk, (length_t)k, counters, matches);
            nodes = counters[CMB_CNT_NODE];
        } else {
            nodes = index.exactMatchesOutput(chunk, matches);
        }
        for (size_t i = 0; i < matches.size(); i++)
            for (const auto& o : matches[i])
                std::cout << i << ' ' << o.getBegin() << ' ' << o.getEnd() << ' ' << o.getDistance() << ' ' << (o.isRevCompl() ? 1 : 0) << "\n";
        std::cerr << "nodes " << nodes << "\n";
        // the first read once more, character by character from its middle: right with ...Forward, then left with ...Backward
        if (k == 0 && !chunk.empty() && chunk[0].size() >= 2) {
            const std::string& s = chunk[0];
            auto code = [](char c) -> length_t { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 0; };
            SARangePair cur = index.getCompleteRange(), next;
            const size_t mid = s.size() / 2;
            bool alive = true;
            for (size_t i = mid; alive && i < s.size(); i++) alive = code(s[i]) && index.findRangesWithExtraCharForward(code(s[i]), cur, next), cur = alive ? next : cur;
            for (size_t i = mid; alive && i-- > 0;) alive = code(s[i]) && index.findRangesWithExtraCharBackward(code(s[i]), cur, next), cur = alive ? next : cur;
            if (alive) {
                std::vector<length_t> pos;
                index.getTextPositionsFromSARange(cur, pos);
                std::sort(pos.begin(), pos.end());
                std::cerr << "walk " << cur.getOriginalDepth() << ' ' << cur.width() << ' ' << pos.size();
                for (auto p : pos) std::cerr << ' ' << p;
                std::cerr << "\n";
            } else {
                std::cerr << "walk dead\n";
            }
        }
    } catch (const std::exception& e) {
        std::cerr << "Fatal error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
