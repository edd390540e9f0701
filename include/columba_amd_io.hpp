// Host-side I/O around the C-ABI: a FASTA / FASTQ chunk reader, an ordered SAM writer and the chunk loop of a
// Columba-style aligner.  Host C++ only; all matching happens behind include/columba_amd.h.
//
// Mirrors (reference, src/):
//   SequenceRecord::readFromFileFASTQ / readFromFileFASTA   fastq.cpp:43-146 (records; multi-line FASTA; "*" quality)
//   Reader::getNextChunk                                    fastq.cpp:395      (chunks of records)
//   OutputWriter::writerThread (header + ordered chunks)    fastq.cpp:567-660  (@HD, @PG, the @SQ lines of <base>.headerSN.bin)
//   processChunk / threadEntrySingleEnd                     parallel.cpp:67-111
#pragma once
#include "columba_amd.h"

#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace columba_amd {

struct SequenceRecord { // fastq.h SequenceRecord: identifier line as read (with @ or >), sequence, quality ("*" for FASTA)
    std::string seqID, read, qual;
};

class Reader { // fastq.h Reader (single-end; plain text)
    std::ifstream in;
    std::string fileName;
    bool fastq = false;
    bool typeKnown = false;

    static void chomp(std::string& s) {
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
    }
    bool skipBlankLines() {
        for (;;) {
            const int c = in.peek();
            if (c == EOF) return false;
            if (c != '\n' && c != '\r') return true;
            std::string dummy;
            std::getline(in, dummy);
        }
    }

  public:
    explicit Reader(const std::string& file) : in(file), fileName(file) {
        if (!in) throw std::runtime_error("Cannot open file: " + file);
    }
    // one record; false at the end of the file
    bool next(SequenceRecord& r) {
        r.seqID.clear();
        r.read.clear();
        r.qual.clear();
        if (!skipBlankLines()) return false;
        const int c = in.peek();
        if (!typeKnown) {
            fastq = c == '@';
            typeKnown = true;
        }
        if (fastq) { // fastq.cpp:43-99
            if (c != '@') throw std::ios::failure("File " + fileName + " doesn't appear to be in FastQ format");
            std::string plus;
            std::getline(in, r.seqID);
            std::getline(in, r.read);
            std::getline(in, plus);
            std::getline(in, r.qual);
            chomp(r.seqID);
            chomp(r.read);
            chomp(r.qual);
            return !r.read.empty();
        }
        if (c != '>') throw std::ios::failure("File " + fileName + " doesn't appear to be in Fasta format");
        std::getline(in, r.seqID); // fastq.cpp:101-146
        chomp(r.seqID);
        std::string line;
        while (in.good() && in.peek() != '>' && in.peek() != EOF) {
            std::getline(in, line);
            chomp(line);
            r.read += line;
        }
        r.qual = "*";
        return !r.read.empty();
    }
    // up to n records; false when the file is exhausted and nothing was read
    bool getNextChunk(std::vector<SequenceRecord>& chunk, size_t n) {
        chunk.clear();
        SequenceRecord r;
        while (chunk.size() < n && next(r)) chunk.push_back(r);
        return !chunk.empty();
    }
};

class OutputWriter { // fastq.h OutputWriter: SAM header, then the chunks in the order of their ids
    std::ofstream out;
    std::map<size_t, std::string> pending;
    size_t nextChunkID = 0;

  public:
    OutputWriter(const std::string& file, const std::string& headerFile, const std::string& commandLine) : out(file) {
        if (!out) throw std::runtime_error("Cannot open file: " + file);
        out << "@HD\tVN:1.6\tSO:queryname\n"; // fastq.cpp:579-583
        out << "@PG\tID:Columba-amd\tPN:Columba\tCL:" << commandLine << "\n";
        std::ifstream hs(headerFile, std::ios::binary);
        std::string line;
        while (hs && std::getline(hs, line)) out << line << "\n";
    }
    void commitChunk(size_t id, std::string&& text) { // chunks may arrive out of order; they leave in order
        pending.emplace(id, std::move(text));
        for (auto it = pending.begin(); it != pending.end() && it->first == nextChunkID; it = pending.erase(it), nextChunkID++)
            out << it->second;
    }
    void flush() { out.flush(); }
};

// sequence names of an index: <base>.sna (size_t length + bytes per name, indexinterface.cpp:175-195)
inline std::vector<std::string> readSequenceNames(const std::string& base) {
    std::vector<std::string> names;
    std::ifstream f(base + ".sna", std::ios::binary);
    while (f) {
        size_t len = 0;
        f.read(reinterpret_cast<char*>(&len), sizeof(len));
        if (!f) break;
        std::string s(len, '\0');
        f.read(&s[0], (std::streamsize)len);
        if (!f) break;
        names.push_back(s);
    }
    return names;
}

} // namespace columba_amd
