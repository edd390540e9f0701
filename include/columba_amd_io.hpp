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
#if defined(__has_include)
#if __has_include(<zlib.h>)
#include <zlib.h> // gz-compressed read files (the reference: HAVE_ZLIB, seqfile.cpp); link with -lz
#define COLUMBA_AMD_HAVE_ZLIB 1
#endif
#endif

namespace columba_amd {

// A read file as a stream of lines, plain or gz-compressed by its extension (SeqFile, seqfile.{h,cpp}: ".gz" selects zlib)
class LineSource {
    std::ifstream in;
#ifdef COLUMBA_AMD_HAVE_ZLIB
    gzFile gz = nullptr;
#endif
    bool compressed = false;
    std::string buf;
    size_t pos = 0;
    bool exhausted = false;

    bool fill() { // more bytes into buf; false at the end of the file
        if (exhausted) return false;
        buf.erase(0, pos);
        pos = 0;
        char tmp[1 << 16];
        long n = 0;
        if (compressed) {
#ifdef COLUMBA_AMD_HAVE_ZLIB
            n = gzread(gz, tmp, sizeof(tmp));
            if (n < 0) throw std::ios::failure("error while reading a gz-compressed file");
#endif
        } else {
            in.read(tmp, sizeof(tmp));
            n = (long)in.gcount();
        }
        if (n <= 0) {
            exhausted = true;
            return false;
        }
        buf.append(tmp, (size_t)n);
        return true;
    }

  public:
    explicit LineSource(const std::string& file) {
        compressed = file.size() > 3 && file.compare(file.size() - 3, 3, ".gz") == 0;
        if (compressed) {
#ifdef COLUMBA_AMD_HAVE_ZLIB
            gz = gzopen(file.c_str(), "rb");
            if (!gz) throw std::runtime_error("Cannot open file: " + file);
#else
            throw std::runtime_error("gz-compressed input needs zlib (built without it): " + file);
#endif
        } else {
            in.open(file, std::ios::binary);
            if (!in) throw std::runtime_error("Cannot open file: " + file);
        }
    }
    LineSource(const LineSource&) = delete;
    ~LineSource() {
#ifdef COLUMBA_AMD_HAVE_ZLIB
        if (gz) gzclose(gz);
#endif
    }
    int peek() { // the next character, EOF at the end
        if (pos >= buf.size() && !fill()) return EOF;
        return (unsigned char)buf[pos];
    }
    bool getline(std::string& line) { // without the newline; false at the end of the file
        line.clear();
        for (;;) {
            const size_t nl = buf.find('\n', pos);
            if (nl != std::string::npos) {
                line.append(buf, pos, nl - pos);
                pos = nl + 1;
                return true;
            }
            line.append(buf, pos, std::string::npos);
            pos = buf.size();
            if (!fill()) return !line.empty();
        }
    }
    bool good() { return peek() != EOF; }
};

struct SequenceRecord { // fastq.h SequenceRecord: identifier line as read (with @ or >), sequence, quality ("*" for FASTA)
    std::string seqID, read, qual;
};

class Reader { // fastq.h Reader (single-end; plain text or .gz)
    LineSource in;
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
            in.getline(dummy);
        }
    }

  public:
    explicit Reader(const std::string& file) : in(file), fileName(file) {}
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
            in.getline(r.seqID);
            in.getline(r.read);
            in.getline(plus);
            in.getline(r.qual);
            chomp(r.seqID);
            chomp(r.read);
            chomp(r.qual);
            return !r.read.empty();
        }
        if (c != '>') throw std::ios::failure("File " + fileName + " doesn't appear to be in Fasta format");
        in.getline(r.seqID); // fastq.cpp:101-146
        chomp(r.seqID);
        std::string line;
        while (in.good() && in.peek() != '>' && in.peek() != EOF) {
            in.getline(line);
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

class OutputWriter { // fastq.h OutputWriter: SAM header, then the chunks in the order of their ids; ".sam" or ".sam.gz" (fastq.cpp:466-493)
    std::ofstream out;
#ifdef COLUMBA_AMD_HAVE_ZLIB
    gzFile gz = nullptr;
#endif
    std::map<size_t, std::string> pending;
    size_t nextChunkID = 0;
    void put(const std::string& s) {
#ifdef COLUMBA_AMD_HAVE_ZLIB
        if (gz) {
            for (size_t o = 0; o < s.size();) { // (gzwrite takes an unsigned count)
                const unsigned n = (unsigned)(s.size() - o < (1u << 30) ? s.size() - o : (1u << 30));
                if (gzwrite(gz, s.data() + o, n) != (int)n) throw std::ios::failure("error while writing a gz-compressed file");
                o += n;
            }
            return;
        }
#endif
        out << s;
    }

  public:
    OutputWriter(const std::string& file, const std::string& headerFile, const std::string& commandLine) {
        std::string ext = file.size() >= 7 ? file.substr(file.size() - 7) : std::string();
        for (auto& c : ext) c = (char)toupper((unsigned char)c);
        if (ext == ".SAM.GZ") {
#ifdef COLUMBA_AMD_HAVE_ZLIB
            gz = gzopen(file.c_str(), "wb");
            if (!gz) throw std::runtime_error("Cannot open file: " + file);
#else
            throw std::runtime_error("gz-compressed output needs zlib (built without it): " + file);
#endif
        } else {
            out.open(file);
            if (!out) throw std::runtime_error("Cannot open file: " + file);
        }
        put("@HD\tVN:1.6\tSO:queryname\n"); // fastq.cpp:579-583
        put("@PG\tID:Columba-amd\tPN:Columba\tCL:" + commandLine + "\n");
        std::ifstream hs(headerFile, std::ios::binary);
        std::string line;
        while (hs && std::getline(hs, line)) put(line + "\n");
    }
    OutputWriter(const OutputWriter&) = delete;
    ~OutputWriter() {
#ifdef COLUMBA_AMD_HAVE_ZLIB
        if (gz) gzclose(gz);
#endif
    }
    void commitChunk(size_t id, std::string&& text) { // chunks may arrive out of order; they leave in order
        pending.emplace(id, std::move(text));
        for (auto it = pending.begin(); it != pending.end() && it->first == nextChunkID; it = pending.erase(it), nextChunkID++)
            put(it->second);
    }
    void flush() {
#ifdef COLUMBA_AMD_HAVE_ZLIB
        if (gz) {
            gzflush(gz, Z_SYNC_FLUSH);
            return;
        }
#endif
        out.flush();
    }
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
