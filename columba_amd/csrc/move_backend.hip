// C-ABI of the run-length compressed backend (include/columba_amd.h, section "b-move") on top of move_dev.hpp.
// gfx950 only.  A translation unit of its own, linked into libcolumba_amd.so.
#include "../../include/columba_amd.h"
#include "host_schemes.hpp"
#include "move_search.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace cmb {
int failWith(int code, const std::string& msg); // columba_amd.hip: sets the calling thread's cmb_last_error
}
using namespace cmb;

#define MV_HIPCHK(expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace {

// a growing host array in PAGE-LOCKED memory (the occurrence records of a batch: 10^6 reads of BASELINE configs[4] leave 1.15 GB — into
// a std::vector the device-to-host copy went through bounce buffers and the resize zeroed every record first); contents are kept when
// it grows, the capacity is kept between runs
template <typename T> struct MvHostVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    MvHostVec() {}
    MvHostVec(const MvHostVec&) = delete;
    MvHostVec& operator=(const MvHostVec&) = delete;
    ~MvHostVec() {
        if (p) (void)hipHostFree(p);
    }
    void resize(size_t count) {
        if (count > cap) {
            const size_t want = count + count / 2 + 1024;
            T* q = nullptr;
            MV_HIPCHK(hipHostMalloc((void**)&q, want * sizeof(T), hipHostMallocDefault));
            if (n) memcpy(q, p, n * sizeof(T));
            if (p) (void)hipHostFree(p);
            p = q;
            cap = want;
        }
        n = count;
    }
    void assign(const std::vector<T>& v) {
        n = 0;
        resize(v.size());
        if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    }
    void clear() { n = 0; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
    const T* data() const { return p; }
    const T& operator[](size_t i) const { return p[i]; }
};

template <typename T> struct MvBuf {
    T* p = nullptr;
    size_t n = 0;
    MvBuf() {}
    MvBuf(const MvBuf&) = delete;
    MvBuf& operator=(const MvBuf&) = delete;
    ~MvBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        MV_HIPCHK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
        n = count;
    }
    void upload(const T* h, size_t count) {
        alloc(count);
        if (count) MV_HIPCHK(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
    }
    size_t bytes() const { return n * sizeof(T); }
};

unsigned gridFor(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256 + 1, 256 * 32); }

struct PosSetHost {
    MvBuf<uint64_t> pos, dir;
    uint32_t shift = 0;
    uint64_t count = 0;
    PosSet dev() const { return PosSet{pos.p, count, dir.p, shift}; }
    // elements below `limit`, strictly increasing; about eight elements per directory bucket
    void build(const uint64_t* h, uint64_t cnt, uint64_t limit, uint32_t* dFlags) {
        count = cnt;
        pos.upload(h, cnt);
        hipLaunchKernelGGL(k_posset_check, dim3(gridFor(cnt)), dim3(256), 0, 0, pos.p, cnt, limit, dFlags);
        shift = 0;
        while (shift < 40 && (limit >> shift) > std::max<uint64_t>(cnt / 8, 1)) shift++;
        const uint64_t buckets = (limit >> shift) + 1;
        dir.alloc(buckets + 1);
        hipLaunchKernelGGL(k_posset_dir, dim3(gridFor(buckets + 1)), dim3(256), 0, 0, pos.p, cnt, shift, buckets, dir.p);
        MV_HIPCHK(hipGetLastError());
    }
    size_t bytes() const { return pos.bytes() + dir.bytes(); }
};

struct TableHost {
    MvBuf<uint4> rows;
    MvBuf<uint64_t> smpF, smpL;
    uint64_t runs = 0, zeroCharPos = 0;
    MoveTable dev() const { return MoveTable{rows.p, runs, zeroCharPos, smpF.p, smpL.p}; }
    size_t bytes() const { return rows.bytes() + smpF.bytes() + smpL.bytes(); }
};

uint32_t bitsFor(double v) { return (uint32_t)std::ceil(std::log2(v)); } // moverepr.h:44-46

} // namespace

namespace cmb { // (columba_amd.hip: the text kernels and k_cigar live in that translation unit)
int moveCigarsOnText(cmb_index* textIndex, hipStream_t s, const uint64_t* offs, const uint32_t* G, uint32_t gw, uint32_t nReads, uint32_t maxLen,
                     uint32_t k, int gapless, const void* occs, const uint32_t* occRead, uint64_t nOcc, void* aln, uint16_t* ops, uint32_t stride,
                     uint32_t* flagWord);
const std::vector<uint32_t>& seqStartsOfIndex(const cmb_index* textIndex);
} // namespace cmb

struct cmb_move_index {
    int device = 0;
    uint64_t n = 0;
    TableHost tab[2];
    bool hasLocate = false;
    PosSetHost predFirst, predLast, plcpPos;
    MvBuf<uint64_t> firstToRun, lastToRun, plcpSum;
    MoveDev d{};
    // the k-mer table of the search (populateTable), built on first use for the word size asked for
    MvBuf<MoveRangeRec> kmer;
    uint32_t kmerSize = 0;
    std::mutex kmerMutex;
    // optional: the text itself (cmb_move_attach_text), for the CIGARs of the occurrences — codes and a 2-bit copy
    cmb_index* textIndex = nullptr; // (cmb_index_create_text_only: codes, 2-bit copy, sequence starts)
    ~cmb_move_index() {
        if (textIndex) cmb_index_destroy(textIndex);
    }
};

static void bindMoveDev(cmb_move_index* ix) { // MoveDev pointers from the owning buffers
    ix->d = MoveDev{};
    ix->d.n = ix->n;
    ix->d.fwd = ix->tab[0].dev();
    ix->d.rev = ix->tab[1].dev();
    if (ix->hasLocate) {
        ix->d.predFirst = ix->predFirst.dev();
        ix->d.predLast = ix->predLast.dev();
        ix->d.plcpPos = ix->plcpPos.dev();
        ix->d.firstToRun = ix->firstToRun.p;
        ix->d.lastToRun = ix->lastToRun.p;
        ix->d.plcpSum = ix->plcpSum.p;
    }
}

// one .LFBP file (moverepr.cpp:103-168) -> 16-byte rows in HBM, checked
static void loadTable(TableHost& t, const uint8_t* file, uint64_t fileBytes, uint32_t lengthBits, const uint64_t* smpF,
                      const uint64_t* smpL, uint64_t& nOut, uint32_t* dFlags, const char* what) {
    const size_t W = lengthBits / 8;
    if (fileBytes < 3 * W) throw std::invalid_argument(std::string(what) + ": shorter than its header");
    uint64_t hdr[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) std::memcpy(&hdr[i], file + i * W, W);
    const uint64_t n = hdr[0], runs = hdr[1];
    if (n < 2 || runs < 2 || runs > n) throw std::invalid_argument(std::string(what) + ": implausible text size / number of runs");
    const uint32_t bitsN = bitsFor((double)n), bitsR = bitsFor((double)runs);
    if (bitsN > 40) throw std::length_error(std::string(what) + ": texts of 2^40 characters and more are not supported");
    const uint32_t rowBytes = (3 + 2 * bitsN + bitsR + 7) / 8;
    const uint64_t body = (uint64_t)rowBytes * (runs + 1);
    if (fileBytes < 3 * W + body) throw std::invalid_argument(std::string(what) + ": truncated (" + std::to_string(fileBytes) + " bytes, " + std::to_string(3 * W + body) + " expected)");
    if (hdr[2] >= n) throw std::invalid_argument(std::string(what) + ": position of the sentinel outside the text");
    MvBuf<uint8_t> packed;
    packed.upload(file + 3 * W, body);
    t.rows.alloc(runs + 2);
    MV_HIPCHK(hipMemset(t.rows.p + runs + 1, 0xFF, sizeof(uint4)));
    hipLaunchKernelGGL(k_move_unpack, dim3(gridFor(runs + 1)), dim3(256), 0, 0, packed.p, runs + 1, rowBytes, bitsN, bitsR, t.rows.p);
    hipLaunchKernelGGL(k_move_gaps, dim3(gridFor(runs)), dim3(256), 0, 0, t.rows.p, runs);
    hipLaunchKernelGGL(k_move_check, dim3(gridFor(runs + 1)), dim3(256), 0, 0, t.rows.p, runs, n, dFlags);
    MV_HIPCHK(hipGetLastError());
    MV_HIPCHK(hipDeviceSynchronize());
    t.runs = runs;
    t.zeroCharPos = hdr[2];
    t.smpF.upload(smpF, runs);
    t.smpL.upload(smpL, runs);
    nOut = n;
}

extern "C" int cmb_move_create(const cmb_move_desc* desc, int device, cmb_move_index** out) {
    if (!desc || !out || !desc->lfbp || !desc->rev_lfbp || !desc->samples_first || !desc->samples_last || !desc->rev_samples_first ||
        !desc->rev_samples_last)
        return failWith(CMB_ERR_INVALID, "bad argument");
    if (desc->length_bits != 64 && desc->length_bits != 32) return failWith(CMB_ERR_INVALID, "length_bits must be 64 or 32");
    const bool loc = desc->pred_first || desc->first_to_run || desc->pred_last || desc->last_to_run || desc->plcp_pos || desc->plcp_sum;
    if (loc && !(desc->pred_first && desc->first_to_run && desc->pred_last && desc->last_to_run && desc->plcp_pos && desc->plcp_sum && desc->n_plcp))
        return failWith(CMB_ERR_INVALID, "the locate arrays come together or not at all");
    *out = nullptr;
    try {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
            return failWith(CMB_ERR_DEVICE, "no such GPU (the move tables live in HBM; there is no CPU path)");
        MV_HIPCHK(hipSetDevice(device));
        std::unique_ptr<cmb_move_index> ix(new cmb_move_index());
        ix->device = device;
        MvBuf<uint32_t> flags;
        flags.alloc(4);
        MV_HIPCHK(hipMemset(flags.p, 0, 4 * sizeof(uint32_t)));
        uint64_t nF = 0, nR = 0;
        loadTable(ix->tab[0], desc->lfbp, desc->lfbp_bytes, desc->length_bits, desc->samples_first, desc->samples_last, nF, flags.p, ".LFBP");
        loadTable(ix->tab[1], desc->rev_lfbp, desc->rev_lfbp_bytes, desc->length_bits, desc->rev_samples_first, desc->rev_samples_last, nR,
                  flags.p + 1, ".rev.LFBP");
        if (nF != nR) return failWith(CMB_ERR_INVALID, "the two move tables are of different texts");
        ix->n = nF;
        if (loc) {
            const uint64_t r = ix->tab[0].runs;
            ix->predFirst.build(desc->pred_first, r, ix->n, flags.p + 2);
            ix->predLast.build(desc->pred_last, r, ix->n, flags.p + 2);
            ix->plcpPos.build(desc->plcp_pos, desc->n_plcp, ix->n, flags.p + 2);
            ix->firstToRun.upload(desc->first_to_run, r);
            ix->lastToRun.upload(desc->last_to_run, r);
            ix->plcpSum.upload(desc->plcp_sum, desc->n_plcp);
            hipLaunchKernelGGL(k_run_map_check, dim3(gridFor(r)), dim3(256), 0, 0, ix->firstToRun.p, r, r, flags.p + 2);
            hipLaunchKernelGGL(k_run_map_check, dim3(gridFor(r)), dim3(256), 0, 0, ix->lastToRun.p, r, r, flags.p + 2);
            if (desc->plcp_pos[0] != 0) return failWith(CMB_ERR_INVALID, "the PLCP samples must start at position 0");
            ix->hasLocate = true;
        }
        MV_HIPCHK(hipGetLastError());
        uint32_t h[4];
        MV_HIPCHK(hipMemcpy(h, flags.p, sizeof(h), hipMemcpyDeviceToHost));
        if (h[0] || h[1])
            return failWith(CMB_ERR_INVALID, "inconsistent move table (" + std::to_string(h[0]) + " rows of .LFBP, " + std::to_string(h[1]) +
                                                 " rows of .rev.LFBP break the order of the runs, the LF targets or the terminating row)");
        if (h[2]) return failWith(CMB_ERR_INVALID, "inconsistent locate arrays (positions not increasing / outside the text, or run numbers outside the table)");
        bindMoveDev(ix.get());
        *out = ix.release();
        return CMB_OK;
    } catch (const std::invalid_argument& e) {
        return failWith(CMB_ERR_INVALID, e.what());
    } catch (const std::length_error& e) {
        return failWith(CMB_ERR_UNSUPPORTED, e.what());
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" void cmb_move_destroy(cmb_move_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    delete idx;
}

// The text beside the index (optional): what the alignments of the occurrences, the trimming at sequence ends and the SAM records are
// computed on (a text-only cmb_index: cmb_index_create_text_only).  seq_starts / n_seqs as in cmb_index_desc (the starts and the final
// n - 1), or NULL.
extern "C" int cmb_move_attach_text(cmb_move_index* idx, const char* text, uint64_t n, const uint32_t* seq_starts, uint32_t n_seqs) {
    if (!idx || !text) return failWith(CMB_ERR_INVALID, "null argument");
    if (n != idx->n && n + 1 != idx->n) return failWith(CMB_ERR_INVALID, "the text does not have the length of the indexed text");
    if (idx->n >= 0xFFFFFF00ull) return failWith(CMB_ERR_UNSUPPORTED, "alignments on texts of 2^32 characters and more are not implemented");
    if (idx->textIndex) cmb_index_destroy(idx->textIndex);
    idx->textIndex = nullptr;
    std::string t(text, text + n);
    if (n + 1 == idx->n) t.push_back('$');
    return cmb_index_create_text_only(t.data(), t.size(), seq_starts, n_seqs, idx->device, &idx->textIndex);
}
extern "C" cmb_index* cmb_move_text_index(const cmb_move_index* idx) { return idx ? idx->textIndex : nullptr; }

extern "C" uint64_t cmb_move_device_bytes(const cmb_move_index* idx) {
    if (!idx) return 0;
    return idx->tab[0].bytes() + idx->tab[1].bytes() + idx->predFirst.bytes() + idx->predLast.bytes() + idx->plcpPos.bytes() +
           idx->firstToRun.bytes() + idx->lastToRun.bytes() + idx->plcpSum.bytes();
}

extern "C" int cmb_move_info(const cmb_move_index* idx, uint64_t* text_length, uint64_t* runs, uint64_t* rev_runs) {
    if (!idx) return failWith(CMB_ERR_INVALID, "bad argument");
    if (text_length) *text_length = idx->n;
    if (runs) *runs = idx->tab[0].runs;
    if (rev_runs) *rev_runs = idx->tab[1].runs;
    return CMB_OK;
}

// BMove::getCompleteRange (bmove.h:369-373) with getInitialToehold (bmove.h:139-142)
extern "C" int cmb_move_complete_range(const cmb_move_index* idx, cmb_move_range* out) {
    if (!idx || !out) return failWith(CMB_ERR_INVALID, "bad argument");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        uint64_t last = 0;
        MV_HIPCHK(hipMemcpy(&last, idx->tab[0].smpL.p + idx->tab[0].runs - 1, sizeof(last), hipMemcpyDeviceToHost));
        std::memset(out, 0, sizeof(*out));
        out->begin = 0, out->end = idx->n, out->begin_run = 0, out->end_run = idx->tab[0].runs - 1;
        out->rev_begin = 0, out->rev_end = idx->n, out->rev_begin_run = 0, out->rev_end_run = idx->tab[1].runs - 1;
        out->toehold = last - 1;
        out->runs_valid = out->rev_runs_valid = 1;
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_move_rows(const cmb_move_index* idx, int rev, uint64_t first, uint64_t count, uint64_t* out) {
    if (!idx || !out || rev < 0 || rev > 1 || first + count > idx->tab[rev].runs + 1) return failWith(CMB_ERR_INVALID, "bad argument");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        std::vector<uint4> h(count);
        if (count) MV_HIPCHK(hipMemcpy(h.data(), idx->tab[rev].rows.p + first, count * sizeof(uint4), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < count; i++) {
            const MoveRow r = unpackMoveRow(h[i]);
            out[4 * i] = r.head, out[4 * i + 1] = r.in, out[4 * i + 2] = r.out, out[4 * i + 3] = r.outRun;
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

static_assert(sizeof(cmb_move_range) == sizeof(MoveRangeRec), "cmb_move_range layout");

extern "C" int cmb_move_extend_batch(const cmb_move_index* idx, int mode, const cmb_move_range* parents, uint64_t n, cmb_move_range* children,
                                     uint8_t* ok) {
    if (!idx || mode < 0 || mode > 2 || (n && (!parents || !children || !ok))) return failWith(CMB_ERR_INVALID, "bad argument");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        MvBuf<MoveRangeRec> din, dout;
        MvBuf<uint8_t> dok;
        MvBuf<uint32_t> bad;
        din.upload((const MoveRangeRec*)parents, n);
        dout.alloc(4 * n);
        dok.alloc(4 * n);
        bad.alloc(1);
        MV_HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
        if (n) hipLaunchKernelGGL(k_move_extend, dim3(gridFor(n)), dim3(256), 0, 0, idx->d, mode, din.p, n, dout.p, dok.p, bad.p);
        MV_HIPCHK(hipGetLastError());
        uint32_t hb = 0;
        MV_HIPCHK(hipMemcpy(&hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost));
        if (hb)
            return failWith(CMB_ERR_INVALID, std::to_string(hb) + " of the parents are not ranges of this index (empty, outside the text, or "
                                                                  "with run indices that do not enclose them)");
        if (n) {
            MV_HIPCHK(hipMemcpy(children, dout.p, 4 * n * sizeof(MoveRangeRec), hipMemcpyDeviceToHost));
            MV_HIPCHK(hipMemcpy(ok, dok.p, 4 * n, hipMemcpyDeviceToHost));
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_move_extend_bench(const cmb_move_index* idx, int mode, const void* d_parents, uint64_t n, void* d_children, void* d_ok,
                                     uint32_t iters, float* avg_ms) {
    if (!idx || mode < 0 || mode > 2 || !d_parents || !d_children || !d_ok || !avg_ms || !iters || !n) return failWith(CMB_ERR_INVALID, "bad argument");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        hipStream_t s;
        MV_HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t a, b;
        MV_HIPCHK(hipEventCreate(&a));
        MV_HIPCHK(hipEventCreate(&b));
        MvBuf<uint32_t> bad;
        bad.alloc(1);
        MV_HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
        auto launch = [&]() {
            hipLaunchKernelGGL(k_move_extend, dim3(gridFor(n)), dim3(256), 0, s, idx->d, mode, (const MoveRangeRec*)d_parents, n,
                               (MoveRangeRec*)d_children, (uint8_t*)d_ok, bad.p);
        };
        launch(); // warm-up
        MV_HIPCHK(hipEventRecord(a, s));
        for (uint32_t i = 0; i < iters; i++) launch();
        MV_HIPCHK(hipEventRecord(b, s));
        MV_HIPCHK(hipEventSynchronize(b));
        float ms = 0;
        MV_HIPCHK(hipEventElapsedTime(&ms, a, b));
        *avg_ms = ms / iters;
        MV_HIPCHK(hipEventDestroy(a));
        MV_HIPCHK(hipEventDestroy(b));
        MV_HIPCHK(hipStreamDestroy(s));
        uint32_t hb = 0;
        MV_HIPCHK(hipMemcpy(&hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost));
        if (hb) return failWith(CMB_ERR_INVALID, "parents that are not ranges of this index");
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_move_locate_batch(const cmb_move_index* idx, const cmb_move_range* ranges, uint64_t n, const uint64_t* offsets,
                                     uint64_t* positions) {
    if (!idx || (n && (!ranges || !offsets || !positions))) return failWith(CMB_ERR_INVALID, "bad argument");
    if (!idx->hasLocate) return failWith(CMB_ERR_INVALID, "this index was created without the locate arrays");
    for (uint64_t i = 0; i < n; i++)
        if (ranges[i].end <= ranges[i].begin || offsets[i + 1] - offsets[i] != ranges[i].end - ranges[i].begin || offsets[i + 1] < offsets[i])
            return failWith(CMB_ERR_INVALID, "offsets[i + 1] - offsets[i] must be the width of range i");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        MvBuf<MoveRangeRec> din;
        MvBuf<uint64_t> doff, dpos;
        MvBuf<uint32_t> bad;
        din.upload((const MoveRangeRec*)ranges, n);
        doff.upload(offsets, n + 1);
        const uint64_t total = n ? offsets[n] - offsets[0] : 0;
        dpos.alloc(total);
        bad.alloc(1);
        MV_HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
        if (n) hipLaunchKernelGGL(k_move_locate, dim3(gridFor(n)), dim3(256), 0, 0, idx->d, din.p, n, doff.p, offsets[0], dpos.p, bad.p, false);
        MV_HIPCHK(hipGetLastError());
        uint32_t hb = 0;
        MV_HIPCHK(hipMemcpy(&hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost));
        if (hb)
            return failWith(CMB_ERR_INVALID, std::to_string(hb) + " of the ranges do not belong to this index (toehold, depth and width "
                                                                  "do not describe one suffix array interval)");
        if (total) MV_HIPCHK(hipMemcpy(positions + offsets[0], dpos.p, total * sizeof(uint64_t), hipMemcpyDeviceToHost));
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

static thread_local float g_exactMs[3] = {0, 0, 0};
static thread_local unsigned long long g_exactExpansions = 0; // extensions attempted by the calling thread's last cmb_move_match_exact
// device time of the calling thread's last cmb_move_match_exact: [0] k_move_exact, [1] prefix sum, [2] k_move_locate + k_move_occ
extern "C" int cmb_move_last_timings(float* ms, uint32_t n) {
    if (!ms) return failWith(CMB_ERR_INVALID, "bad argument");
    for (uint32_t i = 0; i < n && i < 3; i++) ms[i] = g_exactMs[i];
    return CMB_OK;
}

// k = 0 on the b-move index (SearchStrategy::matchApproxAllMap with maxED = 0, searchstrategy.cpp:499-510)
extern "C" int cmb_move_match_exact(const cmb_move_index* idx, const char* reads, const uint64_t* read_offsets, uint64_t n_reads,
                                    cmb_move_occ* occ_out, uint64_t occ_cap, uint64_t* occ_offsets, uint64_t* n_occ, uint64_t* counters) {
    if (!idx || !read_offsets || !n_occ || (n_reads && !reads) || (occ_cap && !occ_out)) return failWith(CMB_ERR_INVALID, "bad argument");
    if (!idx->hasLocate) return failWith(CMB_ERR_INVALID, "this index was created without the locate arrays");
    static_assert(sizeof(cmb_move_occ) == sizeof(MoveOccRec), "cmb_move_occ layout");
    if (n_reads >= (1ull << 30)) return failWith(CMB_ERR_UNSUPPORTED, "2^30 reads and more per call");
    for (uint64_t i = 0; i < n_reads; i++)
        if (read_offsets[i + 1] < read_offsets[i]) return failWith(CMB_ERR_INVALID, "read offsets must not decrease");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        const uint64_t nTasks = 2 * n_reads, nChars = read_offsets[n_reads] - read_offsets[0];
        MvBuf<uint8_t> dReads;
        MvBuf<uint64_t> dOff, dWidth, dTaskOff, dPos, dReadOcc;
        MvBuf<MoveRangeRec> dRanges;
        MvBuf<unsigned long long> dNodes;
        MvBuf<uint32_t> bad;
        dReads.upload((const uint8_t*)reads + read_offsets[0], nChars);
        std::vector<uint64_t> off(read_offsets, read_offsets + n_reads + 1);
        for (auto& o : off) o -= read_offsets[0];
        dOff.upload(off.data(), n_reads + 1);
        dWidth.alloc(nTasks + 1);
        dTaskOff.alloc(nTasks + 1);
        dRanges.alloc(nTasks);
        dNodes.alloc(2);
        bad.alloc(1);
        MV_HIPCHK(hipMemset(dNodes.p, 0, 2 * sizeof(unsigned long long)));
        MV_HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
        MV_HIPCHK(hipMemset(dWidth.p, 0, (nTasks + 1) * sizeof(uint64_t)));
        hipEvent_t ev[4];
        for (auto& e : ev) MV_HIPCHK(hipEventCreate(&e));
        struct EvGuard {
            hipEvent_t* e;
            ~EvGuard() {
                for (int i = 0; i < 4; i++) (void)hipEventDestroy(e[i]);
            }
        } evGuard{ev};
        g_exactMs[0] = g_exactMs[1] = g_exactMs[2] = 0;
        MV_HIPCHK(hipEventRecord(ev[0], 0));
        if (nTasks) hipLaunchKernelGGL(k_move_exact, dim3(gridFor(nTasks)), dim3(256), 0, 0, idx->d, dReads.p, dOff.p, nTasks, dRanges.p, dWidth.p, dNodes.p);
        MV_HIPCHK(hipGetLastError());
        MV_HIPCHK(hipEventRecord(ev[1], 0));
        // offsets of the tasks' occurrences: exclusive prefix sum over nTasks + 1 widths (the last one is zero)
        size_t tmpBytes = 0;
        MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, dWidth.p, dTaskOff.p, (int)(nTasks + 1)));
        MvBuf<uint8_t> tmp;
        tmp.alloc(tmpBytes);
        MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmpBytes, dWidth.p, dTaskOff.p, (int)(nTasks + 1)));
        MV_HIPCHK(hipEventRecord(ev[2], 0));
        uint64_t total = 0;
        MV_HIPCHK(hipMemcpy(&total, dTaskOff.p + nTasks, sizeof(total), hipMemcpyDeviceToHost));
        MV_HIPCHK(hipEventElapsedTime(&g_exactMs[0], ev[0], ev[1]));
        MV_HIPCHK(hipEventElapsedTime(&g_exactMs[1], ev[1], ev[2]));
        unsigned long long nodes = 0, nodes2[2] = {0, 0};
        MV_HIPCHK(hipMemcpy(nodes2, dNodes.p, sizeof(nodes2), hipMemcpyDeviceToHost));
        nodes = nodes2[0];
        g_exactExpansions = nodes2[1];
        *n_occ = total;
        if (counters) counters[0] = nodes, counters[1] = total;
        if (occ_offsets) {
            dReadOcc.alloc(n_reads + 1);
            hipLaunchKernelGGL(k_move_read_offsets, dim3(gridFor(n_reads + 1)), dim3(256), 0, 0, dTaskOff.p, n_reads, dReadOcc.p);
            MV_HIPCHK(hipMemcpy(occ_offsets, dReadOcc.p, (n_reads + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        }
        if (total > occ_cap)
            return failWith(CMB_ERR_OVERFLOW, std::to_string(total) + " occurrences, room for " + std::to_string(occ_cap) + " (n_occ holds the number needed)");
        if (total) {
            dPos.alloc(total);
            MvBuf<MoveOccRec> dOcc;
            dOcc.alloc(total);
            MV_HIPCHK(hipEventRecord(ev[2], 0));
            hipLaunchKernelGGL(k_move_locate, dim3(gridFor(nTasks)), dim3(256), 0, 0, idx->d, dRanges.p, nTasks, dTaskOff.p, (uint64_t)0, dPos.p, bad.p, true);
            hipLaunchKernelGGL(k_move_occ, dim3(gridFor(total)), dim3(256), 0, 0, dPos.p, dTaskOff.p, nTasks, total, dOff.p, dOcc.p);
            MV_HIPCHK(hipGetLastError());
            MV_HIPCHK(hipEventRecord(ev[3], 0));
            MV_HIPCHK(hipEventSynchronize(ev[3]));
            MV_HIPCHK(hipEventElapsedTime(&g_exactMs[2], ev[2], ev[3]));
            uint32_t hb = 0;
            MV_HIPCHK(hipMemcpy(&hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost));
            if (hb) return failWith(CMB_ERR_INTERNAL, std::to_string(hb) + " ranges whose phi chains do not have the width of the range (inconsistent locate arrays)");
            MV_HIPCHK(hipMemcpy(occ_out, dOcc.p, total * sizeof(MoveOccRec), hipMemcpyDeviceToHost));
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

// IndexInterface::populateTable of the RLC flavour (indexinterface.cpp:294-335)
extern "C" int cmb_move_kmer_table(const cmb_move_index* idx, uint32_t word_size, cmb_move_range* out) {
    if (!idx || !out || word_size > 12) return failWith(CMB_ERR_INVALID, "bad argument (k-mer size up to 12)");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        const uint64_t total = 1ull << (2 * word_size);
        MvBuf<MoveRangeRec> d;
        d.alloc(total);
        hipLaunchKernelGGL(k_move_kmer_table, dim3(gridFor(total)), dim3(256), 0, 0, idx->d, word_size, d.p);
        MV_HIPCHK(hipGetLastError());
        MV_HIPCHK(hipMemcpy(out, d.p, total * sizeof(MoveRangeRec), hipMemcpyDeviceToHost));
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

// ---- replication of a built index on other GPUs: the DEVICE layout travels (as for the FM-index, cmb_index_layout_of ...) ----
namespace {
struct MvArrayRef {
    void** p;
    size_t* n;
    size_t elem;
};
void moveArrays(cmb_move_index* ix, MvArrayRef out[CMB_MOVE_DEV_ARRAYS]) {
    int k = 0;
    for (int t = 0; t < 2; t++) {
        out[k++] = {(void**)&ix->tab[t].rows.p, &ix->tab[t].rows.n, sizeof(uint4)};
        out[k++] = {(void**)&ix->tab[t].smpF.p, &ix->tab[t].smpF.n, sizeof(uint64_t)};
        out[k++] = {(void**)&ix->tab[t].smpL.p, &ix->tab[t].smpL.n, sizeof(uint64_t)};
    }
    PosSetHost* sets[3] = {&ix->predFirst, &ix->predLast, &ix->plcpPos};
    for (auto* ps : sets) {
        out[k++] = {(void**)&ps->pos.p, &ps->pos.n, sizeof(uint64_t)};
        out[k++] = {(void**)&ps->dir.p, &ps->dir.n, sizeof(uint64_t)};
    }
    out[k++] = {(void**)&ix->firstToRun.p, &ix->firstToRun.n, sizeof(uint64_t)};
    out[k++] = {(void**)&ix->lastToRun.p, &ix->lastToRun.n, sizeof(uint64_t)};
    out[k++] = {(void**)&ix->plcpSum.p, &ix->plcpSum.n, sizeof(uint64_t)};
}
} // namespace

extern "C" int cmb_move_layout_of(const cmb_move_index* idx, cmb_move_layout* out) {
    if (!idx || !out) return failWith(CMB_ERR_INVALID, "bad argument");
    std::memset(out, 0, sizeof(*out));
    out->text_length = idx->n;
    for (int t = 0; t < 2; t++) out->runs[t] = idx->tab[t].runs, out->zero_char_pos[t] = idx->tab[t].zeroCharPos;
    out->has_locate = idx->hasLocate;
    const PosSetHost* sets[3] = {&idx->predFirst, &idx->predLast, &idx->plcpPos};
    for (int i = 0; i < 3; i++) out->set_count[i] = sets[i]->count, out->set_shift[i] = sets[i]->shift;
    MvArrayRef a[CMB_MOVE_DEV_ARRAYS];
    moveArrays(const_cast<cmb_move_index*>(idx), a);
    for (int i = 0; i < CMB_MOVE_DEV_ARRAYS; i++) out->bytes[i] = *a[i].p ? (uint64_t)(*a[i].n * a[i].elem) : 0;
    return CMB_OK;
}

extern "C" int cmb_move_create_empty(const cmb_move_layout* L, int device, cmb_move_index** out) {
    if (!L || !out) return failWith(CMB_ERR_INVALID, "bad argument");
    if (L->text_length < 2 || L->text_length >= (1ull << 40) || L->runs[0] < 2 || L->runs[1] < 2) return failWith(CMB_ERR_INVALID, "move index layout out of range");
    *out = nullptr;
    try {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
            return failWith(CMB_ERR_DEVICE, "no such GPU (the move tables live in HBM; there is no CPU path)");
        MV_HIPCHK(hipSetDevice(device));
        std::unique_ptr<cmb_move_index> ix(new cmb_move_index());
        ix->device = device;
        ix->n = L->text_length;
        ix->hasLocate = L->has_locate != 0;
        for (int t = 0; t < 2; t++) ix->tab[t].runs = L->runs[t], ix->tab[t].zeroCharPos = L->zero_char_pos[t];
        PosSetHost* sets[3] = {&ix->predFirst, &ix->predLast, &ix->plcpPos};
        for (int i = 0; i < 3; i++) sets[i]->count = L->set_count[i], sets[i]->shift = L->set_shift[i];
        MvArrayRef a[CMB_MOVE_DEV_ARRAYS];
        moveArrays(ix.get(), a);
        for (int i = 0; i < CMB_MOVE_DEV_ARRAYS; i++) {
            if (L->bytes[i] % a[i].elem) return failWith(CMB_ERR_INVALID, "move index layout: array size is not a whole number of elements");
            if (L->bytes[i] == 0) continue; // absent (an index without the locate arrays)
            MV_HIPCHK(hipMalloc(a[i].p, L->bytes[i]));
            *a[i].n = L->bytes[i] / a[i].elem;
        }
        // sizes must fit what the kernels index
        bool okSizes = true;
        for (int t = 0; t < 2; t++)
            okSizes = okSizes && ix->tab[t].rows.n >= ix->tab[t].runs + 2 && ix->tab[t].smpF.n >= ix->tab[t].runs && ix->tab[t].smpL.n >= ix->tab[t].runs;
        if (ix->hasLocate) {
            for (int i = 0; i < 3; i++)
                okSizes = okSizes && sets[i]->shift < 41 && sets[i]->pos.n >= sets[i]->count && sets[i]->dir.n >= (ix->n >> sets[i]->shift) + 2 && sets[i]->count >= 1;
            okSizes = okSizes && ix->firstToRun.n >= ix->predFirst.count && ix->lastToRun.n >= ix->predLast.count && ix->plcpSum.n >= ix->plcpPos.count &&
                      ix->predFirst.count == ix->tab[0].runs && ix->predLast.count == ix->tab[0].runs;
        }
        if (!okSizes) return failWith(CMB_ERR_INVALID, "move index layout: array sizes do not fit the numbers of runs / samples");
        bindMoveDev(ix.get());
        *out = ix.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_move_device_arrays(cmb_move_index* idx, void** ptrs, uint64_t* bytes) {
    if (!idx || !ptrs || !bytes) return failWith(CMB_ERR_INVALID, "bad argument");
    MvArrayRef a[CMB_MOVE_DEV_ARRAYS];
    moveArrays(idx, a);
    for (int i = 0; i < CMB_MOVE_DEV_ARRAYS; i++) {
        ptrs[i] = *a[i].p;
        bytes[i] = *a[i].p ? (uint64_t)(*a[i].n * a[i].elem) : 0;
    }
    return CMB_OK;
}

// the checks of cmb_move_create on arrays that arrived through a collective
extern "C" int cmb_move_validate(cmb_move_index* idx) {
    if (!idx) return failWith(CMB_ERR_INVALID, "bad argument");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        MvBuf<uint32_t> flags;
        flags.alloc(4);
        MV_HIPCHK(hipMemset(flags.p, 0, 4 * sizeof(uint32_t)));
        for (int t = 0; t < 2; t++)
            hipLaunchKernelGGL(k_move_check, dim3(gridFor(idx->tab[t].runs + 1)), dim3(256), 0, 0, idx->tab[t].rows.p, idx->tab[t].runs, idx->n, flags.p + t);
        if (idx->hasLocate) {
            const PosSetHost* sets[3] = {&idx->predFirst, &idx->predLast, &idx->plcpPos};
            for (auto* ps : sets) hipLaunchKernelGGL(k_posset_check, dim3(gridFor(ps->count)), dim3(256), 0, 0, ps->pos.p, ps->count, idx->n, flags.p + 2);
            const uint64_t r = idx->tab[0].runs;
            hipLaunchKernelGGL(k_run_map_check, dim3(gridFor(r)), dim3(256), 0, 0, idx->firstToRun.p, r, r, flags.p + 2);
            hipLaunchKernelGGL(k_run_map_check, dim3(gridFor(r)), dim3(256), 0, 0, idx->lastToRun.p, r, r, flags.p + 2);
            for (auto* ps : sets) hipLaunchKernelGGL(k_posset_dir_check, dim3(gridFor((idx->n >> ps->shift) + 2)), dim3(256), 0, 0, ps->pos.p, ps->count, ps->shift, (idx->n >> ps->shift) + 1, ps->dir.p, flags.p + 3);
        }
        MV_HIPCHK(hipGetLastError());
        uint32_t h[4];
        MV_HIPCHK(hipMemcpy(h, flags.p, sizeof(h), hipMemcpyDeviceToHost));
        if (idx->hasLocate) { // plcpAt reads plcpSum[rank - 1]: position 0 must be a PLCP run start (cmb_move_create checks the same)
            uint64_t first = ~0ull;
            if (idx->plcpPos.count) MV_HIPCHK(hipMemcpy(&first, idx->plcpPos.pos.p, sizeof(first), hipMemcpyDeviceToHost));
            if (first != 0) return failWith(CMB_ERR_INVALID, "inconsistent move index arrays (the PLCP run starts do not begin at position 0)");
        }
        if (h[0] || h[1] || h[2] || h[3])
            return failWith(CMB_ERR_INVALID, "inconsistent move index arrays (" + std::to_string(h[0]) + " / " + std::to_string(h[1]) + " table rows, " +
                                                 std::to_string(h[2]) + " locate entries, " + std::to_string(h[3]) + " directory entries)");
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}


// =====================================================================================================================
// The approximate search on the b-move index (move_search.hpp): SearchStrategy::matchApprox in ALL mode for a chunk of
// reads (searchstrategy.cpp:495-535 with the RUN_LENGTH_COMPRESSION branches).  BASELINE.json configs[4].
// =====================================================================================================================
struct cmb_move_batch {
    cmb_move_index* ix = nullptr;
    uint32_t k = 0, nReads = 0, maxLen = 0, gw = 0, kmerSize = 0;
    int metric = 1;
    DevStrategyK hostStrat{};
    // more than 8 parts (the greedy schemes under Hamming distance at 8 ... 13 errors): the wide tables, as on the FM-index
    bool wide = false;
    bool noSmallMatrix = false; // a phase of an earlier run did not fit the 32-bit in-index matrix (GeoN32): this batch stays on GeoN
    DevStrategyKT<MAXP_WIDE> hostStratW{};
    MvBuf<DevStrategyKT<MAXP_WIDE>> stratW;
    MvBuf<PartOutT<MAXP_WIDE>> partsW;
    uint32_t sNumParts = 0, sPartition = 0, sNSchemes = 0, sMaxSearches = 0;
    hipStream_t stream = nullptr;
    std::vector<uint8_t> hostReads; // (k = 0 goes through cmb_move_match_exact, which takes host buffers)
    std::vector<uint64_t> hostOffs;
    MvBuf<uint8_t> reads, seq, psel, sortTmp;
    MvBuf<uint64_t> offs;
    MvBuf<uint32_t> G, cnt, bfsCnt, fmIdxA, fmIdxB, keep, slot, vals, valsB, bad;
    MvBuf<DevStrategyK> strat;
    MvBuf<PartOut> parts;
    MvBuf<MoveRangeRec> exr, locRanges;
    MvBuf<MvTask> tasks;
    MvBuf<uint4> Q[2], Ev[2], F, C, A, locMeta;
    MvBuf<unsigned long long> counters, blockCnt, keysA, keysB;
    MvBuf<MvFmRec> fm;
    MvBuf<uint64_t> widths, locWidths, locOff, positions, readCnt, readOff;
    MvBuf<MoveOccOut> out;
    size_t qCap = 0, evCap = 0, fCap = 0, cCap = 0, aCap = 0;
    // naive backtracking (k_mvs_naive): node double buffer, nodes per pass, the survivors of its own filter pass per read x strand
    MvBuf<uint4> nvQ[2];
    MvBuf<uint32_t> nvCnt;
    size_t nvQCap = 0;
    MvBuf<MoveOccOut> naiveOut;
    MvBuf<uint64_t> naiveOff;
    // BEST mode's strata (cmb_move_match_best): every strand of a read filtered by itself, as mapRead does (searchstrategy.h:490-523)
    bool perStrand = false;
    MvBuf<uint64_t> rsOff;
    // alignments of the final occurrences (cmb_move_batch_want_alignments; needs cmb_move_attach_text)
    bool wantAln = false;
    uint32_t alnStride = 0;
    MvBuf<uint4> occ32, alnRec;
    MvBuf<uint32_t> occRead;
    MvBuf<uint16_t> alnOps;
    std::vector<uint4> hAlnRec; // {seqId, seqBegin, nOps, spans} per occurrence
    std::vector<uint16_t> hAlnOps;
    // results
    MvHostVec<cmb_move_occ> occs;
    std::vector<uint64_t> occOffs;
    uint64_t cnts[CMB_CNT_MAX];
    std::vector<std::pair<const char*, float>> times;
    bool done = false;
    // the occurrence records of a slice travel to the host on a stream of their own while the next slice is matched
    hipStream_t copyStream = nullptr;
    hipEvent_t outReady = nullptr, copyDone = nullptr;
    bool copyPending = false;
    void waitForCopies() {
        if (copyPending) (void)hipStreamSynchronize(copyStream);
        copyPending = false;
    }
    // A large chunk is matched as two halves side by side, each a batch of its own (pools, streams, host thread): the frontier search of
    // one half is a chain of dependent round trips that leaves line rate unused, which the other half's partitioning, locate and filter
    // kernels take (the FM-index batches do the same with three sub-batches).  The parent holds no device memory.
    std::vector<cmb_move_batch*> subs;
    std::vector<uint32_t> subLo;
    ~cmb_move_batch() {
        for (cmb_move_batch* c : subs) delete c;
        waitForCopies(); // (the records land in page-locked memory this object frees)
        if (copyStream) (void)hipStreamDestroy(copyStream);
        if (outReady) (void)hipEventDestroy(outReady);
        if (copyDone) (void)hipEventDestroy(copyDone);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static int ensureKmerTable(cmb_move_index* ix, uint32_t ws) {
    std::lock_guard<std::mutex> lock(ix->kmerMutex);
    if (ix->kmerSize == ws && ix->kmer.p) return CMB_OK;
    if (ws < 1 || ws > 12) return failWith(CMB_ERR_INVALID, "k-mer size of the b-move search must be 1 .. 12");
    const uint64_t entries = 1ull << (2 * ws);
    ix->kmer.alloc(entries);
    hipLaunchKernelGGL(k_move_kmer_table, dim3(gridFor(entries)), dim3(256), 0, 0, ix->d, ws, ix->kmer.p);
    MV_HIPCHK(hipGetLastError());
    MV_HIPCHK(hipDeviceSynchronize());
    ix->kmerSize = ws;
    return CMB_OK;
}

static int moveBatchCreateOne(cmb_move_index* idx, const cmb_strategy* st, uint32_t max_distance, uint32_t kmer_size, const char* seqs,
                              const uint64_t* offs, uint32_t n_reads, cmb_move_batch** out);

extern "C" int cmb_move_batch_create(cmb_move_index* idx, const cmb_strategy* st, uint32_t max_distance, uint32_t kmer_size, const char* seqs,
                                     const uint64_t* offs, uint32_t n_reads, cmb_move_batch** out) {
    if (!idx || !st || !offs || !out || (!seqs && n_reads)) return failWith(CMB_ERR_INVALID, "null argument");
    // two halves from 2^19 reads (two full slices at 5 errors and more); CMB_MOVE_SUBBATCHES=n overrides (1: one batch)
    // (measured on the configs[4] stand-in, 10^6 x 250 bp at 6 errors: 1 / 2 / 3 parts 737 / 675 / 668 ms per step; four parts of 2^18 reads
    // in flight ran out of the 288 GB — two halves hold their pools twice, so they are only taken with most of the HBM free)
    uint32_t S = max_distance > 0 && n_reads >= (1u << 19) ? 2u : 1u;
    if (S > 1) {
        size_t freeB = 0, totalB = 0;
        if (hipSetDevice(idx->device) != hipSuccess || hipMemGetInfo(&freeB, &totalB) != hipSuccess || freeB < (size_t)160 << 30) S = 1;
    }
    if (const char* e = getenv("CMB_MOVE_SUBBATCHES")) S = max_distance > 0 ? (uint32_t)std::min(4, std::max(1, atoi(e))) : 1u;
    if (S > n_reads) S = 1;
    if (S == 1) return moveBatchCreateOne(idx, st, max_distance, kmer_size, seqs, offs, n_reads, out);
    if (n_reads >= (1u << 23)) return failWith(CMB_ERR_UNSUPPORTED, "2^23 reads and more per b-move batch (24-bit read numbers in the filter keys)");
    std::unique_ptr<cmb_move_batch> parent(new cmb_move_batch());
    parent->ix = idx, parent->k = max_distance, parent->nReads = n_reads, parent->metric = st->metric, parent->kmerSize = kmer_size;
    for (uint32_t j = 0; j < S; j++) {
        const uint32_t lo = (uint32_t)((uint64_t)n_reads * j / S), hi = (uint32_t)((uint64_t)n_reads * (j + 1) / S);
        cmb_move_batch* c = nullptr;
        const int rc = moveBatchCreateOne(idx, st, max_distance, kmer_size, seqs, offs + lo, hi - lo, &c); // (offsets are rebased there)
        if (rc != CMB_OK) return rc;
        parent->subs.push_back(c);
        parent->subLo.push_back(lo);
    }
    *out = parent.release();
    return CMB_OK;
}

static int moveBatchCreateOne(cmb_move_index* idx, const cmb_strategy* st, uint32_t max_distance, uint32_t kmer_size, const char* seqs,
                              const uint64_t* offs, uint32_t n_reads, cmb_move_batch** out) {
    if (!idx || !st || !offs || !out || (!seqs && n_reads)) return failWith(CMB_ERR_INVALID, "null argument");
    if (!idx->hasLocate) return failWith(CMB_ERR_INVALID, "this index was created without the locate arrays");
    if (n_reads >= (1u << 23)) return failWith(CMB_ERR_UNSUPPORTED, "2^23 reads and more per b-move batch (24-bit read numbers in the filter keys)");
    if (idx->n >> 40) return failWith(CMB_ERR_UNSUPPORTED, "texts of 2^40 characters and more");
    try {
        MV_HIPCHK(hipSetDevice(idx->device));
        std::unique_ptr<cmb_move_batch> b(new cmb_move_batch());
        b->ix = idx;
        b->k = max_distance;
        b->nReads = n_reads;
        b->metric = st->metric;
        b->kmerSize = kmer_size;
        if (max_distance > 0) {
            if (max_distance > 13) return failWith(CMB_ERR_UNSUPPORTED, "more than 13 errors (MAX_K, definitions.h:50)");
            try {
                // (edit distance beyond 7 errors: the wide record geometries of the frontier, GeoW / GeoX, on the wide tables)
                b->wide = st->numPartsFor(max_distance) > (uint32_t)MAXP || (st->metric == CMB_METRIC_EDIT && max_distance > 7);
                if (b->wide) {
                    b->hostStratW = st->flatten<MAXP_WIDE>(max_distance);
                    b->sNumParts = b->hostStratW.numParts, b->sPartition = b->hostStratW.partition, b->sNSchemes = b->hostStratW.nSchemes;
                    for (int i = 0; i < b->hostStratW.nSchemes; i++) b->sMaxSearches = std::max<uint32_t>(b->sMaxSearches, b->hostStratW.sch[i].nSearches);
                } else {
                    b->hostStrat = st->flatten(max_distance);
                    b->sNumParts = b->hostStrat.numParts, b->sPartition = b->hostStrat.partition, b->sNSchemes = b->hostStrat.nSchemes;
                    for (int i = 0; i < b->hostStrat.nSchemes; i++) b->sMaxSearches = std::max<uint32_t>(b->sMaxSearches, b->hostStrat.sch[i].nSearches);
                }
            } catch (const std::exception& e) {
                return failWith(CMB_ERR_INVALID, e.what());
            }
            const int rc = ensureKmerTable(idx, kmer_size);
            if (rc != CMB_OK) return rc;
        }
        uint32_t maxLen = 1;
        for (uint32_t i = 0; i < n_reads; i++) {
            if (offs[i + 1] < offs[i]) return failWith(CMB_ERR_INVALID, "read offsets must be non-decreasing");
            maxLen = std::max<uint32_t>(maxLen, (uint32_t)(offs[i + 1] - offs[i]));
        }
        if (maxLen > (uint32_t)MAX_READ) return failWith(CMB_ERR_UNSUPPORTED, "reads longer than " + std::to_string(MAX_READ) + " are not supported");
        maxLen = (maxLen + 15u) & ~15u;
        b->maxLen = maxLen;
        b->gw = gWords(maxLen);
        b->hostOffs.assign(offs, offs + n_reads + 1);
        for (auto& o : b->hostOffs) o -= offs[0];
        b->hostReads.assign((const uint8_t*)seqs + offs[0], (const uint8_t*)seqs + offs[n_reads]);
        MV_HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        MV_HIPCHK(hipStreamCreateWithFlags(&b->copyStream, hipStreamNonBlocking));
        MV_HIPCHK(hipEventCreateWithFlags(&b->outReady, hipEventDisableTiming));
        MV_HIPCHK(hipEventCreateWithFlags(&b->copyDone, hipEventDisableTiming));
        b->reads.upload(b->hostReads.data(), b->hostReads.size());
        b->offs.upload(b->hostOffs.data(), n_reads + 1);
        if (max_distance > 0) { // (per-read scratch is sized per slice, at the first run)
            if (b->wide) b->stratW.upload(&b->hostStratW, 1);
            else b->strat.upload(&b->hostStrat, 1);
        }
        b->cnt.alloc(8);
        b->counters.alloc(CMB_CNT_MAX);
        b->bad.alloc(1);
        *out = b.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" void cmb_move_batch_destroy(cmb_move_batch* b) { delete b; }

namespace {
struct MvTimer {
    hipStream_t s;
    hipEvent_t a, b;
    std::vector<std::pair<const char*, float>>& out;
    MvTimer(hipStream_t st, std::vector<std::pair<const char*, float>>& o) : s(st), out(o) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
    }
    ~MvTimer() {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
    void begin() { (void)hipEventRecord(a, s); }
    void end(const char* name) {
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        for (auto& t : out)
            if (!strcmp(t.first, name)) {
                t.second += ms;
                return;
            }
        out.push_back({name, ms});
    }
};
} // namespace

static int runSlice(cmb_move_batch* b, uint32_t lo, uint32_t hi);

// The pools of the frontier search hold up to ~100 KB per read at k = 6 / 250 bp (600 final-column records of 96 bytes, 90
// contexts of 512 bytes): a large chunk is matched as consecutive SLICES that reuse one set of pools (CMB_MOVE_SLICE=n reads per
// slice; results and counters are those of the whole chunk).
static int moveBatchRunOne(cmb_move_batch* b);

extern "C" int cmb_move_batch_run(cmb_move_batch* b) {
    if (!b) return failWith(CMB_ERR_INVALID, "null argument");
    if (b->subs.empty()) return moveBatchRunOne(b);
    b->done = false;
    std::vector<int> rc(b->subs.size(), CMB_OK);
    std::vector<std::string> err(b->subs.size());
    // CMB_MOVE_SERIAL_SUBBATCHES: one part after the other — every kernel has the device to itself, which is what per-kernel timings want
    // (bench.py times its steps concurrently and takes the kernel table from extra serial steps, as for the FM-index batches)
    if (getenv("CMB_MOVE_SERIAL_SUBBATCHES")) {
        for (size_t j = 0; j < b->subs.size(); j++) {
            rc[j] = moveBatchRunOne(b->subs[j]);
            if (rc[j] != CMB_OK) return rc[j];
        }
    } else {
        std::vector<std::thread> th;
        for (size_t j = 0; j < b->subs.size(); j++)
            th.emplace_back([&, j] {
                rc[j] = moveBatchRunOne(b->subs[j]);
                if (rc[j] != CMB_OK) err[j] = cmb_last_error(); // (the message lives in the worker's thread-local storage)
            });
        for (auto& t : th) t.join();
    }
    for (size_t j = 0; j < rc.size(); j++)
        if (rc[j] != CMB_OK)
            return failWith(rc[j], err[j] + (err[j].find("out of memory") != std::string::npos
                                                 ? " (the chunk is matched as " + std::to_string(b->subs.size()) + " concurrent parts with pools of their own: "
                                                   "CMB_MOVE_SUBBATCHES=1 matches it as one batch)"
                                                 : ""));
    memset(b->cnts, 0, sizeof(b->cnts));
    b->times.clear();
    for (const cmb_move_batch* c : b->subs) {
        for (int i = 0; i < CMB_CNT_MAX; i++) b->cnts[i] += c->cnts[i];
        for (const auto& t : c->times) { // busy time per kernel group, summed over the halves (they overlap on the device)
            bool found = false;
            for (auto& u : b->times)
                if (!strcmp(u.first, t.first)) u.second += t.second, found = true;
            if (!found) b->times.push_back(t);
        }
    }
    b->done = true;
    return CMB_OK;
}

static int moveBatchRunOne(cmb_move_batch* b) {
    b->done = false;
    b->waitForCopies(); // (a run that failed half-way may have left one behind)
    b->times.clear();
    memset(b->cnts, 0, sizeof(b->cnts));
    b->occOffs.assign((size_t)b->nReads + 1, 0);
    b->occs.clear();
    b->hAlnRec.clear();
    b->hAlnOps.clear();
    b->alnStride = 2u * b->k + 3u;
    uint32_t slice = b->k >= 5 ? (1u << 18) : b->k >= 3 ? (1u << 19) : (1u << 20);
    if (getenv("CMB_MOVE_SLICE")) slice = (uint32_t)std::max(1, atoi(getenv("CMB_MOVE_SLICE")));
    if (b->k == 0) slice = b->nReads ? b->nReads : 1;
    if (b->nReads > slice) { // slices of equal size (a short last slice runs its levels on a mostly empty device)
        const uint32_t nSl = (b->nReads + slice - 1) / slice;
        slice = (b->nReads + nSl - 1) / nSl;
    }
    for (uint32_t lo = 0; lo < b->nReads; lo += slice) {
        const int rc = runSlice(b, lo, std::min<uint64_t>((uint64_t)lo + slice, b->nReads));
        if (rc != CMB_OK) {
            b->waitForCopies();
            return rc;
        }
    }
    b->waitForCopies();
    b->done = true;
    return CMB_OK;
}

static int runSlice(cmb_move_batch* b, uint32_t lo, uint32_t hi) {
    try {
        cmb_move_index* ix = b->ix;
        MV_HIPCHK(hipSetDevice(ix->device));
        hipStream_t s = b->stream;
        const uint32_t nReads = hi - lo, tasksRS = 2 * nReads;
        const uint64_t* dOffs = b->offs.p + lo; // (the kernels number the slice's reads from 0; offsets stay absolute)
        if (b->k == 0) { // exactMatchesOutput of both strands (searchstrategy.cpp:499-510): the k = 0 path of this backend
            uint64_t nOcc = 0, c2[2] = {0, 0};
            std::vector<cmb_move_occ> tmp((size_t)nReads * 8 + 1024);
            int rc = cmb_move_match_exact(ix, (const char*)b->hostReads.data(), b->hostOffs.data(), nReads, tmp.data(), tmp.size(), b->occOffs.data(), &nOcc, c2);
            if (rc == CMB_ERR_OVERFLOW) {
                tmp.resize(nOcc);
                rc = cmb_move_match_exact(ix, (const char*)b->hostReads.data(), b->hostOffs.data(), nReads, tmp.data(), tmp.size(), b->occOffs.data(), &nOcc, c2);
            }
            if (rc != CMB_OK) return rc;
            tmp.resize(nOcc);
            b->occs.assign(tmp);
            if (b->wantAln) { // an exact match aligns as <length>M; its sequence by position (findSeqName, indexinterface.cpp:799-832)
                const std::vector<uint32_t>& sp = cmb::seqStartsOfIndex(ix->textIndex);
                b->hAlnRec.resize(nOcc);
                b->hAlnOps.assign((size_t)nOcc * b->alnStride, 0);
                for (uint64_t i = 0; i < nOcc; i++) {
                    const cmb_move_occ& o = b->occs[i];
                    if (sp.size() < 2) { // one sequence
                        b->hAlnRec[i] = make_uint4(0u, (uint32_t)o.begin, 1u, 0u);
                    } else {
                        const uint32_t id = (uint32_t)(std::upper_bound(sp.begin(), sp.end() - 1, (uint32_t)o.begin) - sp.begin()) - 1;
                        b->hAlnRec[i] = make_uint4(id, (uint32_t)o.begin - sp[id], 1u, o.end > sp[id + 1] ? 1u : 0u);
                    }
                    b->hAlnOps[i * b->alnStride] = (uint16_t)(((o.end - o.begin) << 2) | 0u);
                }
            }
            b->cnts[CMB_CNT_NODE] = c2[0];
            b->cnts[CMB_CNT_EXPANSIONS] = g_exactExpansions;
            b->cnts[CMB_CNT_TOTAL_REPORTED] = c2[1];
            b->cnts[CMB_CNT_LOCATED_ROWS] = c2[1];
            float ms[3];
            (void)cmb_move_last_timings(ms, 3);
            b->times.push_back({"k_move_exact", ms[0]});
            b->times.push_back({"locate", ms[1] + ms[2]});
            return CMB_OK;
        }
        MvTimer tm(s, b->times);
        Queues q{};
        q.cnt = b->cnt.p;
        q.counters = b->counters.p;
        MvSearchIndex sx{ix->d, ix->kmer.p, ix->kmerSize};
        if (ix->kmerSize != b->kmerSize) {
            const int rc = ensureKmerTable(ix, b->kmerSize);
            if (rc != CMB_OK) return rc;
            sx.kmer = ix->kmer.p;
            sx.kmerSize = ix->kmerSize;
        }
        uint32_t hcnt[8];
        if ((b->wide ? b->partsW.n : b->parts.n) < (size_t)2 * nReads) { // per-read scratch of a slice
            b->seq.alloc((size_t)2 * nReads * b->maxLen);
            b->G.alloc((size_t)nReads * 8 * b->gw);
            if (b->wide) b->partsW.alloc((size_t)2 * nReads);
            else b->parts.alloc((size_t)2 * nReads);
            b->psel.alloc((size_t)2 * nReads);
            b->exr.alloc((size_t)2 * nReads * b->sNumParts);
            b->tasks.alloc((size_t)2 * nReads * b->sMaxSearches + 64);
        }
        // ---- read preparation
        tm.begin();
        MV_HIPCHK(hipMemsetAsync(b->G.p, 0, (size_t)nReads * 8 * b->gw * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_mvs_prep, dim3(gridFor((uint64_t)nReads * b->gw)), dim3(256), 0, s, b->reads.p, dOffs, nReads, b->maxLen, b->gw, b->seq.p, b->G.p);
        tm.end("k_prep");
        const uint32_t P = b->sNumParts, maxSearches = b->sMaxSearches;
        uint32_t nFm = 0;
        bool hasNaive = false; // reads of the slice are matched by naive backtracking (k_mvs_parts marked them in psel)
        for (int attempt = 0;; attempt++) {
            if (attempt >= 60) return failWith(CMB_ERR_INTERNAL, "work queues keep overflowing");
            MV_HIPCHK(hipMemsetAsync(b->cnt.p, 0, 8 * sizeof(uint32_t), s));
            MV_HIPCHK(hipMemsetAsync(b->counters.p, 0, CMB_CNT_MAX * sizeof(unsigned long long), s));
            if (!b->fm.n) b->fm.alloc((size_t)nReads * 16 + 4096);
            q.fmCap = (uint32_t)std::min<size_t>(b->fm.n, 0xFFFFFFF0u);
            // ---- prologue
            tm.begin();
            {
                const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)tasksRS + 63) / 64, 256 * 64);
                const uint64_t nWork = (uint64_t)tasksRS * maxSearches;
                const unsigned gridE = (unsigned)std::min<uint64_t>((nWork + 63) / 64, 256 * 64);
                const size_t exLdsBytes = (size_t)P * 3 * 64 * sizeof(uint4);
                if (b->wide) {
                    auto kp = b->sPartition == 0 ? k_mvs_parts<0, MAXP_WIDE> : b->sPartition == 1 ? k_mvs_parts<1, MAXP_WIDE> : k_mvs_parts<2, MAXP_WIDE>;
                    hipLaunchKernelGGL(kp, dim3(grid), dim3(64), exLdsBytes, s, sx, b->stratW.p, nReads, b->maxLen, b->seq.p, dOffs, b->partsW.p, b->exr.p,
                                       b->psel.p, q);
                    hipLaunchKernelGGL(k_mvs_exact<MAXP_WIDE>, dim3(gridE), dim3(64), 0, s, sx, b->stratW.p, nReads, b->maxLen, maxSearches, b->seq.p, b->partsW.p,
                                       b->exr.p, b->psel.p, b->tasks.p, (uint32_t)std::min<size_t>(b->tasks.n, 0xFFFFFFF0u), q);
                } else {
                    auto kp = b->sPartition == 0 ? k_mvs_parts<0, MAXP> : b->sPartition == 1 ? k_mvs_parts<1, MAXP> : k_mvs_parts<2, MAXP>;
                    hipLaunchKernelGGL(kp, dim3(grid), dim3(64), exLdsBytes, s, sx, b->strat.p, nReads, b->maxLen, b->seq.p, dOffs, b->parts.p, b->exr.p,
                                       b->psel.p, q);
                    hipLaunchKernelGGL(k_mvs_exact<MAXP>, dim3(gridE), dim3(64), 0, s, sx, b->strat.p, nReads, b->maxLen, maxSearches, b->seq.p, b->parts.p,
                                       b->exr.p, b->psel.p, b->tasks.p, (uint32_t)std::min<size_t>(b->tasks.n, 0xFFFFFFF0u), q);
                }
            }
            tm.end("k_partition");
            MV_HIPCHK(hipGetLastError());
            MV_HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            MV_HIPCHK(hipStreamSynchronize(s));
            hasNaive = (hcnt[3] & FLAG_UNSUPPORTED_READ) != 0;
            if (hcnt[3] & FLAG_SEED_OVERLAP)
                return failWith(CMB_ERR_INVALID, "dynamic partitioning: the seeds of a read overlap — the k-mer size is too large for the seeding "
                                                 "positions of this search strategy at this read length");
            if (hcnt[3] & FLAG_DFS_OVERFLOW) return failWith(CMB_ERR_INTERNAL, "task queue too small");
            if (hasNaive) {
                // ---- reads not longer than the number of parts / one-part strategies: naive backtracking (k_mvs_naive)
                tm.begin();
                const bool edit = b->metric == CMB_METRIC_EDIT;
                const uint32_t maxPassN = b->maxLen + 2 * b->k + 4;
                constexpr uint32_t PU = MvTraits::PAIR_U4;
                if (!b->nvQCap) b->nvQCap = getenv("CMB_TEST_SMALL_POOLS") ? 64 : 16384;
                for (int j = 0; j < 2; j++)
                    if (b->nvQ[j].n < (PU + 2) * b->nvQCap) b->nvQ[j].alloc((PU + 2) * b->nvQCap);
                const size_t cntWords = (size_t)maxPassN + 2;
                if (b->nvCnt.n < cntWords) b->nvCnt.alloc(cntWords);
                MV_HIPCHK(hipMemsetAsync(b->nvCnt.p, 0, cntWords * sizeof(uint32_t), s));
                MvHbfsBufs N{};
                N.Q[0] = b->nvQ[0].p;
                N.Q[1] = b->nvQ[1].p;
                N.qCap = (uint32_t)std::min<size_t>(b->nvQ[0].n / (PU + 2), 0xFFFFFFF0u);
                N.nq = b->nvCnt.p;
                N.blockCnt = nullptr;
                N.fmX = b->fm.p;
                auto kNaiveStart = edit ? k_mvs_naive<true, true, false> : k_mvs_naive<false, true, false>;
                auto kNaivePass = edit ? k_mvs_naive<true, false, false> : k_mvs_naive<false, false, false>;
                if (edit && b->k > MX_MAX_ED) kNaiveStart = k_mvs_naive<true, true, true>, kNaivePass = k_mvs_naive<true, false, true>;
                hipLaunchKernelGGL(kNaiveStart, dim3((tasksRS + 255) / 256), dim3(256), 0, s, ix->d, N, 0u,
                                   (const uint8_t*)b->psel.p, tasksRS, dOffs, b->gw, b->G.p, b->seq.p, b->maxLen, b->k, q);
                std::vector<uint32_t> hc(cntWords);
                uint32_t pass = 0, peakQ = 0;
                bool drained = false;
                while (!drained && pass < maxPassN) {
                    const uint32_t upTo = std::min(pass + 16u, maxPassN);
                    for (; pass < upTo; pass++)
                        hipLaunchKernelGGL(kNaivePass, dim3(BFS_GRID), dim3(256), 0, s, ix->d, N, pass,
                                           (const uint8_t*)b->psel.p, tasksRS, dOffs, b->gw, b->G.p, b->seq.p, b->maxLen, b->k, q);
                    MV_HIPCHK(hipMemcpyAsync(hc.data(), b->nvCnt.p, cntWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    MV_HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                    MV_HIPCHK(hipStreamSynchronize(s));
                    if (hcnt[3] & MVS_NAIVE_STOP) break;
                    drained = hc[pass] == 0;
                }
                for (uint32_t p2 = 0; p2 <= pass && p2 < cntWords; p2++) peakQ = std::max(peakQ, hc[p2]);
                tm.end("k_naive");
                MV_HIPCHK(hipGetLastError());
                if (hcnt[3] & MVS_NAIVE_STOP) {
                    // (the search stopped at the overflow: it has only counted what it needed up to there)
                    if (hcnt[3] & FLAG_NAIVE_Q) b->nvQCap = std::max<size_t>(4 * b->nvQCap, (size_t)peakQ + peakQ / 4);
                    if (hcnt[3] & FLAG_FMOCC_OVERFLOW) b->fm.alloc(std::max<size_t>(8 * b->fm.n, (size_t)hcnt[1] + hcnt[1] / 4 + 1024));
                    continue;
                }
                if (!drained) return failWith(CMB_ERR_INTERNAL, "the naive search did not finish within its pass bound");
            }
            const uint32_t nTasks = hcnt[5];
            if (nTasks && b->metric != CMB_METRIC_EDIT) {
                // ---- Hamming distance: the frontier without a matrix (k_mvs_hbfs), one row per pass
                tm.begin();
                const uint32_t maxPass = b->maxLen + 2 * (b->wide ? MAXP_WIDE : MAXP) + 16;
                constexpr uint32_t PU = MvTraits::PAIR_U4;
                if (!b->qCap) b->qCap = (getenv("CMB_TEST_SMALL_POOLS") ? 0 : (size_t)nReads * 8) + 1024;
                b->qCap = std::max<size_t>(b->qCap, (size_t)nTasks + 1024);
                for (int j = 0; j < 2; j++)
                    if (b->Q[j].n < (PU + 1) * b->qCap) b->Q[j].alloc((PU + 1) * b->qCap);
                const size_t cntWords = (size_t)maxPass + 2;
                if (b->bfsCnt.n < cntWords) b->bfsCnt.alloc(cntWords);
                if (b->blockCnt.n < (size_t)BFS_GRID * 4) b->blockCnt.alloc((size_t)BFS_GRID * 4);
                MV_HIPCHK(hipMemsetAsync(b->bfsCnt.p, 0, cntWords * sizeof(uint32_t), s));
                MV_HIPCHK(hipMemsetAsync(b->blockCnt.p, 0, (size_t)BFS_GRID * 4 * sizeof(unsigned long long), s));
                MvHbfsBufs H{};
                H.Q[0] = b->Q[0].p;
                H.Q[1] = b->Q[1].p;
                H.qCap = (uint32_t)std::min<size_t>(b->Q[0].n / (PU + 1), 0xFFFFFFF0u);
                H.nq = b->bfsCnt.p;
                H.blockCnt = b->blockCnt.p;
                H.fmX = b->fm.p;
                if (b->wide)
                    hipLaunchKernelGGL((k_mvs_hbfs<true, MAXP_WIDE>), dim3(std::min<uint32_t>((nTasks + 255) / 256, BFS_GRID)), dim3(256), 0, s, ix->d, b->stratW.p,
                                       H, 0u, b->tasks.p, nTasks, b->maxLen, b->seq.p, b->partsW.p, q);
                else
                    hipLaunchKernelGGL((k_mvs_hbfs<true, MAXP>), dim3(std::min<uint32_t>((nTasks + 255) / 256, BFS_GRID)), dim3(256), 0, s, ix->d, b->strat.p, H, 0u,
                                       b->tasks.p, nTasks, b->maxLen, b->seq.p, b->parts.p, q);
                std::vector<uint32_t> hc(cntWords);
                uint32_t pass = 0, peakQ = 0;
                bool drained = false;
                while (!drained && pass < maxPass) {
                    const uint32_t upTo = std::min(pass + 16u, maxPass);
                    for (; pass < upTo; pass++)
                        if (b->wide)
                            hipLaunchKernelGGL((k_mvs_hbfs<false, MAXP_WIDE>), dim3(BFS_GRID), dim3(256), 0, s, ix->d, b->stratW.p, H, pass, (const MvTask*)nullptr,
                                               0u, b->maxLen, b->seq.p, b->partsW.p, q);
                        else
                            hipLaunchKernelGGL((k_mvs_hbfs<false, MAXP>), dim3(BFS_GRID), dim3(256), 0, s, ix->d, b->strat.p, H, pass, (const MvTask*)nullptr, 0u,
                                               b->maxLen, b->seq.p, b->parts.p, q);
                    MV_HIPCHK(hipMemcpyAsync(hc.data(), b->bfsCnt.p, cntWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    MV_HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                    MV_HIPCHK(hipStreamSynchronize(s));
                    if (hcnt[3] & BFS_STOP) break;
                    drained = hc[pass] == 0;
                }
                for (uint32_t p2 = 0; p2 <= pass && p2 < cntWords; p2++) peakQ = std::max(peakQ, hc[p2]);
                MvBufs Bf{};
                Bf.blockCnt = b->blockCnt.p;
                hipLaunchKernelGGL(k_mvs_finish, dim3(1), dim3(256), 0, s, Bf, q);
                tm.end("k_dfs");
                MV_HIPCHK(hipGetLastError());
                if (hcnt[3] & (FLAG_BFS_Q | FLAG_FMOCC_OVERFLOW)) {
                    if (hcnt[3] & FLAG_BFS_Q) b->qCap = std::max<size_t>(2 * b->qCap, (size_t)peakQ + peakQ / 4);
                    if (hcnt[3] & FLAG_FMOCC_OVERFLOW) b->fm.alloc(std::max<size_t>(2 * b->fm.n, (size_t)hcnt[1] + hcnt[1] / 4 + 1024));
                    continue;
                }
                if (!drained) return failWith(CMB_ERR_INTERNAL, "frontier search did not finish within its pass bound");
            } else if (nTasks) {
                tm.begin();
                const uint32_t maxPass = 2 * b->maxLen + 8 * (b->wide ? MAXP_WIDE : MAXP) + 64;
                const bool geoX = b->wide && b->k > MX_MAX_ED; // (11 ... 13 errors: the in-index matrix with 16-row blocks)
                const uint32_t pkU4 = b->wide ? GeoW::PK_U4 : GeoN::PK_U4; // planes of a node's final-column pack / uint4 of an event's
                if (!b->qCap) {
                    const size_t slack = getenv("CMB_TEST_SMALL_POOLS") ? 64 : 65536;
                    const size_t per = getenv("CMB_TEST_SMALL_POOLS") ? 0 : 1;
                    b->qCap = per * (size_t)nReads * 4 + slack;
                    b->evCap = per * (size_t)nReads + slack;
                    b->fCap = per * (size_t)nReads * 16 + slack;
                    b->cCap = per * (size_t)nReads * 4 + slack;
                    b->aCap = per * (size_t)nReads * 32 + slack;
                }
                b->qCap = std::max<size_t>(b->qCap, (size_t)nTasks + 1024);
                constexpr uint32_t PU = MvTraits::PAIR_U4;
                for (int j = 0; j < 2; j++) {
                    if (b->Q[j].n < (PU + 2 + pkU4) * b->qCap) b->Q[j].alloc((PU + 2 + pkU4) * b->qCap);
                    if (b->Ev[j].n < (1 + pkU4) * b->evCap) b->Ev[j].alloc((1 + pkU4) * b->evCap);
                }
                if (b->F.n < (PU + 1) * b->fCap) b->F.alloc((PU + 1) * b->fCap);
                const uint32_t ctxU4 = geoX ? CTX_U4_X : ctxU4For(b->maxLen);
                if (b->C.n < (size_t)ctxU4 * b->cCap) b->C.alloc((size_t)ctxU4 * b->cCap);
                if (b->A.n < b->aCap) b->A.alloc(b->aCap);
                const size_t cntWords = 2 * ((size_t)maxPass + 2) + 4;
                if (b->bfsCnt.n < cntWords) b->bfsCnt.alloc(cntWords);
                if (b->blockCnt.n < (size_t)BFS_GRID * 4) b->blockCnt.alloc((size_t)BFS_GRID * 4);
                MV_HIPCHK(hipMemsetAsync(b->bfsCnt.p, 0, cntWords * sizeof(uint32_t), s));
                MV_HIPCHK(hipMemsetAsync(b->blockCnt.p, 0, (size_t)BFS_GRID * 4 * sizeof(unsigned long long), s));
                MvBufs B{};
                for (int j = 0; j < 2; j++) {
                    B.Q[j] = b->Q[j].p;
                    B.Ev[j] = b->Ev[j].p;
                }
                B.F = b->F.p;
                B.C = b->C.p;
                B.A = b->A.p;
                B.qCap = (uint32_t)std::min<size_t>(b->Q[0].n / (PU + 2 + pkU4), 0xFFFFFFF0u);
                B.evCap = (uint32_t)std::min<size_t>(b->Ev[0].n / (1 + pkU4), 0xFFFFFFF0u);
                B.fCap = (uint32_t)std::min<size_t>(b->F.n / (PU + 1), 0xFFFFFFF0u);
                B.cCap = (uint32_t)std::min<size_t>(b->C.n / ctxU4, 0xFFFFFFF0u);
                B.ctxU4 = ctxU4;
                B.ctxMblk = geoX ? CTX_MBLK_X : ctxMblkFor(b->maxLen);
                B.aCap = (uint32_t)std::min<size_t>(b->A.n, 0xFFFFFFF0u);
                // text and run counts below 2^32 (the reference's default build of length_t): the expanding blocks work on 32-bit positions with
                // the children in slots (mvExpandSlots; CMB_MOVE_POS64=1: the general 40-bit path)
                const bool smallPos = !b->wide && ix->d.n < 0xFFFFFFF0ull && ix->d.fwd.runs < 0xFFFFFFF0ull && ix->d.rev.runs < 0xFFFFFFF0ull && !getenv("CMB_MOVE_POS64");
                B.chain = getenv("CMB_MVS_CHAIN") ? (uint32_t)std::max(1, atoi(getenv("CMB_MVS_CHAIN"))) : (smallPos ? MVS_CHAIN_SMALL : MVS_CHAIN);
                B.gridX = getenv("CMB_MVS_GRID") ? (uint32_t)std::min<int>(BFS_GRID, std::max(1, atoi(getenv("CMB_MVS_GRID"))))
                                                 : (b->wide ? MVS_GRID_X_WIDE : smallPos ? MVS_GRID_X_SMALL : MVS_GRID_X);
                B.gridEv = BFS_GRID_EV;
                B.nq = b->bfsCnt.p;
                B.ne = b->bfsCnt.p + (maxPass + 2);
                B.pool = b->bfsCnt.p + 2 * (maxPass + 2);
                B.blockCnt = b->blockCnt.p;
                B.fmX = b->fm.p;
                B.rowSteps = nullptr;
                B.narrowWv = getenv("CMB_TEST_NARROW_WV") ? (uint32_t)std::max(0, atoi(getenv("CMB_TEST_NARROW_WV"))) : 0xFFFFu;
                // up to 6 errors the in-index matrix runs on 32-bit words (GeoN32, dev_matrix.hpp: MXS_*) unless a phase did not fit it (CMB_MATRIX64=1: never)
                const bool small32 = !b->wide && b->k <= MXS_MAX_ED && !b->noSmallMatrix && !getenv("CMB_MATRIX64");
                const dim3 gStart(std::min<uint32_t>((nTasks + 255) / 256, BFS_GRID));
                if (geoX)
                    hipLaunchKernelGGL(k_mvs_start<GeoX>, gStart, dim3(256), 0, s, b->stratW.p, B, b->tasks.p, nTasks, dOffs, b->gw, b->G.p, b->partsW.p, q);
                else if (b->wide)
                    hipLaunchKernelGGL(k_mvs_start<GeoW>, gStart, dim3(256), 0, s, b->stratW.p, B, b->tasks.p, nTasks, dOffs, b->gw, b->G.p, b->partsW.p, q);
                else if (small32)
                    hipLaunchKernelGGL(k_mvs_start<GeoN32>, gStart, dim3(256), 0, s, b->strat.p, B, b->tasks.p, nTasks, dOffs, b->gw, b->G.p, b->parts.p, q);
                else
                    hipLaunchKernelGGL(k_mvs_start<GeoN>, gStart, dim3(256), 0, s, b->strat.p, B, b->tasks.p, nTasks, dOffs, b->gw, b->G.p, b->parts.p, q);
                std::vector<uint32_t> hc(cntWords);
                uint32_t pass = 0, peakQ = 0, peakEv = 0;
                bool drained = false;
                while (!drained && pass < maxPass) {
                    const uint32_t upTo = std::min(pass + 16u, maxPass);
                    for (; pass < upTo; pass++)
                        if (geoX)
                            hipLaunchKernelGGL(k_mvs_pass<GeoX>, dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d, b->stratW.p, B, pass, dOffs, b->gw, b->G.p,
                                               b->partsW.p, q);
                        else if (b->wide)
                            hipLaunchKernelGGL(k_mvs_pass<GeoW>, dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d, b->stratW.p, B, pass, dOffs, b->gw, b->G.p,
                                               b->partsW.p, q);
                        else if (small32)
                            hipLaunchKernelGGL((smallPos ? k_mvs_pass<GeoN32, true> : k_mvs_pass<GeoN32, false>), dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d,
                                               b->strat.p, B, pass, dOffs, b->gw, b->G.p, b->parts.p, q);
                        else
                            hipLaunchKernelGGL((smallPos ? k_mvs_pass<GeoN, true> : k_mvs_pass<GeoN, false>), dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d,
                                               b->strat.p, B, pass, dOffs, b->gw, b->G.p, b->parts.p, q);
                    MV_HIPCHK(hipMemcpyAsync(hc.data(), b->bfsCnt.p, cntWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    MV_HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                    MV_HIPCHK(hipStreamSynchronize(s));
                    if (hcnt[3] & BFS_STOP) break;
                    drained = hc[pass] == 0 && hc[maxPass + 2 + pass] == 0;
                }
                for (uint32_t p2 = 0; p2 <= pass && p2 < maxPass + 2; p2++) {
                    peakQ = std::max(peakQ, hc[p2]);
                    peakEv = std::max(peakEv, hc[maxPass + 2 + p2]);
                }
                const uint32_t* pool = hc.data() + 2 * (maxPass + 2);
                if (getenv("CMB_VERBOSE"))
                    fprintf(stderr, "[mvs] %u tasks, %u passes, peak frontier %u, peak events %u, F %u, contexts %u, arena %u\n", nTasks, pass, peakQ, peakEv,
                            pool[0], pool[1], pool[2]);
                hipLaunchKernelGGL(k_mvs_finish, dim3(1), dim3(256), 0, s, B, q);
                tm.end("k_dfs");
                MV_HIPCHK(hipGetLastError());
                if (hcnt[3] & FLAG_NARROW_MATRIX) { // (a first column wider than the small matrix holds: once more on the reference's words)
                    b->noSmallMatrix = true;
                    continue;
                }
                if (hcnt[3] & FLAG_CAPACITY) return failWith(CMB_ERR_INTERNAL, "device search capacity exceeded (band width / descendants)");
                if (hcnt[3] & (FLAG_BFS_Q | FLAG_BFS_EV | FLAG_BFS_F | FLAG_BFS_CTX | FLAG_BFS_ARENA | FLAG_FMOCC_OVERFLOW)) {
                    if (hcnt[3] & FLAG_BFS_Q) b->qCap = std::max<size_t>(2 * b->qCap, (size_t)peakQ + peakQ / 4);
                    if (hcnt[3] & FLAG_BFS_EV) b->evCap = std::max<size_t>(2 * b->evCap, (size_t)peakEv + peakEv / 4);
                    if (hcnt[3] & FLAG_BFS_F) b->fCap = std::max<size_t>(2 * b->fCap, (size_t)pool[0] + pool[0] / 4);
                    if (hcnt[3] & FLAG_BFS_CTX) b->cCap = std::max<size_t>(2 * b->cCap, (size_t)pool[1] + pool[1] / 4);
                    if (hcnt[3] & FLAG_BFS_ARENA) b->aCap = std::max<size_t>(2 * b->aCap, (size_t)pool[2] + pool[2] / 4);
                    if (hcnt[3] & FLAG_FMOCC_OVERFLOW) b->fm.alloc(std::max<size_t>(2 * b->fm.n, (size_t)hcnt[1] + hcnt[1] / 4 + 1024));
                    continue;
                }
                if (!drained) return failWith(CMB_ERR_INTERNAL, "frontier search did not finish within its pass bound");
            }
            nFm = hcnt[1];
            break;
        }
        // ---- in-index occurrences: de-duplicate, locate, sort, filter
        uint64_t totalPos = 0;
        uint32_t nUniq = 0;
        tm.begin();
        if (nFm) {
            if (b->keysA.n < nFm) {
                const size_t c = (size_t)nFm + nFm / 4 + 256;
                b->keysA.alloc(c), b->keysB.alloc(c), b->fmIdxA.alloc(c), b->fmIdxB.alloc(c), b->keep.alloc(c + 1), b->slot.alloc(c + 1), b->widths.alloc(c + 1);
            }
            hipLaunchKernelGGL(k_mvs_fm_keys, dim3(gridFor(nFm)), dim3(256), 0, s, b->fm.p, nFm, b->keysA.p, b->fmIdxA.p);
            size_t tb = 0;
            MV_HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, b->keysA.p, b->keysB.p, b->fmIdxA.p, b->fmIdxB.p, (int)nFm, 0, 64, s));
            if (b->sortTmp.n < tb) b->sortTmp.alloc(tb + tb / 4);
            MV_HIPCHK(hipcub::DeviceRadixSort::SortPairs(b->sortTmp.p, tb, b->keysA.p, b->keysB.p, b->fmIdxA.p, b->fmIdxB.p, (int)nFm, 0, 64, s));
            hipLaunchKernelGGL(k_mvs_fm_unique, dim3(gridFor(nFm)), dim3(256), 0, s, b->fm.p, b->fmIdxB.p, nFm, b->keep.p, b->widths.p);
            MV_HIPCHK(hipMemsetAsync(b->keep.p + nFm, 0, sizeof(uint32_t), s));
            tb = 0;
            MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, b->keep.p, b->slot.p, (int)(nFm + 1), s));
            if (b->sortTmp.n < tb) b->sortTmp.alloc(tb + tb / 4);
            MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->sortTmp.p, tb, b->keep.p, b->slot.p, (int)(nFm + 1), s));
            MV_HIPCHK(hipMemcpyAsync(&nUniq, b->slot.p + nFm, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MV_HIPCHK(hipStreamSynchronize(s));
            if (nUniq) {
                if (b->locRanges.n < nUniq) {
                    const size_t c = (size_t)nUniq + nUniq / 4 + 256;
                    b->locRanges.alloc(c), b->locMeta.alloc(c), b->locWidths.alloc(c + 1), b->locOff.alloc(c + 1);
                }
                hipLaunchKernelGGL(k_mvs_fm_compact, dim3(gridFor(nFm)), dim3(256), 0, s, b->fm.p, b->fmIdxB.p, b->keep.p, b->slot.p, nFm, b->locRanges.p,
                                   b->locMeta.p, b->locWidths.p);
                MV_HIPCHK(hipMemsetAsync(b->locWidths.p + nUniq, 0, sizeof(uint64_t), s));
                tb = 0;
                MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, b->locWidths.p, b->locOff.p, (int)(nUniq + 1), s));
                if (b->sortTmp.n < tb) b->sortTmp.alloc(tb + tb / 4);
                MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->sortTmp.p, tb, b->locWidths.p, b->locOff.p, (int)(nUniq + 1), s));
                MV_HIPCHK(hipMemcpyAsync(&totalPos, b->locOff.p + nUniq, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                MV_HIPCHK(hipStreamSynchronize(s));
            }
        }
        tm.end("fm_unique");
        if (totalPos >= (1ull << 31)) return failWith(CMB_ERR_UNSUPPORTED, "2^31 and more text positions in one b-move batch");
        tm.begin();
        if (totalPos) {
            if (b->positions.n < totalPos) {
                const size_t c = totalPos + totalPos / 4 + 256;
                b->positions.alloc(c), b->keysA.alloc(std::max(c, b->keysA.n)), b->keysB.alloc(std::max(c, b->keysB.n)), b->vals.alloc(c), b->valsB.alloc(c);
            }
            if (b->keysA.n < totalPos) b->keysA.alloc(totalPos + 256), b->keysB.alloc(totalPos + 256);
            MV_HIPCHK(hipMemsetAsync(b->bad.p, 0, sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_move_locate, dim3(gridFor(nUniq)), dim3(256), 0, s, ix->d, b->locRanges.p, (uint64_t)nUniq, b->locOff.p, (uint64_t)0,
                               b->positions.p, b->bad.p, true);
        }
        tm.end("locate");
        tm.begin();
        if (b->readCnt.n < (size_t)2 * nReads + 1) b->readCnt.alloc((size_t)2 * nReads + 1), b->readOff.alloc((size_t)2 * nReads + 1);
        uint64_t nOut = 0, nNaiveKept = 0;
        const uint32_t window = b->metric == CMB_METRIC_EDIT ? b->k : 0u;
        const uint32_t uniqueOnly = b->metric == CMB_METRIC_EDIT ? 0u : 1u; // (getTextOccHamming, indexinterface.cpp:1331-1371: no redundancy filter)
        auto sortAndFilter = [&](uint64_t nKeys, uint32_t nGroups, MvBuf<MoveOccOut>& dst, uint64_t& kept) {
            size_t tb = 0;
            MV_HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, b->keysA.p, b->keysB.p, b->vals.p, b->valsB.p, (int)nKeys, 0, 64, s));
            if (b->sortTmp.n < tb) b->sortTmp.alloc(tb + tb / 4);
            MV_HIPCHK(hipcub::DeviceRadixSort::SortPairs(b->sortTmp.p, tb, b->keysA.p, b->keysB.p, b->vals.p, b->valsB.p, (int)nKeys, 0, 64, s));
            MV_HIPCHK(hipMemsetAsync(b->readCnt.p, 0, ((size_t)nGroups + 1) * sizeof(uint64_t), s));
            hipLaunchKernelGGL(k_mvs_filter<false>, dim3(gridFor(nGroups)), dim3(256), 0, s, b->keysB.p, b->valsB.p, nKeys, nGroups, window, b->readCnt.p,
                               (const uint64_t*)nullptr, (MoveOccOut*)nullptr, uniqueOnly);
            tb = 0;
            MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, b->readCnt.p, b->readOff.p, (int)(nGroups + 1), s));
            if (b->sortTmp.n < tb) b->sortTmp.alloc(tb + tb / 4);
            MV_HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->sortTmp.p, tb, b->readCnt.p, b->readOff.p, (int)(nGroups + 1), s));
            MV_HIPCHK(hipMemcpyAsync(&kept, b->readOff.p + nGroups, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            MV_HIPCHK(hipStreamSynchronize(s));
            if (dst.n < kept) dst.alloc(kept + kept / 4 + 256);
            if (kept)
                hipLaunchKernelGGL(k_mvs_filter<true>, dim3(gridFor(nGroups)), dim3(256), 0, s, b->keysB.p, b->valsB.p, nKeys, nGroups, window, b->readCnt.p,
                                   b->readOff.p, dst.p, uniqueOnly);
        };
        if (totalPos) {
            if (hasNaive) { // the naive path's own filter pass, per read x strand (indexinterface.cpp:1137, :1205)
                hipLaunchKernelGGL(k_mvs_text_keys, dim3(gridFor(totalPos)), dim3(256), 0, s, b->positions.p, b->locOff.p, nUniq, totalPos, b->locMeta.p,
                                   b->keysA.p, b->vals.p, b->bad.p, (const uint8_t*)b->psel.p, 1);
                sortAndFilter(totalPos, 2 * nReads, b->naiveOut, nNaiveKept);
                if (b->naiveOff.n < (size_t)2 * nReads + 1) b->naiveOff.alloc((size_t)2 * nReads + 1);
                MV_HIPCHK(hipMemcpyAsync(b->naiveOff.p, b->readOff.p, ((size_t)2 * nReads + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
                const uint64_t need = totalPos + nNaiveKept;
                if (need >= (1ull << 31)) return failWith(CMB_ERR_UNSUPPORTED, "2^31 and more text positions in one b-move batch");
                if (b->keysA.n < need || b->vals.n < need) {
                    MV_HIPCHK(hipStreamSynchronize(s));
                    b->keysA.alloc(need + 256), b->keysB.alloc(need + 256), b->vals.alloc(need + 256), b->valsB.alloc(need + 256);
                }
            }
            const int perStrand = b->perStrand ? 1 : 0;
            hipLaunchKernelGGL(k_mvs_text_keys, dim3(gridFor(totalPos)), dim3(256), 0, s, b->positions.p, b->locOff.p, nUniq, totalPos, b->locMeta.p, b->keysA.p,
                               b->vals.p, b->bad.p, (const uint8_t*)b->psel.p, hasNaive ? 2 : 0, perStrand);
            if (nNaiveKept)
                hipLaunchKernelGGL(k_mvs_occ_keys, dim3(gridFor(2 * nReads)), dim3(256), 0, s, b->naiveOut.p, b->naiveOff.p, 2 * nReads, b->keysA.p + totalPos,
                                   b->vals.p + totalPos, perStrand);
            // (the previous slice's records may still be on their way out of b->out: this stream waits for them before the buffer is
            // written or enlarged — sortAndFilter synchronises the stream before it allocates)
            if (b->copyPending) MV_HIPCHK(hipStreamWaitEvent(s, b->copyDone, 0));
            sortAndFilter(totalPos + nNaiveKept, perStrand ? 2 * nReads : nReads, b->out, nOut);
            if (perStrand) { // per read again: the two strands of a read are neighbouring groups
                if (b->rsOff.n < (size_t)2 * nReads + 1) b->rsOff.alloc((size_t)2 * nReads + 1);
                MV_HIPCHK(hipMemcpyAsync(b->rsOff.p, b->readOff.p, ((size_t)2 * nReads + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(k_mvs_read_offsets, dim3(gridFor(nReads + 1)), dim3(256), 0, s, b->rsOff.p, nReads, b->readOff.p);
            }
        }
        tm.end("filter");
        MV_HIPCHK(hipGetLastError());
        if (b->wantAln && nOut) { // CIGAR and sequence of every final occurrence: findCIGAR on text[begin, end) (k_cigar)
            tm.begin();
            if (b->occ32.n < nOut) {
                const size_t c = nOut + nOut / 4 + 256;
                b->occ32.alloc(c), b->alnRec.alloc(c), b->occRead.alloc(c), b->alnOps.alloc(c * b->alnStride);
            }
            hipLaunchKernelGGL(k_mvs_occ32, dim3(gridFor(nReads)), dim3(256), 0, s, b->out.p, b->readOff.p, nReads, b->occ32.p, b->occRead.p);
            uint32_t flagBefore = 0;
            MV_HIPCHK(hipMemcpyAsync(&flagBefore, b->cnt.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            const int rc = cmb::moveCigarsOnText(ix->textIndex, s, dOffs, b->G.p, b->gw, nReads, b->maxLen, b->k, b->metric == CMB_METRIC_EDIT ? 0 : 1,
                                                 b->occ32.p, b->occRead.p, nOut, b->alnRec.p, b->alnOps.p, b->alnStride, b->cnt.p + 3);
            if (rc != CMB_OK) return rc;
            const size_t base = b->hAlnRec.size();
            b->hAlnRec.resize(base + nOut);
            b->hAlnOps.resize((base + nOut) * b->alnStride);
            MV_HIPCHK(hipMemcpyAsync(b->hAlnRec.data() + base, b->alnRec.p, nOut * sizeof(uint4), hipMemcpyDeviceToHost, s));
            MV_HIPCHK(hipMemcpyAsync(b->hAlnOps.data() + base * b->alnStride, b->alnOps.p, nOut * b->alnStride * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
            uint32_t flagAfter = 0;
            MV_HIPCHK(hipMemcpyAsync(&flagAfter, b->cnt.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MV_HIPCHK(hipStreamSynchronize(s));
            tm.end("k_cigar");
            if ((flagAfter & ~flagBefore) & FLAG_CAPACITY)
                return failWith(CMB_ERR_INTERNAL, "a CIGAR traceback left the band or an occurrence is not an alignment within its distance");
        }
        uint32_t hb = 0;
        MV_HIPCHK(hipMemcpyAsync(&hb, b->bad.p, sizeof(hb), hipMemcpyDeviceToHost, s));
        unsigned long long hc64[CMB_CNT_MAX];
        MV_HIPCHK(hipMemcpyAsync(hc64, b->counters.p, sizeof(hc64), hipMemcpyDeviceToHost, s));
        static_assert(sizeof(cmb_move_occ) == sizeof(MoveOccOut), "cmb_move_occ layout");
        const size_t occBase = b->occs.size();
        if (occBase + nOut > b->occs.cap) b->waitForCopies(); // (the page-locked vector moves when it grows)
        b->occs.resize(occBase + nOut);
        std::vector<uint64_t> sliceOffs((size_t)nReads + 1, 0);
        if (nOut) { // on the copy stream, behind everything this slice has queued: the next slice starts while the records travel
            MV_HIPCHK(hipEventRecord(b->outReady, s));
            MV_HIPCHK(hipStreamWaitEvent(b->copyStream, b->outReady, 0));
            MV_HIPCHK(hipMemcpyAsync(b->occs.data() + occBase, b->out.p, nOut * sizeof(MoveOccOut), hipMemcpyDeviceToHost, b->copyStream));
            MV_HIPCHK(hipEventRecord(b->copyDone, b->copyStream));
            b->copyPending = true;
        }
        if (totalPos) MV_HIPCHK(hipMemcpyAsync(sliceOffs.data(), b->readOff.p, ((size_t)nReads + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        MV_HIPCHK(hipStreamSynchronize(s));
        for (uint32_t i = 0; i <= nReads; i++) b->occOffs[(size_t)lo + i] = occBase + sliceOffs[i];
        if (totalPos && hb)
            return failWith(CMB_ERR_INTERNAL, std::to_string(hb) + " occurrences whose phi chains do not have the width of their range, or whose "
                                                                   "fields do not fit the filter keys");
        for (int i = 0; i < CMB_CNT_MAX; i++) b->cnts[i] += hc64[i];
        b->cnts[CMB_CNT_TOTAL_REPORTED] += totalPos + nNaiveKept; // (the naive path's survivors are reported again as text occurrences of the read)
        b->cnts[CMB_CNT_LOCATED_ROWS] += totalPos;
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_move_batch_result_size(const cmb_move_batch* b, uint64_t* n_occ) {
    if (!b || !n_occ) return failWith(CMB_ERR_INVALID, "null argument");
    if (!b->done) return failWith(CMB_ERR_INVALID, "batch has not been run");
    *n_occ = b->occs.size();
    for (const cmb_move_batch* c : b->subs) *n_occ += c->occs.size();
    return CMB_OK;
}
extern "C" int cmb_move_batch_results(const cmb_move_batch* b, cmb_move_occ* out, uint64_t out_cap, uint64_t* out_offs, uint64_t* counters) {
    if (!b) return failWith(CMB_ERR_INVALID, "null argument");
    if (!b->done) return failWith(CMB_ERR_INVALID, "batch has not been run");
    if (!b->subs.empty()) { // the halves' lists one after the other, offsets rebased
        uint64_t total = 0;
        for (const cmb_move_batch* c : b->subs) total += c->occs.size();
        if (total > out_cap) return failWith(CMB_ERR_OVERFLOW, "output buffer too small");
        uint64_t base = 0;
        for (size_t j = 0; j < b->subs.size(); j++) {
            const cmb_move_batch* c = b->subs[j];
            if (out && !c->occs.empty()) memcpy(out + base, c->occs.data(), c->occs.size() * sizeof(cmb_move_occ));
            if (out_offs)
                for (uint32_t i = 0; i <= c->nReads; i++) out_offs[(size_t)b->subLo[j] + i] = base + c->occOffs[i];
            base += c->occs.size();
        }
        if (counters) memcpy(counters, b->cnts, sizeof(b->cnts));
        return CMB_OK;
    }
    if (b->occs.size() > out_cap) return failWith(CMB_ERR_OVERFLOW, "output buffer too small");
    if (out && !b->occs.empty()) memcpy(out, b->occs.data(), b->occs.size() * sizeof(cmb_move_occ));
    if (out_offs) memcpy(out_offs, b->occOffs.data(), b->occOffs.size() * sizeof(uint64_t));
    if (counters) memcpy(counters, b->cnts, sizeof(b->cnts));
    return CMB_OK;
}
extern "C" int cmb_move_batch_filter_per_strand(cmb_move_batch* b, int on) {
    if (!b) return failWith(CMB_ERR_INVALID, "null argument");
    b->perStrand = on != 0;
    for (cmb_move_batch* c : b->subs) c->perStrand = on != 0;
    return CMB_OK;
}
extern "C" int cmb_move_batch_want_alignments(cmb_move_batch* b, int on) {
    if (!b) return failWith(CMB_ERR_INVALID, "null argument");
    if (on && !b->ix->textIndex) return failWith(CMB_ERR_INVALID, "alignments need the text beside the index (cmb_move_attach_text)");
    b->wantAln = on != 0;
    for (cmb_move_batch* c : b->subs) c->wantAln = on != 0;
    return CMB_OK;
}
// as cmb_batch_alignments: one record per occurrence of cmb_move_batch_results, CIGAR runs (length << 2 | op, op 0 M / 1 I / 2 D) in a pool
extern "C" int cmb_move_batch_alignments(const cmb_move_batch* b, cmb_aln* out, uint64_t cap, uint16_t* cigar_ops, uint64_t ops_cap, uint64_t* n_ops) {
    if (!b) return failWith(CMB_ERR_INVALID, "null argument");
    if (!b->done) return failWith(CMB_ERR_INVALID, "batch has not been run");
    if (!b->wantAln) return failWith(CMB_ERR_INVALID, "alignments were not requested (cmb_move_batch_want_alignments)");
    std::vector<const cmb_move_batch*> parts(b->subs.begin(), b->subs.end()); // (the halves' records one after the other, as their occurrences)
    if (parts.empty()) parts.push_back(b);
    uint64_t total = 0, ops = 0;
    for (const cmb_move_batch* c : parts) {
        total += c->hAlnRec.size();
        for (const uint4& r : c->hAlnRec) ops += r.z;
    }
    if (n_ops) *n_ops = ops;
    if (cap < total || ops_cap < ops) return failWith(CMB_ERR_OVERFLOW, "output buffer too small");
    uint64_t po = 0, at = 0;
    for (const cmb_move_batch* c : parts)
        for (uint64_t i = 0; i < c->hAlnRec.size(); i++, at++) {
            const uint4& r = c->hAlnRec[i];
            out[at] = cmb_aln{r.x, r.y, po, (uint16_t)r.z, (uint16_t)r.w};
            const uint16_t* src = c->hAlnOps.data() + i * c->alnStride;
            for (uint32_t j = 0; j < r.z; j++) cigar_ops[po + j] = src[r.z - 1 - j]; // (stored end to begin)
            po += r.z;
        }
    return CMB_OK;
}
extern "C" int cmb_move_batch_timings(const cmb_move_batch* b, const char** names, float* ms, uint32_t cap) {
    if (!b || !names || !ms) return -1;
    uint32_t n = 0;
    for (const auto& t : b->times) {
        if (n >= cap) break;
        names[n] = t.first;
        ms[n] = t.second;
        n++;
    }
    return (int)n;
}
extern "C" int cmb_move_match_batch(cmb_move_index* idx, const cmb_strategy* st, uint32_t max_distance, uint32_t kmer_size, const char* seqs,
                                    const uint64_t* offs, uint32_t n_reads, cmb_move_occ* out, uint64_t out_cap, uint64_t* out_offs,
                                    uint64_t* counters, uint64_t* needed) {
    cmb_move_batch* b = nullptr;
    int rc = cmb_move_batch_create(idx, st, max_distance, kmer_size, seqs, offs, n_reads, &b);
    if (rc != CMB_OK) return rc;
    std::unique_ptr<cmb_move_batch> guard(b);
    rc = cmb_move_batch_run(b);
    if (rc != CMB_OK) return rc;
    uint64_t nOcc = 0;
    (void)cmb_move_batch_result_size(b, &nOcc);
    if (needed) *needed = nOcc;
    if (nOcc > out_cap) return failWith(CMB_ERR_OVERFLOW, "output buffer too small (needed holds the required number of records)");
    return cmb_move_batch_results(b, out, out_cap, out_offs, counters);
}
