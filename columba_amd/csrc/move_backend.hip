// C-ABI of the run-length compressed backend (include/columba_amd.h, section "b-move") on top of move_dev.hpp.
// gfx950 only.  A translation unit of its own, linked into libcolumba_amd.so.
#include "../../include/columba_amd.h"
#include "move_dev.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
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

struct cmb_move_index {
    int device = 0;
    uint64_t n = 0;
    TableHost tab[2];
    bool hasLocate = false;
    PosSetHost predFirst, predLast, plcpPos;
    MvBuf<uint64_t> firstToRun, lastToRun, plcpSum;
    MoveDev d{};
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
        dNodes.alloc(1);
        bad.alloc(1);
        MV_HIPCHK(hipMemset(dNodes.p, 0, sizeof(unsigned long long)));
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
        unsigned long long nodes = 0;
        MV_HIPCHK(hipMemcpy(&nodes, dNodes.p, sizeof(nodes), hipMemcpyDeviceToHost));
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
        if (h[0] || h[1] || h[2] || h[3])
            return failWith(CMB_ERR_INVALID, "inconsistent move index arrays (" + std::to_string(h[0]) + " / " + std::to_string(h[1]) + " table rows, " +
                                                 std::to_string(h[2]) + " locate entries, " + std::to_string(h[3]) + " directory entries)");
        return CMB_OK;
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
}
