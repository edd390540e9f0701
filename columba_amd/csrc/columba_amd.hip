// C-ABI implementation (include/columba_amd.h) on top of the HIP kernels.  gfx950 only.
#include "../../include/columba_amd.h"
#include "host_sam.hpp"
#include "host_schemes.hpp"
#include "kernels.hpp"

#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace cmb;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
namespace cmb {
int failWith(int code, const std::string& msg) { return fail(code, msg); } // for the library's other translation units
}
#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void upload(const T* h, size_t count) {
        alloc(count);
        if (count) HIPCHK(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
    }
    void uploadPadded(const T* h, size_t count, size_t pad) { // `pad` zeroed elements behind the data (k_prep reads 16-byte chunks)
        alloc(count + pad);
        if (count) HIPCHK(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(p + count, 0, pad * sizeof(T)));
    }
    size_t bytes() const { return n * sizeof(T); }
};

// page-locked host memory (results are copied at PCIe speed, asynchronously)
template <typename T> struct PinnedBuf {
    T* p = nullptr;
    size_t cap = 0, n = 0;
    PinnedBuf() {}
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    void resize(size_t count) { // contents are not kept
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = count + count / 4 + 64;
            HIPCHK(hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault));
            cap = want;
        }
        n = count;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
    const T* data() const { return p; }
};

// ------------------------------------------------------------------------------------ index
struct cmb_index {
    int device = 0;
    uint32_t saSparseness = 0;
    DevIndex d{};
    DevBuf<uint4> blkF, blkR; // 32-byte rank blocks
    DevBuf<uint32_t> saSamples;
    DevBuf<uint8_t> text;
    DevBuf<uint32_t> text2;
    DevBuf<uint4> kmer;
    std::vector<uint32_t> seqStarts;
    DevBuf<uint32_t> seqStartsDev; // the same on the device (k_cigar: sequence assignment); [0, n - 1] if none were given
    uint32_t nSeqsDev = 0;
    uint64_t bytes = 0;
    bool textOnly = false; // cmb_index_create_text_only: no BWT, no suffix array — alignments and SAM records only
    void uploadSeqStarts() {
        std::vector<uint32_t> v = seqStarts;
        if (v.size() < 2) v = {0u, d.n ? d.n - 1u : 0u};
        seqStartsDev.upload(v.data(), v.size());
        nSeqsDev = (uint32_t)v.size() - 1u;
    }
};

#ifdef CMB_BFS_STATS
extern "C" int cmb_debug_bfs_stats(unsigned long long* out, int reset) { // diagnostic build only
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(cmb::g_bfsStats), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(cmb::g_bfsStats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef CMB_BOUNDS
extern "C" int cmb_debug_oob(unsigned long long* out) { // diagnostic build only (tools/bounds_check.sh)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cmb::g_oob), 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef CMB_STAGE_STATS
extern "C" int cmb_debug_stage_stats(unsigned long long* out, int reset) { // diagnostic build only
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(cmb::g_stageStats), 24 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(cmb::g_stageStats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
extern "C" const char* cmb_last_error(void) { return g_err.c_str(); }
extern "C" const char* cmb_version(void) { return "columba_amd 0.1 (gfx950)"; }

static void useDevice(int device) { HIPCHK(hipSetDevice(device)); }

// k_check_index on the arrays of an index (at creation, and after its arrays were filled by a collective)
static int probeIndex(cmb_index* ix) {
    DevBuf<uint32_t> bad;
    bad.alloc(1);
    HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
    const uint32_t nProbe = std::min<uint32_t>(ix->d.n, 1u << 20);
    if (!nProbe) return CMB_OK;
    hipLaunchKernelGGL(k_check_index, dim3((nProbe + 255) / 256), dim3(256), 0, 0, ix->d, ix->saSparseness, nProbe, (uint32_t)std::min<size_t>(ix->saSamples.n, 0xFFFFFFFFu), bad.p);
    HIPCHK(hipGetLastError());
    uint32_t hb = 0;
    HIPCHK(hipMemcpy(&hb, bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (hb)
        return fail(CMB_ERR_INVALID, "the index arrays do not belong together: " + std::to_string(hb) + " of " + std::to_string(nProbe) +
                                         " probed suffix-array rows do not reach a sampled row within " + std::to_string(ix->saSparseness) +
                                         " LF steps (wrong sparseness, or bit vectors / samples / BWT of different texts)");
    return CMB_OK;
}
extern "C" int cmb_index_validate(cmb_index* idx) {
    if (!idx) return fail(CMB_ERR_INVALID, "null argument");
    try {
        useDevice(idx->device);
        return probeIndex(idx);
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_index_create(const cmb_index_desc* desc, int device, cmb_index** out) {
    if (!desc || !out) return fail(CMB_ERR_INVALID, "null argument");
    if (desc->text_length == 0 || desc->text_length >= 0xFFFFFFFFull)
        return fail(CMB_ERR_UNSUPPORTED, "text length must fit a 32-bit length_t");
    if (desc->kmer_size > 15) return fail(CMB_ERR_INVALID, "k-mer size > 15 (the reference's -K option takes 0 ... 15, alignparameters.cpp:449)");
    if (desc->sa_sparseness == 0 || (desc->sa_sparseness & (desc->sa_sparseness - 1)))
        return fail(CMB_ERR_INVALID, "suffix array sparseness must be a power of two");
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(CMB_ERR_DEVICE, "no HIP device available (the product path has no CPU fallback)");
        useDevice(device);
        std::unique_ptr<cmb_index> ix(new cmb_index());
        ix->device = device;
        ix->saSparseness = desc->sa_sparseness;
        const uint64_t n = desc->text_length, N = n + 1;
        const uint64_t bvW = 4 * ((N + 63) / 64), cW = 8 * ((N + 511) / 512);
        const uint64_t nBlocks = N / RANK_BLOCK + 1; // rank(N) must be answerable
        for (int dir = 0; dir < 2; dir++) {        // re-pack the reference arrays, one direction at a time
            DevBuf<uint64_t> bv, cnt;
            bv.upload(dir ? desc->bv_rev : desc->bv_fwd, bvW);
            cnt.upload(dir ? desc->cnt_rev : desc->cnt_fwd, cW);
            DevBuf<uint4>& blk = dir ? ix->blkR : ix->blkF;
            blk.alloc(nBlocks * 2);
            hipLaunchKernelGGL(k_relayout, dim3((unsigned)((nBlocks + 255) / 256)), dim3(256), 0, 0, bv.p, cnt.p, N,
                               nBlocks, blk.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
        }
        const uint64_t saW = (n + 63) / 64;
        { // the sparse suffix array's bitvector + rank9 counts ride in slot 3 of the forward blocks (k_relayout_sa)
            DevBuf<uint64_t> bv, cnt;
            bv.upload(desc->sa_bv, saW);
            cnt.upload(desc->sa_bv_counts, (saW + 7) / 4);
            hipLaunchKernelGGL(k_relayout_sa, dim3((unsigned)((nBlocks + 255) / 256)), dim3(256), 0, 0, bv.p, cnt.p, saW,
                               nBlocks, ix->blkF.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
        }
        ix->saSamples.upload(desc->sa_samples, desc->n_samples);
        // padded: the verification kernels read (unaligned) 16-byte chunks up to two chunks ahead, and lanes whose
        // candidate has ended keep prefetching while their wavefront runs (at most MAX_READ + 3 k rows + 48 bytes)
        constexpr uint64_t TEXT_PAD = 640;
        static_assert(TEXT_PAD >= (uint64_t)VROWS + 64, "text padding must cover the longest verification window");
        ix->text.alloc(n + TEXT_PAD);
        HIPCHK(hipMemset(ix->text.p, 0, n + TEXT_PAD));
        HIPCHK(hipMemcpy(ix->text.p, desc->text, n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_encode_text, dim3(4096), dim3(256), 0, 0, ix->text.p, n, n + TEXT_PAD); // ASCII -> codes 0..4
        HIPCHK(hipGetLastError());
        bool packedOk = false;
        { // 2-bit copy for the matrix kernels (k_pack_text); the words past the text's padding are never used
            const uint64_t nWords = (n + TEXT_PAD) / 16;
            ix->text2.alloc(nWords + 16);
            HIPCHK(hipMemset(ix->text2.p, 0, (nWords + 16) * sizeof(uint32_t)));
            DevBuf<uint32_t> bad;
            bad.alloc(1);
            HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
            hipLaunchKernelGGL(k_pack_text, dim3(4096), dim3(256), 0, 0, ix->text.p, n, nWords, ix->text2.p, bad.p);
            HIPCHK(hipGetLastError());
            uint32_t hb = 0;
            HIPCHK(hipMemcpy(&hb, bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
            packedOk = hb == 0 && !getenv("CMB_TEXT_BYTES");
            if (!packedOk) ix->text2.release();
        }
        ix->kmer.alloc(1ull << (2 * desc->kmer_size));
        if (desc->seq_starts) ix->seqStarts.assign(desc->seq_starts, desc->seq_starts + desc->n_seqs);
        DevIndex& d = ix->d;
        d.n = (uint32_t)n;
        for (int i = 0; i < 5; i++) d.counts[i] = (uint32_t)desc->counts[i];
        d.fwd = DevBWT{ix->blkF.p, (uint32_t)desc->dollar_pos_fwd};
        d.rev = DevBWT{ix->blkR.p, (uint32_t)desc->dollar_pos_rev};
        d.saSamples = ix->saSamples.p;
        d.text = ix->text.p;
        d.text2 = packedOk ? ix->text2.p : nullptr;
        d.kmer = ix->kmer.p;
        d.kmerSize = desc->kmer_size;
        d.switchPoint = desc->in_text_switch;
        const uint32_t total = 1u << (2 * desc->kmer_size);
        hipLaunchKernelGGL(k_kmer_table, dim3((total + 255) / 256), dim3(256), 0, 0, d, ix->kmer.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        if (int rc = probeIndex(ix.get())) return rc; // the arrays must belong together, or findSA would never end
        ix->bytes = ix->blkF.bytes() + ix->blkR.bytes() + ix->saSamples.bytes() + ix->text.bytes() + ix->text2.bytes() + ix->kmer.bytes();
        ix->uploadSeqStarts();
        *out = ix.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}
// ---- replication: layout description, empty twin, raw device arrays (include/columba_amd.h)
namespace {
struct ArrayRef {
    void** p;
    size_t* n;
    size_t elem;
};
// the seven device arrays of an index in the order of cmb_index_layout::bytes
inline void indexArrays(cmb_index* ix, ArrayRef out[CMB_DEV_ARRAYS]) {
    out[0] = {(void**)&ix->blkF.p, &ix->blkF.n, sizeof(uint4)};
    out[1] = {(void**)&ix->blkR.p, &ix->blkR.n, sizeof(uint4)};
    out[2] = {(void**)&ix->saSamples.p, &ix->saSamples.n, sizeof(uint32_t)};
    out[3] = {(void**)&ix->text.p, &ix->text.n, sizeof(uint8_t)};
    out[4] = {(void**)&ix->text2.p, &ix->text2.n, sizeof(uint32_t)};
    out[5] = {(void**)&ix->kmer.p, &ix->kmer.n, sizeof(uint4)};
}
inline void bindDevIndex(cmb_index* ix) { // DevIndex pointers from the owning buffers
    DevIndex& d = ix->d;
    d.fwd.blk = ix->blkF.p;
    d.rev.blk = ix->blkR.p;
    d.saSamples = ix->saSamples.p;
    d.text = ix->text.p;
    d.text2 = ix->text2.n ? ix->text2.p : nullptr;
    d.kmer = ix->kmer.p;
}
} // namespace

extern "C" int cmb_index_layout_of(const cmb_index* idx, cmb_index_layout* out) {
    if (!idx || !out) return fail(CMB_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    out->text_length = idx->d.n;
    for (int i = 0; i < 5; i++) out->counts[i] = idx->d.counts[i];
    out->dollar_pos_fwd = idx->d.fwd.dollarPos;
    out->dollar_pos_rev = idx->d.rev.dollarPos;
    out->n_samples = idx->saSamples.n;
    out->sa_sparseness = idx->saSparseness;
    out->kmer_size = idx->d.kmerSize;
    out->in_text_switch = idx->d.switchPoint;
    out->n_seqs = (uint32_t)idx->seqStarts.size();
    ArrayRef a[CMB_DEV_ARRAYS];
    indexArrays(const_cast<cmb_index*>(idx), a);
    for (int i = 0; i < CMB_DEV_ARRAYS; i++) out->bytes[i] = *a[i].p ? (uint64_t)(*a[i].n * a[i].elem) : 0;
    return CMB_OK;
}
extern "C" int cmb_index_seq_starts(const cmb_index* idx, uint32_t* out) {
    if (!idx || (!out && !idx->seqStarts.empty())) return fail(CMB_ERR_INVALID, "null argument");
    if (!idx->seqStarts.empty()) memcpy(out, idx->seqStarts.data(), idx->seqStarts.size() * sizeof(uint32_t));
    return CMB_OK;
}
// An index that holds ONLY the text (codes + 2-bit copy) and the sequence starts: what alignments, trimming at sequence ends and SAM
// records need of an index — for the b-move flavour, whose own index has no text (cmb_move_attach_text).  No batch can be created on it.
extern "C" int cmb_index_create_text_only(const char* text, uint64_t n, const uint32_t* seq_starts, uint32_t n_seqs, int device,
                                          cmb_index** out) {
    if (!text || !out) return fail(CMB_ERR_INVALID, "null argument");
    if (n == 0 || n >= 0xFFFFFF00ull) return fail(CMB_ERR_UNSUPPORTED, "text length must fit a 32-bit length_t");
    try {
        useDevice(device);
        std::unique_ptr<cmb_index> ix(new cmb_index());
        ix->device = device;
        ix->textOnly = true;
        constexpr uint64_t TEXT_PAD = 640;
        ix->text.alloc(n + TEXT_PAD);
        HIPCHK(hipMemset(ix->text.p, 0, n + TEXT_PAD));
        HIPCHK(hipMemcpy(ix->text.p, text, n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_encode_text, dim3(4096), dim3(256), 0, 0, ix->text.p, n, n + TEXT_PAD);
        const uint64_t nWords = (n + TEXT_PAD) / 16;
        ix->text2.alloc(nWords + 16);
        HIPCHK(hipMemset(ix->text2.p, 0, (nWords + 16) * sizeof(uint32_t)));
        DevBuf<uint32_t> bad;
        bad.alloc(1);
        HIPCHK(hipMemset(bad.p, 0, sizeof(uint32_t)));
        hipLaunchKernelGGL(k_pack_text, dim3(4096), dim3(256), 0, 0, ix->text.p, n, nWords, ix->text2.p, bad.p);
        HIPCHK(hipGetLastError());
        uint32_t hb = 0;
        HIPCHK(hipMemcpy(&hb, bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (hb) ix->text2.release();
        if (seq_starts && n_seqs) ix->seqStarts.assign(seq_starts, seq_starts + n_seqs); // (entries: the starts and the final n - 1, as cmb_index_desc)
        ix->d.n = (uint32_t)n;
        ix->d.text = ix->text.p;
        ix->d.text2 = hb ? nullptr : ix->text2.p;
        ix->uploadSeqStarts();
        ix->bytes = ix->text.bytes() + ix->text2.bytes();
        *out = ix.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_index_create_empty(const cmb_index_layout* L, const uint32_t* seq_starts, int device, cmb_index** out) {
    if (!L || !out) return fail(CMB_ERR_INVALID, "null argument");
    if (L->text_length == 0 || L->text_length >= 0xFFFFFFFFull || L->kmer_size > 15)
        return fail(CMB_ERR_INVALID, "index layout out of range");
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(CMB_ERR_DEVICE, "no HIP device available (the product path has no CPU fallback)");
        useDevice(device);
        std::unique_ptr<cmb_index> ix(new cmb_index());
        ix->device = device;
        ix->saSparseness = L->sa_sparseness;
        ArrayRef a[CMB_DEV_ARRAYS];
        indexArrays(ix.get(), a);
        {   // the sizes must be those cmb_index_create gives an index of this text length (a kernel indexes them by position)
            const uint64_t n = L->text_length, rankBytes = ((n + 1) / RANK_BLOCK + 1) * 2 * sizeof(uint4);
            if (L->sa_sparseness == 0 || (L->sa_sparseness & (L->sa_sparseness - 1)))
                return fail(CMB_ERR_INVALID, "index layout: suffix array sparseness must be a power of two");
            if (L->bytes[0] != rankBytes || L->bytes[1] != rankBytes)
                return fail(CMB_ERR_INVALID, "index layout: rank blocks do not have the size of this text length");
            if (L->bytes[2] < ((n + L->sa_sparseness - 1) / L->sa_sparseness) * sizeof(uint32_t))
                return fail(CMB_ERR_INVALID, "index layout: fewer suffix array samples than text length / sparseness");
            if (L->bytes[3] < n) return fail(CMB_ERR_INVALID, "index layout: text shorter than the text length");
            if (L->bytes[4] != 0 && L->bytes[4] < (n + 15) / 16 * sizeof(uint32_t))
                return fail(CMB_ERR_INVALID, "index layout: 2-bit text shorter than the text length");
            if (L->bytes[5] != (sizeof(uint4) << (2 * L->kmer_size)))
                return fail(CMB_ERR_INVALID, "index layout: k-mer table does not have 4^k entries");
        }
        uint64_t total = 0;
        for (int i = 0; i < CMB_DEV_ARRAYS; i++) {
            if (L->bytes[i] % a[i].elem) return fail(CMB_ERR_INVALID, "index layout: array size is not a whole number of elements");
            if (L->bytes[i] == 0) continue; // absent (2-bit text)
            HIPCHK(hipMalloc(a[i].p, L->bytes[i]));
            *a[i].n = L->bytes[i] / a[i].elem;
            total += L->bytes[i];
        }
        if (!ix->blkF.p || !ix->blkR.p || !ix->text.p || !ix->kmer.p)
            return fail(CMB_ERR_INVALID, "index layout: a required array is missing");
        DevIndex& d = ix->d;
        d.n = (uint32_t)L->text_length;
        for (int i = 0; i < 5; i++) d.counts[i] = (uint32_t)L->counts[i];
        d.fwd.dollarPos = (uint32_t)L->dollar_pos_fwd;
        d.rev.dollarPos = (uint32_t)L->dollar_pos_rev;
        d.kmerSize = L->kmer_size;
        d.switchPoint = L->in_text_switch;
        bindDevIndex(ix.get());
        if (seq_starts && L->n_seqs) ix->seqStarts.assign(seq_starts, seq_starts + L->n_seqs);
        ix->uploadSeqStarts();
        ix->bytes = total;
        *out = ix.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}
extern "C" int cmb_index_device_arrays(cmb_index* idx, void** ptrs, uint64_t* bytes) {
    if (!idx || !ptrs || !bytes) return fail(CMB_ERR_INVALID, "null argument");
    ArrayRef a[CMB_DEV_ARRAYS];
    indexArrays(idx, a);
    for (int i = 0; i < CMB_DEV_ARRAYS; i++) {
        ptrs[i] = *a[i].p;
        bytes[i] = *a[i].p ? (uint64_t)(*a[i].n * a[i].elem) : 0;
    }
    return CMB_OK;
}

extern "C" void cmb_index_destroy(cmb_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    delete idx;
}
extern "C" uint64_t cmb_index_device_bytes(const cmb_index* idx) { return idx ? idx->bytes : 0; }
extern "C" int cmb_index_kmer_table(const cmb_index* idx, uint32_t* out) {
    if (!idx || !out) return fail(CMB_ERR_INVALID, "null argument");
    try {
        useDevice(idx->device);
        HIPCHK(hipMemcpy(out, idx->kmer.p, idx->kmer.bytes(), hipMemcpyDeviceToHost));
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// --------------------------------------------------------------------------------- strategy
extern "C" int cmb_strategy_create(int metric, int partition, uint32_t kmer_cutoff, cmb_strategy** out) {
    if (!out || metric < 0 || metric > 1 || partition < 0 || partition > 2)
        return fail(CMB_ERR_INVALID, "bad metric / partition strategy");
    cmb_strategy* s = new cmb_strategy();
    s->metric = metric;
    s->partition = partition;
    s->kmerCutOff = kmer_cutoff;
    *out = s;
    return CMB_OK;
}
extern "C" int cmb_strategy_create_named(const char* name, int metric, int partition, cmb_strategy** out) {
    cmb_strategy* s = nullptr;
    int rc = cmb_strategy_create(metric, partition, 20, &s);
    if (rc) return rc;
    try {
        fillNamed(*s, name ? name : "");
    } catch (const std::exception& e) {
        delete s;
        return fail(CMB_ERR_INVALID, e.what());
    }
    *out = s;
    return CMB_OK;
}
extern "C" int cmb_strategy_create_from_dir(const char* dir, int mode, int metric, int partition,
                                            cmb_strategy** out) {
    if (mode < CMB_DIR_CUSTOM || mode > CMB_DIR_CUSTOM_DYNAMIC) return fail(CMB_ERR_INVALID, "bad scheme directory mode");
    cmb_strategy* s = nullptr;
    int rc = cmb_strategy_create(metric, partition, 20, &s);
    if (rc) return rc;
    try {
        if (mode == CMB_DIR_MULTIPLE) fillFromMultipleDir(*s, dir ? dir : "");
        else fillFromCustomDir(*s, dir ? dir : "", mode == CMB_DIR_CUSTOM_DYNAMIC);
    } catch (const std::exception& e) {
        delete s;
        return fail(CMB_ERR_INVALID, e.what());
    }
    *out = s;
    return CMB_OK;
}
extern "C" int cmb_strategy_add_scheme(cmb_strategy* s, uint32_t k, uint32_t n_searches, uint32_t n_parts,
                                       const uint32_t* pi, const uint32_t* L, const uint32_t* U) {
    if (!s || !pi || !L || !U) return fail(CMB_ERR_INVALID, "null argument");
    try {
        HostScheme sch;
        sch.k = k;
        for (uint32_t i = 0; i < n_searches; i++) {
            HostSearch h;
            h.pi.assign(pi + i * n_parts, pi + (i + 1) * n_parts);
            h.L.assign(L + i * n_parts, L + (i + 1) * n_parts);
            h.U.assign(U + i * n_parts, U + (i + 1) * n_parts);
            h.sIdx = i;
            sch.searches.push_back(h);
        }
        sch.finalize();
        for (const auto& h : sch.searches) (void)toDevSearch(h);
        s->schemes[k].push_back(sch);
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_INVALID, e.what());
    }
}
extern "C" int cmb_strategy_set_partition_params(cmb_strategy* s, uint32_t k, const double* seeding,
                                                 uint32_t n_seeding, const uint64_t* weights, uint32_t n_weights,
                                                 const double* begins, uint32_t n_begins) {
    if (!s) return fail(CMB_ERR_INVALID, "null argument");
    PartitionParams pp;
    if (n_seeding) pp.seeding.assign(seeding, seeding + n_seeding);
    if (n_weights) pp.weights.assign(weights, weights + n_weights);
    if (n_begins) pp.begins.assign(begins, begins + n_begins);
    s->params[k] = pp;
    return CMB_OK;
}
extern "C" void cmb_strategy_destroy(cmb_strategy* s) { delete s; }
extern "C" int cmb_strategy_describe(const cmb_strategy* s, uint32_t k, uint32_t* n_schemes, uint32_t* n_parts,
                                     uint32_t* critical_parts, uint32_t cap) {
    if (!s) return fail(CMB_ERR_INVALID, "null argument");
    auto it = s->schemes.find(k);
    if (it == s->schemes.end() || it->second.empty())
        return fail(CMB_ERR_INVALID, "the search strategy does not support distance " + std::to_string(k));
    if (n_schemes) *n_schemes = (uint32_t)it->second.size();
    if (n_parts) *n_parts = it->second.front().numParts();
    for (uint32_t i = 0; critical_parts && i < cap && i < it->second.size(); i++)
        critical_parts[i] = it->second[i].critical;
    return CMB_OK;
}
extern "C" int cmb_strategy_export_scheme(const cmb_strategy* s, uint32_t k, uint32_t scheme, uint32_t* pi, uint32_t* L,
                                          uint32_t* U, uint32_t cap, uint32_t* n_searches, uint32_t* n_parts) {
    if (!s) return fail(CMB_ERR_INVALID, "null argument");
    auto it = s->schemes.find(k);
    if (it == s->schemes.end() || scheme >= it->second.size())
        return fail(CMB_ERR_INVALID, "the search strategy has no such scheme for distance " + std::to_string(k));
    const HostScheme& h = it->second[scheme];
    const uint32_t ns = (uint32_t)h.searches.size(), np = h.numParts();
    if (n_searches) *n_searches = ns;
    if (n_parts) *n_parts = np;
    if ((uint64_t)ns * np > cap) return fail(CMB_ERR_OVERFLOW, "output arrays too small");
    for (uint32_t i = 0; i < ns; i++)
        for (uint32_t j = 0; j < np; j++) {
            if (pi) pi[i * np + j] = h.searches[i].pi[j];
            if (L) L[i * np + j] = h.searches[i].L[j];
            if (U) U[i * np + j] = h.searches[i].U[j];
        }
    return CMB_OK;
}
extern "C" int cmb_strategy_export_partition(const cmb_strategy* s, uint32_t k, double* seeding, uint64_t* weights,
                                             double* begins, uint32_t cap, uint32_t* kmer_cutoff) {
    if (!s) return fail(CMB_ERR_INVALID, "null argument");
    try {
        const PartitionParams d = s->partitionFor(k);
        const uint32_t P = (uint32_t)d.weights.size();
        if (P > cap) return fail(CMB_ERR_OVERFLOW, "output arrays too small");
        for (uint32_t i = 0; i + 2 < P && seeding; i++) seeding[i] = d.seeding[i];
        for (uint32_t i = 0; i < P && weights; i++) weights[i] = d.weights[i];
        for (uint32_t i = 0; i + 1 < P && begins; i++) begins[i] = d.begins[i];
        if (kmer_cutoff) *kmer_cutoff = s->kmerCutOff;
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_INVALID, e.what());
    }
}

// ------------------------------------------------------------------------------------ batch
struct KernelTime {
    const char* name;
    float ms;
};

struct cmb_batch {
    cmb_index* ix = nullptr;
    uint32_t k = 0, nReads = 0, maxLen = 0, gw = 0;
    uint32_t minLen = 0; // shortest read of the batch and of every chunk staged since (picks the verification stages
                         // that carry final-column code)
    int metric = 1;
    DevStrategyK hostStrat{};
    // schemes with more than MAXP parts (the greedy schemes for 8 ... 13 errors): tables of MAXP_WIDE parts, run by the wide instances
    // of k_parts / k_exact / k_hbfs (Hamming distance; the edit-distance matcher stops at 7 errors)
    bool wide = false;
    bool wideEdit = false; // edit distance 8 ... 13: wide records AND the wide layout of the filter keys
    bool geoX = false;     // ... 11 ... 13: the in-index matrix with 16-row blocks (GeoX), the in-text matrix with 8-row blocks
    bool noSmallMatrix = false; // a phase of an earlier run did not fit the 32-bit in-index matrix (GeoN32): this batch stays on GeoN
    DevStrategyKT<MAXP_WIDE> hostStratW{};
    DevBuf<DevStrategyKT<MAXP_WIDE>> stratW;
    DevBuf<PartOutT<MAXP_WIDE>> partsW;
    uint32_t sNumParts = 0, sPartition = 0, sNSchemes = 0, sMaxSearches = 0; // of whichever table the batch runs on
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> reads, seq;
    DevBuf<uint32_t> rec; // read records for k_parts / k_exact (k_prep)
    DevBuf<uint4> exr;    // exact-match range pair of every part: [part][read x strand] (k_parts -> k_exact)
    DevBuf<uint8_t> psel; // selected scheme per read x strand (bit 7: nothing to search)
    uint32_t recW = 0;
    DevBuf<uint64_t> offs;
    DevBuf<uint32_t> G;
    DevBuf<uint4> mfull; // match words of the full reads per 32-row block (k_match_words)
    DevBuf<DevStrategyK> strat;
    // frontier search (dev_bfs_edit.hpp): node / event double buffers, F records, contexts, list arena
    DevBuf<uint4> bfsQ[2], bfsEv[2], bfsF, bfsC, bfsA;
    DevBuf<uint32_t> bfsCnt;               // nq[passes], ne[passes], pool[4]
    DevBuf<unsigned long long> bfsBlockCnt; // [BFS_GRID_CNT][4]
    size_t bfsQCap = 0, bfsEvCap = 0, bfsFCap = 0, bfsCCap = 0, bfsACap = 0;
    // naive backtracking (dev_bfs_naive.hpp): node double buffer, nodes per pass
    DevBuf<uint4> nvQ[2];
    DevBuf<uint32_t> nvCnt;
    size_t nvQCap = 0;
    bool hasNaive = false; // reads of the running chunk take that path (k_parts marked them in psel)
    DevBuf<PartOut> parts;
    DevBuf<DfsTask> dfs;
    DevBuf<uint64_t> vW; // packed trace rows [row][slot]
    DevBuf<uint4> tbq;
    DevBuf<uint8_t> dpSlab; // k_verify_wide: the band rows of one candidate per slot
    DevBuf<uint32_t> dpList, dpWork; // k_wide_filter: the keys that go on to k_verify_wide; [0] next key, [1] number of those
    DevBuf<uint4> items;
    DevBuf<FMOccRec> fm, fmUniq;
    DevBuf<TextOccRec> text;
    DevBuf<uint4> fout;
    DevBuf<unsigned long long> keysA, keysB;
    DevBuf<uint32_t> fcounts, fsegB, fsegE, frank;
    DevBuf<uint64_t> foffs;
    DevBuf<unsigned long long> fmKeysA, fmKeysB;
    DevBuf<uint32_t> fmIdxA, fmIdxB, fmN;
    DevBuf<uint8_t> sortTmp, scanTmp;
    DevBuf<unsigned long long> vkeysA, vkeysB; // verification keys (k_verify) / sorted, then distinct
    DevBuf<uint32_t> vcounts, vruns;           // multiplicities of the distinct keys / number of runs
    DevBuf<uint4> vsA[2], vsB[2];              // staged verification: survivor lists (ping-pong)
    DevBuf<uint32_t> vsC[2], vsN;
    DevBuf<uint32_t> cnt;
    DevBuf<unsigned long long> counters;
    uint32_t nSlots = 0;
    std::vector<uint64_t> hostOffs;
    bool perStrand = false; // every strand filtered by itself (BEST mode: mapRead, searchstrategy.h:490-523)
    bool allowUnsupported = false; // (cmb_batch_allow_unsupported: no effect any more)
    // alignments of the final occurrences (cmb_batch_want_alignments): CIGAR runs + sequence assignment
    bool wantAln = false;
    DevBuf<uint32_t> foutRead;
    DevBuf<uint16_t> alnOps;
    DevBuf<AlnRec> alnRec;
    uint32_t alnStride = 0;
    PinnedBuf<uint16_t> hAlnOps;
    PinnedBuf<AlnRec> hAlnRec;
    // cmb_verify_batch_staged: candidates given by the caller take the place of the search's in-text items, and the
    // raw text occurrences (before the filter) are what is handed back
    std::vector<uint4> presetItems;
    std::vector<TextOccRec> presetOut;
    // A large batch is a COMPOSITE of sub-batches that run concurrently, one host thread + HIP stream each: the
    // stages of different sub-batches drift apart, so VALU-bound matrix kernels of one overlap memory-bound
    // extension kernels of another (starting them one after the other on purpose was measured slower: the
    // device idles at both ends).
    std::vector<cmb_batch*> subs;
    std::vector<uint32_t> subBound; // composite: first read of every sub-batch (+ total)
    // the NEXT chunk of reads, uploaded on a stream of its own while this one is being matched (cmb_batch_stage_reads);
    // cmb_batch_run swaps it in
    DevBuf<uint8_t> readsStage;
    DevBuf<uint64_t> offsStage;
    std::vector<uint64_t> hostOffsStage, hostOffsActive;
    // the offsets of a registered chunk once more in PAGE-LOCKED memory, two buffers taking turns: the upload a run starts must not stall
    // the host (from pageable memory hipMemcpyAsync copies through bounce buffers before it returns — milliseconds in which the run has
    // not launched a kernel yet), and the next registration must not overwrite offsets the copy engine is still reading
    PinnedBuf<uint64_t> offsPin[2];
    int pinNext = 0, pinStaged = 0;
    hipStream_t copyStream = nullptr;
    hipEvent_t copyDone = nullptr;
    bool staged = false;                 // readsStage / offsStage hold an uploaded chunk (or its upload is in flight)
    const char* pendingSeqs = nullptr;   // a registered chunk whose upload the next run starts
    bool pending = false;
    cmb_batch* parent = nullptr;
    uint32_t subIndex = 0;
    // results
    PinnedBuf<cmb_occ> occs;
    PinnedBuf<uint64_t> occOffs;
    uint64_t cnts[CMB_CNT_MAX];
    std::vector<KernelTime> times;
    bool done = false;
    ~cmb_batch() {
        for (cmb_batch* c : subs) delete c;
        if (copyDone) (void)hipEventDestroy(copyDone);
        if (copyStream) (void)hipStreamDestroy(copyStream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static int batchCreateOne(cmb_index* idx, const cmb_strategy* st, uint32_t max_distance, const char* seqs,
                          const uint64_t* offs, uint32_t n_reads, cmb_batch** out);
constexpr uint32_t MAX_SUB_READS = 1u << 23; // reads per sub-batch (keys: kernels.hpp k_pack_keys, packVerifyKey)

extern "C" int cmb_batch_create(cmb_index* idx, const cmb_strategy* st, uint32_t max_distance, const char* seqs,
                                const uint64_t* offs, uint32_t n_reads, cmb_batch** out) {
    if (!idx || !st || !offs || !out || (!seqs && n_reads)) return fail(CMB_ERR_INVALID, "null argument");
    if (idx->textOnly) return fail(CMB_ERR_INVALID, "this index holds only the text (cmb_index_create_text_only): nothing can be matched on it");
    if (n_reads >= 0x7FFFFFFFu) return fail(CMB_ERR_INVALID, "too many reads in one batch");
    // sub-batches: 3 from 8 M reads, 2 from 2 M (below that the fixed cost per frontier level dominates; measured
    // on 10 M reads: 1 -> 26.4, 2 -> 30.4, 3 -> 30.6, 4 -> 29.5 M reads/s); CMB_SUBBATCHES overrides
    uint32_t S = n_reads >= 8000000u ? 3u : n_reads >= 2000000u ? 2u : 1u;
    if (getenv("CMB_SUBBATCHES")) S = (uint32_t)std::max(1, atoi(getenv("CMB_SUBBATCHES")));
    // a sub-batch holds fewer than 2^24 reads (24-bit read number of the filter key, 25 bits of read x strand in the
    // verification key): larger batches are cut into more sub-batches up front, never refused after the work is done
    // (22 group bits in the wide filter keys; from 11 errors on the frontier of a read can hold millions of nodes on a large reference —
    // 20 000 reads at 12 errors on 3 Gbp overflowed pools of 2^32 slots —, so those sub-batches are small and run three at a time)
    const uint32_t maxSub = st->metric != CMB_METRIC_EDIT || max_distance <= 7 ? MAX_SUB_READS : max_distance <= MX_MAX_ED ? (1u << 20) : (1u << 11);
    S = std::max<uint32_t>(S, (uint32_t)(((uint64_t)n_reads + maxSub - 1) / maxSub));
    S = std::min<uint32_t>(S, std::max<uint32_t>(n_reads, 1u));
    if (S <= 1) return batchCreateOne(idx, st, max_distance, seqs, offs, n_reads, out);
    std::unique_ptr<cmb_batch> parent(new cmb_batch());
    parent->ix = idx;
    parent->k = max_distance;
    parent->nReads = n_reads;
    parent->metric = st->metric;
    // slice bounds: equal shares, or CMB_SUBBATCH_SPLIT="w0,w1,..." (relative weights; unequal sub-batches reach
    // their stages at different times)
    std::vector<double> wts(S, 1.0);
    if (const char* sp = getenv("CMB_SUBBATCH_SPLIT")) {
        std::vector<double> w;
        for (const char* q = sp; *q;) {
            char* e = nullptr;
            const double v = strtod(q, &e);
            if (e == q) break;
            if (v > 0) w.push_back(v);
            q = *e == ',' ? e + 1 : e;
        }
        if (w.size() == S) wts = w;
    }
    double wsum = 0;
    for (double v : wts) wsum += v;
    std::vector<uint32_t> bound(S + 1, 0);
    {
        double acc = 0;
        for (uint32_t j = 0; j < S; j++) {
            acc += wts[j];
            bound[j + 1] = j + 1 == S ? n_reads : (uint32_t)((double)n_reads * (acc / wsum));
        }
    }
    for (uint32_t j = 0; j < S; j++) {
        const uint32_t lo = bound[j], hi = std::max(bound[j + 1], bound[j]);
        std::vector<uint64_t> o(hi - lo + 1);
        for (uint32_t i = lo; i <= hi; i++) {
            if (i > lo && offs[i] < offs[i - 1]) return fail(CMB_ERR_INVALID, "read offsets must be non-decreasing");
            o[i - lo] = offs[i] - offs[lo];
        }
        cmb_batch* c = nullptr;
        const int rc = batchCreateOne(idx, st, max_distance, seqs + offs[lo], o.data(), hi - lo, &c);
        if (rc != CMB_OK) return rc;
        c->parent = parent.get();
        c->subIndex = j;
        parent->subs.push_back(c);
    }
    parent->subBound = bound;
    *out = parent.release();
    return CMB_OK;
}

static int batchCreateOne(cmb_index* idx, const cmb_strategy* st, uint32_t max_distance, const char* seqs,
                          const uint64_t* offs, uint32_t n_reads, cmb_batch** out) {
    try {
        useDevice(idx->device);
        std::unique_ptr<cmb_batch> b(new cmb_batch());
        b->ix = idx;
        b->k = max_distance;
        b->nReads = n_reads;
        b->metric = st->metric;
        if (n_reads >= (1u << 24)) // (only reachable through CMB_SUBBATCH_SPLIT weights: checked before any work)
            return fail(CMB_ERR_UNSUPPORTED, "more than 2^24 reads in one sub-batch");
        if (max_distance > 0) {
            // in-text verification: nZeros + maxED = 3k+1 must fit the first column of the in-text matrix (the reference
            // switches to its 128-bit matrix at k = 7, fmindex.h:240-246; here: 64-bit words / 16-row blocks, LEFT = 22)
            // edit distance beyond 7 errors: the wide record geometry of the frontier (GeoW up to 10 errors, the reach of the reference's
            // 64-bit in-index matrix; GeoX with its 16-row blocks beyond); the staged in-text matrices stop at 7, candidates are verified by
            // k_wide_filter + k_verify_wide
            b->wideEdit = st->metric == CMB_METRIC_EDIT && 3 * max_distance + 1 > MXW_LEFT;
            b->geoX = b->wideEdit && max_distance > MX_MAX_ED; // (beyond the reference's 64-bit in-index matrix: dev_matrix.hpp, MXN_*)
            try {
                b->wide = st->numPartsFor(max_distance) > (uint32_t)MAXP || b->wideEdit;
                if (max_distance > 13) return fail(CMB_ERR_UNSUPPORTED, "more than 13 errors (MAX_K, definitions.h:50)");
                if (b->wide) {
                    b->hostStratW = st->flatten<MAXP_WIDE>(max_distance);
                    b->sNumParts = b->hostStratW.numParts, b->sPartition = b->hostStratW.partition, b->sNSchemes = b->hostStratW.nSchemes;
                    for (int i = 0; i < b->hostStratW.nSchemes; i++) b->sMaxSearches = std::max<uint32_t>(b->sMaxSearches, b->hostStratW.sch[i].nSearches);
                } else {
                    b->hostStrat = st->flatten<MAXP>(max_distance);
                    b->sNumParts = b->hostStrat.numParts, b->sPartition = b->hostStrat.partition, b->sNSchemes = b->hostStrat.nSchemes;
                    for (int i = 0; i < b->hostStrat.nSchemes; i++) b->sMaxSearches = std::max<uint32_t>(b->sMaxSearches, b->hostStrat.sch[i].nSearches);
                }
            } catch (const std::exception& e) {
                return fail(CMB_ERR_INVALID, e.what());
            }
        }
        uint32_t maxLen = 1, minLen = ~0u;
        for (uint32_t i = 0; i < n_reads; i++) {
            if (offs[i + 1] < offs[i]) return fail(CMB_ERR_INVALID, "read offsets must be non-decreasing");
            maxLen = std::max<uint32_t>(maxLen, (uint32_t)(offs[i + 1] - offs[i]));
            minLen = std::min<uint32_t>(minLen, (uint32_t)(offs[i + 1] - offs[i]));
        }
        b->minLen = n_reads ? minLen : 0u;
        if (maxLen > (uint32_t)MAX_READ)
            return fail(CMB_ERR_UNSUPPORTED, "reads longer than " + std::to_string(MAX_READ) + " are not supported");
        // rows of per-read arrays are padded to a multiple of 16 bytes (16-byte stores in k_prep); the number of
        // 32-character words per read does not change
        maxLen = (maxLen + 15u) & ~15u;
        b->maxLen = maxLen;
        b->gw = gWords(maxLen);
        b->hostOffs.assign(offs, offs + n_reads + 1);
        HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        b->reads.uploadPadded((const uint8_t*)seqs, offs[n_reads], 64);
        b->offs.upload(offs, n_reads + 1);
        b->seq.alloc((size_t)2 * n_reads * maxLen);
        b->G.alloc((size_t)n_reads * 8 * b->gw); // eight bit-strings per read (both strands use them: gString)
        b->recW = ((1 + 2 * ((maxLen + 31) / 32)) + 3) / 4 * 4;
        b->rec.alloc((size_t)2 * n_reads * b->recW);
        if (b->wide) {
            b->stratW.upload(&b->hostStratW, 1);
            b->partsW.alloc((size_t)2 * n_reads);
        } else {
            b->strat.upload(&b->hostStrat, 1);
            b->parts.alloc((size_t)2 * n_reads);
        }
        b->dfs.alloc((size_t)n_reads * 2 + 4096);
        b->items.alloc((size_t)n_reads * 64 + 4096);
        b->fm.alloc((size_t)n_reads * 8 + 4096);
        b->text.alloc((size_t)n_reads * 48 + 4096);
        b->cnt.alloc(8);
        b->counters.alloc(CMB_CNT_MAX);
        *out = b.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

namespace {

struct Timer {
    hipStream_t s;
    hipEvent_t a, b;
    std::vector<KernelTime>& out;
    Timer(hipStream_t st, std::vector<KernelTime>& o) : s(st), out(o) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
    }
    ~Timer() {
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
            if (!strcmp(t.name, name)) {
                t.ms += ms;
                return;
            }
        out.push_back({name, ms});
    }
};

struct HostOcc {
    uint32_t begin, end, dist, strand;
};

// ordering of TextOcc (indexhelpers.h:779-795) with the strand as final, deterministic tie-break
inline bool occLess(const HostOcc& a, const HostOcc& b) {
    if (a.begin != b.begin) return a.begin < b.begin;
    if (a.dist != b.dist) return a.dist < b.dist;
    const uint32_t wa = a.end > a.begin ? a.end - a.begin : 0, wb = b.end > b.begin ? b.end - b.begin : 0;
    if (wa != wb) return wa < wb;
    return a.strand < b.strand;
}

} // namespace

static int batchRunOne(cmb_batch* b);

extern "C" int cmb_batch_run(cmb_batch* b) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    if (b->subs.empty()) return batchRunOne(b);
    b->done = false;
    std::vector<int> rc(b->subs.size(), CMB_OK);
    std::vector<std::string> err(b->subs.size());
    // CMB_SERIAL_SUBBATCHES: one sub-batch after the other — every kernel has the device to itself, which is what
    // per-kernel timings and counters want (bench.py times its throughput steps concurrently and takes the
    // kernel table from one extra serial step)
    const bool serial = getenv("CMB_SERIAL_SUBBATCHES") != nullptr;
    // workers take the sub-batches in order, three at a time (the number that pays, cmb_batch_create); CMB_MAX_CONCURRENT=m: at most m
    const char* mcEnv = getenv("CMB_MAX_CONCURRENT");
    size_t nWorkers = serial ? 1 : std::min<size_t>(b->subs.size(), 3);
    if (mcEnv && !serial) nWorkers = std::min<size_t>(b->subs.size(), std::max(1, atoi(mcEnv)));
    // (beyond 7 errors, where the sub-batches are many and their pools large, a finished sub-batch gives its pools back: the next one allocates what it needs)
    const bool manySubs = b->subs.size() > 3 && b->k > 7;
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (size_t w = 0; w < nWorkers; w++)
        th.emplace_back([&] {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= b->subs.size()) break;
                cmb_batch* c = b->subs[j];
                rc[j] = batchRunOne(c);
                if (rc[j] != CMB_OK) err[j] = cmb_last_error(); // (the message lives in the worker's thread-local storage)
                if (manySubs) { // (results, alignments and counters are on the host; a second run allocates again)
                    for (int q2 = 0; q2 < 2; q2++) c->bfsQ[q2].release(), c->bfsEv[q2].release();
                    c->bfsF.release(), c->bfsC.release(), c->bfsA.release(), c->dpSlab.release(), c->dpList.release(), c->vW.release();
                    c->sortTmp.release(), c->scanTmp.release(), c->vkeysA.release(), c->vkeysB.release();
                }
            }
        });
    for (auto& t : th) t.join();
    for (size_t j = 0; j < rc.size(); j++)
        if (rc[j] != CMB_OK) return fail(rc[j], err[j]);
    memset(b->cnts, 0, sizeof(b->cnts));
    b->times.clear();
    for (cmb_batch* c : b->subs) {
        for (int i = 0; i < CMB_CNT_MAX; i++) b->cnts[i] += c->cnts[i];
        for (const auto& t : c->times) { // busy time per kernel group, summed over the concurrent sub-batches
            bool found = false;
            for (auto& u : b->times)
                if (!strcmp(u.name, t.name)) {
                    u.ms += t.ms;
                    found = true;
                }
            if (!found) b->times.push_back(t);
        }
    }
    b->done = true;
    return CMB_OK;
}

// Streaming: the reads of the NEXT chunk (same number of reads, none longer than the batch was created for) are copied
// to the device on a stream of their own while the current chunk is matched; the next cmb_batch_run takes them.  The
// host memory must stay valid until that run has started (page-locked memory makes the copy asynchronous and fast).
extern "C" int cmb_batch_stage_reads(cmb_batch* b, const char* seqs, const uint64_t* offs, uint32_t n_reads) {
    if (!b || !offs || (!seqs && n_reads)) return fail(CMB_ERR_INVALID, "null argument");
    if (n_reads != b->nReads) return fail(CMB_ERR_INVALID, "a staged chunk must hold as many reads as the batch was created with");
    if (!b->subs.empty()) {
        for (size_t j = 0; j < b->subs.size(); j++) {
            const uint32_t lo = b->subBound[j], hi = b->subBound[j + 1];
            std::vector<uint64_t> o(hi - lo + 1);
            for (uint32_t i = lo; i <= hi; i++) o[i - lo] = offs[i] - offs[lo];
            const int rc = cmb_batch_stage_reads(b->subs[j], seqs + offs[lo], o.data(), hi - lo);
            if (rc != CMB_OK) return rc;
        }
        return CMB_OK;
    }
    uint32_t maxLen = 1, minLen = ~0u;
    for (uint32_t i = 0; i < n_reads; i++) {
        if (offs[i + 1] < offs[i]) return fail(CMB_ERR_INVALID, "read offsets must be non-decreasing");
        maxLen = std::max<uint32_t>(maxLen, (uint32_t)(offs[i + 1] - offs[i]));
        minLen = std::min<uint32_t>(minLen, (uint32_t)(offs[i + 1] - offs[i]));
    }
    b->minLen = std::min(b->minLen, n_reads ? minLen : 0u);
    if (maxLen > b->maxLen) return fail(CMB_ERR_INVALID, "a staged read is longer than the batch was created for");
    if (b->pending) return fail(CMB_ERR_INVALID, "a chunk is already registered for the next run");
    b->hostOffsStage.assign(offs, offs + n_reads + 1);
    b->offsPin[b->pinNext].resize((size_t)n_reads + 1);
    memcpy(b->offsPin[b->pinNext].p, offs, ((size_t)n_reads + 1) * sizeof(uint64_t));
    b->pinStaged = b->pinNext;
    b->pinNext ^= 1;
    b->pendingSeqs = seqs;
    b->pending = true;
    return CMB_OK;
}
// start the upload of the registered chunk on the copy stream (called by a run, right after its own reads are in place)
static void startStagedUpload(cmb_batch* b) {
    if (!b->copyStream) {
        HIPCHK(hipStreamCreateWithFlags(&b->copyStream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->copyDone, hipEventDisableTiming));
    }
    const uint32_t n = b->nReads;
    const size_t nChars = b->hostOffsStage[n];
    if (b->readsStage.n < nChars) b->readsStage.alloc(nChars + nChars / 16 + 256);
    if (b->offsStage.n < (size_t)n + 1) b->offsStage.alloc((size_t)n + 1);
    if (nChars) HIPCHK(hipMemcpyAsync(b->readsStage.p, b->pendingSeqs, nChars, hipMemcpyHostToDevice, b->copyStream));
    HIPCHK(hipMemcpyAsync(b->offsStage.p, b->offsPin[b->pinStaged].p, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                          b->copyStream));
    HIPCHK(hipEventRecord(b->copyDone, b->copyStream));
    b->pending = false;
    b->staged = true;
}

static int batchRunOne(cmb_batch* b) {
    try {
        cmb_index* ix = b->ix;
        useDevice(ix->device);
        hipStream_t s = b->stream;
        if (b->staged) { // the chunk uploaded during the previous run becomes the batch's reads
            HIPCHK(hipEventSynchronize(b->copyDone));
            std::swap(b->reads.p, b->readsStage.p);
            std::swap(b->reads.n, b->readsStage.n);
            std::swap(b->offs.p, b->offsStage.p);
            std::swap(b->offs.n, b->offsStage.n);
            b->hostOffs.swap(b->hostOffsActive);
            b->staged = false;
        }
        if (b->pending) { // the chunk registered since travels to the device while this run's kernels execute
            startStagedUpload(b);
            b->hostOffsActive.swap(b->hostOffsStage); // (offsets of the chunk in flight: the batch's from the next run on; the next registration refills hostOffsStage)
        }
        b->times.clear();
        b->done = false;
        b->hasNaive = false;
        Timer tm(s, b->times);
        const uint32_t nReads = b->nReads;
        const uint32_t tasks = 2 * nReads;
        if (nReads == 0) {
            b->occs.resize(0);
            b->occOffs.resize(1);
            b->occOffs.p[0] = 0;
            memset(b->cnts, 0, sizeof(b->cnts));
            b->done = true;
            return CMB_OK;
        }
        uint32_t hcnt[8];
        const bool verbose = getenv("CMB_VERBOSE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!verbose) return;
            auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[host] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        Queues q{};
        q.cnt = b->cnt.p;
        q.counters = b->counters.p;

        HIPCHK(hipMemsetAsync(b->counters.p, 0, CMB_CNT_MAX * sizeof(unsigned long long), s));
        tm.begin();
        {
            HIPCHK(hipMemsetAsync(b->G.p, 0, b->G.bytes(), s));
            HIPCHK(hipMemsetAsync(b->rec.p, 0, b->rec.bytes(), s));
            const uint32_t chunks = (b->maxLen + 31) / 32;
            const uint64_t nthr = (uint64_t)nReads * chunks;
            hipLaunchKernelGGL(k_prep, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, b->reads.p, b->offs.p,
                               nReads, b->maxLen, b->gw, chunks, b->seq.p, b->G.p, b->rec.p, b->recW);
        }
        MFull mf{nullptr, 0};
        if (b->metric == CMB_METRIC_EDIT && b->k > 0) {
            mf.nBlk = mfullBlocks(b->maxLen);
            const uint64_t nW = (uint64_t)tasks * mf.nBlk;
            if (b->mfull.n < 2 * nW) b->mfull.alloc(2 * nW);
            hipLaunchKernelGGL(k_match_words, dim3((unsigned)((nW + 255) / 256)), dim3(256), 0, s, b->G.p, b->gw, b->offs.p, tasks,
                               mf.nBlk, b->mfull.p);
            mf.p = b->mfull.p;
        }
        tm.end("k_prep");

        // ---- prologue + DFS (re-run with larger queues if they overflow: nothing is truncated)
        const uint32_t pCap = getenv("CMB_P_SLOTS") ? (uint32_t)std::max(256, atoi(getenv("CMB_P_SLOTS"))) / 256u * 256u : 256u * 32768u; // (one lane per read x strand
        // up to 8 M: k_parts 65.7 ms with 512 k lanes looping over the tasks, 65.0 / 62.1 / 62.5 ms with 1 M / 4 M / 8 M)
        const uint32_t pSlots = std::min<uint32_t>(((tasks + 255) / 256) * 256, pCap);
        const bool preset = !b->presetItems.empty();
        if (preset) { // (no search: the candidates come from the caller)
            if (b->items.n < b->presetItems.size()) b->items.alloc(b->presetItems.size() + 1024);
            HIPCHK(hipMemsetAsync(b->cnt.p, 0, 8 * sizeof(uint32_t), s));
            HIPCHK(hipMemcpyAsync(b->items.p, b->presetItems.data(), b->presetItems.size() * sizeof(uint4), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            memset(hcnt, 0, sizeof(hcnt));
            hcnt[0] = (uint32_t)b->presetItems.size();
            q.items = b->items.p;
            q.itemCap = (uint32_t)std::min<size_t>(b->items.n, 0xFFFFFFF0u);
        }
        int bfsAttempts = 0;
        for (int attempt = 0; !preset; attempt++) {
            HIPCHK(hipMemsetAsync(b->cnt.p, 0, 8 * sizeof(uint32_t), s));
            if (attempt) {
                HIPCHK(hipMemsetAsync(b->counters.p, 0, CMB_CNT_MAX * sizeof(unsigned long long), s));
            }
            q.items = b->items.p;
            q.itemCap = (uint32_t)std::min<size_t>(b->items.n, 0xFFFFFFF0u);
            q.fm = b->fm.p;
            q.fmCap = (uint32_t)std::min<size_t>(b->fm.n, 0xFFFFFFF0u);
            q.text = b->text.p;
            q.textCap = (uint32_t)std::min<size_t>(b->text.n, 0xFFFFFFF0u);
            const uint32_t dfsCap = (uint32_t)std::min<size_t>(b->dfs.n, 0xFFFFFFF0u);
            tm.begin();
            const uint32_t pParts = b->k ? b->sNumParts : 1;
            const uint32_t stratBytes = (uint32_t)(((b->wide ? sizeof(DevStrategyKT<MAXP_WIDE>) : sizeof(DevStrategyK)) + 15) / 16 * 16);
            const uint32_t rdWords = 2 * ((b->maxLen + 31) / 32);
            uint32_t nSlots = 1; // k = 0: one exact search per read x strand
            const bool longReads = b->maxLen > 256;
            if (b->k) {
                if (b->exr.n < (size_t)pParts * tasks) {
                    b->exr.alloc((size_t)pParts * tasks);
                    b->psel.alloc(tasks);
                }
                const size_t pLds = stratBytes + (5 * pParts + rdWords) * 256 * sizeof(uint32_t);
                if (b->wide) {
                    constexpr int W = MAXP_WIDE;
                    auto kp = longReads ? (b->sPartition == 0 ? k_parts<0, true, W> : b->sPartition == 1 ? k_parts<1, true, W> : k_parts<2, true, W>)
                                        : (b->sPartition == 0 ? k_parts<0, false, W> : b->sPartition == 1 ? k_parts<1, false, W> : k_parts<2, false, W>);
                    if (pLds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pLds));
                    hipLaunchKernelGGL(kp, dim3(pSlots / 256), dim3(256), pLds, s, ix->d, b->stratW.p, nReads, b->k, b->maxLen, b->seq.p,
                                       (const uint4*)b->rec.p, b->recW / 4, b->partsW.p, b->exr.p, b->psel.p, q);
                } else {
                    auto kp = longReads ? (b->sPartition == 0 ? k_parts<0, true> : b->sPartition == 1 ? k_parts<1, true> : k_parts<2, true>)
                                        : (b->sPartition == 0 ? k_parts<0, false> : b->sPartition == 1 ? k_parts<1, false> : k_parts<2, false>);
                    hipLaunchKernelGGL(kp, dim3(pSlots / 256), dim3(256), pLds, s, ix->d, b->strat.p, nReads,
                                   b->k, b->maxLen, b->seq.p, (const uint4*)b->rec.p, b->recW / 4, b->parts.p, b->exr.p,
                                   b->psel.p, q);
                }
                nSlots = b->sMaxSearches + 1; // + the part-level pre-verification
            }
            const uint64_t eTasks = (uint64_t)tasks * nSlots;
            const uint32_t eSlots = (uint32_t)std::min<uint64_t>(((eTasks + 255) / 256) * 256, 256ull * 4096ull);
            const size_t eLds = stratBytes + (pParts + rdWords) * 256 * sizeof(uint32_t);
            if (b->wide)
                hipLaunchKernelGGL((longReads ? k_exact<true, MAXP_WIDE> : k_exact<false, MAXP_WIDE>), dim3(eSlots / 256), dim3(256), eLds, s, ix->d,
                                   b->stratW.p, nReads, b->k, b->maxLen, nSlots, b->seq.p, (const uint4*)b->rec.p, b->recW / 4, b->partsW.p,
                                   b->exr.p, b->psel.p, b->dfs.p, dfsCap, q);
            else
                hipLaunchKernelGGL(longReads ? k_exact<true> : k_exact<false>, dim3(eSlots / 256), dim3(256), eLds,
                               s, ix->d, b->strat.p, nReads, b->k, b->maxLen, nSlots, b->seq.p, (const uint4*)b->rec.p,
                               b->recW / 4, b->parts.p, b->exr.p, b->psel.p, b->dfs.p, dfsCap, q);
            tm.end("k_partition");
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            uint32_t flags = hcnt[3];
            // ---- reads not longer than the number of parts (and every read of a one-part strategy): naive backtracking
            // (searchstrategy.cpp:148-152, :442-459 -> dev_bfs_naive.hpp); k_parts marked them in psel
            b->hasNaive = (flags & FLAG_UNSUPPORTED_READ) != 0;
            if (b->hasNaive) {
                tm.begin();
                const uint32_t maxPassN = b->maxLen + 2 * b->k + 4; // (rows of the matrix: len + 2 k + 1 at most)
                if (!b->nvQCap) b->nvQCap = getenv("CMB_TEST_SMALL_POOLS") ? 64 : 16384;
                for (int j = 0; j < 2; j++)
                    if (b->nvQ[j].n < 3 * b->nvQCap) b->nvQ[j].alloc(3 * b->nvQCap);
                const size_t cntWords = (size_t)maxPassN + 2;
                if (b->nvCnt.n < cntWords) b->nvCnt.alloc(cntWords);
                HIPCHK(hipMemsetAsync(b->nvCnt.p, 0, cntWords * sizeof(uint32_t), s));
                NaiveBufs N{};
                N.Q[0] = b->nvQ[0].p;
                N.Q[1] = b->nvQ[1].p;
                N.qCap = (uint32_t)std::min<size_t>(b->nvQ[0].n / 3, 0xFFFFFFF0u);
                N.nq = b->nvCnt.p;
                const bool edit = b->metric == CMB_METRIC_EDIT;
                hipLaunchKernelGGL(k_naive_start, dim3((tasks + 255) / 256), dim3(256), 0, s, ix->d, b->psel.p, b->offs.p, tasks,
                                   b->k, edit ? 0u : 1u, N, q, b->geoX ? 1u : 0u);
                std::vector<uint32_t> hc(cntWords);
                uint32_t pass = 0, peakQ = 0;
                bool drained = false;
                auto kNaive = edit ? k_naive_pass<true, false> : k_naive_pass<false, false>;
                if (b->geoX) kNaive = k_naive_pass<true, true>;
                while (!drained && pass < maxPassN) {
                    const uint32_t upTo = std::min(pass + 16u, maxPassN);
                    for (; pass < upTo; pass++)
                        hipLaunchKernelGGL(kNaive, dim3(BFS_GRID), dim3(256), 0, s, ix->d, N,
                                           pass, b->offs.p, b->gw, b->G.p, b->seq.p, b->maxLen, b->k, q);
                    HIPCHK(hipMemcpyAsync(hc.data(), b->nvCnt.p, cntWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                    HIPCHK(hipStreamSynchronize(s));
                    if (hcnt[3] & (FLAG_NAIVE_Q | FLAG_ITEM_OVERFLOW | FLAG_FMOCC_OVERFLOW)) break;
                    drained = hc[pass] == 0;
                }
                for (uint32_t p2 = 0; p2 <= pass && p2 < cntWords; p2++) peakQ = std::max(peakQ, hc[p2]);
                if (verbose) fprintf(stderr, "[naive] %u roots, %u passes, peak frontier %u\n", hc[0], pass, peakQ);
                tm.end("k_naive");
                HIPCHK(hipGetLastError());
                flags = hcnt[3];
                if (flags & FLAG_NAIVE_Q) {
                    if (attempt >= 60) return fail(CMB_ERR_INTERNAL, "the naive search's frontier keeps overflowing");
                    b->nvQCap = std::max<size_t>(4 * b->nvQCap, (size_t)peakQ + peakQ / 4);
                    continue;
                }
                if (!drained && !(flags & (FLAG_ITEM_OVERFLOW | FLAG_FMOCC_OVERFLOW)))
                    return fail(CMB_ERR_INTERNAL, "the naive search did not finish within its pass bound");
            }
            if (flags & FLAG_SEED_OVERLAP)
                return fail(CMB_ERR_INVALID,
                            "dynamic partitioning: the seeds of a read overlap — the k-mer size of the index is too large "
                            "for the seeding positions of this search strategy at this read length (the reference caps "
                            "the k-mer size, e.g. at 4 for kuch2 and 01*0: alignparameters.cpp:1070-1114, :1275-1278)");
            const uint32_t nDfs = hcnt[5];
            if (!(flags & (FLAG_ITEM_OVERFLOW | FLAG_DFS_OVERFLOW)) && nDfs) {
                tm.begin();
                if (b->metric == CMB_METRIC_EDIT) {
                    // ---- frontier search: start pass, then (expand, events) per level until both queues drain
                    const uint32_t maxPass = 2 * b->maxLen + 8 * (b->wide ? MAXP_WIDE : MAXP) + 64;
                    // node planes / uint4 per event of the record geometry (dev_bfs_edit.hpp: GeoN, GeoW)
                    const size_t qPlanes = 3 + (b->wide ? GeoW::PK_U4 : GeoN::PK_U4), evU4 = 1 + (b->wide ? GeoW::PK_U4 : GeoN::PK_U4);
                    if (!b->bfsQCap) {
                        // first guess; every pool grows (and the search re-runs) when it turns out too small.
                        // CMB_TEST_SMALL_POOLS starts from almost nothing so that tests exercise that path.
                        const size_t slack = getenv("CMB_TEST_SMALL_POOLS") ? 64 : 65536;
                        const size_t per = getenv("CMB_TEST_SMALL_POOLS") ? 0 : 1;
                        b->bfsQCap = per * (size_t)nReads * 2 + slack;
                        b->bfsEvCap = per * (size_t)nReads / 2 + slack;
                        b->bfsFCap = per * (size_t)nReads * 16 + slack;
                        b->bfsCCap = per * (size_t)nReads * 2 + slack;
                        b->bfsACap = per * (size_t)nReads * 8 + slack;
                    }
                    b->bfsQCap = std::max<size_t>(b->bfsQCap, (size_t)nDfs + 1024);
                    for (int j = 0; j < 2; j++) {
                        if (b->bfsQ[j].n < qPlanes * b->bfsQCap) b->bfsQ[j].alloc(qPlanes * b->bfsQCap);
                        if (b->bfsEv[j].n < evU4 * b->bfsEvCap) b->bfsEv[j].alloc(evU4 * b->bfsEvCap);
                    }
                    if (b->bfsF.n < F_U4 * b->bfsFCap) b->bfsF.alloc(F_U4 * b->bfsFCap);
                    // (GeoX: blocks of 16 rows — up to 32 of them for 480 + 26 rows, two uint4 of match words each)
                    const uint32_t ctxU4 = b->geoX ? CTX_U4_X : ctxU4For(b->maxLen);
                    if (b->bfsC.n < (size_t)ctxU4 * b->bfsCCap) b->bfsC.alloc((size_t)ctxU4 * b->bfsCCap);
                    if (b->bfsA.n < b->bfsACap) b->bfsA.alloc(b->bfsACap);
                    const size_t cntWords = 2 * ((size_t)maxPass + 2) + 4;
                    if (b->bfsCnt.n < cntWords) b->bfsCnt.alloc(cntWords);
                    if (b->bfsBlockCnt.n < (size_t)BFS_GRID_CNT * 4) b->bfsBlockCnt.alloc((size_t)BFS_GRID_CNT * 4);
                    HIPCHK(hipMemsetAsync(b->bfsCnt.p, 0, cntWords * sizeof(uint32_t), s));
                    HIPCHK(hipMemsetAsync(b->bfsBlockCnt.p, 0, (size_t)BFS_GRID_CNT * 4 * sizeof(unsigned long long), s));
                    BfsBufs B{};
                    for (int j = 0; j < 2; j++) {
                        B.Q[j] = b->bfsQ[j].p;
                        B.Ev[j] = b->bfsEv[j].p;
                    }
                    B.F = b->bfsF.p;
                    B.C = b->bfsC.p;
                    B.A = b->bfsA.p;
                    B.qCap = (uint32_t)std::min<size_t>(b->bfsQ[0].n / qPlanes, 0xFFFFFFF0u);
                    B.evCap = (uint32_t)std::min<size_t>(b->bfsEv[0].n / evU4, 0xFFFFFFF0u);
                    B.fCap = (uint32_t)std::min<size_t>(b->bfsF.n / F_U4, 0xFFFFFFF0u);
                    B.cCap = (uint32_t)std::min<size_t>(b->bfsC.n / ctxU4, 0xFFFFFFF0u);
                    B.ctxU4 = ctxU4;
                    B.ctxMblk = b->geoX ? CTX_MBLK_X : ctxMblkFor(b->maxLen);
                    B.aCap = (uint32_t)std::min<size_t>(b->bfsA.n, 0xFFFFFFF0u);
                    B.chain = getenv("CMB_BFS_CHAIN") ? (uint32_t)std::max(1, atoi(getenv("CMB_BFS_CHAIN"))) : BFS_CHAIN;
                    B.gridX = getenv("CMB_BFS_GRID") ? (uint32_t)std::min<int>(BFS_GRID_CNT, std::max(1, atoi(getenv("CMB_BFS_GRID"))))
                                                     : BFS_GRID_X;
                    B.gridEv = getenv("CMB_BFS_GRID_EV") ? (uint32_t)std::max(1, atoi(getenv("CMB_BFS_GRID_EV"))) : BFS_GRID_EV;
                    // (CMB_TEST_NARROW_WV: tests lower the bound so that the re-run on the 64-bit geometry is exercised)
                    B.narrowWv = getenv("CMB_TEST_NARROW_WV") ? (uint32_t)std::max(0, atoi(getenv("CMB_TEST_NARROW_WV"))) : 0xFFFFu;
                    B.nq = b->bfsCnt.p;
                    B.ne = b->bfsCnt.p + (maxPass + 2);
                    B.pool = b->bfsCnt.p + 2 * (maxPass + 2);
                    B.blockCnt = b->bfsBlockCnt.p;
                    // up to 6 errors the frontier carries the in-index matrix on 32-bit words (GeoN32, dev_matrix.hpp: MXS_*); CMB_MATRIX64=1
                    // keeps the reference's 64-bit words (GeoN), as does a batch one of whose phases did not fit the small matrix
                    const bool small32 = !b->wide && b->k <= MXS_MAX_ED && !b->noSmallMatrix && !getenv("CMB_MATRIX64");
                    if (b->geoX)
                        hipLaunchKernelGGL(k_bfs_start<GeoX>, dim3(std::min<uint32_t>((nDfs + 255) / 256, BFS_GRID)), dim3(256), 0, s,
                                           ix->d, b->stratW.p, B, b->dfs.p, nDfs, b->offs.p, b->gw, b->G.p, b->partsW.p, q);
                    else if (b->wide)
                        hipLaunchKernelGGL(k_bfs_start<GeoW>, dim3(std::min<uint32_t>((nDfs + 255) / 256, BFS_GRID)), dim3(256), 0, s,
                                           ix->d, b->stratW.p, B, b->dfs.p, nDfs, b->offs.p, b->gw, b->G.p, b->partsW.p, q);
                    else if (small32)
                        hipLaunchKernelGGL(k_bfs_start<GeoN32>, dim3(std::min<uint32_t>((nDfs + 255) / 256, BFS_GRID)), dim3(256), 0, s,
                                           ix->d, b->strat.p, B, b->dfs.p, nDfs, b->offs.p, b->gw, b->G.p, b->parts.p, q);
                    else
                        hipLaunchKernelGGL(k_bfs_start<GeoN>, dim3(std::min<uint32_t>((nDfs + 255) / 256, BFS_GRID)), dim3(256), 0, s,
                                       ix->d, b->strat.p, B, b->dfs.p, nDfs, b->offs.p, b->gw, b->G.p, b->parts.p, q);
                    std::vector<uint32_t> hc(cntWords);
                    // passes between two looks at the queue sizes (a pass on a drained frontier costs the device ~4 us, a look costs the host a round trip)
                    const uint32_t CHECK = getenv("CMB_BFS_CHECK") ? (uint32_t)std::max(1, atoi(getenv("CMB_BFS_CHECK"))) : 16;
                    uint32_t pass = 0, peakQ = 0, peakEv = 0;
                    bool drained = false;
                    while (!drained && pass < maxPass) {
                        const uint32_t upTo = std::min(pass + CHECK, maxPass);
                        for (; pass < upTo; pass++) {
                            if (b->geoX)
                                hipLaunchKernelGGL(k_bfs_pass<GeoX>, dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d, b->stratW.p, B,
                                                   pass, b->offs.p, b->gw, b->G.p, b->partsW.p, q);
                            else if (b->wide)
                                hipLaunchKernelGGL(k_bfs_pass<GeoW>, dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d, b->stratW.p, B,
                                                   pass, b->offs.p, b->gw, b->G.p, b->partsW.p, q);
                            else if (small32)
                                hipLaunchKernelGGL(k_bfs_pass<GeoN32>, dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d, b->strat.p, B,
                                                   pass, b->offs.p, b->gw, b->G.p, b->parts.p, q);
                            else
                                hipLaunchKernelGGL(k_bfs_pass<GeoN>, dim3(B.gridX + B.gridEv), dim3(256), 0, s, ix->d, b->strat.p, B,
                                               pass, b->offs.p, b->gw, b->G.p, b->parts.p, q);
                        }
                        HIPCHK(hipMemcpyAsync(hc.data(), b->bfsCnt.p, cntWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                        HIPCHK(hipStreamSynchronize(s));
                        if (hcnt[3] & BFS_STOP) break;
                        drained = hc[pass] == 0 && hc[maxPass + 2 + pass] == 0;
                    }
                    for (uint32_t p2 = 0; p2 <= pass && p2 < maxPass + 2; p2++) {
                        peakQ = std::max(peakQ, hc[p2]);
                        peakEv = std::max(peakEv, hc[maxPass + 2 + p2]);
                    }
                    const uint32_t* pool = hc.data() + 2 * (maxPass + 2);
                    if (getenv("CMB_VERBOSE"))
                        fprintf(stderr, "[bfs] %u tasks, %u passes, peak frontier %u, peak events %u, F %u, contexts %u, arena %u\n",
                                nDfs, pass, peakQ, peakEv, pool[0], pool[1], pool[2]);
                    if (getenv("CMB_VERBOSE") && atoi(getenv("CMB_VERBOSE")) > 1)
                        for (uint32_t p2 = 0; p2 <= pass; p2++)
                            fprintf(stderr, "  pass %u: %u nodes, %u events\n", p2, hc[p2], hc[maxPass + 2 + p2]);
                    hipLaunchKernelGGL(k_bfs_finish, dim3(1), dim3(256), 0, s, B, q);
                    if (hcnt[3] & FLAG_NARROW_MATRIX) { // (a first column wider than the small matrix holds: once more on the reference's words)
                        if (verbose) fprintf(stderr, "[bfs] a phase did not fit the 32-bit in-index matrix: re-running on 64-bit words\n");
                        b->noSmallMatrix = true;
                        HIPCHK(hipStreamSynchronize(s));
                        tm.end("k_dfs");
                        continue;
                    }
                    if (hcnt[3] & (FLAG_BFS_Q | FLAG_BFS_EV | FLAG_BFS_F | FLAG_BFS_CTX | FLAG_BFS_ARENA)) {
                        // a pool was too small: grow what was asked for (at least x2) and run the search again
                        // (an attempt only shows the demand up to the pass that overflowed: beyond 7 errors, where the first guesses are far
                        // off — 64 reads at 12 errors on 3 Gbp took eleven attempts at x 2 —, the pools grow x 4)
                        if (++bfsAttempts >= 48) return fail(CMB_ERR_INTERNAL, "frontier pools keep overflowing");
                        const size_t gf = b->k > 7 ? 4 : 2;
                        if (hcnt[3] & FLAG_BFS_Q) b->bfsQCap = std::max<size_t>(gf * b->bfsQCap, (size_t)peakQ + peakQ / 4);
                        if (hcnt[3] & FLAG_BFS_EV) b->bfsEvCap = std::max<size_t>(gf * b->bfsEvCap, (size_t)peakEv + peakEv / 4);
                        if (hcnt[3] & FLAG_BFS_F) b->bfsFCap = std::max<size_t>(gf * b->bfsFCap, (size_t)pool[0] + pool[0] / 4);
                        if (hcnt[3] & FLAG_BFS_CTX) b->bfsCCap = std::max<size_t>(gf * b->bfsCCap, (size_t)pool[1] + pool[1] / 4);
                        if (hcnt[3] & FLAG_BFS_ARENA) b->bfsACap = std::max<size_t>(gf * b->bfsACap, (size_t)pool[2] + pool[2] / 4);
                        HIPCHK(hipStreamSynchronize(s));
                        tm.end("k_dfs");
                        continue;
                    }
                    if (!drained && !(hcnt[3] & BFS_STOP))
                        return fail(CMB_ERR_INTERNAL, "frontier search did not finish within its pass bound");
                } else {
                    // ---- Hamming distance: the same frontier idea without a matrix (dev_bfs_hamming.hpp)
                    const uint32_t maxPass = b->maxLen + 2 * (b->wide ? MAXP_WIDE : MAXP) + 16;
                    if (!b->bfsQCap) b->bfsQCap = (getenv("CMB_TEST_SMALL_POOLS") ? 0 : (size_t)nReads * 4) + 1024;
                    b->bfsQCap = std::max<size_t>(b->bfsQCap, (size_t)nDfs + 1024);
                    for (int j = 0; j < 2; j++)
                        if (b->bfsQ[j].n < 2 * b->bfsQCap) b->bfsQ[j].alloc(2 * b->bfsQCap);
                    const size_t cntWords = (size_t)maxPass + 2;
                    if (b->bfsCnt.n < cntWords) b->bfsCnt.alloc(cntWords);
                    if (b->bfsBlockCnt.n < (size_t)BFS_GRID_CNT * 4) b->bfsBlockCnt.alloc((size_t)BFS_GRID_CNT * 4);
                    HIPCHK(hipMemsetAsync(b->bfsCnt.p, 0, cntWords * sizeof(uint32_t), s));
                    HIPCHK(hipMemsetAsync(b->bfsBlockCnt.p, 0, (size_t)BFS_GRID_CNT * 4 * sizeof(unsigned long long), s));
                    HbfsBufs H{};
                    H.Q[0] = b->bfsQ[0].p;
                    H.Q[1] = b->bfsQ[1].p;
                    H.qCap = (uint32_t)std::min<size_t>(b->bfsQ[0].n / 2, 0xFFFFFFF0u);
                    H.nq = b->bfsCnt.p;
                    H.blockCnt = b->bfsBlockCnt.p;
                    const uint32_t hLds = stratBytes;
                    if (b->wide)
                        hipLaunchKernelGGL((k_hbfs<true, MAXP_WIDE>), dim3(std::min<uint32_t>((nDfs + 255) / 256, BFS_GRID)), dim3(256), hLds, s,
                                           ix->d, b->stratW.p, H, 0u, b->dfs.p, nDfs, b->maxLen, b->seq.p, b->partsW.p, q);
                    else
                        hipLaunchKernelGGL(k_hbfs<true>, dim3(std::min<uint32_t>((nDfs + 255) / 256, BFS_GRID)), dim3(256), hLds, s,
                                       ix->d, b->strat.p, H, 0u, b->dfs.p, nDfs, b->maxLen, b->seq.p, b->parts.p, q);
                    std::vector<uint32_t> hc(cntWords);
                    uint32_t pass = 0, peakQ = 0;
                    bool drained = false;
                    while (!drained && pass < maxPass) {
                        const uint32_t upTo = std::min(pass + 16u, maxPass);
                        for (; pass < upTo; pass++)
                            if (b->wide)
                                hipLaunchKernelGGL((k_hbfs<false, MAXP_WIDE>), dim3(BFS_GRID), dim3(256), hLds, s, ix->d, b->stratW.p, H, pass,
                                                   (const DfsTask*)nullptr, 0u, b->maxLen, b->seq.p, b->partsW.p, q);
                            else
                                hipLaunchKernelGGL(k_hbfs<false>, dim3(BFS_GRID), dim3(256), hLds, s, ix->d, b->strat.p, H, pass,
                                               (const DfsTask*)nullptr, 0u, b->maxLen, b->seq.p, b->parts.p, q);
                        HIPCHK(hipMemcpyAsync(hc.data(), b->bfsCnt.p, cntWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                        HIPCHK(hipStreamSynchronize(s));
                        if (hcnt[3] & BFS_STOP) break;
                        drained = hc[pass] == 0;
                    }
                    for (uint32_t p2 = 0; p2 <= pass && p2 < cntWords; p2++) peakQ = std::max(peakQ, hc[p2]);
                    if (getenv("CMB_VERBOSE")) fprintf(stderr, "[hbfs] %u tasks, %u passes, peak frontier %u\n", nDfs, pass, peakQ);
                    BfsBufs Bf{};
                    Bf.blockCnt = b->bfsBlockCnt.p;
                    hipLaunchKernelGGL(k_bfs_finish, dim3(1), dim3(256), 0, s, Bf, q);
                    if (hcnt[3] & FLAG_BFS_Q) {
                        if (++bfsAttempts >= 24) return fail(CMB_ERR_INTERNAL, "frontier keeps overflowing");
                        b->bfsQCap = std::max<size_t>(2 * b->bfsQCap, (size_t)peakQ + peakQ / 4);
                        HIPCHK(hipStreamSynchronize(s));
                        tm.end("k_dfs");
                        continue;
                    }
                    if (!drained && !(hcnt[3] & BFS_STOP))
                        return fail(CMB_ERR_INTERNAL, "frontier search did not finish within its pass bound");
                }
                tm.end("k_dfs");
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                flags = hcnt[3];
            }
            if (flags & FLAG_CAPACITY)
                return fail(CMB_ERR_INTERNAL, "device search capacity exceeded (band width / descendants / stack)");
            if (flags & (FLAG_ITEM_OVERFLOW | FLAG_FMOCC_OVERFLOW | FLAG_DFS_OVERFLOW)) {
                if (attempt >= 60) return fail(CMB_ERR_INTERNAL, "work queues keep overflowing");
                // (a search that stopped at the overflow has only counted what it needed up to there: the naive search of very
                // short reads, which match all over the text, asks for orders of magnitude more than the first guess)
                const size_t grow = b->hasNaive ? 8 : 2;
                if (hcnt[0] > q.itemCap) b->items.alloc((size_t)hcnt[0] * grow + 1024);
                if (hcnt[1] > q.fmCap) b->fm.alloc((size_t)hcnt[1] * grow + 1024);
                if (hcnt[5] > dfsCap) b->dfs.alloc((size_t)hcnt[5] + hcnt[5] / 8 + 1024);
                continue;
            }
            break;
        }
        const uint32_t nItems = hcnt[0], nFm = hcnt[1];
        lap("prep + search");

        // ---- de-duplicate the in-index occurrences (Occurrences::eraseDoublesFM, indexhelpers.h:2135-2146)
        uint32_t nFmUniq = 0;
        if (nFm) {
            if (b->fmKeysA.n < nFm) {
                b->fmKeysA.alloc((size_t)nFm + nFm / 4 + 256);
                b->fmKeysB.alloc(b->fmKeysA.n);
                b->fmIdxA.alloc(b->fmKeysA.n);
                b->fmIdxB.alloc(b->fmKeysA.n);
                b->fmUniq.alloc(b->fmKeysA.n);
            }
            if (b->fmN.n < 1) b->fmN.alloc(1);
            HIPCHK(hipMemsetAsync(b->fmN.p, 0, sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_fm_keys, dim3((nFm + 255) / 256), dim3(256), 0, s, b->fm.p, nFm, b->fmKeysA.p, b->fmIdxA.p);
            size_t tmpBytes = 0;
            HIPCHK(rocprim::radix_sort_pairs(nullptr, tmpBytes, b->fmKeysA.p, b->fmKeysB.p, b->fmIdxA.p, b->fmIdxB.p, nFm, 0,
                                             64, s));
            if (b->sortTmp.n < tmpBytes) b->sortTmp.alloc(tmpBytes + 256);
            HIPCHK(rocprim::radix_sort_pairs(b->sortTmp.p, tmpBytes, b->fmKeysA.p, b->fmKeysB.p, b->fmIdxA.p, b->fmIdxB.p,
                                             nFm, 0, 64, s));
            hipLaunchKernelGGL(k_fm_unique, dim3((nFm + 255) / 256), dim3(256), 0, s, b->fm.p, b->fmKeysB.p, b->fmIdxB.p, nFm,
                               b->fmUniq.p, b->fmN.p, q);
            HIPCHK(hipMemcpyAsync(&nFmUniq, b->fmN.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        lap("fm occurrences");

        // ---- locate + verify; text queue retried on overflow
        // traceback task queue: at most one task per item + the chunk slack of every k_verify wavefront
        // (a chunk of 256 is retired when the wavefront's next group does not fit: at most 63 holes per chunk)
        const size_t tbNeed = (size_t)nItems + (size_t)nItems / 2 + (size_t)(256u * 2048u / 64u + 1) * 256u;
        if (b->tbq.n < tbNeed) b->tbq.alloc(tbNeed + nItems / 8);
        for (int attempt = 0;; attempt++) {
            q.text = b->text.p;
            q.textCap = (uint32_t)std::min<size_t>(b->text.n, 0xFFFFFFF0u);
            uint32_t zero[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(b->cnt.p + 2, zero, sizeof(zero), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(b->cnt.p + 7, zero, sizeof(uint32_t), hipMemcpyHostToDevice, s));
            unsigned long long keep[CMB_CNT_MAX];
            HIPCHK(hipMemcpyAsync(keep, b->counters.p, sizeof(keep), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (nItems) {
                // (lanes of k_verify: 128 k / 256 k / 512 k -> 34.8 / 32.0 / 35.9 ms of the locate group on the headline workload, round 4)
                const uint32_t vCap = getenv("CMB_V_SLOTS") ? (uint32_t)std::min(256 * 2048, std::max(256, atoi(getenv("CMB_V_SLOTS")))) / 256u * 256u : 256u * 1024u;
                const uint32_t vSlots = std::min<uint32_t>(((nItems + 255) / 256) * 256, vCap);
                // Edit distance: identical verifications (same read x strand, text window and bounds — the parts of
                // one read seeding the same alignment) are performed once: k_verify only locates and emits a key per
                // candidate, the keys are sorted and run-length encoded, k_verify_edit verifies the distinct ones and
                // scales the counters by the multiplicities.
                // (25 key bits for read x strand; 21 in the key layout of batches at 8 ... 13 errors, whose sub-batches hold at most 2^20 reads)
                const bool dedup = b->metric == CMB_METRIC_EDIT && b->k > 0 && 2ull * nReads < (b->wideEdit ? (1ull << 21) : (1ull << 25));
                if (b->wideEdit && !dedup) return fail(CMB_ERR_INTERNAL, "a sub-batch beyond 7 errors holds more reads than its verification keys number");
                const uint32_t tbCap = (uint32_t)std::min<size_t>(b->tbq.n, 0xFFFFFFF0u);
                const char* vGroup = "k_verify";
                tm.begin();
                if (dedup && b->vkeysA.n < nItems) {
                    b->vkeysA.alloc((size_t)nItems + nItems / 8 + 256);
                    b->vkeysB.alloc((size_t)nItems + nItems / 8 + 256);
                    b->vcounts.alloc((size_t)nItems + nItems / 8 + 256);
                    b->vruns.alloc(4);
                }
                auto kv = dedup ? k_verify<true> : k_verify<false>;
                hipLaunchKernelGGL(kv, dim3(vSlots / 256), dim3(256), 0, s, ix->d, b->offs.p, b->maxLen, b->gw,
                                   b->seq.p, mf, b->items.p, nItems, b->tbq.p, tbCap,
                                   dedup ? b->vkeysA.p : (unsigned long long*)nullptr, q, b->wideEdit ? 1u : 0u); // (1: the wide key layout)
                if (dedup) {
                    uint32_t nRuns = 0;
                    size_t tmpBytes = 0;
                    // sorted bits: low VK_LOW bits of the start, the bounds, read x strand (2 nReads < 2^rsBits, so
                    // that the all-ones key of the other items sorts behind every real key) — see packVerifyKey
                    uint32_t rsBits = 1;
                    while ((1ull << rsBits) <= 2ull * nReads) rsBits++;
                    const uint32_t bit0 = 32u - VK_LOW, bit1 = (b->wideEdit ? VKW_RS : 39u) + rsBits;
                    HIPCHK(rocprim::radix_sort_keys(nullptr, tmpBytes, b->vkeysA.p, b->vkeysB.p, nItems, bit0, bit1, s));
                    if (b->sortTmp.n < tmpBytes) b->sortTmp.alloc(tmpBytes + 256);
                    HIPCHK(rocprim::radix_sort_keys(b->sortTmp.p, tmpBytes, b->vkeysA.p, b->vkeysB.p, nItems, bit0, bit1, s));
                    size_t rleBytes = 0;
                    HIPCHK(rocprim::run_length_encode(nullptr, rleBytes, b->vkeysB.p, nItems, b->vkeysA.p, b->vcounts.p,
                                                      b->vruns.p, s));
                    if (b->scanTmp.n < rleBytes) b->scanTmp.alloc(rleBytes + 256);
                    HIPCHK(rocprim::run_length_encode(b->scanTmp.p, rleBytes, b->vkeysB.p, nItems, b->vkeysA.p,
                                                      b->vcounts.p, b->vruns.p, s));
                    HIPCHK(hipMemcpyAsync(&nRuns, b->vruns.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIPCHK(hipStreamSynchronize(s));
                    tm.end("k_verify"); // locate + key sort + run-length encode; the matrix stages are timed apart
                    vGroup = "k_verify_edit";
                    tm.begin();
                    if (getenv("CMB_VERBOSE")) fprintf(stderr, "[verify] %u items, %u distinct keys\n", nItems, nRuns);
                    if (nRuns && b->wideEdit) {
                        // 8 ... 13 errors: the band (up to 53 columns) needs the wide left margins of the 64-bit in-text matrix (dev_matrix.hpp:
                        // MXX_*, MXY_*): k_wide_filter sorts out the candidates that never reach their final column, k_verify_wide<true> verifies the rest
                        const uint32_t fGrid = std::min<uint32_t>((nRuns + 255) / 256, 2048u);
                        // (every key, the slots a wavefront leaves unused when it retires a chunk — fewer than 64 of 256 —, a chunk per wavefront)
                        const size_t listNeed = (size_t)nRuns + nRuns / 3 + (size_t)fGrid * 4 * 256 + 512;
                        if (b->dpList.n < listNeed) b->dpList.alloc(listNeed + listNeed / 8);
                        if (b->dpWork.n < 4) b->dpWork.alloc(4);
                        HIPCHK(hipMemsetAsync(b->dpWork.p, 0, 4 * sizeof(uint32_t), s));
                        auto kFilter = k_wide_filter<WxTen>;
                        auto kWide = k_verify_wide<true, WxTen>;
                        if (b->geoX) kFilter = k_wide_filter<WxThirteen>, kWide = k_verify_wide<true, WxThirteen>;
                        hipLaunchKernelGGL(kFilter, dim3(fGrid), dim3(256), 0, s, ix->d, b->offs.p, b->G.p, b->gw, b->vkeysA.p, b->vcounts.p, nRuns,
                                           b->dpWork.p, b->dpList.p, (uint32_t)std::min<size_t>(b->dpList.n, 0xFFFFFFF0u), q);
                        const uint32_t slotBytes = (vwRows(b->maxLen) + 1u) * VW_ROW_BYTES;
                        // (six wavefronts per SIMD: 1024 SIMDs x 6 x 64 lanes — forward pass and traceback wait for memory; a slot is 3 - 8 KB)
                        const uint32_t dSlots = std::min<uint32_t>(((nRuns + 255) / 256) * 256, getenv("CMB_VW_SLOTS") ? (uint32_t)atoi(getenv("CMB_VW_SLOTS")) : 256u * 1536u);
                        if (b->dpSlab.n < (size_t)slotBytes * dSlots) b->dpSlab.alloc((size_t)slotBytes * dSlots);
                        hipLaunchKernelGGL(kWide, dim3(dSlots / 256), dim3(256), 0, s, ix->d, b->offs.p, b->maxLen, b->seq.p, b->G.p, b->gw,
                                           (const uint4*)nullptr, (uint32_t)std::min<size_t>(b->dpList.n, 0xFFFFFFF0u), b->vkeysA.p, b->vcounts.p, b->dpList.p, b->dpWork.p + 1, b->dpSlab.p, slotBytes, q);
                        if (verbose) {
                            uint32_t hw[2];
                            HIPCHK(hipMemcpyAsync(hw, b->dpWork.p, sizeof(hw), hipMemcpyDeviceToHost, s));
                            HIPCHK(hipStreamSynchronize(s));
                            fprintf(stderr, "[verify] %u list slots (survivors and holes) for %u distinct candidates\n", hw[1], nRuns);
                        }
                    } else if (nRuns) {
                        // staged verification (kernels.hpp: k_verify_stage): one launch per nb 32-row matrix blocks,
                        // survivor lists ping-pong, list sizes stay on the device
                        // nb 32-row blocks per stage: fewer stages re-fetch fewer text lines and move fewer survivor
                        // records, more blocks leave more lanes idle behind candidates that ended (CMB_STAGE_BLOCKS)
                        const char* nbEnv = getenv("CMB_STAGE_BLOCKS");
                        // (round 4, 150 bp reads at 4 errors: 1 / 2 / 3 / 4 / 6 blocks per stage -> 53.1 / 31.4 / 29.4 / 32.0 / 38.0 ms of k_verify_edit)
                        const uint32_t nb = nbEnv ? std::min(8u, std::max(1u, (uint32_t)atoi(nbEnv))) : 3u;
                        const uint32_t nStages = (vRows(b->maxLen) + 32u * nb - 1u) / (32u * nb) + 1u;
                        for (int j = 0; j < 2; j++)
                            if (b->vsC[j].n < nRuns) {
                                b->vsA[j].alloc((size_t)nRuns + nRuns / 8 + 256);
                                b->vsB[j].alloc(b->vsA[j].n);
                                b->vsC[j].alloc(b->vsA[j].n);
                            }
                        if (b->vsN.n < nStages + 2) b->vsN.alloc(nStages + 2);
                        HIPCHK(hipMemsetAsync(b->vsN.p, 0, (nStages + 2) * sizeof(uint32_t), s));
                        const uint32_t listCap = (uint32_t)std::min<size_t>(b->vsC[0].n, 0xFFFFFFF0u);
                        const uint32_t gridCap = getenv("CMB_STAGE_GRID") ? (uint32_t)std::max(256, atoi(getenv("CMB_STAGE_GRID"))) : 8192u;
                        const uint32_t grid = std::min<uint32_t>((nRuns + 255) / 256, gridCap);
                        VStageList L0{b->vsA[0].p, b->vsB[0].p, b->vsC[0].p}, L1{b->vsA[1].p, b->vsB[1].p, b->vsC[1].p};
                        // k <= 4: the matrix on 32-bit words (dev_matrix.hpp); CMB_MATRIX_WIDE=1 keeps the 64-bit words
                        const bool w32 = b->k <= MX32_MAX_ED && !getenv("CMB_MATRIX_WIDE");
                        const bool packed = ix->d.text2 != nullptr; // 2-bit text (cmb_index_create)
                        // stages no read of the batch can reach its final-column rows in (row >= len - maxED - 1,
                        // maxED <= 7) run the instance without final-column code
                        auto needsFinal = [&](uint32_t st) { return 32u * nb * (st + 1u) - 1u + 8u >= b->minLen; };
                        auto stageKernel = [&](bool first, bool fin) {
                            const int sel = (first ? 8 : 0) | (w32 ? 4 : 0) | (packed ? 2 : 0) | (fin ? 1 : 0);
                            switch (sel) {
                            case 0: return k_verify_stage<false, false, false, false>;
                            case 1: return k_verify_stage<false, false, false, true>;
                            case 2: return k_verify_stage<false, false, true, false>;
                            case 3: return k_verify_stage<false, false, true, true>;
                            case 4: return k_verify_stage<false, true, false, false>;
                            case 5: return k_verify_stage<false, true, false, true>;
                            case 6: return k_verify_stage<false, true, true, false>;
                            case 7: return k_verify_stage<false, true, true, true>;
                            case 8: return k_verify_stage<true, false, false, false>;
                            case 9: return k_verify_stage<true, false, false, true>;
                            case 10: return k_verify_stage<true, false, true, false>;
                            case 11: return k_verify_stage<true, false, true, true>;
                            case 12: return k_verify_stage<true, true, false, false>;
                            case 13: return k_verify_stage<true, true, false, true>;
                            case 14: return k_verify_stage<true, true, true, false>;
                            default: return k_verify_stage<true, true, true, true>;
                            }
                        };
                        hipLaunchKernelGGL(stageKernel(true, needsFinal(0)), dim3(grid), dim3(256), 0, s, ix->d, b->offs.p, mf,
                                           b->vkeysA.p, b->vcounts.p, nRuns, L0, L1, b->vsN.p, listCap, 0u, nb, b->tbq.p, tbCap, q);
                        for (uint32_t st = 1; st < nStages; st++)
                            hipLaunchKernelGGL(stageKernel(false, needsFinal(st)), dim3(grid), dim3(256), 0, s, ix->d, b->offs.p, mf,
                                               (const unsigned long long*)nullptr, (const uint32_t*)nullptr, 0u,
                                               (st & 1u) ? L1 : L0, (st & 1u) ? L0 : L1, b->vsN.p, listCap, st, nb, b->tbq.p, tbCap,
                                               q);
                        if (verbose) {
                            std::vector<uint32_t> hn(nStages + 2);
                            HIPCHK(hipMemcpyAsync(hn.data(), b->vsN.p, hn.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                            HIPCHK(hipStreamSynchronize(s));
                            fprintf(stderr, "[verify] survivors per stage:");
                            for (uint32_t st = 1; st <= nStages; st++) fprintf(stderr, " %u", hn[st]);
                            fprintf(stderr, "\n");
                        }
                    }
                }
                tm.end(vGroup);
                HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                const uint32_t nTb = hcnt[7];
                if (nTb) {
                    const uint32_t slotCap = getenv("CMB_TB_SLOTS") ? (uint32_t)std::max(256, atoi(getenv("CMB_TB_SLOTS"))) / 256u * 256u : 512u * 1024u; // (measured 64 k … 2 M slots: 84 / 49 / 34.4 / 32.8 / 31 / 29 ms alone; 512 k best beside other sub-batches)
                    const uint32_t tSlots = std::min<uint32_t>(((nTb + 255) / 256) * 256, slotCap);
                    // 64-byte lines of 16 narrow (k <= 4) or 8 wide trace rows (a group is written whole)
                    const bool narrow = b->k <= TBN_MAX_ED && !getenv("CMB_TRACE_WIDE");
                    const uint32_t tLines = narrow ? (vRows(b->maxLen) + 15u) / 16u + 2u : (vRows(b->maxLen) + 7u) / 8u + 2u;
                    if (b->vW.n < (size_t)tLines * 8 * tSlots) b->vW.alloc((size_t)tLines * 8 * tSlots);
                    VPlanes vp{b->vW.p, tSlots, tLines};
                    tm.begin();
                    auto kTrace = k_traceback<false, false>;
                    if (narrow) kTrace = ix->d.text2 ? k_traceback<true, true> : k_traceback<true, false>;
                    // no final-column row (> len - maxED - 1, maxED <= 7) at or before this row, for any read of the batch
                    const uint32_t rowMin = b->minLen > 8u ? b->minLen - 8u : 0u;
                    hipLaunchKernelGGL(kTrace, dim3(tSlots / 256), dim3(256), 0, s, ix->d, b->offs.p, mf, b->tbq.p, nTb,
                                       vp, q, rowMin);
                    tm.end("k_traceback");
                }
            }
            HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (nFmUniq) {
                tm.begin();
                const uint32_t nb = (uint32_t)std::min<size_t>(((size_t)nFmUniq + 255) / 256, 4096);
                hipLaunchKernelGGL(k_fmocc, dim3(nb), dim3(256), 0, s, ix->d, b->fmUniq.p, nFmUniq, q);
                tm.end("k_fmocc");
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (hcnt[3] & FLAG_TEXT_OVERFLOW) {
                if (attempt >= 3) return fail(CMB_ERR_INTERNAL, "text occurrence queue keeps overflowing");
                b->text.alloc((size_t)hcnt[2] + hcnt[2] / 8 + 1024);
                HIPCHK(hipMemcpy(b->counters.p, keep, sizeof(keep), hipMemcpyHostToDevice));
                uint32_t z = 0; // clear the overflow flag for the retry
                uint32_t fl = hcnt[3] & ~(uint32_t)FLAG_TEXT_OVERFLOW;
                (void)z;
                HIPCHK(hipMemcpy(b->cnt.p + 3, &fl, sizeof(uint32_t), hipMemcpyHostToDevice));
                continue;
            }
            break;
        }
        uint32_t nText = hcnt[2];
        lap("verify + traceback + fmocc");
        if (preset) { // the raw text occurrences and the counters are the result
            if (hcnt[3] & (FLAG_CAPACITY | FLAG_TRACE_RULE)) return fail(CMB_ERR_INTERNAL, "a traceback left the band");
            b->presetOut.resize(nText);
            if (nText) HIPCHK(hipMemcpy(b->presetOut.data(), b->text.p, (size_t)nText * sizeof(TextOccRec), hipMemcpyDeviceToHost));
            unsigned long long hc2[CMB_CNT_MAX];
            HIPCHK(hipMemcpy(hc2, b->counters.p, sizeof(hc2), hipMemcpyDeviceToHost));
            for (int i = 0; i < CMB_CNT_MAX; i++) b->cnts[i] = hc2[i];
            b->done = true;
            return CMB_OK;
        }

        const uint32_t keyLayout = b->metric == CMB_METRIC_HAMMING ? 1u : b->wideEdit ? 2u : 0u; // (kernels.hpp: keyBits)
        const uint32_t groupShift = keyBits(keyLayout).group;
        // ---- naive backtracking: the filter pass of approxMatchesNaive[Hamming] itself, per read x strand (dev_bfs_naive.hpp)
        uint64_t naiveSurvivors = 0;
        if (b->hasNaive && nText) {
            tm.begin();
            const uint32_t nG = 2u * nReads;
            if (nG >= (1u << (64u - groupShift))) return fail(CMB_ERR_UNSUPPORTED, "more than 2^23 reads in a sub-batch with reads matched by naive backtracking");
            if (b->keysA.n < nText) {
                b->keysA.alloc((size_t)nText + nText / 8 + 256);
                b->keysB.alloc((size_t)nText + nText / 8 + 256);
            }
            if (b->fcounts.n < (size_t)nG + 1) {
                b->fcounts.alloc((size_t)nG + 1);
                b->foffs.alloc((size_t)nG + 1);
                b->fsegB.alloc((size_t)nG + 1);
                b->fsegE.alloc((size_t)nG + 1);
            }
            if (b->frank.n < nText) b->frank.alloc((size_t)nText + nText / 8 + 256);
            hipLaunchKernelGGL(k_pack_keys, dim3((nText + 255) / 256), dim3(256), 0, s, b->text.p, nText, b->offs.p, b->k, b->keysA.p,
                               b->cnt.p, 1u, (const uint8_t*)b->psel.p, keyLayout);
            size_t tmpBytes = 0;
            HIPCHK(rocprim::radix_sort_keys(nullptr, tmpBytes, b->keysA.p, b->keysB.p, nText, 0, 64, s));
            if (b->sortTmp.n < tmpBytes) b->sortTmp.alloc(tmpBytes + 256);
            HIPCHK(rocprim::radix_sort_keys(b->sortTmp.p, tmpBytes, b->keysA.p, b->keysB.p, nText, 0, 64, s));
            const int mode = b->metric == CMB_METRIC_HAMMING ? 1 : 2;
            HIPCHK(hipMemsetAsync(b->fcounts.p, 0, ((size_t)nG + 1) * sizeof(uint32_t), s));
            HIPCHK(hipMemsetAsync(b->fsegB.p, 0xFF, ((size_t)nG + 1) * sizeof(uint32_t), s));
            HIPCHK(hipMemsetAsync(b->frank.p, 0xFF, (size_t)nText * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_filter_segments, dim3((nText + 255) / 256), dim3(256), 0, s, b->keysB.p, nText, b->fsegB.p, b->fsegE.p, groupShift);
            hipLaunchKernelGGL(k_filter_mark, dim3((nG + 255) / 256), dim3(256), 0, s, b->keysB.p, nG, b->k, mode, b->fcounts.p,
                               b->frank.p, b->fsegB.p, b->fsegE.p, keyLayout);
            size_t scanBytes = 0;
            HIPCHK(rocprim::exclusive_scan(nullptr, scanBytes, b->fcounts.p, b->foffs.p, (uint64_t)0, (size_t)nG + 1,
                                           rocprim::plus<uint64_t>(), s));
            if (b->scanTmp.n < scanBytes) b->scanTmp.alloc(scanBytes + 256);
            HIPCHK(rocprim::exclusive_scan(b->scanTmp.p, scanBytes, b->fcounts.p, b->foffs.p, (uint64_t)0, (size_t)nG + 1,
                                           rocprim::plus<uint64_t>(), s));
            HIPCHK(hipMemcpyAsync(&naiveSurvivors, b->foffs.p + nG, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (hcnt[3] & FLAG_CAPACITY)
                return fail(CMB_ERR_INTERNAL, "occurrence does not fit the filter key (width / distance range) or a traceback left the band");
            if ((uint64_t)nText + naiveSurvivors >= 0xFFFFFFF0ull) return fail(CMB_ERR_UNSUPPORTED, "too many occurrences in one sub-batch");
            if (b->text.n < (size_t)nText + naiveSurvivors) { // grow, keeping the records
                DevBuf<TextOccRec> bigger;
                bigger.alloc((size_t)nText + naiveSurvivors + 1024);
                HIPCHK(hipMemcpyAsync(bigger.p, b->text.p, (size_t)nText * sizeof(TextOccRec), hipMemcpyDeviceToDevice, s));
                HIPCHK(hipStreamSynchronize(s));
                std::swap(bigger.p, b->text.p);
                std::swap(bigger.n, b->text.n);
            }
            hipLaunchKernelGGL(k_naive_drop, dim3((nText + 255) / 256), dim3(256), 0, s, b->text.p, nText, (const uint8_t*)b->psel.p);
            if (naiveSurvivors)
                hipLaunchKernelGGL(k_naive_keep, dim3((nText + 255) / 256), dim3(256), 0, s, b->keysB.p, nText, b->offs.p, b->k,
                                   b->frank.p, b->foffs.p, b->text.p + nText, keyLayout);
            HIPCHK(hipGetLastError());
            nText += (uint32_t)naiveSurvivors;
            tm.end("k_naive_filter");
        }

        // ---- sort + filter on the device (getUniqueTextOccurrences / getTextOccHamming,
        // indexinterface.cpp:1331-1491): pack -> one 64-bit radix sort -> per-read scan
        unsigned long long hc[CMB_CNT_MAX];
        HIPCHK(hipMemcpy(hc, b->counters.p, sizeof(hc), hipMemcpyDeviceToHost));
        for (int i = 0; i < CMB_CNT_MAX; i++) b->cnts[i] = hc[i];
        b->cnts[1] += naiveSurvivors; // reported once more, as text occurrences of the read (indexinterface.cpp:1333, :1378)
        // groups of the filter: reads, or read x strand when every strand is filtered by itself
        const uint32_t nGroups = b->perStrand ? 2u * nReads : nReads;
        if (nGroups >= (1u << (64u - groupShift))) return fail(CMB_ERR_UNSUPPORTED, "more filter groups in one sub-batch than the keys number");
        {
            tm.begin();
            if (b->keysA.n < nText) {
                b->keysA.alloc((size_t)nText + nText / 8 + 256);
                b->keysB.alloc((size_t)nText + nText / 8 + 256);
            }
            if (b->fcounts.n < (size_t)nGroups + 1) {
                b->fcounts.alloc((size_t)nGroups + 1);
                b->foffs.alloc((size_t)nGroups + 1);
                b->fsegB.alloc((size_t)nGroups + 1);
                b->fsegE.alloc((size_t)nGroups + 1);
            }
            if (nText) {
                hipLaunchKernelGGL(k_pack_keys, dim3((nText + 255) / 256), dim3(256), 0, s, b->text.p, nText, b->offs.p,
                                   b->k, b->keysA.p, b->cnt.p, b->perStrand ? 1u : 0u, (const uint8_t*)nullptr,
                                   keyLayout);
                size_t tmpBytes = 0;
                HIPCHK(rocprim::radix_sort_keys(nullptr, tmpBytes, b->keysA.p, b->keysB.p, nText, 0, 64, s));
                if (b->sortTmp.n < tmpBytes) b->sortTmp.alloc(tmpBytes + 256);
                HIPCHK(rocprim::radix_sort_keys(b->sortTmp.p, tmpBytes, b->keysA.p, b->keysB.p, nText, 0, 64, s));
            }
            const int mode = b->k == 0 ? 0 : (b->metric == CMB_METRIC_HAMMING ? 1 : 2);
            HIPCHK(hipMemsetAsync(b->fcounts.p, 0, ((size_t)nGroups + 1) * sizeof(uint32_t), s));
            HIPCHK(hipMemsetAsync(b->fsegB.p, 0xFF, ((size_t)nGroups + 1) * sizeof(uint32_t), s));
            if (nText)
                hipLaunchKernelGGL(k_filter_segments, dim3((nText + 255) / 256), dim3(256), 0, s, b->keysB.p, nText, b->fsegB.p,
                                   b->fsegE.p, groupShift);
            if (b->frank.n < nText) b->frank.alloc((size_t)nText + nText / 8 + 256);
            if (nText) HIPCHK(hipMemsetAsync(b->frank.p, 0xFF, (size_t)nText * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_filter_mark, dim3((nGroups + 255) / 256), dim3(256), 0, s, b->keysB.p, nGroups, b->k, mode,
                               b->fcounts.p, b->frank.p, b->fsegB.p, b->fsegE.p, keyLayout);
            size_t scanBytes = 0;
            HIPCHK(rocprim::exclusive_scan(nullptr, scanBytes, b->fcounts.p, b->foffs.p, (uint64_t)0, (size_t)nGroups + 1,
                                           rocprim::plus<uint64_t>(), s));
            if (b->scanTmp.n < scanBytes) b->scanTmp.alloc(scanBytes + 256);
            HIPCHK(rocprim::exclusive_scan(b->scanTmp.p, scanBytes, b->fcounts.p, b->foffs.p, (uint64_t)0, (size_t)nGroups + 1,
                                           rocprim::plus<uint64_t>(), s));
            uint64_t total = 0;
            HIPCHK(hipMemcpyAsync(&total, b->foffs.p + nGroups, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (hcnt[3] & FLAG_TRACE_RULE)
                return fail(CMB_ERR_INTERNAL, "a traceback read a band-edge bit the narrow trace rows take as implied");
            if (hcnt[3] & FLAG_CAPACITY)
                return fail(CMB_ERR_INTERNAL, "occurrence does not fit the filter key (width / distance range) or a traceback left the band");
            if (b->fout.n < total) b->fout.alloc((size_t)total + total / 8 + 256);
            if (b->wantAln && b->foutRead.n < total) b->foutRead.alloc((size_t)total + total / 8 + 256);
            if (total)
                hipLaunchKernelGGL(k_filter_write, dim3((nText + 255) / 256), dim3(256), 0, s, b->keysB.p, nText, b->offs.p,
                                   b->k, b->frank.p, b->foffs.p, b->fout.p, b->wantAln ? b->foutRead.p : (uint32_t*)nullptr,
                                   b->perStrand ? 1u : 0u, keyLayout);
            HIPCHK(hipGetLastError());
            tm.end("k_filter");
            lap("filter");
            if (b->wantAln) { // CIGAR + sequence of every final occurrence (k_cigar)
                tm.begin();
                b->alnStride = 2u * b->k + 3u;
                if (b->alnRec.n < total) {
                    b->alnRec.alloc((size_t)total + total / 8 + 256);
                    b->alnOps.alloc(b->alnRec.n * std::max<size_t>(b->alnStride, 2u * 7u + 3u));
                }
                if (total) {
                    const uint32_t cSlots = (uint32_t)std::min<uint64_t>(((total + 255) / 256) * 256, 512u * 1024u);
                    const bool narrow = b->k <= TBN_MAX_ED && !getenv("CMB_TRACE_WIDE");
                    const uint32_t tLines = narrow ? (vRows(b->maxLen) + 15u) / 16u + 2u : (vRows(b->maxLen) + 7u) / 8u + 2u;
                    if (b->vW.n < (size_t)tLines * 8 * cSlots) b->vW.alloc((size_t)tLines * 8 * cSlots);
                    VPlanes vp{b->vW.p, cSlots, tLines};
                    MFull mfc = mf;
                    if (!mfc.p) { // (Hamming / exact batches have no match words: every CIGAR is len x M, nothing is traced)
                        mfc.nBlk = 0;
                    }
                    auto kc = k_cigar<false, false>;
                    if (narrow) kc = ix->d.text2 ? k_cigar<true, true> : k_cigar<true, false>;
                    else if (ix->d.text2) kc = k_cigar<false, true>;
                    if (b->metric == CMB_METRIC_EDIT && b->k > CIGAR_BLOCK_WORDS_MAX_ED) {
                        // (k_cigar's match words reach 9 columns right of the diagonal: kernels.hpp, k_cigar_wide)
                        const uint32_t wSlots = std::min<uint32_t>(cSlots, 256u * 256u);
                        const uint32_t slotBytes = (vwRows(b->maxLen) + 1u) * VW_ROW_BYTES;
                        if (b->dpSlab.n < (size_t)slotBytes * wSlots) b->dpSlab.alloc((size_t)slotBytes * wSlots);
                        hipLaunchKernelGGL(k_cigar_wide, dim3(wSlots / 256), dim3(256), 0, s, ix->d, b->offs.p, b->G.p, b->gw, b->fout.p, b->foutRead.p,
                                           (uint64_t)total, b->dpSlab.p, slotBytes, ix->seqStartsDev.p, ix->nSeqsDev, b->alnOps.p, b->alnStride,
                                           b->alnRec.p, b->cnt.p + 3, 0u);
                    } else
                    hipLaunchKernelGGL(kc, dim3(cSlots / 256), dim3(256), 0, s, ix->d, b->offs.p, mfc, b->fout.p, b->foutRead.p,
                                       (uint64_t)total, vp, ix->seqStartsDev.p, ix->nSeqsDev, b->alnOps.p, b->alnStride, b->alnRec.p,
                                       b->cnt.p + 3, (b->metric != CMB_METRIC_EDIT || b->k == 0) ? 1u : 0u);
                }
                tm.end("k_cigar");
                HIPCHK(hipGetLastError());
            }
            b->occs.resize(total);
            b->occOffs.resize((size_t)nGroups + 1);
            if (total) HIPCHK(hipMemcpyAsync(b->occs.data(), b->fout.p, (size_t)total * sizeof(cmb_occ), hipMemcpyDeviceToHost, s));
            if (b->wantAln) {
                b->hAlnRec.resize(total);
                b->hAlnOps.resize((size_t)total * b->alnStride);
                if (total) {
                    HIPCHK(hipMemcpyAsync(b->hAlnRec.data(), b->alnRec.p, (size_t)total * sizeof(AlnRec), hipMemcpyDeviceToHost, s));
                    HIPCHK(hipMemcpyAsync(b->hAlnOps.data(), b->alnOps.p, (size_t)total * b->alnStride * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
                }
                HIPCHK(hipMemcpyAsync(hcnt, b->cnt.p, sizeof(hcnt), hipMemcpyDeviceToHost, s));
            }
            HIPCHK(hipMemcpyAsync(b->occOffs.data(), b->foffs.p, ((size_t)nGroups + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (b->wantAln && (hcnt[3] & FLAG_CAPACITY))
                return fail(CMB_ERR_INTERNAL, "a CIGAR traceback left the band or an occurrence is not an alignment within its distance");
            lap("results to the host");
            if (verbose) {
                size_t fr = 0, tot = 0;
                (void)hipMemGetInfo(&fr, &tot);
                fprintf(stderr, "[mem] device memory in use: %.1f GB of %.1f GB\n", (tot - fr) / 1e9, tot / 1e9);
            }
        }
        // TOTAL_REPORTED_POSITIONS (indexinterface.cpp:1378,1390 / :1333,1352)
        // (the device counter holds the records k_verify / k_traceback wrote; queue holes are not records)
        b->done = true;
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// ---- what the b-move translation unit borrows from this one (move_backend.hip: cmb_move_attach_text, alignments) -------------
// The b-move index holds no text (the reference builds the CIGAR of that flavour from a matched string that travels with the search,
// indexinterface.h:294-303).  With 288 GB of HBM a 2-bit copy of the text fits beside the index (a quarter byte per character: 50 GB for
// 64 human haplotypes next to ~130 GB of tables), and the matched string of an occurrence IS text[begin, end): findCIGAR on that
// window (k_cigar) gives the reference's CIGAR without carrying anything through the frontier.
namespace cmb {
const std::vector<uint32_t>& seqStartsOfIndex(const cmb_index* textIndex) { return textIndex->seqStarts; }
// CIGAR and sequence assignment of nOcc occurrences {begin, end, distance, strand} of the reads `occRead` on a text: the k_cigar launch
// of batchRunOne for a caller that has the pieces (device pointers throughout; aln: nOcc x 16 bytes {seqId, seqBegin, nOps, spans};
// ops: nOcc x stride, stored end to begin as for cmb_batch_alignments).
int moveCigarsOnText(cmb_index* textIndex, hipStream_t s, const uint64_t* offs, const uint32_t* G, uint32_t gw, uint32_t nReads, uint32_t maxLen,
                     uint32_t k, int gapless, const void* occs, const uint32_t* occRead, uint64_t nOcc, void* aln, uint16_t* ops, uint32_t stride,
                     uint32_t* flagWord) {
    try {
        HIPCHK(hipSetDevice(textIndex->device));
        if (!nOcc) return CMB_OK;
        const DevIndex& d = textIndex->d;
        const uint32_t* text2 = d.text2;
        const uint32_t* seqStartsDev = textIndex->seqStartsDev.p;
        const uint32_t nSeqs = textIndex->nSeqsDev;
        DevBuf<uint4> mfull;
        MFull mf{nullptr, 0};
        if (!gapless) {
            mf.nBlk = mfullBlocks(maxLen);
            const uint64_t nW = (uint64_t)2 * nReads * mf.nBlk;
            mfull.alloc(2 * nW);
            hipLaunchKernelGGL(k_match_words, dim3((unsigned)((nW + 255) / 256)), dim3(256), 0, s, G, gw, offs, 2 * nReads, mf.nBlk, mfull.p);
            mf.p = mfull.p;
        }
        const uint32_t cSlots = (uint32_t)std::min<uint64_t>(((nOcc + 255) / 256) * 256, 512u * 1024u);
        if (!gapless && k > CIGAR_BLOCK_WORDS_MAX_ED) {
            // beyond 9 errors (k_cigar's match words reach 9 columns right of the diagonal): the matrix with the wide left margin, as for
            // the FM-index batches (kernels.hpp: k_cigar_wide; match words straight from the read's bit-strings)
            const uint32_t wSlots = std::min<uint32_t>(cSlots, 256u * 256u);
            const uint32_t slotBytes = (vwRows(maxLen) + 1u) * VW_ROW_BYTES;
            DevBuf<uint8_t> slab;
            slab.alloc((size_t)slotBytes * wSlots);
            hipLaunchKernelGGL(k_cigar_wide, dim3(wSlots / 256), dim3(256), 0, s, d, offs, G, gw, (const uint4*)occs, occRead, nOcc, slab.p, slotBytes,
                               seqStartsDev, nSeqs, ops, stride, (AlnRec*)aln, flagWord, 0u);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s)); // (the slab goes out of scope)
            return CMB_OK;
        }
        const bool narrow = k <= TBN_MAX_ED;
        const uint32_t tLines = narrow ? (vRows(maxLen) + 15u) / 16u + 2u : (vRows(maxLen) + 7u) / 8u + 2u;
        DevBuf<uint64_t> vW;
        vW.alloc((size_t)tLines * 8 * cSlots);
        VPlanes vp{vW.p, cSlots, tLines};
        auto kc = k_cigar<false, false>;
        if (narrow) kc = text2 ? k_cigar<true, true> : k_cigar<true, false>;
        else if (text2) kc = k_cigar<false, true>;
        hipLaunchKernelGGL(kc, dim3(cSlots / 256), dim3(256), 0, s, d, offs, mf, (const uint4*)occs, occRead, nOcc, vp, seqStartsDev, nSeqs, ops,
                           stride, (AlnRec*)aln, flagWord, gapless ? 1u : 0u);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s)); // (the temporaries above go out of scope)
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}
} // namespace cmb

extern "C" int cmb_batch_result_size(const cmb_batch* b, uint64_t* n_occ) {
    if (!b || !n_occ) return fail(CMB_ERR_INVALID, "null argument");
    if (!b->done) return fail(CMB_ERR_INVALID, "batch has not been run");
    uint64_t n = b->occs.size();
    for (const cmb_batch* c : b->subs) n += c->occs.size();
    *n_occ = n;
    return CMB_OK;
}
extern "C" int cmb_batch_results(const cmb_batch* b, cmb_occ* out, uint64_t out_cap, uint64_t* out_offs,
                                 uint64_t* counters) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    if (!b->done) return fail(CMB_ERR_INVALID, "batch has not been run");
    if (!b->subs.empty()) { // composite: the sub-batches' lists one after the other, offsets rebased
        uint64_t total = 0;
        for (const cmb_batch* c : b->subs) total += c->occs.size();
        if (out_cap < total) return fail(CMB_ERR_OVERFLOW, "output buffer too small");
        uint64_t base = 0, read = 0;
        for (const cmb_batch* c : b->subs) {
            if (out && !c->occs.empty()) memcpy(out + base, c->occs.data(), c->occs.size() * sizeof(cmb_occ));
            if (out_offs)
                for (uint32_t i = 0; i < c->nReads; i++) out_offs[read + i] = base + c->occOffs.data()[c->perStrand ? 2 * (size_t)i : i];
            read += c->nReads;
            base += c->occs.size();
        }
        if (out_offs) out_offs[read] = base;
        if (counters) memcpy(counters, b->cnts, sizeof(b->cnts));
        return CMB_OK;
    }
    if (out_cap < b->occs.size()) return fail(CMB_ERR_OVERFLOW, "output buffer too small");
    if (out && !b->occs.empty()) memcpy(out, b->occs.data(), b->occs.size() * sizeof(cmb_occ));
    if (out_offs) {
        if (b->perStrand)
            for (uint32_t i = 0; i <= b->nReads; i++) out_offs[i] = b->occOffs.data()[2 * (size_t)i];
        else memcpy(out_offs, b->occOffs.data(), b->occOffs.size() * sizeof(uint64_t));
    }
    if (counters) memcpy(counters, b->cnts, sizeof(b->cnts));
    return CMB_OK;
}
extern "C" int cmb_batch_filter_per_strand(cmb_batch* b, int on) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    b->perStrand = on != 0;
    for (cmb_batch* c : b->subs) c->perStrand = on != 0;
    b->done = false;
    return CMB_OK;
}
extern "C" int cmb_batch_want_alignments(cmb_batch* b, int on) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    b->wantAln = on != 0;
    for (cmb_batch* c : b->subs) c->wantAln = on != 0;
    b->done = false;
    return CMB_OK;
}
extern "C" int cmb_batch_alignments(const cmb_batch* b, cmb_aln* out, uint64_t cap, uint16_t* cigar_ops, uint64_t ops_cap,
                                    uint64_t* n_ops) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    if (!b->done) return fail(CMB_ERR_INVALID, "batch has not been run");
    if (!b->wantAln) return fail(CMB_ERR_INVALID, "alignments were not requested (cmb_batch_want_alignments)");
    std::vector<const cmb_batch*> parts;
    if (b->subs.empty()) parts.push_back(b);
    else
        for (const cmb_batch* c : b->subs) parts.push_back(c);
    uint64_t total = 0, ops = 0;
    for (const cmb_batch* c : parts) {
        total += c->hAlnRec.size();
        for (size_t i = 0; i < c->hAlnRec.size(); i++) ops += c->hAlnRec.data()[i].nOps;
    }
    if (n_ops) *n_ops = ops;
    if (cap < total || ops_cap < ops) return fail(CMB_ERR_OVERFLOW, "output buffer too small");
    uint64_t o = 0, po = 0;
    for (const cmb_batch* c : parts)
        for (size_t i = 0; i < c->hAlnRec.size(); i++) {
            const AlnRec& r = c->hAlnRec.data()[i];
            out[o] = cmb_aln{r.seqId, r.seqBegin, po, (uint16_t)r.nOps, (uint16_t)r.spans};
            const uint16_t* src = c->hAlnOps.data() + i * c->alnStride;
            for (uint32_t j = 0; j < r.nOps; j++) cigar_ops[po + j] = src[r.nOps - 1 - j]; // (stored end to begin)
            po += r.nOps;
            o++;
        }
    return CMB_OK;
}
extern "C" int cmb_batch_timings(const cmb_batch* b, const char** names, float* ms, uint32_t cap) {
    if (!b) return 0;
    uint32_t n = 0;
    for (const auto& t : b->times) {
        if (n >= cap) break;
        if (names) names[n] = t.name;
        if (ms) ms[n] = t.ms;
        n++;
    }
    return (int)n;
}
// Reads not longer than the number of parts of the search scheme (and every read of a one-part strategy) are matched by naive
// backtracking, as in the reference (searchstrategy.cpp:148-152, :442-459; indexinterface.cpp:1055-1210 -> dev_bfs_naive.hpp).
// cmb_batch_read_status says which reads took that path; cmb_batch_allow_unsupported is kept for older callers and has no effect.
extern "C" int cmb_batch_allow_unsupported(cmb_batch* b, int on) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    b->allowUnsupported = on != 0;
    for (cmb_batch* c : b->subs) c->allowUnsupported = on != 0;
    return CMB_OK;
}
static uint32_t readStatusOne(const cmb_batch* b, uint8_t* status) {
    uint32_t n = 0;
    const uint32_t P = b->k ? b->sNumParts : 0;
    for (uint32_t i = 0; i < b->nReads; i++) {
        const uint64_t len = b->hostOffs[i + 1] - b->hostOffs[i];
        const bool naive = b->k > 0 && (P >= len || P == 1);
        if (status) status[i] = naive ? CMB_READ_NAIVE_FALLBACK : 0;
        n += naive;
    }
    return n;
}
extern "C" int cmb_batch_read_status(const cmb_batch* b, uint8_t* status, uint32_t* n_flagged) {
    if (!b) return fail(CMB_ERR_INVALID, "null argument");
    uint32_t n = 0;
    if (b->subs.empty()) n = readStatusOne(b, status);
    else
        for (size_t j = 0; j < b->subs.size(); j++) n += readStatusOne(b->subs[j], status ? status + b->subBound[j] : nullptr);
    if (n_flagged) *n_flagged = n;
    return CMB_OK;
}

extern "C" void cmb_batch_destroy(cmb_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->ix->device);
    delete b;
}

extern "C" int cmb_match_batch(cmb_index* idx, const cmb_strategy* st, uint32_t max_distance, const char* seqs,
                               const uint64_t* offs, uint32_t n_reads, cmb_occ* out, uint64_t out_cap,
                               uint64_t* out_offs, uint64_t* counters, uint64_t* needed) {
    cmb_batch* b = nullptr;
    int rc = cmb_batch_create(idx, st, max_distance, seqs, offs, n_reads, &b);
    if (rc) return rc;
    rc = cmb_batch_run(b);
    if (rc == CMB_OK) {
        uint64_t n = 0;
        cmb_batch_result_size(b, &n);
        if (needed) *needed = n;
        rc = cmb_batch_results(b, out, out_cap, out_offs, counters);
    }
    cmb_batch_destroy(b);
    return rc;
}

// ------------------------------------------------------------------------- output records (host-only)
static int64_t putString(const std::string& s, char* out, uint64_t cap) {
    if (out && cap > s.size()) memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}
static SamHit toHit(const cmb_sam_hit& h) {
    SamHit r;
    r.seqName = h.seq_name ? h.seq_name : "*";
    r.cigar = cigarString(h.cigar_ops, h.n_ops);
    r.pos0 = h.pos0;
    r.distance = h.distance;
    r.revCompl = h.revcomp != 0;
    return r;
}
extern "C" int64_t cmb_sam_se(const char* read_id, const cmb_sam_hit* hit, int primary, uint32_t n_hits, uint32_t min_score,
                              const char* print_seq, const char* print_qual, char* out, uint64_t cap) {
    if (!read_id || !hit || !print_seq || !print_qual) return fail(CMB_ERR_INVALID, "null argument");
    return putString(samLineSE(read_id, toHit(*hit), primary != 0, n_hits, min_score, print_seq, print_qual), out, cap);
}
extern "C" int64_t cmb_sam_se_xa(const char* read_id, const cmb_sam_hit* hits, uint32_t n, uint32_t n_hits, const char* print_seq,
                                 const char* print_qual, char* out, uint64_t cap) {
    if (!read_id || !hits || n == 0 || !print_seq || !print_qual) return fail(CMB_ERR_INVALID, "null argument");
    std::vector<SamHit> v;
    for (uint32_t i = 0; i < n; i++) v.push_back(toHit(hits[i]));
    return putString(samLineSEWithXA(read_id, v, n_hits, print_seq, print_qual), out, cap);
}
extern "C" int64_t cmb_sam_unmapped_se(const char* read_id, const char* seq, const char* qual, char* out, uint64_t cap) {
    if (!read_id || !seq || !qual) return fail(CMB_ERR_INVALID, "null argument");
    return putString(samLineUnmappedSE(read_id, seq, qual), out, cap);
}
extern "C" int cmb_read_prepare(const char* id, const char* seq, const char* qual, char* id_out, char* seq_out, char* revcomp_out,
                                char* revqual_out) {
    if (!id || !seq) return fail(CMB_ERR_INVALID, "null argument");
    const std::string cid = cleanSeqID(id), cs = cleanReadSeq(seq), rc = revComplWithN(cs);
    std::string rq = qual ? qual : "";
    std::reverse(rq.begin(), rq.end());
    if (id_out) memcpy(id_out, cid.c_str(), cid.size() + 1);
    if (seq_out) memcpy(seq_out, cs.c_str(), cs.size() + 1);
    if (revcomp_out) memcpy(revcomp_out, rc.c_str(), rc.size() + 1);
    if (revqual_out) memcpy(revqual_out, rq.c_str(), rq.size() + 1);
    return CMB_OK;
}

// ------------------------------------------------------------------------- fine-grained hooks
extern "C" int cmb_rank_batch(cmb_index* idx, int rev, const uint32_t* c, const uint64_t* p, uint64_t n,
                              uint64_t* out) {
    if (!idx || (n && (!c || !p || !out))) return fail(CMB_ERR_INVALID, "null argument");
    try {
        useDevice(idx->device);
        for (uint64_t i = 0; i < n; i++)
            if (c[i] > 3 || p[i] > idx->d.n) return fail(CMB_ERR_INVALID, "rank argument out of range");
        DevBuf<uint32_t> dc;
        DevBuf<uint64_t> dp, dout;
        dc.upload(c, n);
        dp.upload(p, n);
        dout.alloc(n);
        if (n) hipLaunchKernelGGL(k_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx->d, rev, dc.p, dp.p, n, dout.p);
        HIPCHK(hipGetLastError());
        if (n) HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

static int checkRanges(const cmb_index* idx, const uint32_t* in, uint64_t n) {
    for (uint64_t i = 0; i < 4 * n; i++)
        if (in[i] > idx->d.n) return fail(CMB_ERR_INVALID, "range bound beyond the text length");
    return CMB_OK;
}

extern "C" int cmb_extend_batch(cmb_index* idx, int mode, const uint32_t* in, uint64_t n, uint32_t* out,
                                uint8_t* ok) {
    if (!idx || mode < 0 || mode > 2 || (n && (!in || !out || !ok))) return fail(CMB_ERR_INVALID, "bad argument");
    if (checkRanges(idx, in, n)) return CMB_ERR_INVALID;
    try {
        useDevice(idx->device);
        DevBuf<uint4> din, dout;
        DevBuf<uint8_t> dok;
        din.upload((const uint4*)in, n);
        dout.alloc(4 * n);
        dok.alloc(4 * n);
        if (n) {
            const unsigned nb = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
            hipLaunchKernelGGL(k_extend, dim3(nb), dim3(256), 0, 0, idx->d, mode, din.p, n, dout.p, dok.p);
        }
        HIPCHK(hipGetLastError());
        if (n) {
            HIPCHK(hipMemcpy(out, dout.p, n * 64, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(ok, dok.p, n * 4, hipMemcpyDeviceToHost));
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_extend_bench(cmb_index* idx, int mode, const void* d_in, uint64_t n, void* d_out, void* d_ok,
                                uint32_t iters, float* avg_ms) {
    if (!idx || mode < 0 || mode > 2 || !d_in || !d_out || !d_ok || !avg_ms || !iters)
        return fail(CMB_ERR_INVALID, "bad argument");
    try {
        useDevice(idx->device);
        hipStream_t s;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        const unsigned nb = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
        auto kern = k_extend;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s, idx->d, mode, (const uint4*)d_in, n, (uint4*)d_out,
                           (uint8_t*)d_ok); // warm-up
        HIPCHK(hipEventRecord(a, s));
        for (uint32_t i = 0; i < iters; i++)
            hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s, idx->d, mode, (const uint4*)d_in, n,
                               (uint4*)d_out, (uint8_t*)d_ok);
        HIPCHK(hipEventRecord(b, s));
        HIPCHK(hipEventSynchronize(b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        *avg_ms = ms / iters;
        HIPCHK(hipGetLastError());
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        (void)hipStreamDestroy(s);
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

extern "C" int cmb_locate_batch(cmb_index* idx, const uint32_t* rows, uint64_t n, uint32_t* out,
                                uint64_t* lf_steps) {
    if (!idx || (n && (!rows || !out))) return fail(CMB_ERR_INVALID, "null argument");
    for (uint64_t i = 0; i < n; i++)
        if (rows[i] >= idx->d.n) return fail(CMB_ERR_INVALID, "suffix array row out of range");
    try {
        useDevice(idx->device);
        DevBuf<uint32_t> dr, dout;
        DevBuf<unsigned long long> dlf;
        dr.upload(rows, n);
        dout.alloc(n);
        dlf.alloc(1);
        HIPCHK(hipMemset(dlf.p, 0, 8));
        if (n) hipLaunchKernelGGL(k_locate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx->d, dr.p, n, dout.p, dlf.p);
        HIPCHK(hipGetLastError());
        if (n) HIPCHK(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
        unsigned long long lf = 0;
        HIPCHK(hipMemcpy(&lf, dlf.p, 8, hipMemcpyDeviceToHost));
        if (lf_steps) *lf_steps = lf;
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// The PRODUCTION edit-distance verification path for one pattern and n start positions: k_verify<KEYS> (one key per
// candidate) -> radix sort + run-length encode (identical candidates are verified once, counters scaled by the
// multiplicity) -> k_verify_stage x stages -> k_traceback (beyond 7 errors: -> k_wide_filter -> k_verify_wide), exactly as cmb_batch_run
// runs it (it IS cmb_batch_run on a one-read batch whose in-text items are given).  Same contract as cmb_verify_batch, incl. duplicates in `starts`.
extern "C" int cmb_verify_batch_staged(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* starts,
                                       uint64_t n, uint32_t max_ed, uint32_t min_ed, int fixed_start, cmb_occ* out,
                                       uint64_t out_cap, uint64_t* n_out, uint64_t* counters) {
    if (!idx || !pattern || (n && !starts) || !n_out) return fail(CMB_ERR_INVALID, "null argument");
    if (plen == 0 || plen > (uint32_t)MAX_READ) return fail(CMB_ERR_UNSUPPORTED, "pattern length not supported");
    if (max_ed == 0 || max_ed > MXN_MAX_ED || min_ed > 15) return fail(CMB_ERR_UNSUPPORTED, "more than 13 errors"); // (8 ... 13: k_wide_filter + k_verify_wide)
    if (n >= (1ull << 31)) return fail(CMB_ERR_INVALID, "too many start positions");
    for (uint64_t i = 0; i < n; i++)
        if (starts[i] > idx->d.n) return fail(CMB_ERR_INVALID, "start position beyond the text");
    *n_out = 0;
    if (n == 0) {
        if (counters) memset(counters, 0, CMB_CNT_MAX * sizeof(uint64_t));
        return CMB_OK;
    }
    cmb_strategy* st = nullptr;
    int rc = cmb_strategy_create_named("columba", CMB_METRIC_EDIT, CMB_PARTITION_UNIFORM, &st); // (schemes for 1..7 errors; unused)
    if (rc) return rc;
    cmb_batch* b = nullptr;
    const uint64_t ho[2] = {0, plen};
    rc = batchCreateOne(idx, st, max_ed, pattern, ho, 1, &b);
    cmb_strategy_destroy(st);
    if (rc) return rc;
    const uint32_t meta = (max_ed << 12) | (min_ed << 16) | ((fixed_start ? 1u : 0u) << 20) | ((uint32_t)ITEM_EDIT << 21) | (1u << 23);
    b->presetItems.resize(n);
    for (uint64_t i = 0; i < n; i++) b->presetItems[i] = make_uint4(0, starts[i], 0, meta); // (bit 23: nothing to locate)
    rc = batchRunOne(b);
    if (rc == CMB_OK) {
        std::vector<TextOccRec> t = b->presetOut;
        t.erase(std::remove_if(t.begin(), t.end(), [](const TextOccRec& x) { return x.rsId == 0xFFFFFFFFu; }), t.end());
        std::sort(t.begin(), t.end(), [](const TextOccRec& x, const TextOccRec& y) {
            if (x.begin != y.begin) return x.begin < y.begin;
            if (x.end != y.end) return x.end < y.end;
            return x.dist < y.dist;
        });
        *n_out = t.size();
        if (t.size() > out_cap) rc = fail(CMB_ERR_OVERFLOW, "output buffer too small");
        else
            for (size_t i = 0; i < t.size(); i++) out[i] = cmb_occ{t[i].begin, t[i].end, t[i].dist, 0};
        if (counters) memcpy(counters, b->cnts, sizeof(b->cnts));
    }
    cmb_batch_destroy(b);
    return rc;
}

// in-text verification hook: FMIndex::inTextVerification(startPos, maxED, minED, ..., pattern,
// fixedStartPos) (fmindex.cpp:267-310) for one pattern.  Runs the production k_prep + k_verify on a
// one-read batch whose items carry the start positions directly (meta bit 23: nothing to locate).
static int verifyDirect(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* starts, const uint32_t* ends,
                        uint64_t n, uint32_t max_ed, uint32_t min_ed, int fixed_start, cmb_occ* out, uint64_t out_cap,
                        uint64_t* n_out, uint64_t* counters);
extern "C" int cmb_verify_batch(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* starts,
                                uint64_t n, uint32_t max_ed, uint32_t min_ed, int fixed_start, cmb_occ* out,
                                uint64_t out_cap, uint64_t* n_out, uint64_t* counters) {
    return verifyDirect(idx, pattern, plen, starts, nullptr, n, max_ed, min_ed, fixed_start, out, out_cap, n_out, counters);
}
// FMIndex::inTextVerificationOneString (fmindex.cpp:312-342): the pattern against ONE text window [start, end) with a
// fixed start (one zero in the first column) — what IndexInterface::findSeqName runs on an occurrence it has trimmed
// to the sequence it lies in (indexinterface.cpp:850-866)
extern "C" int cmb_verify_window(cmb_index* idx, const char* pattern, uint32_t plen, uint32_t start, uint32_t end,
                                 uint32_t max_ed, uint32_t min_ed, cmb_occ* out, uint64_t out_cap, uint64_t* n_out,
                                 uint64_t* counters) {
    if (end <= start) return fail(CMB_ERR_INVALID, "empty text window");
    return verifyDirect(idx, pattern, plen, &start, &end, 1, max_ed, min_ed, 1, out, out_cap, n_out, counters);
}
static int verifyDirect(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* starts, const uint32_t* ends,
                        uint64_t n, uint32_t max_ed, uint32_t min_ed, int fixed_start, cmb_occ* out, uint64_t out_cap,
                        uint64_t* n_out, uint64_t* counters) {
    if (!idx || !pattern || (n && !starts) || !n_out) return fail(CMB_ERR_INVALID, "null argument");
    if (plen == 0 || plen > (uint32_t)MAX_READ) return fail(CMB_ERR_UNSUPPORTED, "pattern length not supported");
    const bool dp = 3 * max_ed + 1 > MXW_LEFT; // beyond 7 errors: k_verify_wide (the band does not fit the bit-parallel in-text matrices)
    if (max_ed > MXN_MAX_ED || min_ed > 15) return fail(CMB_ERR_UNSUPPORTED, "more than 13 errors");
    for (uint64_t i = 0; i < n; i++)
        if (starts[i] > idx->d.n) return fail(CMB_ERR_INVALID, "start position beyond the text");
    try {
        useDevice(idx->device);
        const uint32_t mlen = (plen + 15u) & ~15u; // row stride of the per-read arrays (k_prep stores 16 bytes)
        const uint32_t gw = gWords(mlen);
        DevBuf<uint8_t> reads, seq;
        DevBuf<uint64_t> offs;
        DevBuf<uint32_t> G, cnt;
        DevBuf<uint4> items;
        DevBuf<TextOccRec> text;
        DevBuf<uint64_t> vW;
        DevBuf<uint4> tbq;
        DevBuf<unsigned long long> ctr;
        const uint64_t ho[2] = {0, plen};
        reads.uploadPadded((const uint8_t*)pattern, plen, 64);
        offs.upload(ho, 2);
        seq.alloc(2 * (size_t)mlen);
        G.alloc(8 * (size_t)gw);
        std::vector<uint4> hi(n);
        const uint32_t meta = (max_ed << 12) | (min_ed << 16) | ((fixed_start ? 1u : 0u) << 20) |
                              ((uint32_t)ITEM_EDIT << 21) | (1u << 23);
        for (uint64_t i = 0; i < n; i++) hi[i] = make_uint4(0, starts[i], ends ? ends[i] : 0u, meta);
        items.upload(hi.data(), n);
        // (every wavefront of k_verify / k_traceback may leave one partly used chunk of 256 records)
        const size_t cap = n * 32 + 64 + 2 * (std::min<uint64_t>(std::max<uint64_t>(((n + 255) / 256) * 256, 256), 65536) / 64 + 1) * 256;
        text.alloc(cap);
        cnt.alloc(8);
        ctr.alloc(CMB_CNT_MAX);
        HIPCHK(hipMemset(cnt.p, 0, 32));
        HIPCHK(hipMemset(ctr.p, 0, CMB_CNT_MAX * 8));
        const uint32_t slots = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(((n + 255) / 256) * 256, 256), 65536);
        const bool narrow = max_ed <= TBN_MAX_ED && !getenv("CMB_TRACE_WIDE");
        const uint32_t tLines = narrow ? ((uint32_t)VROWS + 15u) / 16u + 2u : ((uint32_t)VROWS + 7u) / 8u + 2u;
        vW.alloc((size_t)tLines * 8 * slots);
        tbq.alloc(n + (size_t)(slots / 64 + 1) * 256);
        VPlanes vp{vW.p, slots, tLines};
        Queues q{};
        q.text = text.p;
        q.textCap = (uint32_t)cap;
        q.cnt = cnt.p;
        q.counters = ctr.p;
        HIPCHK(hipMemset(G.p, 0, G.bytes()));
        hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, 0, reads.p, offs.p, 1u, mlen, gw, (mlen + 31) / 32, seq.p,
                           G.p, (uint32_t*)nullptr, 0u);
        DevBuf<uint4> mfull;
        MFull mf{nullptr, mfullBlocks(mlen)};
        mfull.alloc((size_t)2 * 2 * mf.nBlk);
        hipLaunchKernelGGL(k_match_words, dim3(1), dim3(256), 0, 0, G.p, gw, offs.p, 2u, mf.nBlk, mfull.p);
        mf.p = mfull.p;
        uint32_t hc[8];
        if (n && dp) {
            DevBuf<uint8_t> slab;
            const uint32_t slotBytes = (vwRows(mlen) + 1u) * VW_ROW_BYTES;
            const uint32_t dSlots = (uint32_t)std::min<uint64_t>(((n + 255) / 256) * 256, 256u * 64u);
            slab.alloc((size_t)slotBytes * dSlots);
            auto kWide = k_verify_wide<false, WxTen>;
            if (max_ed > MX_MAX_ED) kWide = k_verify_wide<false, WxThirteen>;
            hipLaunchKernelGGL(kWide, dim3(dSlots / 256), dim3(256), 0, 0, idx->d, offs.p, mlen, seq.p, G.p, gw, items.p, (uint32_t)n,
                               (const unsigned long long*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, slab.p,
                               slotBytes, q);
            HIPCHK(hipDeviceSynchronize());
        } else if (n) {
            hipLaunchKernelGGL(k_verify<false>, dim3(slots / 256), dim3(256), 0, 0, idx->d, offs.p, mlen, gw, seq.p, mf,
                               items.p, (uint32_t)n, tbq.p, (uint32_t)tbq.n, (unsigned long long*)nullptr, q);
            HIPCHK(hipMemcpy(hc, cnt.p, 32, hipMemcpyDeviceToHost));
            if (hc[7]) {
                auto kTrace = k_traceback<false, false>;
                if (narrow) kTrace = idx->d.text2 ? k_traceback<true, true> : k_traceback<true, false>;
                hipLaunchKernelGGL(kTrace, dim3(slots / 256), dim3(256), 0, 0,
                                   idx->d, offs.p, mf, tbq.p, hc[7], vp, q, 0u);
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(hc, cnt.p, 32, hipMemcpyDeviceToHost));
        if (hc[3] & FLAG_TEXT_OVERFLOW) return fail(CMB_ERR_INTERNAL, "verification output overflow");
        if (hc[3] & (FLAG_CAPACITY | FLAG_TRACE_RULE)) return fail(CMB_ERR_INTERNAL, "a traceback left the band");
        std::vector<TextOccRec> t(hc[2]);
        if (hc[2]) HIPCHK(hipMemcpy(t.data(), text.p, hc[2] * sizeof(TextOccRec), hipMemcpyDeviceToHost));
        t.erase(std::remove_if(t.begin(), t.end(), [](const TextOccRec& x) { return x.rsId == 0xFFFFFFFFu; }),
                t.end()); // holes of the wavefronts' queue chunks
        *n_out = t.size();
        if (t.size() > out_cap) return fail(CMB_ERR_OVERFLOW, "output buffer too small");
        std::sort(t.begin(), t.end(), [](const TextOccRec& x, const TextOccRec& y) {
            if (x.begin != y.begin) return x.begin < y.begin;
            if (x.end != y.end) return x.end < y.end;
            return x.dist < y.dist;
        });
        for (size_t i = 0; i < t.size(); i++) out[i] = cmb_occ{t[i].begin, t[i].end, t[i].dist, 0};
        if (counters) {
            unsigned long long c2[CMB_CNT_MAX];
            HIPCHK(hipMemcpy(c2, ctr.p, sizeof(c2), hipMemcpyDeviceToHost));
            for (int i = 0; i < CMB_CNT_MAX; i++) counters[i] = c2[i];
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// CIGAR of one pattern against n text windows [begin, end) with given distances (IBitParallelED::findCIGAR,
// bitparallelmatrix.h:460-527) on the device: k_cigar over a one-read batch.  ops_out: n x stride run-length
// operations (length << 2 | op) from the begin of the alignment, n_ops_out[i] of them for window i.
extern "C" int cmb_cigar_windows(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* begins,
                                 const uint32_t* ends, const uint32_t* distances, uint64_t n, uint16_t* ops_out,
                                 uint32_t stride, uint32_t* n_ops_out) {
    if (!idx || !pattern || (n && (!begins || !ends || !distances || !ops_out || !n_ops_out))) return fail(CMB_ERR_INVALID, "null argument");
    if (plen == 0 || plen > (uint32_t)MAX_READ) return fail(CMB_ERR_UNSUPPORTED, "pattern length not supported");
    uint32_t maxD = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (ends[i] < begins[i] || ends[i] > idx->d.n) return fail(CMB_ERR_INVALID, "text window out of range");
        if (ends[i] - begins[i] > plen + distances[i]) return fail(CMB_ERR_INVALID, "text window longer than an alignment within the distance");
        maxD = std::max(maxD, distances[i]);
    }
    if (maxD > MXN_MAX_ED) return fail(CMB_ERR_UNSUPPORTED, "more than 13 errors"); // (fixed start: the band of 2 maxD + 1 <= 27 columns fits the in-text matrix)
    if (stride < 2 * maxD + 3) return fail(CMB_ERR_INVALID, "stride must be at least 2 * distance + 3");
    if (n == 0) return CMB_OK;
    try {
        useDevice(idx->device);
        const uint32_t mlen = (plen + 15u) & ~15u, gw = gWords(mlen);
        DevBuf<uint8_t> reads, seq;
        DevBuf<uint64_t> offs, vW;
        DevBuf<uint32_t> G, occRead, flag;
        DevBuf<uint4> occs, mfull;
        DevBuf<uint16_t> ops;
        DevBuf<AlnRec> aln;
        const uint64_t ho[2] = {0, plen};
        reads.uploadPadded((const uint8_t*)pattern, plen, 64);
        offs.upload(ho, 2);
        seq.alloc(2 * (size_t)mlen);
        G.alloc(8 * (size_t)gw);
        HIPCHK(hipMemset(G.p, 0, G.bytes()));
        hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, 0, reads.p, offs.p, 1u, mlen, gw, (mlen + 31) / 32, seq.p, G.p,
                           (uint32_t*)nullptr, 0u);
        MFull mf{nullptr, mfullBlocks(mlen)};
        mfull.alloc((size_t)2 * 2 * mf.nBlk);
        hipLaunchKernelGGL(k_match_words, dim3(1), dim3(256), 0, 0, G.p, gw, offs.p, 2u, mf.nBlk, mfull.p);
        mf.p = mfull.p;
        std::vector<uint4> ho4(n);
        for (uint64_t i = 0; i < n; i++) ho4[i] = make_uint4(begins[i], ends[i], distances[i], 0u);
        occs.upload(ho4.data(), n);
        std::vector<uint32_t> zeros(n, 0u);
        occRead.upload(zeros.data(), n);
        flag.alloc(1);
        HIPCHK(hipMemset(flag.p, 0, sizeof(uint32_t)));
        const uint32_t slots = (uint32_t)std::min<uint64_t>(((n + 255) / 256) * 256, 65536);
        const bool narrow = maxD <= TBN_MAX_ED;
        const uint32_t tLines = narrow ? ((uint32_t)VROWS + 15u) / 16u + 2u : ((uint32_t)VROWS + 7u) / 8u + 2u;
        vW.alloc((size_t)tLines * 8 * slots);
        ops.alloc(n * stride);
        aln.alloc(n);
        VPlanes vp{vW.p, slots, tLines};
        auto kc = k_cigar<false, false>;
        if (narrow) kc = idx->d.text2 ? k_cigar<true, true> : k_cigar<true, false>;
        else if (idx->d.text2) kc = k_cigar<false, true>;
        DevBuf<uint8_t> slab;
        if (maxD > CIGAR_BLOCK_WORDS_MAX_ED) { // (k_cigar's match words reach 9 columns right of the diagonal: kernels.hpp, k_cigar_wide)
            const uint32_t slotBytes = (vwRows(mlen) + 1u) * VW_ROW_BYTES;
            slab.alloc((size_t)slotBytes * slots);
            hipLaunchKernelGGL(k_cigar_wide, dim3(slots / 256), dim3(256), 0, 0, idx->d, offs.p, G.p, gw, occs.p, occRead.p, n, slab.p, slotBytes,
                               idx->seqStartsDev.p, idx->nSeqsDev, ops.p, stride, aln.p, flag.p, 0u);
        } else
        hipLaunchKernelGGL(kc, dim3(slots / 256), dim3(256), 0, 0, idx->d, offs.p, mf, occs.p, occRead.p, n, vp, idx->seqStartsDev.p,
                           idx->nSeqsDev, ops.p, stride, aln.p, flag.p, 0u);
        HIPCHK(hipGetLastError());
        uint32_t hf = 0;
        HIPCHK(hipMemcpy(&hf, flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (hf) return fail(CMB_ERR_INTERNAL, "a CIGAR traceback left the band or a window is not an alignment within its distance");
        std::vector<AlnRec> ha(n);
        std::vector<uint16_t> hops(n * stride);
        HIPCHK(hipMemcpy(ha.data(), aln.p, n * sizeof(AlnRec), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hops.data(), ops.p, n * stride * sizeof(uint16_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) {
            n_ops_out[i] = ha[i].nOps;
            for (uint32_t j = 0; j < ha[i].nOps; j++) ops_out[i * stride + j] = hops[i * stride + ha[i].nOps - 1 - j];
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

namespace {

struct BestOcc {
    cmb_occ occ;
    cmb_aln aln;
    std::vector<uint16_t> ops;
};
struct BestRead {
    uint32_t len = 0, cutOff = 0, best = 0, k = 0, prevK = 0, maxED = 0;
    bool bestFound = false, finished = false;
    std::vector<std::vector<BestOcc>> ov[2];  // [strand][distance]
    std::vector<uint8_t> processed[2];        // [strand][distance]
};

// IndexInterface::findSeqName for an occurrence that runs over the end of its sequence (indexinterface.cpp:833-899):
// trim to the sequence it mostly lies in and verify again inside that window (edit distance only)
static bool trimOccurrence(cmb_index* idx, const std::string& seq, uint32_t largestStratum, int metric, BestOcc& o,
                           uint64_t* counters) {
    if (metric == CMB_METRIC_HAMMING) return false;
    const std::vector<uint32_t>& sp = idx->seqStarts;
    const uint32_t index = o.aln.seq_id;
    if (sp.size() < 2 || index + 1 >= sp.size()) return false;
    uint32_t b = o.occ.begin, e = o.occ.end, seqId = index;
    if (sp[index + 1] - b <= largestStratum) { // option 1: begin is just before the end of the sequence
        seqId = index + 1;
        if (seqId + 1 >= sp.size()) return false;
        b = sp[seqId];
        e = std::min(e, sp[seqId + 1]);
    } else if (e - sp[index + 1] <= largestStratum) { // option 2: end is just over the start of the next sequence
        e = sp[index + 1];
    } else {
        return false;
    }
    if (e <= b) return false;
    cmb_occ res[64];
    uint64_t n = 0, cnt[CMB_CNT_MAX];
    if (cmb_verify_window(idx, seq.data(), (uint32_t)seq.size(), b, e, largestStratum, 0, res, 64, &n, cnt) != CMB_OK) return false;
    for (int i = 0; counters && i < CMB_CNT_MAX; i++) counters[i] += cnt[i];
    if (n == 0) return false;
    // the minimal occurrence under TextOcc::operator< (begin, distance, width; indexhelpers.h:779-795)
    const cmb_occ* bestO = &res[0];
    for (uint64_t i = 1; i < n; i++) {
        const cmb_occ& c = res[i];
        const uint32_t wc = c.end - c.begin, wb = bestO->end - bestO->begin;
        if (c.begin != bestO->begin ? c.begin < bestO->begin
                                    : c.distance != bestO->distance ? c.distance < bestO->distance : wc < wb)
            bestO = &c;
    }
    o.occ.begin = bestO->begin;
    o.occ.end = bestO->end;
    o.occ.distance = bestO->distance;
    o.aln.seq_id = seqId;
    o.aln.seq_begin = bestO->begin - sp[seqId];
    o.aln.spans = 2; // found with trimming
    uint16_t opsBuf[2 * 13 + 3];
    uint32_t nOps = 0;
    o.ops.clear();
    if (cmb_cigar_windows(idx, seq.data(), (uint32_t)seq.size(), &o.occ.begin, &o.occ.end, &o.occ.distance, 1, opsBuf, 2 * 13 + 3,
                          &nOps) != CMB_OK)
        return false;
    o.ops.assign(opsBuf, opsBuf + nOps);
    return true;
}

} // namespace

// IndexInterface::findSeqName for ONE occurrence that runs over the end of its sequence (cmb_aln.spans == 1), for host layers
// that build their own records (paired reads): FOUND_WITH_TRIMMING -> *found = 1 and occ / aln / CIGAR updated; NOT_FOUND -> 0.
extern "C" int cmb_trim_occurrence(cmb_index* idx, const char* pattern, uint32_t plen, uint32_t largest_stratum, int metric, cmb_occ* occ,
                                   cmb_aln* aln, uint16_t* cigar_ops, uint32_t ops_cap, uint32_t* n_ops, int* found) {
    if (!idx || !pattern || !occ || !aln || !found) return fail(CMB_ERR_INVALID, "null argument");
    try {
        BestOcc o;
        o.occ = *occ;
        o.aln = *aln;
        const bool ok = trimOccurrence(idx, std::string(pattern, plen), largest_stratum, metric, o, nullptr);
        *found = ok ? 1 : 0;
        if (ok) {
            if (o.ops.size() > ops_cap) return fail(CMB_ERR_OVERFLOW, "room for the CIGAR operations of the trimmed occurrence");
            *occ = o.occ;
            *aln = o.aln;
            aln->cigar_len = (uint16_t)o.ops.size();
            for (size_t i = 0; i < o.ops.size() && cigar_ops; i++) cigar_ops[i] = o.ops[i];
            if (n_ops) *n_ops = (uint32_t)o.ops.size();
        }
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// SAM text of a whole chunk in ALL mode: matchApproxAllMap D) -> SearchStrategy::generateOutputSingleEnd
// (searchstrategy.cpp:530-533, :1824-1902) -> generateSE_SAM / generateSE_SAM_XATag (searchstrategy.h:1612-1641), from the
// occurrences, CIGARs and sequence assignments the device produced (cmb_batch_want_alignments before cmb_batch_run).
// Occurrences that run over the end of their sequence are trimmed and verified again (findSeqName,
// indexinterface.cpp:833-899) through the device hooks.
// the SAM records of one read from its occurrences with alignments (generateOutputSingleEnd, searchstrategy.cpp:1824-1902)
static void samOfRead(std::string& text, cmb_index* idx, uint32_t k, int metric, const std::string& read, const std::string& revC,
                      const std::string& sid, const std::string& qual, std::vector<BestOcc>& occs, const char* const* seq_names,
                      int unmapped_records, int xa_tag) {
    std::string revQ = qual;
    std::reverse(revQ.begin(), revQ.end());
    std::vector<SamHit> hits;
    for (BestOcc& o : occs) {
        if (o.aln.spans == 1 && !trimOccurrence(idx, o.occ.strand ? revC : read, k, metric, o, nullptr)) continue; // NOT_FOUND
        SamHit h;
        h.seqName = seq_names[o.aln.seq_id];
        h.cigar = cigarString(o.ops.data(), (uint32_t)o.ops.size());
        h.pos0 = o.aln.seq_begin;
        h.distance = o.occ.distance;
        h.revCompl = o.occ.strand != 0;
        hits.push_back(h);
    }
    if (hits.empty()) {
        if (unmapped_records) text += samLineUnmappedSE(sid, read, qual);
        return;
    }
    // primary = the first occurrence of minimal distance, swapped to the front (:1883-1899)
    size_t mi = 0;
    for (size_t j = 1; j < hits.size(); j++)
        if (hits[j].distance < hits[mi].distance) mi = j;
    const uint32_t minScore = hits[mi].distance;
    uint32_t nHits = 0;
    for (const SamHit& h : hits) nHits += h.distance == minScore;
    if (mi != 0) std::swap(hits[0], hits[mi]);
    const bool rcFirst = hits[0].revCompl;
    if (xa_tag) {
        text += samLineSEWithXA(sid, hits, nHits, rcFirst ? revC : read, rcFirst ? revQ : qual);
    } else {
        text += samLineSE(sid, hits[0], true, nHits, minScore, rcFirst ? revC : read, rcFirst ? revQ : qual);
        for (size_t j = 1; j < hits.size(); j++) text += samLineSE(sid, hits[j], false, nHits, minScore, "*", "*");
    }
}

extern "C" int64_t cmb_batch_sam(const cmb_batch* b, const char* seqs, const char* const* read_ids, const char* const* quals,
                                 const char* const* seq_names, int unmapped_records, int xa_tag, char* out, uint64_t cap) {
    if (!b || !seqs || !read_ids || !seq_names) return fail(CMB_ERR_INVALID, "null argument");
    if (!b->done) return fail(CMB_ERR_INVALID, "batch has not been run");
    if (!b->wantAln) return fail(CMB_ERR_INVALID, "alignments were not requested (cmb_batch_want_alignments)");
    try {
        std::vector<const cmb_batch*> parts;
        if (b->subs.empty()) parts.push_back(b);
        else
            for (const cmb_batch* c : b->subs) parts.push_back(c);
        cmb_index* idx = b->ix;
        std::string text;
        uint64_t readBase = 0, charBase = 0;
        for (const cmb_batch* c : parts) {
            for (uint32_t i = 0; i < c->nReads; i++) {
                const uint64_t gi = readBase + i;
                const uint64_t o0 = charBase + c->hostOffs[i], o1 = charBase + c->hostOffs[i + 1];
                const std::string read = cleanReadSeq(std::string(seqs + o0, seqs + o1)), revC = revComplWithN(read);
                const std::string sid = cleanSeqID(read_ids[gi]);
                const std::string qual = quals && quals[gi] ? quals[gi] : "*";
                std::vector<BestOcc> occs;
                const uint64_t q0 = c->occOffs.data()[c->perStrand ? 2 * (size_t)i : i], q1 = c->occOffs.data()[c->perStrand ? 2 * (size_t)i + 2 : i + 1];
                for (uint64_t q2 = q0; q2 < q1; q2++) {
                    BestOcc o;
                    o.occ = c->occs.data()[q2];
                    const AlnRec& ar = c->hAlnRec.data()[q2];
                    o.aln = cmb_aln{ar.seqId, ar.seqBegin, 0, (uint16_t)ar.nOps, (uint16_t)ar.spans, 0};
                    const uint16_t* src = c->hAlnOps.data() + q2 * c->alnStride;
                    for (uint32_t j = 0; j < ar.nOps; j++) o.ops.push_back(src[ar.nOps - 1 - j]);
                    occs.push_back(std::move(o));
                }
                samOfRead(text, idx, c->k, c->metric, read, revC, sid, qual, occs, seq_names, unmapped_records, xa_tag);
            }
            readBase += c->nReads;
            charBase += c->hostOffs[c->nReads];
        }
        return putString(text, out, cap);
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// The same for occurrences and alignments the caller holds (cmb_batch_results + cmb_batch_alignments, or their b-move
// counterparts with the text-only index of cmb_move_text_index): occ_offs[n_reads + 1], aln[i].cigar_off into cigar_ops.
extern "C" int64_t cmb_sam_chunk(cmb_index* idx, uint32_t max_distance, int metric, const char* seqs, const uint64_t* offs, uint32_t n_reads,
                                 const char* const* read_ids, const char* const* quals, const char* const* seq_names, const cmb_occ* occ,
                                 const uint64_t* occ_offs, const cmb_aln* aln, const uint16_t* cigar_ops, int unmapped_records, int xa_tag,
                                 char* out, uint64_t cap) {
    if (!idx || !seqs || !offs || !read_ids || !seq_names || !occ_offs || (occ_offs[n_reads] && (!occ || !aln || !cigar_ops)))
        return fail(CMB_ERR_INVALID, "null argument");
    try {
        std::string text;
        for (uint32_t i = 0; i < n_reads; i++) {
            const std::string read = cleanReadSeq(std::string(seqs + offs[i], seqs + offs[i + 1])), revC = revComplWithN(read);
            const std::string sid = cleanSeqID(read_ids[i]);
            const std::string qual = quals && quals[i] ? quals[i] : "*";
            std::vector<BestOcc> occs;
            for (uint64_t q2 = occ_offs[i]; q2 < occ_offs[i + 1]; q2++) {
                BestOcc o;
                o.occ = occ[q2];
                o.aln = aln[q2];
                o.ops.assign(cigar_ops + aln[q2].cigar_off, cigar_ops + aln[q2].cigar_off + aln[q2].cigar_len);
                occs.push_back(std::move(o));
            }
            samOfRead(text, idx, max_distance, metric, read, revC, sid, qual, occs, seq_names, unmapped_records, xa_tag);
        }
        return putString(text, out, cap);
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}

// ------------------------------------------------------------------------- BEST (+x strata) mode
// SearchStrategy::matchApproxBestPlusX / findBestAlignments / processSeq / mapRead / checkAlignments /
// combineOccVectors (reference src/searchstrategy.cpp:623-760, :791-812, :536-620; src/searchstrategy.h:490-523).
// The reference walks every read through its strata on its own: exact matches, then k = 1, 3, 5, 9, 13 (k + x + 2 below
// 5, + 4 above) up to the identity cut-off, until a stratum holds an alignment; every stratum is one ALL-mode search
// of ONE strand (mapRead) whose occurrences below the first unprocessed distance are dropped.  Here a stratum is one
// device batch over all reads that are still looking at that distance: both strands at once, each strand filtered
// by itself (cmb_batch_filter_per_strand), CIGARs and sequence assignment from the device (k_cigar); the per-read
// bookkeeping below is the reference's.

struct cmb_best {
    std::vector<cmb_occ> occ;
    std::vector<cmb_aln> aln;
    std::vector<uint16_t> ops;
    std::vector<uint64_t> offs;
    std::vector<uint32_t> best, nHits;
    uint64_t cnts[CMB_CNT_MAX];
};

namespace {
// what one stratum returns: the ALL-mode lists of a set of reads at distance k, both strands, every strand filtered by itself
struct StratumOut {
    std::vector<cmb_occ> oc;
    std::vector<cmb_aln> al;
    std::vector<uint16_t> ops;
    std::vector<uint64_t> oo, cnt;
};
typedef std::function<int(const char* cat, const uint64_t* o, uint32_t n, uint32_t k, StratumOut& r)> StratumRunner;

// idx: the index whose text and sequence starts serve the trimming (the FM-index itself, or the text beside a b-move index);
// deviceCap: the largest distance the flavour's device path runs; trimCounters: the in-text counters of a trimming count
// (FM-index flavour: inTextVerificationOneString; the b-move flavour's checkTrimmedMatch counts nothing, indexinterface.cpp:722-796)
int matchBestWith(cmb_index* idx, const cmb_strategy* st, uint32_t deviceCap, bool trimCounters, const StratumRunner& runOn, uint32_t x,
                  uint32_t min_identity, const char* seqs, const uint64_t* offs, uint32_t n_reads, cmb_best** out) {
    if (!idx || !st || !offs || !out || (!seqs && n_reads)) return fail(CMB_ERR_INVALID, "null argument");
    if (min_identity < 50 || min_identity > 100) return fail(CMB_ERR_INVALID, "the minimal identity lies between 50 and 100");
    try {
        // getMaxSupportedDistanceForBestMapping (searchstrategy.h:1864, :2744): the largest k such that 1..k all have a
        // scheme — and that this device runs (deviceCap)
        uint32_t maxSupported = 0;
        while (st->schemes.count(maxSupported + 1) && !st->schemes.at(maxSupported + 1).empty()) maxSupported++;
        maxSupported = std::min<uint32_t>(maxSupported, deviceCap);
        std::unique_ptr<cmb_best> R(new cmb_best());
        memset(R->cnts, 0, sizeof(R->cnts));
        std::vector<BestRead> rd(n_reads);
        std::vector<std::string> fw(n_reads), rc(n_reads);
        for (uint32_t i = 0; i < n_reads; i++) {
            BestRead& r = rd[i];
            r.len = (uint32_t)(offs[i + 1] - offs[i]);
            fw[i] = cleanReadSeq(std::string(seqs + offs[i], seqs + offs[i + 1]));
            rc[i] = revComplWithN(fw[i]);
            r.cutOff = std::min<uint32_t>(std::min<uint32_t>(13u, maxSupported), (r.len * (100 - min_identity)) / 100); // getMaxED (:1797)
            r.best = r.cutOff + 1;
            for (int s2 = 0; s2 < 2; s2++) {
                r.ov[s2].assign(r.cutOff + 1, {});
                r.processed[s2].assign(r.cutOff + 1, 0);
            }
        }
        // one stratum for a set of reads: ALL-mode search of both strands at distance k, every strand filtered by itself
        auto runStratum = [&](const std::vector<uint32_t>& ids, uint32_t k) -> int {
            std::string cat;
            std::vector<uint64_t> o(ids.size() + 1, 0);
            for (size_t j = 0; j < ids.size(); j++) {
                cat.append(seqs + offs[ids[j]], seqs + offs[ids[j] + 1]);
                o[j + 1] = cat.size();
            }
            StratumOut so;
            const int rcode = runOn(cat.data(), o.data(), (uint32_t)ids.size(), k, so);
            if (rcode) return rcode;
            const std::vector<cmb_occ>& oc = so.oc;
            const std::vector<cmb_aln>& al = so.al;
            const std::vector<uint16_t>& ops = so.ops;
            const std::vector<uint64_t>&oo = so.oo, &cnt = so.cnt;
            for (int i = 0; i < CMB_CNT_MAX; i++) R->cnts[i] += cnt[i];
            for (size_t j = 0; j < ids.size(); j++) {
                BestRead& r = rd[ids[j]];
                for (int s2 = 0; s2 < 2; s2++) {
                    if (r.processed[s2][k]) continue; // (hasUpdate: this distance was looked at before)
                    // processSeq (:791-812): minD = the first distance not processed yet
                    uint32_t minD = 0;
                    while (minD < k && r.processed[s2][minD]) minD++;
                    for (uint64_t q2 = oo[j]; q2 < oo[j + 1]; q2++) {
                        if ((int)oc[q2].strand != s2 || oc[q2].distance < minD) continue;
                        BestOcc bo;
                        bo.occ = oc[q2];
                        bo.aln = al[q2];
                        bo.ops.assign(ops.begin() + al[q2].cigar_off, ops.begin() + al[q2].cigar_off + al[q2].cigar_len);
                        r.ov[s2][oc[q2].distance].push_back(std::move(bo));
                    }
                    for (uint32_t d = minD; d <= k; d++) r.processed[s2][d] = 1;
                }
            }
            return CMB_OK;
        };
        // checkAlignments (:536-571): keep what lies inside one sequence; trimmed occurrences move to their new distance
        auto checkAlignments = [&](uint32_t i, int s2, uint32_t l, uint32_t cutOffTrim) {
            BestRead& r = rd[i];
            if (l >= r.ov[s2].size()) return;
            std::vector<BestOcc> assigned, trimmed;
            for (BestOcc& o : r.ov[s2][l]) {
                if (o.aln.spans == 0 || o.aln.spans == 3) { // FOUND (3: checked before)
                    o.aln.spans = 3;
                    assigned.push_back(std::move(o));
                    if (l < r.best) r.best = l;
                } else if (o.aln.spans == 1) {
                    if (trimOccurrence(idx, s2 ? rc[i] : fw[i], cutOffTrim, st->metric, o, trimCounters ? R->cnts : nullptr) && o.occ.distance > l &&
                        o.occ.distance < r.ov[s2].size())
                        trimmed.push_back(std::move(o));
                }
            }
            r.ov[s2][l] = std::move(assigned);
            for (BestOcc& o : trimmed) {
                o.aln.spans = 3; // (removeTrimmingLabel: it is an ordinary assigned occurrence of its new stratum)
                const uint32_t d = o.occ.distance;
                r.ov[s2][d].push_back(std::move(o));
            }
        };
        // ---- exact matches first (x == 0), then the strata
        std::vector<uint32_t> all(n_reads);
        for (uint32_t i = 0; i < n_reads; i++) all[i] = i;
        if (x == 0 && n_reads) {
            int rcode = runStratum(all, 0);
            if (rcode) return rcode;
            for (uint32_t i = 0; i < n_reads; i++) {
                BestRead& r = rd[i];
                if (!r.ov[0][0].empty() || !r.ov[1][0].empty()) {
                    checkAlignments(i, 0, 0, r.cutOff);
                    checkAlignments(i, 1, 0, r.cutOff);
                    if (r.best == 0) r.bestFound = true;
                }
            }
        }
        for (uint32_t i = 0; i < n_reads; i++) {
            BestRead& r = rd[i];
            r.maxED = r.best == 0 ? x : r.cutOff;
            r.prevK = 0;
            r.k = std::max(x, 1u);
            r.finished = r.k > r.maxED;
        }
        std::vector<uint8_t> isFresh(n_reads, 0);
        for (;;) {
            // the reads that look at a stratum now, grouped by its distance
            std::map<uint32_t, std::vector<uint32_t>> byK;
            for (uint32_t i = 0; i < n_reads; i++)
                if (!rd[i].finished) byK[rd[i].k].push_back(i);
            if (byK.empty()) break;
            for (auto& kv : byK) {
                const uint32_t k = kv.first;
                std::vector<uint32_t> need; // (a stratum both strands have been through needs no new search)
                for (uint32_t i : kv.second)
                    if (!rd[i].processed[0][k] || !rd[i].processed[1][k]) need.push_back(i);
                std::fill(isFresh.begin(), isFresh.end(), 0);
                for (uint32_t i : need) isFresh[i] = 1;
                if (!need.empty()) {
                    int rcode = runStratum(need, k);
                    if (rcode) return rcode;
                }
                for (uint32_t i : kv.second) {
                    BestRead& r = rd[i];
                    // hasUpdate (:674-681): a stratum looked at before answers with ITS occurrences only; a new one
                    // (processSeq) with any occurrence at distance 0..k
                    bool update = false;
                    for (int s2 = 0; s2 < 2; s2++) {
                        const bool fresh = isFresh[i] != 0;
                        if (!fresh) update |= !r.ov[s2][k].empty();
                        else
                            for (uint32_t d = 0; d <= k; d++) update |= !r.ov[s2][d].empty();
                    }
                    if (update)
                        for (uint32_t l = r.prevK + 1; l <= std::min(k, r.best + x); l++) {
                            checkAlignments(i, 0, l, r.maxED);
                            checkAlignments(i, 1, l, r.maxED);
                        }
                    if (r.bestFound) {
                        r.finished = true; // this was the last iteration
                        continue;
                    }
                    if (update && r.best < r.cutOff + 1) {
                        r.bestFound = true;
                        if (x == 0) {
                            r.finished = true;
                            continue;
                        }
                        r.prevK = k;
                        r.k = std::min(r.best + x, r.maxED); // check the final x strata
                    } else {
                        if (k == r.maxED) {
                            r.finished = true;
                            continue;
                        }
                        const uint32_t step = k < 5 ? 2 : 4;
                        r.prevK = k;
                        r.k = std::min(k + x + step, r.maxED);
                    }
                }
            }
        }
        // ---- results: combineOccVectors (:573-620) per read
        R->offs.assign(n_reads + 1, 0);
        R->best.assign(n_reads, 0xFFFFFFFFu);
        R->nHits.assign(n_reads, 0);
        // occurrences found with trimming still need their CIGAR: one more (small) pass through the device
        for (uint32_t i = 0; i < n_reads; i++) {
            BestRead& r = rd[i];
            R->offs[i] = R->occ.size();
            if (!r.bestFound) continue;
            R->best[i] = r.best;
            R->nHits[i] = (uint32_t)(r.ov[0][r.best].size() + r.ov[1][r.best].size());
            const uint32_t hi = std::min(r.best + x, r.cutOff);
            for (uint32_t d = r.best; d <= hi; d++)
                for (int s2 = 0; s2 < 2; s2++) {
                    std::vector<BestOcc>& v = r.ov[s2][d];
                    std::stable_sort(v.begin(), v.end(), [](const BestOcc& a, const BestOcc& b2) {
                        return a.aln.seq_id < b2.aln.seq_id || (a.aln.seq_id == b2.aln.seq_id && a.aln.seq_begin < b2.aln.seq_begin);
                    });
                    v.erase(std::unique(v.begin(), v.end(), [](const BestOcc& a, const BestOcc& b2) {
                                return a.aln.seq_id == b2.aln.seq_id && a.aln.seq_begin == b2.aln.seq_begin;
                            }), v.end());
                    for (BestOcc& o : v) {
                        cmb_aln a = o.aln;
                        a.cigar_off = R->ops.size();
                        a.cigar_len = (uint16_t)o.ops.size();
                        a.spans = o.aln.spans == 2 ? 2 : 0;
                        R->ops.insert(R->ops.end(), o.ops.begin(), o.ops.end());
                        R->occ.push_back(o.occ);
                        R->aln.push_back(a);
                    }
                }
        }
        R->offs[n_reads] = R->occ.size();
        *out = R.release();
        return CMB_OK;
    } catch (const std::exception& e) {
        return fail(CMB_ERR_DEVICE, e.what());
    }
}
} // namespace

extern "C" int cmb_match_best(cmb_index* idx, const cmb_strategy* st, uint32_t x, uint32_t min_identity, const char* seqs,
                              const uint64_t* offs, uint32_t n_reads, cmb_best** out) {
    if (!idx || !st) return fail(CMB_ERR_INVALID, "null argument");
    const StratumRunner run = [&](const char* cat, const uint64_t* o, uint32_t n, uint32_t k, StratumOut& r) -> int {
        cmb_batch* b = nullptr;
        int rcode = cmb_batch_create(idx, st, k, cat, o, n, &b);
        if (rcode) return rcode;
        struct Guard {
            cmb_batch* b;
            ~Guard() { cmb_batch_destroy(b); }
        } guard{b};
        cmb_batch_filter_per_strand(b, 1);
        cmb_batch_want_alignments(b, 1);
        if ((rcode = cmb_batch_run(b))) return rcode;
        uint64_t nOcc = 0, nOps = 0;
        cmb_batch_result_size(b, &nOcc);
        r.oc.resize(nOcc ? nOcc : 1);
        r.al.resize(nOcc ? nOcc : 1);
        r.oo.resize((size_t)n + 1);
        r.cnt.assign(CMB_CNT_MAX, 0);
        if ((rcode = cmb_batch_results(b, r.oc.data(), r.oc.size(), r.oo.data(), r.cnt.data()))) return rcode;
        (void)cmb_batch_alignments(b, r.al.data(), 0, nullptr, 0, &nOps);
        r.ops.resize(nOps ? nOps : 1);
        return cmb_batch_alignments(b, r.al.data(), r.al.size(), r.ops.data(), r.ops.size(), &nOps);
    };
    // (MAX_K, definitions.h:50)
    return matchBestWith(idx, st, 13u, true, run, x, min_identity, seqs, offs, n_reads, out);
}
// The same on the b-move index (the reference's RUN_LENGTH_COMPRESSION build runs the same matchApproxBestPlusX): the strata are
// b-move batches, CIGARs and trimming read the matched string of an occurrence from the text beside the index (cmb_move_attach_text).
extern "C" int cmb_move_match_best(cmb_move_index* idx, const cmb_strategy* st, uint32_t x, uint32_t min_identity, uint32_t kmer_size,
                                   const char* seqs, const uint64_t* offs, uint32_t n_reads, cmb_best** out) {
    if (!idx || !st) return fail(CMB_ERR_INVALID, "null argument");
    cmb_index* text = cmb_move_text_index(idx);
    if (!text) return fail(CMB_ERR_INVALID, "BEST mode on the b-move index needs the text beside it (cmb_move_attach_text)");
    const StratumRunner run = [&](const char* cat, const uint64_t* o, uint32_t n, uint32_t k, StratumOut& r) -> int {
        cmb_move_batch* b = nullptr;
        int rcode = cmb_move_batch_create(idx, st, k, kmer_size, cat, o, n, &b);
        if (rcode) return rcode;
        struct Guard {
            cmb_move_batch* b;
            ~Guard() { cmb_move_batch_destroy(b); }
        } guard{b};
        cmb_move_batch_filter_per_strand(b, 1);
        if ((rcode = cmb_move_batch_want_alignments(b, 1))) return rcode;
        if ((rcode = cmb_move_batch_run(b))) return rcode;
        uint64_t nOcc = 0, nOps = 0;
        cmb_move_batch_result_size(b, &nOcc);
        std::vector<cmb_move_occ> mo(nOcc ? nOcc : 1);
        r.al.resize(nOcc ? nOcc : 1);
        r.oo.resize((size_t)n + 1);
        r.cnt.assign(CMB_CNT_MAX, 0);
        if ((rcode = cmb_move_batch_results(b, mo.data(), mo.size(), r.oo.data(), r.cnt.data()))) return rcode;
        r.oc.resize(nOcc ? nOcc : 1);
        for (uint64_t i = 0; i < nOcc; i++) {
            r.oc[i].begin = (uint32_t)mo[i].begin, r.oc[i].end = (uint32_t)mo[i].end; // (texts below 2^32: cmb_move_attach_text)
            r.oc[i].distance = mo[i].distance, r.oc[i].strand = mo[i].strand;
        }
        (void)cmb_move_batch_alignments(b, r.al.data(), 0, nullptr, 0, &nOps);
        r.ops.resize(nOps ? nOps : 1);
        return cmb_move_batch_alignments(b, r.al.data(), r.al.size(), r.ops.data(), r.ops.size(), &nOps);
    };
    // (strata up to 13 errors, the reference's MAX_K, as on the FM-index: the b-move search runs that far since round 3, its alignments since round 4)
    return matchBestWith(text, st, 13u, false, run, x, min_identity, seqs, offs, n_reads, out);
}
extern "C" int cmb_best_sizes(const cmb_best* r, uint64_t* n_occ, uint64_t* n_ops) {
    if (!r) return fail(CMB_ERR_INVALID, "null argument");
    if (n_occ) *n_occ = r->occ.size();
    if (n_ops) *n_ops = r->ops.size();
    return CMB_OK;
}
extern "C" int cmb_best_results(const cmb_best* r, cmb_occ* occ, cmb_aln* aln, uint64_t cap, uint16_t* cigar_ops, uint64_t ops_cap,
                                uint64_t* offs, uint32_t* best, uint32_t* n_hits, uint64_t* counters) {
    if (!r) return fail(CMB_ERR_INVALID, "null argument");
    if (cap < r->occ.size() || ops_cap < r->ops.size()) return fail(CMB_ERR_OVERFLOW, "output buffer too small");
    if (occ && !r->occ.empty()) memcpy(occ, r->occ.data(), r->occ.size() * sizeof(cmb_occ));
    if (aln && !r->aln.empty()) memcpy(aln, r->aln.data(), r->aln.size() * sizeof(cmb_aln));
    if (cigar_ops && !r->ops.empty()) memcpy(cigar_ops, r->ops.data(), r->ops.size() * sizeof(uint16_t));
    if (offs) memcpy(offs, r->offs.data(), r->offs.size() * sizeof(uint64_t));
    if (best) memcpy(best, r->best.data(), r->best.size() * sizeof(uint32_t));
    if (n_hits) memcpy(n_hits, r->nHits.data(), r->nHits.size() * sizeof(uint32_t));
    if (counters) memcpy(counters, r->cnts, sizeof(r->cnts));
    return CMB_OK;
}
extern "C" void cmb_best_destroy(cmb_best* r) { delete r; }
