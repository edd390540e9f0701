/*
 * columba_amd.h — C-ABI of the MI355X-native search-scheme FM-index matcher.
 *
 * This is the drop-in boundary for ONE path of biointec/columba (v2.0.3, Vanilla
 * flavour, 32-bit length_t): approximate matching of single-end reads in ALL mode,
 *      SearchStrategy::matchApprox        src/searchstrategy.h:2021-2024
 *   -> SearchStrategy::matchApproxAllMap  src/searchstrategy.cpp:495-535
 * as called per read from processChunk    src/parallel.cpp:67-78.
 * A batch of reads goes in, per read the post-filter occurrence list
 * {begin,end (concatenated-text coordinates), distance, strand} comes out, exactly what
 * `filterPtr` returns at src/searchstrategy.cpp:529 (before SAM formatting).
 *
 * Plain pointers and sizes only; no C++/torch types.  All functions return CMB_OK (0)
 * or a negative error code; cmb_last_error() gives the message of the calling thread's
 * last failure (the reference throws std::runtime_error, src/fmindex/fmindex.cpp:84,
 * src/indexinterface.cpp:100; the C++ adapter in columba_amd/csrc/host re-throws).
 *
 * Threading: handles are immutable after creation; cmb_batch objects own their stream
 * and scratch, so N host threads may run N batches on one index concurrently (the
 * reference shares one index + strategy between workers, src/parallel.cpp:1143-1146,
 * with all mutable state thread_local, src/indexinterface.cpp:57-71).
 */
#ifndef COLUMBA_AMD_H
#define COLUMBA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMB_OK 0
#define CMB_ERR_INVALID (-1)     /* bad argument / malformed scheme (search.h:559-586) */
#define CMB_ERR_DEVICE (-2)      /* HIP runtime failure / no GPU */
#define CMB_ERR_UNSUPPORTED (-3) /* more than 13 errors (MAX_K of the reference), reads longer than 480, 64-bit length_t on the FM-index path */
#define CMB_ERR_OVERFLOW (-4)    /* caller-provided output buffer too small: nothing truncated silently */
#define CMB_ERR_INTERNAL (-5)    /* device-side capacity exceeded for a read (reported, never silent) */

typedef struct cmb_index cmb_index;       /* device-resident bidirectional FM-index */
typedef struct cmb_strategy cmb_strategy; /* search schemes + partitioning parameters */
typedef struct cmb_batch cmb_batch;       /* a batch of reads resident in HBM + its results */

/* --- index -------------------------------------------------------------------------
 * Host arrays in the reference's in-memory layouts (the members of FMIndex,
 * src/fmindex/fmindex.h:46-58; file formats SURVEY.md §5).  The host keeps ownership;
 * cmb_index_create copies (and re-lays-out) into HBM. */
typedef struct {
    uint64_t text_length;     /* n, including the final '$' (IndexInterface::textLength) */
    const uint8_t* text;      /* .txt.bin payload: n bytes, upper-case ACGT + '$' */
    uint64_t counts[5];       /* cumulative char counts for $,A,C,G,T (indexinterface.cpp:143-150) */
    /* BWTRepresentation<5> of the text: .brt  (bwtrepr.h:113, bitvec.h:378) */
    uint64_t dollar_pos_fwd;
    const uint64_t* bv_fwd;   /* BitvecIntl<4>::bv,     4*ceil((n+1)/64) words (bitvec.h:262) */
    const uint64_t* cnt_fwd;  /* BitvecIntl<4>::counts, 8*ceil((n+1)/512) words (bitvec.h:273) */
    /* BWTRepresentation<5> of the reversed text: .rev.brt */
    uint64_t dollar_pos_rev;
    const uint64_t* bv_rev;
    const uint64_t* cnt_rev;
    /* sparse suffix array: .sa.bv.<s> (Bitvec, bitvec.h:176) and .sa.<s> (suffixArray.h:229) */
    const uint64_t* sa_bv;        /* ceil(n/64) words */
    const uint64_t* sa_bv_counts; /* (ceil(n/64)+7)/4 words, rank9 */
    const uint32_t* sa_samples;   /* samples in SA-row order */
    uint64_t n_samples;
    uint32_t sa_sparseness;       /* power of two (FMIndex ctor, fmindex.h:403) */
    /* sequence start offsets + final n-1 (.pos, indexinterface.cpp:162) — carried for callers */
    const uint32_t* seq_starts;
    uint32_t n_seqs;
    uint32_t kmer_size;           /* word size of the k-mer table (default 10, fmindex.h:404) */
    uint32_t in_text_switch;      /* in-text verification switch point (default 4, alignparameters.h:90) */
} cmb_index_desc;

int cmb_index_create(const cmb_index_desc* desc, int device, cmb_index** out);
void cmb_index_destroy(cmb_index* idx);
/* bytes of HBM held by the index */
uint64_t cmb_index_device_bytes(const cmb_index* idx);
/* --- replication of a built index on other GPUs (the index is read-only and replicated, reads are
 * sharded; the reference's worker threads share one index the same way, src/parallel.cpp:1143-1146).
 * The DEVICE layout is what travels: rank 0 describes its arrays (cmb_index_layout_of), every other rank creates
 * an index with empty arrays of the same sizes (cmb_index_create_empty), all ranks hand the raw device pointers
 * (cmb_index_device_arrays) to the collective library — one broadcast per array straight into the index, no host
 * round trip, no second re-layout. */
#define CMB_DEV_ARRAYS 6 /* rank blocks fwd (with the sampled-row bits of the sparse suffix array), rank blocks rev, SA samples, text codes, 2-bit text, k-mer table */
typedef struct {
    uint64_t text_length;
    uint64_t counts[5];
    uint64_t dollar_pos_fwd, dollar_pos_rev;
    uint64_t n_samples;
    uint32_t sa_sparseness, kmer_size, in_text_switch, n_seqs;
    uint64_t bytes[CMB_DEV_ARRAYS]; /* size of every device array (0: absent, e.g. no 2-bit text) */
} cmb_index_layout;
int cmb_index_layout_of(const cmb_index* idx, cmb_index_layout* out);
int cmb_index_seq_starts(const cmb_index* idx, uint32_t* out /* [n_seqs] */);
int cmb_index_create_empty(const cmb_index_layout* layout, const uint32_t* seq_starts, int device, cmb_index** out);
/* an index that holds ONLY the text (codes + 2-bit copy) and the sequence starts: what alignments (cmb_cigar_windows), trimming at
 * sequence ends (cmb_trim_occurrence) and SAM records need — for the b-move flavour, whose own index has no text.  No batch can be
 * created on it.  seq_starts / n_seqs as in cmb_index_desc. */
int cmb_index_create_text_only(const char* text, uint64_t n, const uint32_t* seq_starts, uint32_t n_seqs, int device, cmb_index** out);
int cmb_index_device_arrays(cmb_index* idx, void** ptrs /* [CMB_DEV_ARRAYS] */, uint64_t* bytes /* [CMB_DEV_ARRAYS] */);
/* after the arrays of an empty twin have been filled: the consistency probe cmb_index_create runs (every probed row of the
 * suffix array reaches a sampled row within sa_sparseness LF steps, FMIndex::findSA, src/fmindex/fmindex.cpp:53-60) */
int cmb_index_validate(cmb_index* idx);
/* copy the device k-mer table (4^kmer_size x {sa.b,sa.e,rev.b,rev.e}) to host: test hook for
 * IndexInterface::populateTable (indexinterface.cpp:294-335) */
int cmb_index_kmer_table(const cmb_index* idx, uint32_t* out /* 4 * 4^kmer_size */);

/* --- strategy ----------------------------------------------------------------------
 * Replaces Parameters::createStrategy (src/parameters/alignparameters.cpp:1313-1376). */
#define CMB_METRIC_HAMMING 0
#define CMB_METRIC_EDIT 1
#define CMB_PARTITION_UNIFORM 0
#define CMB_PARTITION_STATIC 1
#define CMB_PARTITION_DYNAMIC 2

/* built-in strategies, the `-S` option (alignparameters.cpp:1341-1372): "kuch1" (KucherovKPlus1,
 * searchstrategy.h:2829), "kuch2" (KucherovKPlus2, :2918), "kianfar" (OptimalKianfar, :3026), "01*0"
 * (O1StarSearchStrategy, :3115), "pigeon" (PigeonHoleSearchStrategy, :3221), "minU" (MinUSearchStrategy, :3284),
 * "columba" (the CLI default: DynamicColumbaStrategy, :3666 — minU schemes, their mirror images and the "middle"
 * schemes with dynamic selection; minU up to 7 errors, the greedy schemes for 8..13 errors: Hamming distance up to 13, edit
 * distance up to 13 as well — beyond 10 the reference switches to its 128-bit matrices, here: 64-bit words with narrower blocks);
 * plus "multiple_opt" (the schemes of search_schemes/multiple_opt with dynamic selection, as `-d` loads them) */
int cmb_strategy_create_named(const char* name, int metric, int partition, cmb_strategy** out);
/* scheme directories: mode CMB_DIR_CUSTOM = `-c <dir> -nD` (CustomSearchStrategy, searchstrategy.cpp:1990),
 * CMB_DIR_MULTIPLE = `-d <dir>` (MultipleSchemesStrategy::readSchemes, searchstrategy.h:2624),
 * CMB_DIR_CUSTOM_DYNAMIC = `-c <dir>` (DynamicCustomStrategy, searchstrategy.h:3744: every scheme and its mirror
 * image with dynamic selection) */
#define CMB_DIR_CUSTOM 0
#define CMB_DIR_MULTIPLE 1
#define CMB_DIR_CUSTOM_DYNAMIC 2
int cmb_strategy_create_from_dir(const char* dir, int mode, int metric, int partition,
                                 cmb_strategy** out);
/* generic: start empty, then add schemes (row-major nSearches x nParts arrays of pi, L, U) */
int cmb_strategy_create(int metric, int partition, uint32_t kmer_cutoff, cmb_strategy** out);
int cmb_strategy_add_scheme(cmb_strategy* s, uint32_t k, uint32_t n_searches, uint32_t n_parts,
                            const uint32_t* pi, const uint32_t* L, const uint32_t* U);
int cmb_strategy_set_partition_params(cmb_strategy* s, uint32_t k, const double* seeding,
                                      uint32_t n_seeding, const uint64_t* weights, uint32_t n_weights,
                                      const double* begins, uint32_t n_begins);
void cmb_strategy_destroy(cmb_strategy* s);
/* introspection used by parity tests (Search::makeSearch search.h:116-194, critical part :525) */
int cmb_strategy_describe(const cmb_strategy* s, uint32_t k, uint32_t* n_schemes, uint32_t* n_parts,
                          uint32_t* critical_parts /* [n_schemes] */, uint32_t cap);
/* the searches of alternative `scheme` for distance k as row-major n_searches x n_parts arrays (what
 * SearchStrategy::createSearches would hand out, searchstrategy.h:2005); cap = capacity of each array in elements.
 * Host-only: needs no GPU. */
int cmb_strategy_export_scheme(const cmb_strategy* s, uint32_t k, uint32_t scheme, uint32_t* pi, uint32_t* L,
                               uint32_t* U, uint32_t cap, uint32_t* n_searches, uint32_t* n_parts);
/* the partitioning parameters in force for distance k (getSeedingPositions / getWeights / getBegins,
 * searchstrategy.h:1825, :283, :245, or the strategy's overrides) and the k-mer cut-off (:206): seeding[n_parts-2],
 * weights[n_parts], begins[n_parts-1].  Host-only. */
int cmb_strategy_export_partition(const cmb_strategy* s, uint32_t k, double* seeding, uint64_t* weights,
                                  double* begins, uint32_t cap, uint32_t* kmer_cutoff);

/* --- matching ---------------------------------------------------------------------- */
typedef struct {
    uint32_t begin, end; /* [begin,end) in the concatenated text (TextOcc::range) */
    uint32_t distance;   /* edit or Hamming distance */
    uint32_t strand;     /* 0 forward, 1 reverse complement (definitions.h:125) */
} cmb_occ;

/* counters returned per batch (Counters, src/indexhelpers.h:1846-1941, + byte-model counters) */
enum {
    CMB_CNT_NODE = 0,               /* NODE_COUNTER */
    CMB_CNT_TOTAL_REPORTED,         /* TOTAL_REPORTED_POSITIONS */
    CMB_CNT_IN_TEXT_STARTED,
    CMB_CNT_ABORTED_IN_TEXT,
    CMB_CNT_CIGARS_IN_TEXT,         /* traced-back in-text hits */
    CMB_CNT_IMMEDIATE_SWITCH,
    CMB_CNT_SEARCH_STARTED,
    CMB_CNT_EXPANSIONS,             /* E of SURVEY.md §8d: extend calls at one parent */
    CMB_CNT_LF_STEPS,               /* L */
    CMB_CNT_LOCATED_ROWS,           /* R */
    CMB_CNT_TEXT_BYTES,             /* T */
    CMB_CNT_MATRIX_ROWS,
    CMB_CNT_DFS_EXPANSIONS,         /* the part of E performed by the DFS kernel (rest: prologue kernel) */
    CMB_CNT_TABLE_ROWS,             /* b-move backend only: 16-byte move-table rows fetched by run walks, LF and fast-forwards */
    CMB_CNT_DFS_TABLE_ROWS,         /* ... the part of them fetched by the frontier search (rest: prologue kernels) */
    CMB_CNT_MAX
};

/* One-shot host-buffer entry: the body of processChunk's loop (parallel.cpp:67-78) for a whole
 * chunk.  seqs: concatenated read characters (any case; non-ACGT treated as N, reads.h:43-58),
 * offs[nReads+1].  out/outOffs are caller-allocated; on CMB_ERR_OVERFLOW *needed holds the required
 * number of cmb_occ and nothing is written.
 *
 * Per read the list is what SearchStrategy::matchApprox leaves in `matches` before generateOutputSingleEnd
 * (searchstrategy.cpp:529), with three deviations in places where the reference's own result is not a function of its
 * input (a maintainer diffing against `columba` output will meet exactly these and no others):
 *  1. strand label of a tie: when the same (begin, end, distance) is found on both strands, which label survives
 *     Occurrences::eraseDoublesAndSortText is decided by an unstable std::sort (indexhelpers.h:2148-2156, TextOcc::==
 *     ignores the strand, :811); this library always keeps strand 0.
 *  2. order at max_distance = 0: the reference leaves exact matches in suffix-array order, forward strand first
 *     (searchstrategy.cpp:499-510); this library returns them sorted by (begin, strand).  The SET is identical; which of
 *     several exact matches becomes the primary SAM record may therefore differ.
 *  3. counters only: in-index occurrences that are equal under FMOcc::== but separated by the unstable sort of
 *     Occurrences::eraseDoublesFM (indexhelpers.h:2135-2146) are located twice by the reference; this library removes
 *     every duplicate, so CMB_CNT_TOTAL_REPORTED, CMB_CNT_LF_STEPS and CMB_CNT_LOCATED_ROWS can be lower than the
 *     reference's by the repeated work (well below 1 %); the occurrence lists are unaffected. */
int cmb_match_batch(cmb_index* idx, const cmb_strategy* st, uint32_t max_distance, const char* seqs,
                    const uint64_t* offs, uint32_t n_reads, cmb_occ* out, uint64_t out_cap,
                    uint64_t* out_offs /* [n_reads+1] */, uint64_t* counters /* [CMB_CNT_MAX] or NULL */,
                    uint64_t* needed);

/* Resident-batch entries (what bench.py times: reads already in HBM when the timed region starts).
 * A batch of 2 M reads or more is held as 2-3 sub-batches that cmb_batch_run processes concurrently (own HIP
 * stream and host thread each; environment CMB_SUBBATCHES=n overrides the number, CMB_SERIAL_SUBBATCHES runs
 * them one after the other); results and counters are those of the whole batch, in read order.
 * Diagnostic knobs, read per run, none of which changes a result: CMB_MATRIX_WIDE=1 (in-text matrices on 64-bit
 * words also for k <= 4), CMB_TRACE_WIDE=1 (8-byte traceback rows also for k <= 4; that kernel checks the rules the
 * 4-byte rows rely on), CMB_STAGE_BLOCKS=n (32-row blocks per verification launch), CMB_MATRIX64=1 (the frontier's in-index
 * matrix on the reference's 64-bit words instead of the 32-bit words with 8-row blocks it runs on up to 6 errors; a batch one of
 * whose phases does not fit the small matrix switches by itself), CMB_VERBOSE=1.
 * cmb_batch_timings: device time per kernel group of the last run (HIP events on the batch's streams; summed
 * over the sub-batches). */
int cmb_batch_create(cmb_index* idx, const cmb_strategy* st, uint32_t max_distance, const char* seqs,
                     const uint64_t* offs, uint32_t n_reads, cmb_batch** out);
int cmb_batch_run(cmb_batch* b);  /* enqueue + wait: the whole hot path on the batch's stream */
/* Streaming (the reference's reader thread hands chunk after chunk to the workers, src/fastq.cpp:283-395): register the
 * NEXT chunk of reads — as many reads as the batch was created with, none longer.  The next cmb_batch_run copies it to
 * the device on a stream of its own WHILE it matches the chunk the batch holds, and the run after that matches it,
 * with all scratch memory of the batch reused:
 *     create(A); stage(B); run() -> A   [B travels]; stage(C); run() -> B   [C travels]; run() -> C
 * seqs must stay valid until the run that uploads it has returned; page-locked host memory makes the copy asynchronous. */
int cmb_batch_stage_reads(cmb_batch* b, const char* seqs, const uint64_t* offs, uint32_t n_reads);
int cmb_batch_result_size(const cmb_batch* b, uint64_t* n_occ);
int cmb_batch_results(const cmb_batch* b, cmb_occ* out, uint64_t out_cap, uint64_t* out_offs,
                      uint64_t* counters);
/* Alignments of the final occurrences: before cmb_batch_run, ask for them; after it,
 * cmb_batch_alignments gives, parallel to the cmb_occ list, for every occurrence
 *  - its CIGAR as run-length operations (IBitParallelED::findCIGAR, bitparallelmatrix.h:460-527 — for in-text
 *    occurrences the same string as their verification's traceBack, :531-586): cigar_len values of
 *    (length << 2 | op), op 0 = M, 1 = I, 2 = D, from the begin of the alignment, at cigar_ops[cigar_off ...];
 *  - the reference sequence it lies in and its 0-based begin inside it (IndexInterface::findSeqName,
 *    indexinterface.cpp:799-832); spans != 0: the occurrence runs past the end of that sequence (the reference then
 *    trims and re-verifies, or drops it: :833-899 — the C++ adapter does that with cmb_verify_window). */
typedef struct {
    uint32_t seq_id, seq_begin;
    uint64_t cigar_off;
    uint16_t cigar_len, spans;
    uint32_t reserved;
} cmb_aln;
int cmb_batch_want_alignments(cmb_batch* b, int on);
int cmb_batch_alignments(const cmb_batch* b, cmb_aln* out, uint64_t cap, uint16_t* cigar_ops, uint64_t ops_cap,
                         uint64_t* n_ops);
/* every strand filtered by itself instead of the two strands of a read together (what one stratum of BEST mode needs:
 * mapRead works on one strand, src/searchstrategy.h:490-523); the result list of a read then holds the forward
 * strand's occurrences followed by the reverse-complement strand's */
int cmb_batch_filter_per_strand(cmb_batch* b, int on);
/* Reads not longer than the number of parts of the search scheme — and every read under the one-part strategy "naive" — are not
 * matched with a search scheme: the reference falls back to naive backtracking over the whole pattern
 * (SearchStrategy::partition / matchWithSearches, searchstrategy.cpp:148-152, :442-459; IndexInterface::approxMatchesNaive[Hamming],
 * indexinterface.cpp:1055-1209).  The device does the same inside the chunk (dev_bfs_naive.hpp; move_search.hpp: k_mvs_naive for the
 * b-move index), including that path's own filter pass per strand; no read of a valid chunk fails the run.
 * cmb_batch_read_status marks the reads that took that path (status[i] & CMB_READ_NAIVE_FALLBACK) — information only; status may be
 * NULL (count only).  cmb_batch_allow_unsupported is kept for callers written against earlier versions and has no effect. */
#define CMB_READ_NAIVE_FALLBACK 1
int cmb_batch_allow_unsupported(cmb_batch* b, int on);
int cmb_batch_read_status(const cmb_batch* b, uint8_t* status /* [n_reads] */, uint32_t* n_flagged);
/* per-kernel device time of the last cmb_batch_run, measured with hipEvents on the batch's own
 * stream.  names: NUL-separated list; ms[n]. Returns number of kernels. */
int cmb_batch_timings(const cmb_batch* b, const char** names, float* ms, uint32_t cap);
void cmb_batch_destroy(cmb_batch* b);

/* --- BEST (+x strata) mode: the reference's default mapping mode (`-a best`, SearchStrategy::matchApproxBestPlusX,
 * src/searchstrategy.cpp:714-746 over findBestAlignments :623-712) for a whole chunk of single-end reads.
 * Every read is walked through its strata — exact matches, then 1, 3, 5, 9, 13 errors up to the cut-off
 * min(13, largest distance the strategy and the device support, len * (100 - min_identity) / 100) — until one holds an
 * alignment that lies inside one reference sequence; the alignments of the best stratum and of the x strata above
 * it are reported, per stratum the forward strand's (ordered by sequence and begin) before the reverse complement's
 * (combineOccVectors :573-620).  A stratum is one device batch over the reads still looking at that distance.
 * best[i] = best distance of read i (0xFFFFFFFF: unmapped), n_hits[i] = occurrences at that distance. */
typedef struct cmb_best cmb_best;
int cmb_match_best(cmb_index* idx, const cmb_strategy* st, uint32_t x, uint32_t min_identity, const char* seqs,
                   const uint64_t* offs, uint32_t n_reads, cmb_best** out);
int cmb_best_sizes(const cmb_best* r, uint64_t* n_occ, uint64_t* n_ops);
int cmb_best_results(const cmb_best* r, cmb_occ* occ, cmb_aln* aln, uint64_t cap, uint16_t* cigar_ops, uint64_t ops_cap,
                     uint64_t* offs /* [n_reads+1] */, uint32_t* best, uint32_t* n_hits, uint64_t* counters);
void cmb_best_destroy(cmb_best* r);

/* --- output records (host-only; no GPU needed) -----------------------------------------
 * SAM lines of single-end reads as the reference formats them (TextOcc::generateSAMSingleEnd / ...XA /
 * createUnmappedSAMOccurrenceSE, src/indexhelpers.cpp:56-120, :177-200; flags, mapping quality and XA entries:
 * src/indexhelpers.h:321-331, :378-388, :416-421).  Every function returns the length of the line (without the
 * terminating NUL) and writes it, NUL-terminated, if cap is larger than that. */
typedef struct {
    const char* seq_name;       /* reference sequence the occurrence lies in */
    uint32_t pos0;              /* 0-based begin inside that sequence (cmb_aln.seq_begin) */
    uint32_t distance;
    uint32_t revcomp;           /* cmb_occ.strand */
    const uint16_t* cigar_ops;  /* run-length operations of cmb_batch_alignments */
    uint32_t n_ops;
} cmb_sam_hit;
int64_t cmb_sam_se(const char* read_id, const cmb_sam_hit* hit, int primary, uint32_t n_hits, uint32_t min_score,
                   const char* print_seq, const char* print_qual, char* out, uint64_t cap);
/* the first hit as the record, the others in its XA tag (generateSE_SAM_XATag, src/searchstrategy.h:1612-1622) */
int64_t cmb_sam_se_xa(const char* read_id, const cmb_sam_hit* hits, uint32_t n, uint32_t n_hits, const char* print_seq,
                      const char* print_qual, char* out, uint64_t cap);
int64_t cmb_sam_unmapped_se(const char* read_id, const char* seq, const char* qual, char* out, uint64_t cap);
/* --- paired-end records (the records only — the pairing of the mates' occurrences,
 * src/searchstrategy.cpp:746-1820, stays with the caller).  mate = NULL: the mate is not mapped.  print_seq / print_qual as for
 * cmb_sam_se (the read as it aligns; an empty quality prints as "*"). */
/* TextOcc::generateSAMPairedEnd (indexhelpers.cpp:114-166) with getFlagsPE and getMapQPairedEnd (indexhelpers.h:340-371, :396-410) */
int64_t cmb_sam_pe(const char* read_id, const cmb_sam_hit* hit, int first_in_pair, const cmb_sam_hit* mate, uint32_t n_pairs,
                   uint32_t min_score, uint32_t frag_size, int discordant, int primary, const char* print_seq, const char* print_qual,
                   char* out, uint64_t cap);
/* TextOcc::generateSAMUnpaired (indexhelpers.cpp:215-262): an occurrence of a read whose pair could not be formed */
int64_t cmb_sam_unpaired(const char* read_id, const cmb_sam_hit* hit, int first_in_pair, uint32_t n_hits, uint32_t min_score, int primary,
                         const char* print_seq, const char* print_qual, char* out, uint64_t cap);
/* TextOcc::createUnmappedSAMOccurrencePE (indexhelpers.cpp:186-213) */
int64_t cmb_sam_unmapped_pe(const char* read_id, const char* seq, const char* qual, int first_in_pair, int mate_mapped, int mate_revcomp,
                            char* out, uint64_t cap);
/* Pairing of the single-end occurrences of two mates in ALL mode and the SAM text of the pair:
 * SearchStrategy::pairSingleEndedMatchesAll (src/searchstrategy.cpp:1345-1399: concordant pairs by orientation and fragment size,
 * otherwise discordant pairs / unpaired records / one or both mates unmapped) with generateSAMPairedEnd (:1904-1970), lines in the
 * order OutputWriter::writeChunks prints them (src/fastq.cpp:662-702).  The occurrences are the mates' ALL-mode results with their
 * sequence assignment (cmb_batch_alignments). */
#define CMB_ORIENTATION_FR 0 /* definitions.h:122 */
#define CMB_ORIENTATION_RF 1
#define CMB_ORIENTATION_FF 2
typedef struct {
    uint32_t seq_id;           /* assigned sequence; 0xFFFFFFFF: none (findSeqName returned NOT_FOUND) */
    uint32_t begin, end;       /* inside that sequence, after trimming */
    uint32_t index_begin;      /* begin in the concatenated text (TextOcc::getIndexBegin) */
    uint32_t distance, strand; /* strand 0 forward, 1 reverse complement */
    const uint16_t* cigar_ops; /* run-length operations as in cmb_sam_hit */
    uint32_t n_ops;
} cmb_pair_occ;
typedef struct {
    const char* id;      /* cleaned identifier (cmb_read_prepare) */
    const char* seq;     /* cleaned read */
    const char* revcomp; /* its reverse complement */
    const char* qual;    /* may be NULL */
    const char* revqual; /* reversed quality, may be NULL */
    const cmb_pair_occ* occ;
    uint32_t n_occ;
} cmb_pair_read;
typedef struct {
    uint32_t orientation;       /* CMB_ORIENTATION_* */
    uint32_t max_frag, min_frag;
    int discordant_allowed;     /* -nD not given */
    int unmapped_records;       /* records for unmapped reads (SearchStrategy::unmappedSAM) */
} cmb_pair_params;
/* returns the length of the text (written if cap is larger); n_pairs_out (may be NULL): TOTAL_UNIQUE_PAIRS of the pair */
int64_t cmb_pair_sam(const cmb_pair_params* params, const cmb_pair_read* read1, const cmb_pair_read* read2, const char* const* seq_names,
                     char* out, uint64_t cap, uint32_t* n_pairs_out);
/* --- paired-end reads in BEST (+x strata) mode: SearchStrategy::matchApproxPairedEndBestPlusX (src/searchstrategy.cpp:1091-1179) with
 * processCombFR / RF / FF (:936-1062), processComb (:834-912), pairOccurrencesForBestMapping (:1743-1815) and, without a concordant
 * pair, pairDiscordantlyBest (:1664-1741); records by generateSAMPairedEnd (:1904-1970), lines in OutputWriter::writeChunks' order.
 * The reference walks one pair through its strata and calls mapRead (searchstrategy.h:490-519: the ALL-mode search of ONE strand of one
 * mate at one distance) for every stratum it has not looked at yet; which ones those are depends on what the earlier ones held.  Here a
 * chunk of pairs is walked together: cmb_pair_best_advance runs every unfinished pair as far as the lists it holds allow and reports
 * the list each of them waits for; the caller produces the lists — one device batch per distance over the reads asked for, each strand
 * filtered by itself (cmb_batch_filter_per_strand) with alignments (cmb_batch_want_alignments) — hands them in with
 * cmb_pair_best_supply and advances again, until no pair waits (n = 0).  Host code; the device is touched only to trim an occurrence that
 * runs over the end of its sequence (cmb_trim_occurrence on text_index, or the caller's hook). */
typedef struct cmb_pair_best cmb_pair_best;
typedef struct {
    uint32_t pair, mate;   /* mate 0: read 1, 1: read 2 */
    uint32_t strand;       /* 0 forward, 1 reverse complement */
    uint32_t max_distance; /* mapRead's maxED */
} cmb_pair_request;
/* findSeqName for an occurrence that runs over the end of its sequence: 1 = found with trimming (occ / aln / operations updated),
 * 0 = not found, < 0 = failure.  Same contract as cmb_trim_occurrence, which is used when no hook is set. */
typedef int (*cmb_pair_trim_fn)(void* user, uint32_t pair, uint32_t mate, uint32_t strand, uint32_t largest_stratum, cmb_occ* occ, cmb_aln* aln,
                                uint16_t* cigar_ops, uint32_t ops_cap, uint32_t* n_ops);
/* reads1[i] / reads2[i]: the mates of pair i as cmb_read_prepare leaves them (occ / n_occ unused).  max_supported: the largest distance
 * the strategy has schemes for and the device runs (getMaxSupportedDistanceForBestMapping); a read's cut-off is
 * min(max_supported, len * (100 - min_identity) / 100) (getMaxED). */
int cmb_pair_best_create(const cmb_pair_params* params, uint32_t x, uint32_t min_identity, uint32_t max_supported, int metric,
                         cmb_index* text_index, uint32_t n_pairs, const cmb_pair_read* reads1, const cmb_pair_read* reads2,
                         cmb_pair_best** out);
int cmb_pair_best_set_trim(cmb_pair_best* b, cmb_pair_trim_fn fn, void* user);
int cmb_pair_best_cutoff(const cmb_pair_best* b, uint32_t pair, uint32_t mate, uint32_t* cut_off);
/* pairSingleEndedMatchesBest (src/searchstrategy.h:1454-1462) over addSingleEndedForBest (src/searchstrategy.cpp:1064-1089), x = 0: the pair
 * starts from the mates' single-end BEST results (what the parameter-inference phase has computed, src/parallel.cpp:790-810) — occurrences
 * with their sequence assigned (aln[j].spans != 1) and their CIGAR; every stratum of read 1 then counts as looked at, those of read 2 if
 * read2_done.  Before the first cmb_pair_best_advance of that pair. */
int cmb_pair_best_seed(cmb_pair_best* b, uint32_t pair, const cmb_occ* occ1, const cmb_aln* aln1, uint64_t n1, const uint16_t* cigar_ops1,
                       const cmb_occ* occ2, const cmb_aln* aln2, uint64_t n2, const uint16_t* cigar_ops2, int read2_done);
/* at most one request per unfinished pair; CMB_ERR_OVERFLOW (with *n = the number wanted) if cap is too small — nothing is lost, call again */
int cmb_pair_best_advance(cmb_pair_best* b, cmb_pair_request* requests, uint64_t cap, uint64_t* n);
/* the ALL-mode result of that mate at that distance (occurrences of the other strand in the list are skipped, so one device result
 * serves both strands with two calls); aln[j].cigar_off indexes cigar_ops */
int cmb_pair_best_supply(cmb_pair_best* b, uint32_t pair, uint32_t mate, uint32_t strand, uint32_t max_distance, const cmb_occ* occ,
                         const cmb_aln* aln, uint64_t n_occ, const uint16_t* cigar_ops);
/* the records of a finished pair; n_pairs_out (may be NULL): TOTAL_UNIQUE_PAIRS of the pair */
int64_t cmb_pair_best_sam(const cmb_pair_best* b, uint32_t pair, const char* const* seq_names, char* out, uint64_t cap, uint32_t* n_pairs_out);
void cmb_pair_best_destroy(cmb_pair_best* b);
/* Inference of orientation and insert-size bounds from pairs whose two mates map unambiguously (src/parallel.cpp:329-360
 * addFragmentAndOrientation, :402-466 inferPairedEndParameters; the caller selects the pairs as :236-262 / :700-727 do: exactly
 * one occurrence per mate).  Coordinates of the two occurrences as reported; n == 0: nothing inferred (inferred = 0). */
typedef struct {
    uint32_t begin1, end1, strand1;
    uint32_t begin2, end2, strand2;
} cmb_pair_sample;
typedef struct {
    uint64_t n_pairs;
    uint32_t inferred;     /* 0: no sample, the caller keeps its defaults */
    uint32_t orientation;  /* CMB_ORIENTATION_* */
    uint32_t max_insert, min_insert;
    float mean_insert, stddev_insert;
} cmb_pair_inferred;
int cmb_pair_infer(const cmb_pair_sample* samples, uint64_t n, cmb_pair_inferred* out);
/* SAM text of a whole chunk matched in ALL mode (SearchStrategy::generateOutputSingleEnd, src/searchstrategy.cpp:1824-1902:
 * sequence assignment incl. trimming at sequence ends, primary = first occurrence of minimal distance, the others as
 * secondary lines or, with xa_tag, in the primary's XA tag; unmapped_records: a flag-4 record for reads without any).
 * Needs cmb_batch_want_alignments before cmb_batch_run; seqs = the read characters given to cmb_batch_create,
 * read_ids / quals = per read the FASTQ identifier line and quality (quals or an entry may be NULL: "*"),
 * seq_names = names of the reference sequences.  Returns the length of the text (written if cap is larger). */
int64_t cmb_batch_sam(const cmb_batch* b, const char* seqs, const char* const* read_ids, const char* const* quals,
                      const char* const* seq_names, int unmapped_records, int xa_tag, char* out, uint64_t cap);
/* the same for occurrences and alignments the caller holds — cmb_batch_results + cmb_batch_alignments, or cmb_move_batch_results
 * (begin / end narrowed to 32 bits) + cmb_move_batch_alignments with the text-only index of cmb_move_text_index: occ_offs[n_reads + 1],
 * aln[i].cigar_off / cigar_len into cigar_ops.  `idx` serves the trimming of occurrences that run over a sequence end. */
int64_t cmb_sam_chunk(cmb_index* idx, uint32_t max_distance, int metric, const char* seqs, const uint64_t* offs, uint32_t n_reads,
                      const char* const* read_ids, const char* const* quals, const char* const* seq_names, const cmb_occ* occ,
                      const uint64_t* occ_offs, const cmb_aln* aln, const uint16_t* cigar_ops, int unmapped_records, int xa_tag, char* out,
                      uint64_t cap);
/* Read::cleanUpRecord + ReadBundle (src/reads.h:43-58, :97-160): identifier without its first character and without
 * anything from the first space on; upper-case sequence with every non-ACGT character replaced by N; its reverse
 * complement; the reversed quality string.  Output buffers hold strlen(input) + 1 bytes each (any may be NULL). */
int cmb_read_prepare(const char* id, const char* seq, const char* qual, char* id_out, char* seq_out, char* revcomp_out,
                     char* revqual_out);

/* --- fine-grained hooks (parity tests + roofline microbenchmark) ------------------- */
/* BitvecIntl<4>::rank(c,p) (bitvec.h:356), rev = 0 forward BWT / 1 reverse BWT */
int cmb_rank_batch(cmb_index* idx, int rev, const uint32_t* c, const uint64_t* p, uint64_t n,
                   uint64_t* out);
/* all four children of n parents (IndexInterface::extendFMPos, indexinterface.cpp:675-697 over
 * FMIndex::findRangesWithExtraChar{Forward,Backward,BackwardUniDirectional}, fmindex.cpp:137-243).
 * mode 0 forward, 1 backward, 2 uni-directional backward.  in: n x {sa.b,sa.e,rev.b,rev.e};
 * out: n x 4 x 4; ok: n x 4.  Host-buffer variant. */
int cmb_extend_batch(cmb_index* idx, int mode, const uint32_t* in, uint64_t n, uint32_t* out,
                     uint8_t* ok);
/* Device-resident variant for the roofline microbenchmark: d_in/d_out/d_ok are device pointers;
 * runs `iters` launches and returns the average kernel time in ms (hipEvents on the launch stream). */
int cmb_extend_bench(cmb_index* idx, int mode, const void* d_in, uint64_t n, void* d_out, void* d_ok,
                     uint32_t iters, float* avg_ms);
/* FMIndex::findSA (fmindex.cpp:53-60) */
int cmb_locate_batch(cmb_index* idx, const uint32_t* rows, uint64_t n, uint32_t* out,
                     uint64_t* lf_steps);
/* FMIndex::inTextVerification (fmindex.cpp:267-310) for one pattern and n start positions */
int cmb_verify_batch(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* starts,
                     uint64_t n, uint32_t max_ed, uint32_t min_ed, int fixed_start, cmb_occ* out,
                     uint64_t out_cap, uint64_t* n_out, uint64_t* counters);

/* FMIndex::inTextVerificationOneString (fmindex.cpp:312-342): one text window [start, end), fixed start */
int cmb_verify_window(cmb_index* idx, const char* pattern, uint32_t plen, uint32_t start, uint32_t end,
                      uint32_t max_ed, uint32_t min_ed, cmb_occ* out, uint64_t out_cap, uint64_t* n_out,
                      uint64_t* counters);
/* IndexInterface::findSeqName (indexinterface.cpp:833-899) for one occurrence that runs over the end of its sequence (cmb_aln.spans
 * == 1 after cmb_batch_alignments): trimmed to the sequence it mostly lies in and verified again inside that window, as the
 * single-end record path (cmb_batch_sam) does.  pattern: the cleaned read of the occurrence's strand (its reverse complement for
 * strand 1); largest_stratum: the distance the chunk was matched at.  *found = 1 (FOUND_WITH_TRIMMING): occ (begin, end, distance),
 * aln (seq_id, seq_begin, spans = 2, cigar_len) and cigar_ops (*n_ops of them, at most 2 * largest_stratum + 3) are the trimmed
 * occurrence's; *found = 0 (NOT_FOUND): the occurrence is to be dropped.  Edit distance only (Hamming: always NOT_FOUND). */
int cmb_trim_occurrence(cmb_index* idx, const char* pattern, uint32_t plen, uint32_t largest_stratum, int metric, cmb_occ* occ,
                        cmb_aln* aln, uint16_t* cigar_ops, uint32_t ops_cap, uint32_t* n_ops, int* found);
/* IBitParallelED::findCIGAR (bitparallelmatrix.h:460-527) of one pattern against n text windows [begin, end) with given
 * distances: ops_out holds n x stride run-length operations (length << 2 | op, from the begin of the alignment),
 * n_ops_out[i] of them for window i; stride >= 2 * largest distance + 3 */
int cmb_cigar_windows(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* begins, const uint32_t* ends,
                      const uint32_t* distances, uint64_t n, uint16_t* ops_out, uint32_t stride, uint32_t* n_ops_out);
/* the same through the PRODUCTION edit-distance path (keys, de-duplication of identical candidates with counters
 * scaled by their multiplicity, staged matrix blocks, traceback): what cmb_batch_run does with the in-text candidates
 * of a search; cmb_verify_batch runs the one-candidate-per-lane kernel used for Hamming / exact candidates */
int cmb_verify_batch_staged(cmb_index* idx, const char* pattern, uint32_t plen, const uint32_t* starts,
                            uint64_t n, uint32_t max_ed, uint32_t min_ed, int fixed_start, cmb_occ* out,
                            uint64_t out_cap, uint64_t* n_out, uint64_t* counters);

/* --- b-move: the run-length compressed backend ------------------------------------------
 * First stage: the index structures in HBM and the two index operations of that backend as batch hooks — character
 * extension with toehold maintenance (BMove::findRangesWithExtraChar{Forward,Backward,BackwardUniDirectional},
 * src/bmove/bmove.cpp:328-478, over MoveLFReprBP, src/bmove/moverepr.cpp) and locate (BMove::collectTextPositions,
 * bmove.cpp:500-560) — and, on top of them, exact matching (cmb_move_match_exact) and the approximate search
 * (cmb_move_match_batch) end to end.
 * 64-bit positions throughout (the RUN_LENGTH_COMPRESSION flavour builds with 64-bit length_t, CMakeLists.txt:41-63);
 * texts below 2^40 characters. */
typedef struct cmb_move_index cmb_move_index;

typedef struct {
    /* contents of <base>.LFBP and <base>.rev.LFBP exactly as MoveLFReprBP::write leaves them (moverepr.cpp:145-168:
     * textSize, nrOfRuns, zeroCharPos as length_t, then nrOfRuns + 1 bit-packed rows) */
    const uint8_t* lfbp;
    uint64_t lfbp_bytes;
    const uint8_t* rev_lfbp;
    uint64_t rev_lfbp_bytes;
    uint32_t length_bits; /* width of length_t in those files: 64 (the flavour's default) or 32 */
    /* suffix array samples at the run boundaries (buildindex.cpp:942-953; BMove::samplesFirst / samplesLast and the
     * reverse pair, <base>.smpf / .smpl / .rev.smpf / .rev.smpl as plain 64-bit values), nrOfRuns each */
    const uint64_t* samples_first;
    const uint64_t* samples_last;
    const uint64_t* rev_samples_first;
    const uint64_t* rev_samples_last;
    /* locate (all NULL: an index for extension only).  pred_first / pred_last: the marked positions of BMove::predFirst /
     * predLast in increasing order (buildindex.cpp:990-1013), nrOfRuns each; first_to_run / last_to_run
     * (buildindex.cpp:1044-1066).  The PLCP array (bmove/plcp.h) in run-length form: plcp_pos = the positions q, in
     * increasing order and starting with 0, where PLCP[q] != PLCP[q - 1] - 1, and plcp_sum[j] = PLCP[q_j] + q_j;
     * PLCP[i] = plcp_sum[j] - i for the last q_j <= i. */
    const uint64_t* pred_first;
    const uint64_t* first_to_run;
    const uint64_t* pred_last;
    const uint64_t* last_to_run;
    const uint64_t* plcp_pos;
    const uint64_t* plcp_sum;
    uint64_t n_plcp;
} cmb_move_desc;

/* SARangePair of the RLC flavour (indexhelpers.h:137-255, :1040-1260): both ranges with their run indices, the toehold */
typedef struct {
    uint64_t begin, end, begin_run, end_run;                 /* range over the suffix array of the text; end_run inclusive */
    uint64_t rev_begin, rev_end, rev_begin_run, rev_end_run; /* ... of the reversed text */
    uint64_t toehold;                                        /* a text position of one occurrence */
    uint32_t original_depth;
    uint8_t runs_valid, rev_runs_valid; /* run indices exact (otherwise: enclosing, searched on use) */
    uint8_t toehold_represents_end;
    uint8_t reserved;
} cmb_move_range;

/* copies into HBM, converts the rows to the device layout and CHECKS the tables (order of the runs, LF targets,
 * terminating row; increasing locate positions): CMB_ERR_INVALID for files that are not a consistent move table */
int cmb_move_create(const cmb_move_desc* desc, int device, cmb_move_index** out);
void cmb_move_destroy(cmb_move_index* idx);
uint64_t cmb_move_device_bytes(const cmb_move_index* idx);
int cmb_move_info(const cmb_move_index* idx, uint64_t* text_length, uint64_t* runs, uint64_t* rev_runs);
/* BMove::getCompleteRange (bmove.h:369-373) */
int cmb_move_complete_range(const cmb_move_index* idx, cmb_move_range* out);
/* rows first .. first + count - 1 of a table as {head, inputStartPos, outputStartPos, outputStartRun} (4 x count values);
 * row nrOfRuns is the terminating row */
int cmb_move_rows(const cmb_move_index* idx, int rev, uint64_t first, uint64_t count, uint64_t* out);
/* all four children (A, C, G, T) of n parents: children n x 4, ok n x 4.  mode 0 forward, 1 backward,
 * 2 uni-directional backward (as cmb_extend_batch).  Host buffers. */
int cmb_move_extend_batch(const cmb_move_index* idx, int mode, const cmb_move_range* parents, uint64_t n, cmb_move_range* children,
                          uint8_t* ok);
/* device-resident variant for the microbenchmark: average kernel time of `iters` launches (HIP events on the launch stream) */
int cmb_move_extend_bench(const cmb_move_index* idx, int mode, const void* d_parents, uint64_t n, void* d_children, void* d_ok,
                          uint32_t iters, float* avg_ms);
/* IndexInterface::populateTable of the RLC flavour (indexinterface.cpp:294-335): the ranges of all 4^word_size k-mers (key: two
 * bits per character, the first character in the highest bits; SARangePair() for k-mers that do not occur) */
int cmb_move_kmer_table(const cmb_move_index* idx, uint32_t word_size, cmb_move_range* out);
/* text positions of n ranges in the reference's order (the toehold's occurrence, its phi chain, its phi^-1 chain):
 * range i writes end - begin values at positions[offsets[i]]; offsets has n + 1 entries */
int cmb_move_locate_batch(const cmb_move_index* idx, const cmb_move_range* ranges, uint64_t n, const uint64_t* offsets, uint64_t* positions);

/* replication on other GPUs, the DEVICE layout (as cmb_index_layout_of / cmb_index_create_empty / cmb_index_device_arrays for the
 * FM-index): rank 0 describes its arrays, every other rank creates an index with empty arrays of those sizes, a collective per
 * array fills them, cmb_move_validate repeats the consistency checks of cmb_move_create on what arrived. */
#define CMB_MOVE_DEV_ARRAYS 15 /* per direction: rows, samplesFirst, samplesLast; predFirst, predLast, PLCP positions (values + directory each); firstToRun, lastToRun, PLCP sums */
typedef struct {
    uint64_t text_length;
    uint64_t runs[2];          /* text, reversed text */
    uint64_t zero_char_pos[2];
    uint64_t set_count[3];     /* predFirst, predLast, PLCP run starts */
    uint32_t set_shift[3];     /* bucket width (log2) of their directories */
    uint32_t has_locate;
    uint64_t bytes[CMB_MOVE_DEV_ARRAYS]; /* 0: absent (an index without the locate arrays) */
} cmb_move_layout;
int cmb_move_layout_of(const cmb_move_index* idx, cmb_move_layout* out);
int cmb_move_create_empty(const cmb_move_layout* layout, int device, cmb_move_index** out);
int cmb_move_device_arrays(cmb_move_index* idx, void** ptrs /* [CMB_MOVE_DEV_ARRAYS] */, uint64_t* bytes /* [CMB_MOVE_DEV_ARRAYS] */);
int cmb_move_validate(cmb_move_index* idx);

/* k = 0 on the b-move index, end to end: SearchStrategy::matchApproxAllMap with maxED = 0 (searchstrategy.cpp:499-510) =
 * IndexInterface::exactMatchesOutput (indexinterface.cpp:947-1014, RLC branch) of every read and of its reverse complement.
 * Reads as for cmb_match_batch (characters + n_reads + 1 offsets; lower case accepted, a read with anything outside ACGT
 * has no exact occurrence).  Per read the occurrences of the forward strand, then those of the reverse complement, each in
 * the order of BMove::collectTextPositions — the reference's order before its output stage sorts.  occ_offsets (n_reads + 1,
 * may be NULL); counters (may be NULL): [0] NODE_COUNTER, [1] TOTAL_REPORTED_POSITIONS.  CMB_ERR_OVERFLOW with *n_occ =
 * the number needed if occ_cap is too small (nothing is truncated). */
typedef struct {
    uint64_t begin, end; /* text coordinates, end exclusive */
    uint32_t distance;   /* 0 */
    uint32_t strand;     /* 0 forward, 1 reverse complement */
} cmb_move_occ;
int cmb_move_match_exact(const cmb_move_index* idx, const char* reads, const uint64_t* read_offsets, uint64_t n_reads,
                         cmb_move_occ* occ_out, uint64_t occ_cap, uint64_t* occ_offsets, uint64_t* n_occ, uint64_t* counters);
/* The approximate search on the b-move index: SearchStrategy::matchApprox in ALL mode (searchstrategy.cpp:495-535) as the
 * RUN_LENGTH_COMPRESSION flavour compiles it — the same search schemes, partitioning and bit-parallel matrix over ranges that
 * carry run indices and a toehold (indexhelpers.h:137-255, :1040-1260), no in-text verification (switch point 0: bmove.cpp:195-197,
 * indexinterface.cpp:345-348, :516-524, :1306-1325, searchstrategy.cpp:461-477), in-index occurrences located by the phi / phi^-1
 * chains of their toehold (bmove.cpp:500-560), then getUniqueTextOccurrences (indexinterface.cpp:1373-1491).  Edit and Hamming distance,
 * max_distance 1 .. 13 (the reference's MAX_K) for the strategies that have schemes for it (0: the exact path above); kmer_size: word size of the k-mer
 * table used for seeding (populateTable, indexinterface.cpp:294-335; built on the device on first use).  Results: per read the list
 * cmb_match_batch would return on the same text, in 64-bit coordinates.  counters[CMB_CNT_MAX]: NODE_COUNTER, SEARCH_STARTED,
 * EXPANSIONS, MATRIX_ROWS as the reference counts them, TOTAL_REPORTED_POSITIONS / LOCATED_ROWS = located positions,
 * CMB_CNT_TABLE_ROWS = move-table rows fetched.  cmb_move_match_batch: host buffers in and out, CMB_ERR_OVERFLOW with *needed = the
 * number of records if out_cap is too small.  cmb_move_batch_*: reads resident on the device between runs (what bench.py times); a
 * chunk of 2^19 reads or more is matched as two concurrent halves, each with pools, streams and a host thread of its own (results,
 * offsets, counters and alignments are those of the one batch; CMB_MOVE_SUBBATCHES=n at creation overrides, 1 = one batch; fewer than 2^23
 * reads per batch).
 * Reads not longer than the number of parts are matched by naive backtracking, as the reference matches them (searchstrategy.cpp:148-152). */
typedef struct cmb_move_batch cmb_move_batch;
int cmb_move_match_batch(cmb_move_index* idx, const cmb_strategy* st, uint32_t max_distance, uint32_t kmer_size, const char* seqs,
                         const uint64_t* offs, uint32_t n_reads, cmb_move_occ* out, uint64_t out_cap, uint64_t* out_offs /* [n_reads + 1] */,
                         uint64_t* counters /* [CMB_CNT_MAX] or NULL */, uint64_t* needed);
int cmb_move_batch_create(cmb_move_index* idx, const cmb_strategy* st, uint32_t max_distance, uint32_t kmer_size, const char* seqs,
                          const uint64_t* offs, uint32_t n_reads, cmb_move_batch** out);
int cmb_move_batch_run(cmb_move_batch* b);
int cmb_move_batch_result_size(const cmb_move_batch* b, uint64_t* n_occ);
int cmb_move_batch_results(const cmb_move_batch* b, cmb_move_occ* out, uint64_t out_cap, uint64_t* out_offs, uint64_t* counters);
int cmb_move_batch_timings(const cmb_move_batch* b, const char** names, float* ms, uint32_t cap);
void cmb_move_batch_destroy(cmb_move_batch* b);
/* Alignments of the occurrences of a b-move batch.  The reference builds the CIGAR of this flavour from the matched string, which its
 * search carries along (src/indexinterface.h:294-303) because the index holds no text.  That string is text[begin, end) of the
 * occurrence: with the text beside the index in HBM (cmb_move_attach_text: one byte + a quarter byte per character; texts below 2^32
 * characters) the CIGAR is IBitParallelED::findCIGAR of that window — what cmb_batch_alignments computes for the FM-index flavour, same
 * kernel.  seq_starts as for cmb_index_desc (NULL: one sequence).  cmb_move_batch_alignments: records parallel to
 * cmb_move_batch_results, CIGAR runs as for cmb_batch_alignments; an occurrence that runs over the end of its sequence has spans = 1. */
int cmb_move_attach_text(cmb_move_index* idx, const char* text, uint64_t n, const uint32_t* seq_starts, uint32_t n_seqs);
cmb_index* cmb_move_text_index(const cmb_move_index* idx); /* the text-only index behind it (owned by idx), or NULL */
int cmb_move_batch_want_alignments(cmb_move_batch* b, int on);
int cmb_move_batch_alignments(const cmb_move_batch* b, cmb_aln* out, uint64_t cap, uint16_t* cigar_ops, uint64_t ops_cap, uint64_t* n_ops);
/* as cmb_batch_filter_per_strand: the redundancy filter takes every strand of a read by itself (what SearchStrategy::mapRead returns,
 * src/searchstrategy.h:490-523); a read's occurrences are then those of its forward strand followed by those of the other one */
int cmb_move_batch_filter_per_strand(cmb_move_batch* b, int on);
/* BEST (+x strata) mode on the b-move index: cmb_match_best with b-move batches as strata — the reference's RUN_LENGTH_COMPRESSION build
 * runs the same SearchStrategy::matchApproxBestPlusX (src/searchstrategy.cpp:623-746).  Needs cmb_move_attach_text: CIGARs
 * (IndexInterface::generateCIGAR on the matched string, src/indexinterface.h:959-989) and the trimming of occurrences that run over a
 * sequence end (checkTrimmedMatch, src/indexinterface.cpp:722-796, which walks the trimmed part of the matched string — no counters)
 * read text[begin, end).  Cut-off: min(13, what the strategy has schemes for, len * (100 - min_identity) / 100), as on the FM-index.  Results through cmb_best_sizes / cmb_best_results / cmb_best_free. */
int cmb_move_match_best(cmb_move_index* idx, const cmb_strategy* st, uint32_t x, uint32_t min_identity, uint32_t kmer_size,
                        const char* seqs, const uint64_t* offs, uint32_t n_reads, cmb_best** out);
/* device time (ms, HIP events) of the calling thread's last cmb_move_match_exact: [0] the backward extension of all reads,
 * [1] the prefix sum of the widths, [2] locate + occurrence records */
int cmb_move_last_timings(float* ms, uint32_t n);

const char* cmb_last_error(void);
const char* cmb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* COLUMBA_AMD_H */
