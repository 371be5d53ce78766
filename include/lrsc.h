/* include/lrsc.h -- C ABI of the MI355X-native PacBio self-correction hot path.
 *
 * This is the drop-in boundary: the C++ host side (longreadselfcorrect_amd/host,
 * which mirrors the reference's SequenceProcessFramework / PacBioSelfCorrectionProcess
 * interface) and any foreign binding call ONLY these entry points.  Plain pointers and
 * sizes; no C++ or torch types.  Every function returns 0 (LRSC_OK) or a negative
 * lrsc_status and never calls exit(); the caller owns every buffer it passes.
 *
 * Reference interface each group replaces (paths relative to the reference tree):
 *   lrsc_index_*          RLBWT::RLBWT(file) + initializeFMIndex   SuffixTools/RLBWT.cpp:23-32,109-248
 *                         (loaded at StriDe/PacBioSelfCorrection.cpp:155-172)
 *   lrsc_rank             RLBWT::getOcc / getPC                    SuffixTools/RLBWT.h:118-140
 *   lrsc_bwt_chars        RLBWT::getChar                           SuffixTools/RLBWT.h:42-63
 *   lrsc_find_kmers       BWTAlgorithms::findInterval/findBiInterval  SuffixTools/BWTAlgorithms.cpp:14-38
 *   lrsc_kmer_grid        KmerFeature grid of LongReadProbe::getSeqAttribute
 *                         PacBio/LongReadProbe.cpp:139-150, PacBio/KmerFeature.h:37-64,92-99
 *   lrsc_find_seeds       LongReadProbe::searchSeedsWithHybridKmers   PacBio/LongReadProbe.cpp:34-117,187-227
 *   lrsc_extend_walks     LongReadSelfCorrectByOverlap ctor + extendOverlap
 *                         PacBio/LongReadCorrectByOverlap.cpp:17-95,155-211
 *   lrsc_correct_batch    PacBioSelfCorrectionProcess::process     PacBio/PacBioSelfCorrectionProcess.cpp:23-206
 *
 * Threading: an lrsc_index is immutable after lrsc_index_upload and may be shared;
 * an lrsc_ctx is single-threaded (one per host worker / per device) and every call on it
 * is synchronous with respect to the caller unless stated otherwise.
 */
#ifndef LRSC_H
#define LRSC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRSC_ABI_VERSION 2

typedef enum lrsc_status {
    LRSC_OK = 0,
    LRSC_ERR_IO = -1,          /* cannot open / short read                      */
    LRSC_ERR_FORMAT = -2,      /* "BWT file is not properly formatted"          */
    LRSC_ERR_ARG = -3,         /* bad argument (null, range, non-ACGT base ...)  */
    LRSC_ERR_NOMEM = -4,
    LRSC_ERR_DEVICE = -5,      /* HIP runtime error, no device, not uploaded     */
    LRSC_ERR_CAPACITY = -6,    /* caller-provided output buffer too small        */
    LRSC_ERR_UNSUPPORTED = -7,
    LRSC_ERR_LIMIT = -8           /* an internal capacity of the device implementation was exceeded (see lrsc_last_error) */
} lrsc_status;

typedef struct lrsc_index lrsc_index;   /* both strands' FM-index: host image + per-device copies */
typedef struct lrsc_ctx lrsc_ctx;       /* per-device execution context (stream, scratch, params)  */

/* BWTInterval (SuffixTools/BWTInterval.h:19-81): lower > upper == invalid */
typedef struct lrsc_interval { int64_t lower, upper; } lrsc_interval;
/* BiBWTInterval (BWTInterval.h:82-100): fwd = interval of reverse(w) in the .rbwt index,
 * rvc = interval of reverse-complement(w) in the .bwt index */
typedef struct lrsc_biinterval { lrsc_interval fwd, rvc; } lrsc_biinterval;

/* which strand's index a raw rank / char query addresses */
enum { LRSC_BWT = 0, LRSC_RBWT = 1 };

typedef struct lrsc_rank_query {
    int64_t idx;       /* position in the BWT, -1 allowed (Occ == 0), < num_symbols */
    uint8_t base;      /* 'A','C','G','T' */
    uint8_t strand;    /* LRSC_BWT or LRSC_RBWT */
    uint8_t pad[6];
} lrsc_rank_query;

typedef struct lrsc_index_info {
    uint64_t num_strings;        /* reads in the index                        */
    uint64_t num_symbols;        /* BWT length per strand (bases + num_strings) */
    uint64_t num_runs[2];        /* RL units on disk (.bwt, .rbwt)              */
    uint64_t pred_count[2][5];   /* C[$ACGT] per strand                         */
    uint32_t block_bytes;        /* bytes of one device rank block              */
    uint32_t block_symbols;      /* BWT symbols covered by one rank block       */
    uint64_t device_bytes;       /* HBM bytes of both strands once uploaded     */
} lrsc_index_info;

/* Parameters of the correction path: a POD mirror of PacBioSelfCorrectionParameters +
 * FMextendParameters + ProbeParameters (PacBio/PacBioSelfCorrectionProcess.h:24-53,
 * LongReadCorrectByOverlap.h:28-47, LongReadProbe.h:7-40) AFTER the driver's derivation
 * step (StriDe/PacBioSelfCorrection.cpp:195-206).  Fill with lrsc_params_default(). */
typedef struct lrsc_params {
    int32_t pb_coverage;        /* -c, default 90                                  */
    double  error_rate;         /* -e, default 0.15                                */
    int32_t start_kmer_len;     /* derived from -g: {5:17, 10:19, 100:21}          */
    int32_t offset[3];          /* static k-mer size offset per mode (0,1,2)       */
    int32_t mode;               /* -m, default 1 (only used when manual != 0)      */
    int32_t manual;             /* -k/-u/-r given                                  */
    int32_t scan_kmer_len;      /* 19  (LongReadProbe.h:27)                        */
    int32_t kmer_len_up_bound;  /* 50  (LongReadProbe.h:28)                        */
    int32_t radius;             /* 100 (LongReadProbe.h:32)                        */
    float   hh_ratio;           /* 0.6f (LongReadProbe.h:33)                       */
    int32_t next_target;        /* -n, default 1                                   */
    int32_t max_leaves;         /* -l, default 32                                  */
    int32_t idmer_len;          /* -i, default 9                                   */
    int32_t min_kmer_len;       /* -s, default 13                                  */
    int32_t split;              /* --split                                         */
    int32_t no_dp;              /* --nodp                                          */
} lrsc_params;

/* genome = 5, 10 or 100 (the -g option); coverage = -c.  Reproduces the derivation at
 * StriDe/PacBioSelfCorrection.cpp:195-200. Returns LRSC_ERR_ARG for another genome value. */
int lrsc_params_default(int genome, int coverage, lrsc_params* out);

/* ---- index ------------------------------------------------------------------------ */
/* Parse <prefix>.bwt and <prefix>.rbwt (binary RL format, Appendix C of SURVEY.md) and
 * build the HBM rank-block image on the host. */
int lrsc_index_open(const char* bwt_path, const char* rbwt_path, lrsc_index** out);
/* Same, from RL-unit strings already in memory (units: (rank<<5)|run_len). */
int lrsc_index_from_units(const uint8_t* bwt_units, uint64_t n_bwt_units,
                          const uint8_t* rbwt_units, uint64_t n_rbwt_units,
                          uint64_t num_strings, uint64_t num_symbols, lrsc_index** out);
int lrsc_index_info_get(const lrsc_index* idx, lrsc_index_info* out);
/* Copy both strands into `device`'s HBM (idempotent per device). */
int lrsc_index_upload(lrsc_index* idx, int device);
void lrsc_index_close(lrsc_index* idx);

/* ---- index construction (the `stride index -a ropebwt2` step; StriDe/index.cpp:164-213,
 *      SuffixTools/BWTCARopebwt.cpp:160-247) ------------------------------------------------ */
/* Builds the multi-string BWT of the reads (reverse_reads == 0 -> the .bwt payload) or of the
 * reversed reads (reverse_reads != 0 -> the .rbwt payload) on `device` by suffix sorting, sentinels in
 * input order, and returns it as the reference's RL units ((rank<<5)|run, runs <= 31,
 * BWTWriterBinary.cpp:50-71).  *units_out is malloc'ed: release with lrsc_buffer_free.
 * num_symbols == read_off[n_reads] + n_reads. */
int lrsc_build_bwt(const char* reads, const uint64_t* read_off, uint32_t n_reads, int reverse_reads, int device,
                   uint8_t** units_out, uint64_t* n_units_out);
void lrsc_buffer_free(void* p);
/* 30-byte header + units, the reference's binary .bwt/.rbwt format (BWTWriterBinary.cpp:28-46,82-93). */
int lrsc_write_bwt_file(const char* path, const uint8_t* units, uint64_t n_units, uint64_t num_strings,
                        uint64_t num_symbols);

/* ---- context ------------------------------------------------------------------------ */
int lrsc_ctx_create(const lrsc_index* idx, const lrsc_params* params, int device, lrsc_ctx** out);
void lrsc_ctx_destroy(lrsc_ctx* ctx);

/* ---- FM primitives (host buffers in, host buffers out) --------------------------------- */
/* out[i] = Occ(base, idx) = #base in BWT[0..idx] of the chosen strand (RLBWT::getOcc). */
int lrsc_rank(lrsc_ctx* ctx, const lrsc_rank_query* q, uint64_t n, uint64_t* out);
/* out[i] = BWT[idx[i]] of `strand` as '$','A','C','G','T' (RLBWT::getChar). */
int lrsc_bwt_chars(lrsc_ctx* ctx, int strand, const uint64_t* idx, uint64_t n, char* out);
/* n fixed-length k-mers, concatenated (n*k bytes, ACGT). out[i] = findBiInterval(kmer_i):
 * fwd searched in the rbwt with reverse(w), rvc in the bwt with revcomp(w), each stopping at
 * the first invalid interval exactly as BWTAlgorithms.cpp:14-31. */
int lrsc_find_kmers(lrsc_ctx* ctx, const char* kmers, uint32_t k, uint64_t n, lrsc_biinterval* out);

/* ---- LongReadProbe k-mer feature grid -------------------------------------------------- */
/* reads: concatenated ACGT bytes, read_off: n_reads+1 offsets.  ks: ascending k-mer sizes
 * (the "pool", e.g. {5,9,15,17,19}), n_k <= 8.  For read r, position p, pool slot j the record
 * index is (read_off[r] + p) * n_k + j.
 *   out_iv[rec]      bi-interval of the KmerFeature (base slot by findBiInterval, larger slots by
 *                    expand(); KmerFeature.h:37-64,92-99)
 *   out_size[rec]    KmerFeature::size (shorter than ks[j] near the read end: "fake")
 *   out_count[rec*4] base-composition counters A,C,G,T accumulated during the search
 * Any of the three outputs may be NULL. */
int lrsc_kmer_grid(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                   const uint8_t* ks, uint32_t n_k,
                   lrsc_biinterval* out_iv, uint8_t* out_size, uint8_t* out_count);

/* ---- resident batches (inputs already in HBM; what bench.py times) ---------------------- */
typedef struct lrsc_batch lrsc_batch;
/* Upload a batch of reads to the ctx's device; validates ACGT. */
int lrsc_batch_create(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                      lrsc_batch** out);
void lrsc_batch_destroy(lrsc_batch* b);
/* Run the k-mer grid kernel over the whole resident batch, keeping results on the device
 * (compact per-position features only).  Timed internally with HIP events on the ctx stream. */
int lrsc_batch_kmer_grid(lrsc_ctx* ctx, lrsc_batch* b);

/* ---- LongReadProbe seeds ------------------------------------------------------------------ */
/* SeedFeature (PacBio/SeedFeature.h:35-45) as searchSeedsWithHybridKmers leaves it, i.e. after
 * estimateBestKmerSize and removeHitchhikingSeeds.  seedStr == read[start, start+len). */
typedef struct lrsc_seed {
    int32_t start;                /* seedStartPos (seedEndPos = start + len - 1) */
    int32_t len;                  /* seedLen                                      */
    int32_t max_fixed_mer_freq;   /* maxFixedMerFreq                              */
    int32_t is_repeat;            /* isRepeat                                     */
    int32_t start_best_kmer_size; /* startBestKmerSize                            */
    int32_t end_best_kmer_size;   /* endBestKmerSize                              */
    int32_t start_kmer_freq;      /* startKmerFreq                                */
    int32_t end_kmer_freq;        /* endKmerFreq                                  */
} lrsc_seed;
/* KmerThreshold table (PacBio/KmerThreshold.cpp:43-79) for `coverage`: out[mode*52 + k], mode 0..2, k 0..51. */
int lrsc_kmer_thresholds(int coverage, float* out);
/* The same table for k up to `end` (KmerThreshold::initialize(s, end, coverage, ""), e.g. `stride kmerfreq`: end = 100):
 * out[mode*(end+2) + k], k 0..end+1. */
int lrsc_kmer_thresholds_range(int coverage, int end, float* out);
/* Seeds of every read of a resident batch: k-mer grid, LongReadProbe::getSeqAttribute,
 * searchSeedsWithHybridKmers (PacBio/LongReadProbe.cpp:34-227), all on the device. */
int lrsc_batch_find_seeds(lrsc_ctx* ctx, lrsc_batch* b);
/* Copy the result of lrsc_batch_find_seeds to the host: seed_count[n_reads]; seeds (read order, then
 * position order; may be NULL) with capacity `cap` records; *n_seeds = total; attribute (optional) one
 * byte per base: 1 unique, 2 repeat (getSeqAttribute). LRSC_ERR_CAPACITY if cap is too small. */
int lrsc_batch_seeds(lrsc_ctx* ctx, lrsc_batch* b, uint32_t* seed_count, lrsc_seed* seeds, uint64_t cap,
                     uint64_t* n_seeds, int8_t* attribute);
/* Convenience: upload, find, fetch, release. */
int lrsc_find_seeds(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                    uint32_t* seed_count, lrsc_seed* seeds, uint64_t cap, uint64_t* n_seeds, int8_t* attribute);

/* ---- diagnostics of --debugseed / --onlyseed ---------------------------------------------------------------- */
/* What the reference writes per read under <out>/seed and <out>/extend when DebugSeed is set
 * (LongReadProbe.cpp:109-113,123-175,220-225; PacBioSelfCorrectionProcess.cpp:71-75,130-140).  Collection is off by
 * default and costs nothing then; switch it on for a batch BEFORE lrsc_batch_find_seeds. */
enum {
    LRSC_DEBUG_OUTCASTS = 1,   /* seeds dropped by removeHitchhikingSeeds        -> seed/error/<read>.seed */
    LRSC_DEBUG_WALKS    = 2,   /* FM walks that failed, DP fallbacks that failed -> extend/<read>.ext, .dp */
    LRSC_DEBUG_RATIO    = 4    /* getSeqAttribute's repeat ratio per position    -> extend/<read>.log      */
};
int lrsc_batch_set_debug(lrsc_batch* b, int flags);
/* Dropped seeds, laid out like lrsc_batch_seeds (initial-seed order within a read). */
int lrsc_batch_outcast_seeds(lrsc_ctx* ctx, lrsc_batch* b, uint32_t* outcast_count, lrsc_seed* seeds, uint64_t cap,
                             uint64_t* n_seeds);
/* ratio[total_bases]: the value getSeqAttribute compares with 0.02 (undefined for reads shorter than start_kmer_len). */
int lrsc_batch_repeat_ratio(lrsc_ctx* ctx, lrsc_batch* b, float* ratio);
/* After lrsc_batch_correct: one byte per seed in lrsc_batch_seeds order.  For seed i >= 1 of a read, 0 = the walk that
 * targets it succeeded by FM-extension (or it was skipped as a next-target); otherwise bits 0-3 = firstFMExtensionType + 4
 * (3 high error, 2 exceed depth, 1 exceed leaves) and bit 4 = the DP/MSA fallback failed too.  The walk's source is seed i-1. */
int lrsc_batch_walk_log(lrsc_ctx* ctx, lrsc_batch* b, uint8_t* log, uint64_t cap);

/* ---- seed-to-seed FM-extend ----------------------------------------------------------------------- */
/* One LongReadSelfCorrectByOverlap(sourceSeed, strBetweenSrcTarget, targetSeed, disBetweenSrcTarget,
 * initkmersize, maxOverlap, FM_params, min_SA_threshold).extendOverlap(result) call
 * (PacBio/LongReadCorrectByOverlap.cpp:17-95,155-211; built at PacBioSelfCorrectionProcess.cpp:186-190).
 * The three strings are ASCII ACGT, concatenated at seq + seq_off. */
typedef struct lrsc_walk_desc {
    uint64_t seq_off;
    uint32_t src_len;            /* sourceSeed length (>= init_kmer; its last init_kmer bases start the walk) */
    uint32_t path_len;           /* strBetweenSrcTarget length                                                */
    uint32_t trg_len;            /* targetSeed length                                                         */
    int32_t  dis;                /* disBetweenSrcTarget                                                       */
    uint32_t init_kmer;          /* initkmersize                                                              */
    uint32_t max_overlap;        /* maxOverlap                                                                */
    uint32_t min_sa_threshold;   /* min_SA_threshold                                                          */
    uint32_t pad;
} lrsc_walk_desc;
typedef struct lrsc_walk_result {
    int32_t  code;               /* extendOverlap's return value: 1, or -1 high error, -2 depth, -3 leaves, -4 */
    uint32_t steps;              /* iterations of the extension loop (accounting)                              */
    uint64_t out_off;            /* FMWalkResult2::mergedSeq at out_arena + out_off (code > 0 only)            */
    uint32_t out_len;
    uint32_t pad;
} lrsc_walk_result;
/* Runs n independent walks on the device.  out_arena receives the merged sequences back to back
 * (*arena_used bytes); LRSC_ERR_CAPACITY if arena_cap is too small (then *arena_used = bytes needed). */
int lrsc_extend_walks(lrsc_ctx* ctx, const char* seq, uint64_t seq_len, const lrsc_walk_desc* walks, uint32_t n,
                      lrsc_walk_result* results, char* out_arena, uint64_t arena_cap, uint64_t* arena_used);

/* ---- the whole per-read path ------------------------------------------------------------------------ */
/* PacBioSelfCorrectionResult (PacBio/PacBioSelfCorrectionProcess.h:58-94) without the wall-clock timers;
 * correctedStrs = pieces [piece_first, piece_first + n_pieces). */
typedef struct lrsc_read_result {
    int32_t  merge;              /* !pieceVec.empty(): record goes to correct.fa, else the raw read to discard.fa */
    uint32_t n_pieces;           /* 1, or more with --split                                                       */
    uint64_t piece_first;
    int64_t  total_reads_len, corrected_len, total_seed_num, total_walk_num, high_error_num, exceed_depth_num,
             exceed_leave_num, fm_num, dp_num, seed_dis;
    int32_t  status;             /* LRSC_READ_OK, or why this one read could not be corrected (the rest of the batch is unaffected):
                                  * the read then comes back uncorrected (merge = 0, no pieces, counters 0)                 */
    int32_t  pad;
} lrsc_read_result;
/* per-read status: an internal capacity of the device path was exceeded by this read alone.  The reference has no such
 * bounds (its containers grow); a caller that must not lose the read can route it to a CPU path. */
enum lrsc_read_status {
    LRSC_READ_OK = 0,
    LRSC_READ_WALK_QUERY_TOO_LONG = 1,   /* a walk's query (k-mer + gap between two seeds + target seed) >= 65535 bases  */
    LRSC_READ_TOO_LONG = 2,              /* the read's output slot would exceed 4 GB                                     */
    LRSC_READ_FRONTIER_LIMIT = 3,        /* more than 160 terminated lineages / 128 children in one walk                 */
    LRSC_READ_GEOMETRY = 4,              /* seeds overlap / an extension k-mer above 59 or below the idmer size          */
    LRSC_READ_DP_LIMIT = 5,              /* DP fallback: a pile-up beyond its column / consensus capacity (a long query is not a
                                          * limit: alignments beyond the kernel's LDS stage run from a global workspace)   */
    LRSC_READ_OUTPUT_LIMIT = 6,          /* the corrected string outgrew its slot                                        */
    LRSC_READ_INTERNAL = 7               /* FM-extension returned a code the reference treats as impossible (it exits)   */
};
/* PacBioSelfCorrectionProcess::process for every read of the batch (PacBioSelfCorrectionProcess.cpp:23-245): seeds, the chain of
 * seed-to-seed FM-extensions, the DP/MSA fallback of correctByMSAlignment (unless params.no_dp) and the stitching, all on the
 * device, with the ctx's parameters.  Piece p is out[piece_off[p] .. piece_off[p+1]).
 * LRSC_ERR_CAPACITY (with *n_pieces / *out_used = what is needed) if piece_cap / out_cap are too small.  A read that exceeds a
 * per-read capacity does not fail the call: see lrsc_read_result.status. */
int lrsc_correct_reads(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                       lrsc_read_result* results, uint64_t* piece_off, uint64_t piece_cap, char* out, uint64_t out_cap,
                       uint64_t* n_pieces, uint64_t* out_used);
/* The same, for a batch that is already resident on the device (lrsc_batch_create): seeds, every seed-to-seed
 * walk of the batch at once from its predicted source k-mer, one DP round for the failed ones, and a per-read
 * stitch pass that replays initCorrect's chain and re-queues the walks whose source was not the predicted one
 * (bit-identical to running the chain in order) -- all on the device; only the corrected strings and the
 * counters come back.  lrsc_correct_reads is this call on a temporary batch. */
int lrsc_batch_correct(lrsc_ctx* ctx, lrsc_batch* batch, lrsc_read_result* results, uint64_t* piece_off,
                       uint64_t piece_cap, char* out, uint64_t out_cap, uint64_t* n_pieces, uint64_t* out_used);
int lrsc_ctx_get_params(const lrsc_ctx* ctx, lrsc_params* out);

/* ---- DP/MSA fallback: Overlapper::extendMatch ------------------------------------------------------------- */
/* One banded alignment (Thirdparty/overlapper.cpp:421-701): s1 = seq[s1_off .. +s1_len) (the query, DP columns),
 * s2 = seq[s2_off .. +s2_len) (DP rows); start1/start2 = the seed match that centres the band. */
typedef struct lrsc_dp_job {
    uint64_t s1_off, s2_off;
    uint32_t s1_len, s2_len;
    int32_t  start1, start2;
} lrsc_dp_job;
typedef struct lrsc_dp_result {      /* SequenceOverlap (Thirdparty/overlapper.h:69-125) */
    int32_t  match0_start, match0_end, match1_start, match1_end;
    int32_t  score, edit_distance, total_columns;
    uint32_t cigar_len;              /* EXPANDED cigar ('M','I','D' per column) at cigar_arena + cigar_off */
    uint64_t cigar_off;
} lrsc_dp_result;
/* n alignments on the device, one wavefront each.  LRSC_ERR_CAPACITY (with *arena_used = bytes needed) if
 * arena_cap is too small. */
int lrsc_dp_align(lrsc_ctx* ctx, const char* seq, uint64_t seq_len, const lrsc_dp_job* jobs, uint32_t n, int band_width,
                  int match_score, int gap_penalty, int mismatch_penalty, lrsc_dp_result* results, char* cigar_arena,
                  uint64_t arena_cap, uint64_t* arena_used);

/* ---- DP/MSA fallback: buildMultipleAlignment + calculateBaseConsensus ---------------------------------------- */
/* One correctByMSAlignment-style call (PacBio/LongReadOverlap.cpp:17-55 + Thirdparty/multiple_alignment.cpp:517-594):
 * query = seq[seq_off .. +len); reads overlapping its first / last kmer_len bases are retrieved from the index
 * (at most params.pb_coverage rows per interval), aligned to the query (band 200, +1/-1/-8), filtered by
 * min_overlap / min_identity, piled up and the base consensus (min_call_coverage, no trimming) is called. */
typedef struct lrsc_msa_query {
    uint64_t seq_off;
    uint32_t len, kmer_len;
    uint32_t min_overlap;
    int32_t  min_call_coverage;
    double   min_identity;
} lrsc_msa_query;
typedef struct lrsc_msa_result {
    uint32_t n_rows;             /* MultipleAlignment::getNumRows(): 1 + accepted overlaps                    */
    uint32_t n_retrieved;        /* strings LF-walked out of the index                                         */
    uint32_t cons_len;           /* consensus at arena + cons_off                                              */
    uint32_t rows_by_step_walk;  /* diagnostic: rows the kernel added one cigar step at a time (corner cases, or
                                    all of them with LRSC_MSA_BATCH=0) instead of in wavefront-wide passes          */
    uint64_t cons_off;
} lrsc_msa_result;
int lrsc_dp_consensus(lrsc_ctx* ctx, const char* seq, uint64_t seq_len, const lrsc_msa_query* queries, uint32_t n,
                      lrsc_msa_result* results, char* arena, uint64_t arena_cap, uint64_t* arena_used);

/* ---- DP/MSA fallback building blocks ------------------------------------------------------------------- */
/* LongReadOverlap::retrieveStr's LF-walks (PacBio/LongReadOverlap.cpp:696-749): job i starts at BWT row
 * rows[i] of strand[i] and emits at most max_steps[i] characters (stops at a '$' row).  Job i's characters
 * ("ACGT", in walk order) land at out + out_off[i]; out_len[i] receives how many. */
int lrsc_lf_walk(lrsc_ctx* ctx, const uint64_t* rows, const uint8_t* strand, const uint32_t* max_steps,
                 const uint64_t* out_off, uint64_t n, char* out, uint64_t out_cap, uint32_t* out_len);

/* ---- measurement ------------------------------------------------------------------------- */
typedef struct lrsc_kernel_stats {
    uint64_t launches;          /* launches since the last reset                        */
    double   total_ms;          /* sum of HIP-event durations on the ctx stream          */
    uint64_t rank_queries;      /* Occ queries issued (algorithmic count)                */
    uint64_t block_loads;       /* rank-block loads (lower-1/upper in one block count 1) */
    uint64_t table_loads;       /* k-mer interval table look-ups (one 64-byte line each)  */
} lrsc_kernel_stats;
enum { LRSC_K_RANK = 0, LRSC_K_FIND = 1, LRSC_K_GRID = 2, LRSC_K_SEEDS = 3, LRSC_K_EXTEND = 4, LRSC_K_LF = 5, LRSC_K_DP = 6, LRSC_K_MSA = 7, LRSC_K_COUNT = 8 };
int lrsc_ctx_stats(lrsc_ctx* ctx, int kernel, lrsc_kernel_stats* out);
int lrsc_ctx_stats_reset(lrsc_ctx* ctx);
/* Block until everything queued on the ctx stream is done. */
int lrsc_ctx_sync(lrsc_ctx* ctx);

const char* lrsc_strerror(int status);
/* last detailed message for the calling thread (e.g. the HIP error string) */
const char* lrsc_last_error(void);
int lrsc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LRSC_H */
