// dp_dev.h -- host/device interface of the DP/MSA fallback kernels (dp_align.hip, dp_msa.hip):
// correctByMSAlignment (PacBio/PacBioSelfCorrectionProcess.cpp:208-245) =
//   LongReadOverlap::retrieveStr LF-walks -> Overlapper::extendMatch banded DP -> MultipleAlignment consensus.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "kernels.h"

namespace lrsc {

constexpr uint32_t kDpMaxBand = 255;          // cells per band column (band_width | 1); 4 per lane, 64 B of trace per column
constexpr uint32_t kDpTraceStride = 64;       // bytes of 2-bit traceback decisions per DP column
constexpr uint32_t kDpMaxSeq = 60000;         // longest s1 / s2 (LDS-staged)

// one banded alignment: s1 = columns (the query), s2 = rows (a retrieved read substring)
struct DpJob {
    uint64_t s1_off, s2_off;      // into DpAlignArgs::codes
    uint64_t ops_off;             // into DpAlignArgs::ops (capacity s1_len + s2_len + 1)
    uint32_t s1_len, s2_len;
    int32_t start1, start2;       // seed match positions (set the band centre, overlapper.cpp:441-442)
    uint32_t mode;                // "ignore identical sequence" (LongReadOverlap.cpp:629-633): 1 = skip if s2 starts with s1,
                                  // 2 = skip if s2 ends with s1, 0 = always align
    uint32_t req;                 // owning DpRequest (thresholds for `accept`), if DpAlignArgs::reqs is set
};

struct DpAlignOut {               // SequenceOverlap (Thirdparty/overlapper.h:69-125)
    int32_t m0s, m0e, m1s, m1e;   // match[0] (s1) and match[1] (s2) start / end
    int32_t score, edit_distance, total_columns;
    uint32_t n_ops;               // expanded cigar, stored LAST op first: op t of the cigar is ops[n_ops - 1 - t]
    uint32_t accept;              // aligned, overlap length >= min_overlap and identity >= min_identity (LongReadOverlap.cpp:645-655)
    uint32_t skipped;
    uint32_t t_fill, t_trace;     // profiling: ticks spent in the DP fill / the traceback
};

// one correctByMSAlignment call (PacBioSelfCorrectionProcess.cpp:208-245)
struct DpRequest {
    uint64_t q_off;               // query = src k-mer + raw segment + target seed, codes
    uint64_t str_off, ops_off;    // n_str slots of str_cap / ops_cap bytes
    uint64_t job_first;           // first DpJob / DpAlignOut of this request
    uint64_t cons_off;            // consensus codes (capacity cons_cap)
    uint64_t row_lo[4];           // retrieveStr start rows: 0 = fwd interval of the source k-mer (rbwt), 1 = its rvc interval (bwt),
    uint32_t cnt[4];              //                         2 / 3 = the same for the reverse-complemented target k-mer
    uint32_t lq, k, max_len, str_cap, ops_cap, cons_cap, n_str;
    uint32_t w_cols;              // column capacity of the multiple alignment (query + gap columns), grown on overflow
    uint32_t min_overlap, coverage;
    int32_t min_call_coverage;
    double min_identity;
};

struct DpMsaOut {
    uint32_t n_rows;              // MultipleAlignment::getNumRows(): 1 + accepted overlaps
    uint32_t cons_len;
    uint32_t error;               // 1 = column capacity exceeded
    uint32_t pad;
    uint32_t kc_total, kc_stage, kc_insert, n_insert;   // profiling: kilo-ticks in the whole request / staging / gap insertion; insertions
};

struct DpPipeArgs {
    const uint8_t* codes;         // queries
    DpRequest* reqs;
    uint32_t n_reqs;
    uint64_t n_jobs;
    uint8_t* strings;
    DpJob* jobs;
    const DpAlignOut* align;
    const uint8_t* ops;
    uint8_t* cons;
    DpMsaOut* msa;
    uint32_t lds_bytes;           // dynamic LDS of the MSA kernel (sized for the largest request)
    const uint32_t* req_list;     // optional: run the MSA kernel for these n_list requests only (size buckets, overflow retries)
    uint32_t n_list;
    uint8_t* msa_ws;              // set: state in this global workspace (lds_bytes per workgroup) instead of LDS
    uint32_t* work_ctr;           // optional, zeroed: the MSA kernel's wavefronts take their next request from it (after their first)
    uint32_t row_batch;           // MSA rows added in wavefront-wide passes (default); 0: the step-by-step walk only
    DevCounters* ctr;
};

struct DpAlignArgs {
    const uint8_t* codes;         // s1 sequences
    const uint8_t* strings;       // s2 sequences (may equal codes)
    const DpJob* jobs;
    uint32_t n_jobs;
    uint32_t band_width;          // as passed to extendMatch (200)
    int32_t match_score, gap_penalty, mismatch_penalty;
    uint8_t* ops;                 // 'M' 'I' 'D'
    DpAlignOut* out;
    uint8_t* trace;               // n_waves x trace_stride
    uint64_t trace_stride;        // (max s1_len + 1) * kDpTraceStride
    uint32_t max_s1, max_s2;      // staging sizes
    const DpRequest* reqs;        // optional
    // staging of the two sequences: LDS (seq_ws == nullptr; jobs that do not fit lds_cap are left to a second launch) or, for the few
    // alignments beyond it (a raw segment of tens of kb between two seeds), a per-wavefront slice of this global workspace
    uint8_t* seq_ws;
    uint64_t seq_ws_stride;
    uint32_t lds_cap;
    uint32_t only_long;           // global variant: do only the jobs the LDS launch skipped
};

// n_waves = gridDim.x; every wave loops over jobs wave, wave + n_waves, ...
hipError_t launch_dp_align(const DpAlignArgs& a, uint32_t n_waves, hipStream_t stream);
// bytes of sequence staging one alignment needs (LDS or global slice)
constexpr uint32_t dp_align_stage_bytes(uint32_t s1_len, uint32_t s2_len) { return ((s1_len + 2 + 3) & ~3u) + 264u + ((s2_len + 3) & ~3u) + 16u; }
constexpr uint32_t kDpAlignLdsCap = 64u * 1024u;
// lane per (request, direction): the two seed k-mers' bi-intervals -> row_lo / cnt
hipError_t launch_dp_seeds(const FmIndexDev& fm, const DpPipeArgs& a, hipStream_t stream);
// lane per retrieved string: LF-walk (retrieveStr), writes the string in its final orientation and its DpJob
hipError_t launch_dp_retrieve(const FmIndexDev& fm, const DpPipeArgs& a, hipStream_t stream);
// wavefront per request: MultipleAlignment::addOverlap for every accepted overlap + calculateBaseConsensus
hipError_t launch_dp_msa(const DpPipeArgs& a, hipStream_t stream);
uint32_t dp_msa_lds_bytes(uint32_t w_cols, uint32_t lq, uint32_t str_cap, uint32_t ops_cap, uint32_t n_str);
// initial column capacity of one multiple alignment: the query plus the gap columns insertions may open
constexpr uint32_t dp_msa_columns(uint32_t lq) { return 3 * lq + 128; }
constexpr uint32_t dp_cons_capacity(uint32_t lq) { return 2 * lq + 128; }
uint32_t dp_msa_waves(const DpPipeArgs& a, bool global);

} // namespace lrsc
