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
    uint32_t skip, pad;           // skip != 0: "identical sequence", not aligned (LongReadOverlap.cpp:629-633)
};

struct DpAlignOut {               // SequenceOverlap (Thirdparty/overlapper.h:69-125)
    int32_t m0s, m0e, m1s, m1e;   // match[0] (s1) and match[1] (s2) start / end
    int32_t score, edit_distance, total_columns;
    uint32_t n_ops;               // expanded cigar, stored LAST op first: op t of the cigar is ops[n_ops - 1 - t]
};

struct DpAlignArgs {
    const uint8_t* codes;
    const DpJob* jobs;
    uint32_t n_jobs;
    uint32_t band_width;          // as passed to extendMatch (200)
    int32_t match_score, gap_penalty, mismatch_penalty;
    uint8_t* ops;                 // 'M' 'I' 'D'
    DpAlignOut* out;
    uint8_t* trace;               // n_waves x trace_stride
    uint64_t trace_stride;        // (max s1_len + 1) * kDpTraceStride
    uint32_t max_s1, max_s2;      // LDS staging sizes
};

// n_waves = gridDim.x; every wave loops over jobs wave, wave + n_waves, ...
hipError_t launch_dp_align(const DpAlignArgs& a, uint32_t n_waves, hipStream_t stream);

} // namespace lrsc
