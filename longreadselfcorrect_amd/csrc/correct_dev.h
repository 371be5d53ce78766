// correct_dev.h -- host/device interface of the persistent per-read correction kernel (correct_dev.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "dp_dev.h"
#include "extend.h"

namespace lrsc {

enum { LRSC_WALK_ERR_GEOMETRY = -103, LRSC_WALK_ERR_OUTPUT = -104, LRSC_WALK_ERR_CODE = -105, LRSC_WALK_ERR_DP = -106 };
enum : uint32_t { kReadDone = 0, kReadParked = 1, kReadYield = 2 };   // ReadOut::state: parked = waiting for the DP stage's
                                                                        // answer; yield = walk budget of this launch used up

constexpr uint32_t kMaxInitK = 59;       // initk + 2 = maxOverlap must stay below the 64-character suffix window

// per-read bounds found by correct_plan_kernel: the host sizes the read's workspace from them
struct ReadPlan {
    uint32_t gap_max;        // longest raw segment any walk of this read can be asked to bridge
    uint32_t lq_max;         // longest m_query = kMaxInitK + gap + target seed
};

// per-read workspace layout (bytes from CorrectArgs::workspace + ws_off) and output slots
struct ReadWork {
    uint64_t ws_off;
    uint64_t out_off;        // into out_codes
    uint64_t piece_off;      // into piece_start
    uint32_t lq_max, pathw, out_cap, piece_cap;
    uint32_t o_item9f, o_item9r, o_next9f, o_next9r, o_head9, o_head5, o_next5, o_flags5, o_term, o_leaves, o_rings,
        o_paths, o_results, o_query, o_best, o_dpq;
};

struct ReadOut {             // PacBioSelfCorrectionResult (PacBioSelfCorrectionProcess.h:58-94) without the strings
    int64_t c[10];           // totalReadsLen, correctedLen, totalSeedNum, totalWalkNum, highErrorNum, exceedDepthNum,
                             // exceedLeaveNum, FMNum, DPNum, seedDis
    uint64_t steps;
    uint64_t cyc[4];         // s_memtime ticks in: query + prepare, trees + root, extension loop, stitching (LRSC_CORRECT_PROFILE)
    uint32_t n_pieces, out_len, merge;
    int32_t error;
    // chain state of a parked read (pieceVec.back()'s SeedFeature fields + iterTarget) and its DP request
    uint32_t state, it;
    int32_t s_seed_len, s_end, s_end_best, s_max_fixed, s_is_repeat;
    uint32_t dp_k, dp_lq;
    int64_t dp_total_freq;       // source.maxFixedMerFreq + target.maxFixedMerFreq
    uint64_t cyc_step[8];        // LRSC_CORRECT_PROFILE, lane kernel: ticks inside the extension step: extendLeaves (of which refine, attempToExtend,
                                 // getFMIndexExtensions), PrunedBySeedSupport, materialise + commit, isTerminated
};

struct CorrectArgs {
    const uint8_t* codes;            // batch reads, 2-bit codes one per byte
    const uint64_t* read_off;
    const int32_t* seeds;            // kSeedInts per seed, slab of read r at seed_slab(read_off[r], r, min_k)
    const uint32_t* seed_count;
    const uint32_t* order;           // launch order (similar lengths share a wavefront)
    const ReadWork* work;
    uint32_t n_reads, min_k;
    uint32_t reads_per_wave;         // 1..64, power of two: 64 / reads_per_wave lanes apart
    uint32_t occupancy;              // wavefronts per SIMD the kernel variant is compiled for: 2 (216 VGPRs) or 4 (128, spills)
    uint8_t* workspace;
    uint8_t* out_codes;
    uint32_t* piece_start;
    ReadOut* out;
    ReadPlan* plan;                  // plan kernel only
    // FMextendParameters / PacBioSelfCorrectionParameters
    uint32_t seed_size, min_overlap, max_leaves;
    int32_t start_kmer_len, next_target, split, no_dp;
    // second and later launches: reads parked on a DP request pick up the answer
    uint32_t resume;
    uint32_t* queue;                 // optional work queue (zeroed before the launch): next slot of `order` to hand out
    uint32_t queue_waves;            // wavefronts to launch when the queue is used
    uint32_t profile;                // per-phase tick counters in ReadOut::cyc (LRSC_CORRECT_PROFILE)
    uint32_t setup_quorum_pct;       // lanes of a wavefront (in %) that must be between walks before they set the next ones up
    uint32_t slow_gate_sweeps;       // ... sweeps a lane with a wide frontier waits at most for company before its general commit runs
    uint32_t step_gate_pct;          // state-machine kernel: lanes inside a walk (in %) that must be at the step gate before it opens
    uint32_t max_steps;              // ... or extension steps: no new walk is started past this budget
    uint32_t max_walks;              // walks a read may run per launch before it yields (0 = no limit); keeps DP rounds even
    const uint32_t* dp_index;        // read -> request
    const DpRequest* dp_reqs;
    const DpMsaOut* dp_msa;
    const uint8_t* dp_cons;
    uint64_t pb_coverage;
    double pacbio_error_rate;
    const double* freqs_of_kmer_size;
    DevCounters* ctr;
    // --debugseed: one byte per seed (same slab indexing as `seeds`), written for the target seed of every walk the FM-extension
    // gave up on: (first FM result code + 4) | 0x10 if the DP fallback failed too (the lines of extend/<read>.ext and .dp)
    uint8_t* walk_log;
    // debugging aid (LRSC_SM_TRACE): the state-machine kernel records (pc, request, answer) of read `trace_read` per sweep
    uint32_t dbg_flags;              // LRSC_SM_DBG: timing ablations (work done twice; results unchanged)
    unsigned long long* prof;        // LRSC_SM_PROFILE: 16 tick / count totals per wavefront (state-machine kernel)
    uint32_t* trace;
    uint32_t trace_cap, trace_read;
};

hipError_t launch_correct_plan(const CorrectArgs& a, hipStream_t stream);
hipError_t launch_correct_reads(const FmIndexDev& fm, const CorrectArgs& a, hipStream_t stream);
// the wavefront-convergent state-machine form (correct_sm.hip): d_fm / d_args are device copies of fm / a
hipError_t launch_correct_sm(const FmIndexDev* d_fm, const CorrectArgs* d_args, const CorrectArgs& a, bool wide, hipStream_t stream, const FmIndexDev& fm);
// out_codes -> ASCII, packed at dst + dst_off[r]
hipError_t launch_correct_gather(const CorrectArgs& a, const uint64_t* dst_off, char* dst, hipStream_t stream);

} // namespace lrsc
