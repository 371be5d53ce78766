// extend.h -- host/device interface of the FM-extend kernels (extend.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "kernels.h"

namespace lrsc {

constexpr uint32_t kMaxChildren = 128;       // 4 extensions x 32 leaves
constexpr uint32_t kMaxResults = 160;        // result slots per walk (one per terminated lineage)
enum { LRSC_WALK_ERR_CHILDREN = -101, LRSC_WALK_ERR_RESULTS = -102 };

struct WalkResultRec {
    double error_rate;
    uint32_t path_len;
    uint32_t match_i;
};

// per-walk geometry + workspace offsets (bytes from ExtendArgs::workspace + ws_off)
struct WalkWork {
    uint64_t codes_off;      // m_query codes: beginning k-mer | raw read segment | target seed
    uint64_t ws_off;
    uint64_t out_off;        // bytes into out_paths (2-bit packed path of the chosen result)
    uint32_t lq, initk, path_len, trg_len;
    int32_t dis;
    uint32_t max_overlap, min_sa, pathw;
    uint32_t o_item9f, o_item9r, o_next9f, o_next9r, o_head9, o_head5, o_next5, o_flags5, o_term, o_leaves, o_rings,
        o_paths, o_results;
};

struct WalkOut {
    int32_t code;            // extendOverlap's return: 1, -1, -2, -3, -4 (or a LRSC_WALK_ERR_*)
    uint32_t path_len;       // characters of the walked path (starts with the beginning k-mer)
    uint32_t match_i;        // target offset i of the terminating 13-mer: merged = path + target[i + 13 ..]
    uint32_t steps;
};

struct ExtendArgs {
    const uint8_t* codes;
    const WalkWork* work;
    const uint64_t* q_off;          // n_walks + 1: prefix sums of lq (prepare kernel)
    const uint32_t* chunk_walk;
    const uint32_t* order;          // optional launch order (long walks first)
    uint64_t total_q;
    uint32_t n_walks;
    uint8_t* workspace;
    uint8_t* out_paths;
    WalkOut* out;
    // FMextendParameters (LongReadCorrectByOverlap.h:28-47)
    uint32_t seed_size, min_overlap, max_leaves;
    uint64_t pb_coverage;
    double pacbio_error_rate;
    const double* freqs_of_kmer_size;   // [101], pow() table computed on the host (.cpp:68-70)
    DevCounters* ctr;
};

size_t leaf_bytes(bool wide);
hipError_t launch_walk_prepare(const FmIndexDev& fm, const ExtendArgs& a, hipStream_t stream);
hipError_t launch_walk_extend(const FmIndexDev& fm, const ExtendArgs& a, hipStream_t stream);

} // namespace lrsc
