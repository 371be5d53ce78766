// correct_dev.h -- constants and small records shared by the correction flow (wp.h / wp.hip) and the host code.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "dp_dev.h"
#include "extend.h"

namespace lrsc {

enum { LRSC_WALK_ERR_GEOMETRY = -103, LRSC_WALK_ERR_OUTPUT = -104, LRSC_WALK_ERR_CODE = -105, LRSC_WALK_ERR_DP = -106 };
enum : uint32_t { kReadDone = 0, kReadParked = 1 };   // WpRead::state: parked = waiting for a re-queued walk / DP answer

constexpr uint32_t kMaxInitK = 59;       // initk + 2 = maxOverlap must stay below the 64-character suffix window

// per-read bounds found by the bounds kernel: the host sizes the read's output slot from them and skips reads beyond a capacity
struct ReadPlan {
    uint32_t gap_max;        // longest raw segment any walk of this read can be asked to bridge
    uint32_t lq_max;         // longest m_query = kMaxInitK + gap + target seed
};

} // namespace lrsc
