// wp.h -- host/device interface of the WALK-PARALLEL correction flow (wp.hip): every (read, seed pair) walk of a
// resident batch runs at once from a PREDICTED source k-mer, all failed walks go through one DP round, and a per-read
// stitch pass verifies each walk's assumed source against the string actually accumulated and re-queues the mismatches.
//
// Why this is exact: initCorrect (PacBio/PacBioSelfCorrectionProcess.cpp:56-157) chains the walks of a read only through
// `source` = pieceVec.back(): its seedStr tail (:170), its seedLen (:166) and five fields that SeedFeature::append copies
// from the TARGET seed (SeedFeature.h:22-33).  After an FM success the accumulated string ends in the target seed's own
// tail (LongReadCorrectByOverlap.cpp:849-851), after the raw fallback it ends at target.seedEndPos (:146), --split restarts
// from the target seed (:143): the source k-mer of walk j is, almost always, the last k characters of seed j-1 as it stands
// in the read (measured: 100 % on the bench-shaped sets, 90 % on the repeat-rich set, profiles/r03_spec_hit_rate.json).
// A walk (or DP answer) is only used by the stitch pass if the (k, source k-mer, next) it was computed for are the true ones.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "correct_dev.h"
#include "dp_dev.h"
#include "extend.h"

namespace lrsc {

enum : uint32_t { kWpFmValid = 1u, kWpDpValid = 2u, kWpGeomBad = 4u };
enum : uint32_t { kWpReqFm = 0u, kWpReqDp = 1u };

// One per (read, target seed `it` >= 1): the latest FM-extension attempt from the source that ends at seed it-1 towards seed
// it + next, and the DP fallback's answer for (source, seed it).  Identity = what the results were computed for.
struct alignas(16) WpSlot {
    uint32_t read, it;
    // FM attempt identity: source k-mer (forward orientation, 2 bits per character, last character in the low bits of lo)
    uint64_t src_lo, src_hi;
    uint8_t k, next, rtou, flags;
    // DP identity (always for next = 0)
    uint8_t dp_k, max_front, pad1, pad2;     // max_front: widest frontier of the attempt (profiling)
    uint64_t dp_src_lo, dp_src_hi;
    // geometry of the FM attempt: m_query = k | gap | target (after the repeat-to-unique swap the target is k long)
    uint32_t lq, gap, trg_len, pathw;
    uint32_t dp_lq;
    int32_t dp_total_freq;        // source.maxFixedMerFreq + target.maxFixedMerFreq
    uint8_t* q;                   // walk query codes
    uint8_t* dpq;                 // forward query of correctByMSAlignment (== q unless rtou)
    uint8_t* prep;                // prepared tables of the walk (temporary: valid from prepare to the end of the extend launch)
    uint32_t* path;               // result path, 2-bit packed, pathw words
    // FM result
    int32_t code;
    uint32_t path_len, match_i, steps;
    // DP result
    uint32_t dp_rows, dp_cons_len, dp_error, leaf_steps;   // leaf_steps: frontier leaves summed over the attempt's steps (profiling)
    const uint8_t* dp_cons;
};

// header of a walk's prepared state (at WpSlot::prep), followed by the tables (offsets from wp_prep_layout)
struct alignas(16) WpStatic {
    uint64_t root[4];             // root k-mer bi-interval
    uint64_t tmask0, tmask1;
    uint32_t n9f, n9r;
    uint32_t pad[2];
};

struct WpPrepLayout { uint32_t item9f, item9r, term, next9f, next9r, head9, head5, next5, flags5, total; };
__host__ __device__ inline WpPrepLayout wp_prep_layout(uint32_t lq, uint32_t trg_len, uint32_t seed_size, uint32_t min_overlap, uint32_t psz)
{
    const uint32_t n9 = lq >= seed_size ? lq - seed_size + 1 : 0, n5 = lq >= 5 ? lq - 4 : 0;
    const uint32_t nT = trg_len >= min_overlap ? trg_len - min_overlap + 1 : 0;
    WpPrepLayout L;
    uint32_t o = (uint32_t)sizeof(WpStatic);
    L.item9f = o; o += n9 * 16u;
    L.item9r = o; o += n9 * 16u;
    L.term = o;   o = (o + nT * 4u * psz + 15u) & ~15u;
    L.next9f = o; o += n9 * 2u;
    L.next9r = o; o += n9 * 2u;
    L.head9 = (o + 1u) & ~1u; o = L.head9 + 1024u;
    L.head5 = o;  o += 2048u;
    L.next5 = o;  o += n5 * 2u;
    L.flags5 = o; o += n5;
    L.total = (o + 63u) & ~63u;
    return L;
}

// per-lane dynamic workspace of the extension kernel (leaf frontier, error-history rings, result records, path slots)
struct WpLaneLayout { uint32_t leaves, rings, results, paths, total; };
__host__ __device__ inline WpLaneLayout wp_lane_layout(uint32_t lbytes, uint32_t pathw)
{
    WpLaneLayout L;
    uint32_t o = 0;
    L.leaves = o;  o = (o + (32u + kMaxChildren) * lbytes + 15u) & ~15u;
    L.rings = o;   o += 32u * 100u * 8u;
    L.results = o; o += kMaxResults * (uint32_t)sizeof(WalkResultRec);
    L.paths = o;   o += (32u + kMaxResults) * pathw * 4u;
    L.total = (o + 63u) & ~63u;
    return L;
}

// per-read layout: its slots, its output slot
struct WpReadWork {
    uint64_t out_off;            // into out_codes
    uint64_t piece_off;          // into piece_start
    uint64_t slot_first;         // WpSlot of target seed 1 (slot of seed `it` = slot_first + it - 1)
    uint32_t out_cap, piece_cap;
    uint32_t n_seeds;            // 0: skipped (a capacity) or fewer than two seeds
    uint32_t pad;
};

// stitch state of a read between rounds + its PacBioSelfCorrectionResult counters
struct WpRead {
    int64_t c[10];               // totalReadsLen, correctedLen, totalSeedNum, totalWalkNum, highErrorNum, exceedDepthNum, exceedLeaveNum, FMNum, DPNum, seedDis
    uint64_t steps;
    uint32_t n_pieces, out_len, merge;
    int32_t error;
    uint32_t state;              // kReadDone, or kReadParked = waiting for a re-queued walk / DP answer
    uint32_t it;
    int32_t next, first_type;
    int32_t s_seed_len, s_end, s_end_best, s_max_fixed, s_is_repeat;
    uint32_t started;
};

// a failed walk / explicit request handed to the DP stage
struct WpDpItem {
    uint64_t q;                  // device address of the forward query
    uint32_t slot, lq, k;
    int32_t total_freq;
};

struct WpRequest { uint32_t slot, kind; };

struct WpArgs {
    // batch
    const uint8_t* codes;
    const uint64_t* read_off;
    const int32_t* seeds;
    const uint32_t* seed_count;
    uint32_t n_reads, min_k;
    uint32_t r0, r1;             // the reads of this pass: [r0, r1) (a batch is cut into read ranges that fit the arenas)
    uint64_t slot_base;          // first slot of read r0: with list == nullptr entry i is slot slot_base + i
    const WpReadWork* work;
    WpRead* reads;
    WpSlot* slots;
    uint64_t n_slots;
    // parameters
    uint32_t seed_size, min_overlap, max_leaves;
    int32_t start_kmer_len, next_target, split, no_dp;
    uint64_t pb_coverage;
    double pacbio_error_rate;
    const double* freqs_of_kmer_size;
    uint32_t psz, lbytes;
    // work list of this round (nullptr = every slot) and the sizes the arenas are cut by
    const uint32_t* list;
    const WpRequest* reqs;       // rounds >= 1: what the stitch pass asked for (list[i] == reqs[i].slot)
    uint32_t n_list;
    uint64_t *sz_q, *sz_prep, *sz_path;      // per list entry: bytes needed; after the scan: offsets
    uint8_t *arena_q, *arena_prep, *arena_path;
    // plan statistics: [0] slots with pathw > kWpPathwSmall, [1] with pathw > kWpPathwMid, [2] max pathw
    uint32_t* plan_stats;
    uint32_t* sort_key;          // per slot: pathw (launch order: long walks first)
    // extension launch
    uint32_t* queue;             // next list entry to hand out
    uint8_t* lane_ws;
    uint32_t lane_ws_bytes, lane_pathw, n_lanes;
    uint32_t lane_stride;        // 1: every lane of a wavefront runs walks; 64: one walk per wavefront (the few very long walks: a lane-per-walk
                                 // wavefront advances at the pace of its slowest lane, and these are thousands of wide steps long)
    uint32_t leaves_in_lds;      // lane_stride 64 launches: the frontier's leaf buffers in LDS instead of the lane workspace
    uint32_t general_quorum_pct, general_max_wait;   // one-kernel form: lanes that need the general step wait for company (see wp_extend_kernel)
    uint32_t auto_dp;            // a failed walk with next == 0 goes to the DP stage in the same round
    WpDpItem* dp_items;
    uint32_t* n_dp_items;
    uint32_t dp_items_cap;
    // DP results of this round
    const DpRequest* dp_reqs;
    const DpMsaOut* dp_msa;
    const uint8_t* dp_cons;      // persistent copy of the stage's consensus buffer
    uint32_t n_dp;
    // stitch
    uint8_t* out_codes;
    uint32_t* piece_start;
    WpRequest* req_out;
    uint32_t* n_req_out;
    uint32_t req_cap;
    uint8_t* walk_log;
    DevCounters* ctr;
    unsigned long long* prof;    // LRSC_CORRECT_PROFILE: 16 tick totals of the extension kernel (per-lane wall ticks summed over lanes)
};

constexpr uint32_t kWpCoopLeaves = 8;   // wp_coop.hip: a helper lane's private frontier (its leaf + up to four children, with room to spare)
constexpr uint32_t kWpPathwSmall = 64, kWpPathwMid = 256;

// ---- the two-class schedule of the extension (wp_fast_kernel / wp_general_kernel) ---------------------------------------------
// A walk in flight owns a CONTEXT: its dynamic workspace (WpLaneLayout) headed by a WpCtx with the few scalars of the Walk
// object that change from step to step.  Between launches every walk in flight sits in one of three lists:
//   FAST     its frontier is one leaf: the next step is of the single-leaf kind (Walk::step_fast, leaf in registers)
//   GENERAL  anything else (several leaves, no accepted base, ...): the general step over the frontier in memory
//   FREE     contexts without a walk
// wp_fast_kernel's lanes pull FAST entries (then fresh walks onto FREE contexts) and step them until they leave the single-leaf
// regime, end, or use up a step budget; wp_general_kernel's lanes do the same for GENERAL entries.  Each kernel therefore runs
// ONE kind of step in all its lanes, instead of every wavefront paying for both kinds in every iteration.
struct WpCtx {
    uint32_t slot;
    uint32_t currentLength, currentKmerSize;
    uint32_t n_cur, n_results;
    uint32_t ring_free, path_free;
    uint32_t steps;
    uint32_t cur_is_small, ended;
    int32_t error;
    uint32_t pad;
};
struct WpList { uint32_t* items; uint32_t n, cursor; };
enum { kWpFastA = 0, kWpFastB = 1, kWpGenA = 2, kWpGenB = 3, kWpFreeA = 4, kWpFreeB = 5, kWpLists = 6 };
struct WpSched {
    WpList list[kWpLists];
    uint32_t fresh_cursor, n_fresh;      // fresh walks: entries [0, n_fresh) of WpSchedArgs::fresh
    uint32_t finished;
    uint32_t round;
};
struct WpSchedArgs {
    WpSched* sched;
    const uint32_t* fresh;               // slot indexes of the walks to run (launch order)
    uint8_t* ctx_ws;                     // n_ctx contexts of ctx_bytes each: WpCtx, then the WpLaneLayout regions (pathw = ctx_pathw)
    uint32_t ctx_bytes, ctx_pathw, n_ctx;
    uint32_t budget;                     // steps a lane gives one walk before it hands it back
    uint32_t quorum_pct;                 // idle lanes (in %) of a wavefront that wait for company before they pull their next walks
};

// bounds of every walk a read can be asked for (initCorrect's two loops, PacBioSelfCorrectionProcess.cpp:78-157): the source always
// ends where seed it-1 ends and the target is seed it + next, next < nextTarget
hipError_t launch_wp_bounds(const WpArgs& a, ReadPlan* plan, hipStream_t stream);
hipError_t launch_wp_plan(const WpArgs& a, hipStream_t stream);
hipError_t launch_wp_materialize(const WpArgs& a, hipStream_t stream);
hipError_t launch_wp_prepare(const FmIndexDev& fm, const WpArgs& a, hipStream_t stream);
hipError_t launch_wp_begin(const FmIndexDev& fm, const WpArgs& a, hipStream_t stream);
hipError_t launch_wp_extend_coop(const FmIndexDev& fm, const WpArgs& a, uint8_t* coop_ws, hipStream_t stream);   // wp_coop.hip
hipError_t launch_wp_extend(const FmIndexDev& fm, const WpArgs& a, hipStream_t stream);
// one round of the two-class schedule: fast kernel, list rotation, general kernel, list rotation
hipError_t launch_wp_sched_init(const WpSchedArgs& sa, uint32_t* list_storage, uint32_t n_fresh, hipStream_t stream);
hipError_t launch_wp_sched_round(const FmIndexDev& fm, const WpArgs& a, const WpSchedArgs& fast, const WpSchedArgs& general, uint32_t n_lanes_fast,
                                 uint32_t n_lanes_general, hipStream_t stream);
hipError_t launch_wp_dp_collect(const WpArgs& a, const WpDpItem* items, hipStream_t stream);
hipError_t launch_wp_stitch(const WpArgs& a, hipStream_t stream);
hipError_t launch_wp_gather(const WpArgs& a, const uint64_t* dst_off, char* dst, hipStream_t stream);
// exclusive prefix sums (hipCUB) of n 64-bit values in place; *total = sum (device scalar); tmp grows as needed
hipError_t wp_scan(uint64_t* v, uint64_t n, void** tmp, size_t* tmp_cap, hipStream_t stream);
// list[] = slot indexes ordered by sort_key descending (hipCUB radix sort)
hipError_t wp_sort_list(const uint32_t* keys, uint32_t* keys_tmp, uint32_t* list, uint32_t* list_tmp, uint32_t n, void** tmp, size_t* tmp_cap, hipStream_t stream);

} // namespace lrsc
