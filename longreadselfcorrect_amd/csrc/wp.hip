// wp.hip -- the walk-parallel correction flow (see wp.h for why it is exact).
//
//   wp_plan_kernel         lane per read (round 0) / per request (later rounds): identity + geometry of every walk, arena sizes
//   wp_materialize_kernel  wavefront per walk: m_query = source k-mer | raw read segment | target seed (reverse-complemented for a
//                          repeat-to-unique walk, PacBioSelfCorrectionProcess.cpp:176-184) and the forward DP query
//   wp_prepare_kernel      wavefront per walk, lane per query offset: bi-intervals of every 5-mer, 9-mer and target 13-mer
//                          (LongReadCorrectByOverlap.cpp:82-94,127-152) -- one k-mer table entry each
//   wp_begin_kernel        lane per walk: interval "trees" (introsort + chains), 5-mer chains, isTerminated filter, root interval
//   wp_extend_kernel       persistent lanes, each pulls walks from a queue and runs extendOverlap (.cpp:155-211, walk_device.h);
//                          a failed walk is handed to the DP stage of the same round
//   wp_dp_collect_kernel   DP answers -> slots
//   wp_stitch_kernel       wavefront per read: initCorrect's chain (PacBioSelfCorrectionProcess.cpp:78-152) over the finished
//                          walks; a walk whose assumed source k-mer is not the true tail of the accumulated string is re-queued
//
// This translation unit compiles walk_device.h with ALL of the walk inlined into its kernels (the two defines below).  With the
// big pieces of the walk as calls (the header's default, kept by extend.hip and wp_coop.hip) the Walk object's address escapes into
// every call as `this`, so the object lives in scratch memory and every field access is a memory round trip of its own -- for a
// lane that has nothing else in flight, most of a step's latency.  Counted on the ISA of wp_extend_kernel's call tree (narrow
// layout): 1 864 FLAT + 1 012 scratch memory instructions with calls, 633 + 364 all inline, for 16 % more instructions (29 k).
// Same source, same arithmetic in the same order: on the bench workload (4.79 M walks per two steps, both rank-block layouts) every
// counter down to the number of rank queries is identical, and the default flow goes from 78 to 93 corrected Mbases/s.
#define LRSC_WALK_FN __device__ __forceinline__
#define LRSC_WALK_NOINLINE
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "walk_device.h"
#include "wp.h"

namespace lrsc {

#ifndef LRSC_WP_EXTEND_OCC
#define LRSC_WP_EXTEND_OCC 2          // wavefronts per SIMD the extension kernel is compiled for (capi.cpp sizes its lanes to match)
#endif

// ---------------------------------------------------------------------------------------
// geometry of one FM attempt (correctByFMExtension, PacBioSelfCorrectionProcess.cpp:162-190)
// ---------------------------------------------------------------------------------------
struct WpGeom {
    int k, interval, trg_len, T_start, T_len;
    bool rtou, bad;
    uint32_t lq, pathw;
};
__device__ __forceinline__ WpGeom wp_geometry(const WpArgs& a, int S_seedLen, int S_end, int S_endBest, bool S_isRepeat, const int32_t* T)
{
    WpGeom g;
    g.T_start = T[0]; g.T_len = T[1];
    const bool T_isRepeat = (T[3] & 1) != 0;
    g.interval = g.T_start - S_end - 1;
    int k = (S_endBest < T[4] ? S_endBest : T[4]) - 2;                     // min(source.endBest, target.startBest) - 2
    if(S_isRepeat || T_isRepeat) {
        k = S_seedLen < g.T_len ? S_seedLen : g.T_len;
        k = k < a.start_kmer_len + 2 ? k : a.start_kmer_len + 2;
    }
    g.k = k;
    g.rtou = S_isRepeat && !T_isRepeat;
    g.trg_len = g.rtou ? k : g.T_len;
    g.bad = k < (int)a.seed_size || k > (int)kMaxInitK || k > S_seedLen || g.interval < 0 || g.trg_len < (int)a.min_overlap;
    g.lq = 0; g.pathw = 0;
    if(!g.bad) {
        g.lq = (uint32_t)(k + g.interval + g.trg_len);
        const double maxLength = (1.2 * (g.interval + 10)) + (double)(2 * (uint64_t)k);
        g.pathw = (uint32_t)(((uint64_t)maxLength + 4 + 15) / 16 + 1);
        if(g.lq >= 65535u) g.bad = true;
    }
    return g;
}

__device__ __forceinline__ uint32_t packed_char(uint64_t lo, uint64_t hi, uint32_t k, uint32_t t)     // t = 0: first character
{
    const uint32_t back = k - 1 - t;
    return back < 32 ? (uint32_t)(lo >> (2 * back)) & 3u : (uint32_t)(hi >> (2 * (back - 32))) & 3u;
}
__device__ __forceinline__ void pack_chars(const uint8_t* p, uint32_t k, uint64_t& lo, uint64_t& hi)
{
    lo = 0; hi = 0;
    for(uint32_t t = 0; t < k; ++t) { hi = (hi << 2) | (lo >> 62); lo = (lo << 2) | p[t]; }
}

__device__ __forceinline__ uint32_t al16(uint32_t x) { return (x + 15u) & ~15u; }

// bytes a slot's attempt needs in the three arenas
__device__ __forceinline__ void wp_sizes(const WpArgs& a, const WpSlot& s, uint32_t kind, uint64_t& q, uint64_t& prep, uint64_t& path)
{
    q = 0; prep = 0; path = 0;
    if(kind == kWpReqDp) { q = al16(s.dp_lq); return; }
    if(s.flags & kWpGeomBad) return;
    q = al16(s.lq) + ((s.rtou && s.next == 0) ? al16(s.dp_lq) : 0u);
    prep = wp_prep_layout(s.lq, s.trg_len, a.seed_size, a.min_overlap, a.psz).total;
    path = (uint64_t)s.pathw * 4u;
}

// ---------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wp_bounds_kernel(WpArgs a, ReadPlan* plan)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if(r >= a.n_reads) return;
    const uint64_t rs = a.read_off[r];
    const uint32_t n_seeds = a.seed_count[r];
    const int32_t* seeds = a.seeds + seed_slab(rs, r, a.min_k) * kSeedInts;
    uint32_t gap_max = 0, lq_max = 0;
    for(uint32_t it = 1; it < n_seeds; ++it) {
        const int s_end = seeds[(uint64_t)(it - 1) * kSeedInts] + seeds[(uint64_t)(it - 1) * kSeedInts + 1] - 1;
        for(int next = 0; next < a.next_target && it + (uint32_t)next < n_seeds; ++next) {
            const int32_t* T = seeds + (uint64_t)(it + (uint32_t)next) * kSeedInts;
            const int gap = T[0] - s_end - 1;
            if(gap < 0) continue;                               // the stitch pass reports it
            if((uint32_t)gap > gap_max) gap_max = (uint32_t)gap;
            const uint32_t lq = kMaxInitK + (uint32_t)gap + (uint32_t)T[1];
            if(lq > lq_max) lq_max = lq;
        }
    }
    plan[r].gap_max = gap_max;
    plan[r].lq_max = lq_max;
}

__global__ __launch_bounds__(256) void wp_plan_kernel(WpArgs a)
{
    const uint32_t r = a.r0 + blockIdx.x * 256 + threadIdx.x;
    if(r >= a.r1) return;
    const WpReadWork rw = a.work[r];
    if(rw.n_seeds < 2) return;
    const uint64_t rs = a.read_off[r];
    const uint8_t* read = a.codes + rs;
    const int32_t* seeds = a.seeds + seed_slab(rs, r, a.min_k) * kSeedInts;
    uint32_t n_mid = 0, n_big = 0, pw_max = 0;
    for(uint32_t it = 1; it < rw.n_seeds; ++it) {
        const int32_t* S = seeds + (uint64_t)(it - 1) * kSeedInts;
        const int32_t* T = seeds + (uint64_t)it * kSeedInts;
        const uint64_t si = rw.slot_first + it - 1;
        WpSlot& s = a.slots[si];
        // predicted source: seed it-1 as it stands in the read
        const WpGeom g = wp_geometry(a, S[1], S[0] + S[1] - 1, S[5], (S[3] & 1) != 0, T);
        s.read = r; s.it = it;
        s.k = (uint8_t)(g.bad ? 0 : g.k); s.next = 0; s.rtou = g.rtou ? 1 : 0;
        s.flags = g.bad ? (uint8_t)kWpGeomBad : 0;
        s.src_lo = 0; s.src_hi = 0;
        if(!g.bad) pack_chars(read + S[0] + S[1] - g.k, (uint32_t)g.k, s.src_lo, s.src_hi);
        s.dp_k = s.k; s.dp_src_lo = s.src_lo; s.dp_src_hi = s.src_hi;
        s.lq = g.lq; s.gap = g.bad ? 0u : (uint32_t)g.interval; s.trg_len = g.bad ? 0u : (uint32_t)g.trg_len; s.pathw = g.pathw;
        s.dp_lq = g.bad ? 0u : (uint32_t)(g.k + g.interval + g.T_len);
        s.dp_total_freq = S[2] + T[2];
        s.q = nullptr; s.dpq = nullptr; s.prep = nullptr; s.path = nullptr;
        s.code = 0; s.path_len = 0; s.match_i = 0; s.steps = 0;
        s.dp_rows = 0; s.dp_cons_len = 0; s.dp_error = 0; s.dp_cons = nullptr;
        const uint64_t li = si - a.slot_base;
        wp_sizes(a, s, kWpReqFm, a.sz_q[li], a.sz_prep[li], a.sz_path[li]);
        a.sort_key[li] = g.pathw;
        if(g.pathw > kWpPathwSmall) ++n_mid;
        if(g.pathw > kWpPathwMid) ++n_big;
        if(g.pathw > pw_max) pw_max = g.pathw;
    }
    if(n_mid) atomicAdd(&a.plan_stats[0], n_mid);
    if(n_big) atomicAdd(&a.plan_stats[1], n_big);
    atomicMax(&a.plan_stats[2], pw_max);
}

// later rounds: the stitch pass wrote the new identity and geometry into the slot; only the sizes are left
__global__ __launch_bounds__(256) void wp_plan_requests_kernel(WpArgs a)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if(i >= a.n_list) return;
    const WpSlot& s = a.slots[a.list[i]];
    const uint32_t kind = a.reqs[i].kind;
    wp_sizes(a, s, kind, a.sz_q[i], a.sz_prep[i], a.sz_path[i]);
    if(kind == kWpReqFm && !(s.flags & kWpGeomBad)) atomicMax(&a.plan_stats[2], s.pathw);
}

// ---------------------------------------------------------------------------------------
// materialize
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void wp_materialize_kernel(WpArgs a)
{
    const uint32_t lane = threadIdx.x;
    for(uint32_t i = blockIdx.x; i < a.n_list; i += gridDim.x) {
        const uint32_t si = a.list ? a.list[i] : (uint32_t)a.slot_base + i;
        WpSlot& s = a.slots[si];
        const uint32_t kind = a.reqs ? a.reqs[i].kind : (uint32_t)kWpReqFm;
        const uint64_t rs = a.read_off[s.read];
        const uint8_t* read = a.codes + rs;
        const int32_t* seeds = a.seeds + seed_slab(rs, s.read, a.min_k) * kSeedInts;
        const int32_t* S = seeds + (uint64_t)(s.it - 1) * kSeedInts;
        const int S_end = S[0] + S[1] - 1;
        if(kind == kWpReqDp) {
            // forward query of correctByMSAlignment: source k-mer | raw segment | target seed `it` (:219-223)
            const int32_t* T0 = seeds + (uint64_t)s.it * kSeedInts;
            uint8_t* dq = a.arena_q + a.sz_q[i];
            const uint32_t k = s.dp_k, iv = (uint32_t)(T0[0] - S_end - 1), tl = (uint32_t)T0[1];
            for(uint32_t t = lane; t < k; t += 64) dq[t] = (uint8_t)packed_char(s.dp_src_lo, s.dp_src_hi, k, t);
            for(uint32_t t = lane; t < iv; t += 64) dq[k + t] = read[S_end + 1 + t];
            for(uint32_t t = lane; t < tl; t += 64) dq[k + iv + t] = read[T0[0] + t];
            if(lane == 0) {
                s.dpq = dq;
                s.flags &= (uint8_t)~kWpDpValid;
                const uint32_t j = atomicAdd(a.n_dp_items, 1u);
                if(j < a.dp_items_cap) {
                    WpDpItem d; d.q = (uint64_t)dq; d.slot = si; d.lq = s.dp_lq; d.k = k; d.total_freq = s.dp_total_freq;
                    a.dp_items[j] = d;
                }
            }
            continue;
        }
        if(s.flags & kWpGeomBad) continue;
        const int32_t* T = seeds + (uint64_t)(s.it + s.next) * kSeedInts;
        const uint32_t k = s.k, iv = s.gap;
        const int T_start = T[0], T_len = T[1];
        uint8_t* q = a.arena_q + a.sz_q[i];
        if(!s.rtou) {
            for(uint32_t t = lane; t < k; t += 64) q[t] = (uint8_t)packed_char(s.src_lo, s.src_hi, k, t);
            for(uint32_t t = lane; t < iv; t += 64) q[k + t] = read[S_end + 1 + t];
            for(uint32_t t = lane; t < (uint32_t)T_len; t += 64) q[k + iv + t] = read[T_start + t];
        } else {
            // src <-> trg swapped and everything reverse-complemented: the walk starts from revcomp(target[0..k))
            for(uint32_t t = lane; t < k; t += 64) q[t] = (uint8_t)(3 - read[T_start + k - 1 - t]);
            for(uint32_t t = lane; t < iv; t += 64) q[k + t] = (uint8_t)(3 - read[S_end + iv - t]);
            for(uint32_t t = lane; t < k; t += 64) q[k + iv + t] = (uint8_t)(3u - packed_char(s.src_lo, s.src_hi, k, k - 1 - t));
        }
        uint8_t* dq = q;
        if(s.next == 0 && s.rtou) {
            dq = q + al16(s.lq);
            for(uint32_t t = lane; t < k; t += 64) dq[t] = (uint8_t)packed_char(s.src_lo, s.src_hi, k, t);
            for(uint32_t t = lane; t < iv; t += 64) dq[k + t] = read[S_end + 1 + t];
            for(uint32_t t = lane; t < (uint32_t)T_len; t += 64) dq[k + iv + t] = read[T_start + t];
        }
        if(lane == 0) {
            s.q = q;
            s.prep = a.arena_prep + a.sz_prep[i];
            s.path = reinterpret_cast<uint32_t*>(a.arena_path + a.sz_path[i]);
            s.flags &= (uint8_t)~kWpFmValid;
            if(s.next == 0) {
                s.dpq = dq;
                s.dp_k = s.k; s.dp_src_lo = s.src_lo; s.dp_src_hi = s.src_hi;
                s.flags &= (uint8_t)~kWpDpValid;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// prepare + begin
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__device__ __forceinline__ void wp_bind_static(Walk<WIDE>& W, const WpArgs& a, const WpSlot& s)
{
    using P = typename Lay<WIDE>::pos_t;
    const WpPrepLayout L = wp_prep_layout(s.lq, s.trg_len, a.seed_size, a.min_overlap, a.psz);
    uint8_t* ws = s.prep;
    W.q = s.q;
    W.Lq = s.lq; W.initk = s.k; W.path_len = s.gap; W.trg_len = s.trg_len; W.dis = (int32_t)s.gap;
    W.it9f = reinterpret_cast<SortItem*>(ws + L.item9f);
    W.it9r = reinterpret_cast<SortItem*>(ws + L.item9r);
    W.next9f = reinterpret_cast<uint16_t*>(ws + L.next9f);
    W.next9r = reinterpret_cast<uint16_t*>(ws + L.next9r);
    W.head9f = reinterpret_cast<uint16_t*>(ws + L.head9);
    W.head9r = W.head9f + 256;
    W.head5 = reinterpret_cast<uint16_t*>(ws + L.head5);
    W.next5 = reinterpret_cast<uint16_t*>(ws + L.next5);
    W.flags5 = ws + L.flags5;
    W.term = reinterpret_cast<const P*>(ws + L.term);
    W.n_term = s.trg_len >= a.min_overlap ? s.trg_len - a.min_overlap + 1 : 0;
}

template <bool WIDE>
__global__ __launch_bounds__(64) void wp_prepare_kernel(FmIndexDev fm, WpArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
    const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
    uint32_t cnt_rank = 0, cnt_blk = 0;
    for(uint32_t i = blockIdx.x; i < a.n_list; i += gridDim.x) {
        if(a.reqs && a.reqs[i].kind != kWpReqFm) continue;
        const WpSlot& s = a.slots[a.list ? a.list[i] : (uint32_t)a.slot_base + i];
        if(s.flags & kWpGeomBad) continue;
        const WpPrepLayout L = wp_prep_layout(s.lq, s.trg_len, a.seed_size, a.min_overlap, a.psz);
        uint8_t* ws = s.prep;
        for(uint32_t o = threadIdx.x; o < s.lq; o += 64)
            prepare_offset<WIDE>(fm, sf, sr, mtab, s.q, o, s.lq, (uint32_t)s.k + s.gap, a.seed_size, a.min_overlap,
                                 reinterpret_cast<SortItem*>(ws + L.item9f), reinterpret_cast<SortItem*>(ws + L.item9r), ws + L.flags5,
                                 reinterpret_cast<P*>(ws + L.term), cnt_rank, cnt_blk);
    }
    flush_counters(a.ctr, cnt_rank, cnt_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(64) void wp_begin_kernel(FmIndexDev fm, WpArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(i < a.n_list && !(a.reqs && a.reqs[i].kind != kWpReqFm)) {
        const WpSlot& s = a.slots[a.list ? a.list[i] : (uint32_t)a.slot_base + i];
        if(!(s.flags & kWpGeomBad)) {
            Walk<WIDE> W;
            W.sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
            W.sR = strand_consts<P>(fm.strand[LRSC_BWT]);
            W.fm = &fm; W.mtab = mtab;
            W.seedSize = a.seed_size; W.minOverlap = a.min_overlap;
            W.n_rank = 0; W.n_blk = 0; W.prof = nullptr;
            wp_bind_static<WIDE>(W, a, s);
            W.begin_static();
            Leaf<P> root;
            root.suf_lo = 0; root.suf_hi = 0;
            for(uint32_t t = 0; t < s.k; ++t) suf_push(root, s.q[t]);
            W.find_suffix(root, s.k);
            WpStatic* H = reinterpret_cast<WpStatic*>(s.prep);
            H->root[0] = root.flo; H->root[1] = root.fhi; H->root[2] = root.rlo; H->root[3] = root.rhi;
            H->tmask0 = W.tmask0; H->tmask1 = W.tmask1; H->n9f = W.n9f; H->n9r = W.n9r;
            n_rank = W.n_rank; n_blk = W.n_blk;
        }
    }
    flush_counters(a.ctr, n_rank, n_blk);
}

// ---------------------------------------------------------------------------------------
// extend: persistent lanes over a queue of walks
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(64, LRSC_WP_EXTEND_OCC) void wp_extend_kernel(FmIndexDev fm, WpArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint32_t stride = a.lane_stride ? a.lane_stride : 1u;
    const bool owner = (threadIdx.x % stride) == 0;
    const uint32_t lane_id = owner ? (blockIdx.x * 64 + threadIdx.x) / stride : 0xFFFFFFFFu;
    Walk<WIDE> W;
    W.sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
    W.sR = strand_consts<P>(fm.strand[LRSC_BWT]);
    W.fm = &fm; W.mtab = mtab;
    W.seedSize = a.seed_size; W.minOverlap = a.min_overlap; W.maxLeaves = a.max_leaves;
    W.PBcoverage = a.pb_coverage; W.PacBioErrorRate = a.pacbio_error_rate; W.errorRate = 0.25; W.localK = 100;
    W.freqsOfKmerSize = a.freqs_of_kmer_size;
    const WpLaneLayout LL = wp_lane_layout(a.lbytes, a.lane_pathw);
    uint8_t* lws = a.lane_ws + (uint64_t)(owner ? lane_id : 0u) * a.lane_ws_bytes;
    // one walk per wavefront (lane_stride 64): the frontier's leaves live in LDS -- the general step is mostly leaf bookkeeping, and
    // these launches are the walks with thousands of wide steps (every access here is a FLAT one, so the code is the same)
    extern __shared__ __attribute__((aligned(16))) uint8_t wp_dyn_lds[];
    Leaf<P>* const leaf_base = stride == 64u && a.leaves_in_lds ? reinterpret_cast<Leaf<P>*>(wp_dyn_lds) : reinterpret_cast<Leaf<P>*>(lws + LL.leaves);
    W.rings = reinterpret_cast<double*>(lws + LL.rings);
    W.results = reinterpret_cast<WalkResultRec*>(lws + LL.results);
    W.paths = reinterpret_cast<uint32_t*>(lws + LL.paths);
    W.pathw = a.lane_pathw;
    W.rpaths = W.paths + (uint64_t)32 * a.lane_pathw;
    W.n_rank = 0; W.n_blk = 0; W.steps = 0; W.leaf_steps = 0; W.error = 0; W.cyc_setup = 0; W.cyc_loop = 0; W.prof = nullptr; W.profile = false;
    const uint64_t min_SA = a.pb_coverage > 60 ? (uint64_t)((a.pb_coverage / 60) * 3) : 3;

    bool in_walk = false;
    uint32_t si = 0;
    uint64_t steps0 = 0;
    // single-leaf fast path (walk_device.h): the one leaf of the frontier in registers
    Leaf<P> L;
    uint32_t pw = 0;
    bool fast = false;
    uint32_t gen_wait = 0;
    bool want_gen = false;
    // profiling (a.prof): per-lane wall ticks inside the step's regions (Walk::tock slots 0-6), 8 = refill, 9 = finish, 10 = whole loop
    uint64_t pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t t_refill = 0, t_finish = 0, n_fast = 0;
    const bool prof = a.prof != nullptr;
    if(prof) W.prof = pr;
    const uint64_t t_loop0 = prof ? __builtin_readcyclecounter() : 0;
    if(lane_id < a.n_lanes)
    while(true) {
        if(!in_walk) {
            const uint64_t tr0 = prof ? __builtin_readcyclecounter() : 0;
            const uint32_t i = atomicAdd(a.queue, 1u);
            if(i >= a.n_list) break;
            if(a.reqs && a.reqs[i].kind != kWpReqFm) continue;
            si = a.list ? a.list[i] : (uint32_t)a.slot_base + i;
            const WpSlot& s = a.slots[si];
            if(s.flags & kWpGeomBad) continue;
            wp_bind_static<WIDE>(W, a, s);
            const WpStatic* H = reinterpret_cast<const WpStatic*>(s.prep);
            W.n9f = H->n9f; W.n9r = H->n9r; W.tmask0 = H->tmask0; W.tmask1 = H->tmask1;
            W.maxOverlap = (uint32_t)s.k + 2;
            W.min_SA_threshold = min_SA;
            // .cpp:55-58,78-79: double expressions truncated to size_t
            if((int32_t)s.gap > 100) W.maxIndelSize = (uint64_t)((int32_t)s.gap * 0.2); else W.maxIndelSize = 20;
            W.maxLength = (uint64_t)((1.2 * ((int32_t)s.gap + 10)) + (double)(2 * (uint64_t)s.k));
            W.minLength = (uint64_t)((0.8 * ((int32_t)s.gap - 20)) + (double)(2 * (uint64_t)s.k));
            W.cur = leaf_base; W.nxt = leaf_base + 32; W.leaf_small = leaf_base;
            W.error = 0;
            steps0 = W.steps; W.leaf_steps = 0; W.max_front = 1;
            const P riv[4] = {(P)H->root[0], (P)H->root[1], (P)H->root[2], (P)H->root[3]};
            W.begin_root(riv);
            in_walk = true;
            fast = false; want_gen = false; gen_wait = 0;
            if(prof) t_refill += __builtin_readcyclecounter() - tr0;
        }
        if(!fast && !want_gen && W.can_fast()) { W.enter_fast(L, pw); fast = true; }
        int r = 2;
        if(fast) {
            const uint64_t tq = prof ? __builtin_readcyclecounter() : 0;
            r = W.step_fast(L, pw);
            if(r != 1) fast = false;
            if(prof) { pr[7] += __builtin_readcyclecounter() - tq; if(r == 1) ++n_fast; }
        }
        const uint32_t n_fast_now = (uint32_t)__builtin_popcountll(__ballot(r == 1));     // lanes still in the single-leaf regime
        if(r == 2) {
            // The general step stalls every lane of the wavefront that is in the single-leaf regime, and costs the same whether
            // one lane needs it or thirty: a lane that needs it waits (a few iterations at most) until a share of the wavefront's
            // busy lanes does, so that the wavefront pays for it once for many lanes instead of in every iteration.
            const uint32_t n_need = (uint32_t)__builtin_popcountll(__ballot(true));
            const uint32_t n_busy = n_need + n_fast_now;
            // (one decision for all of them: when the share is there, or when one of them has waited its limit)
            const bool waited = __ballot(gen_wait >= a.general_max_wait) != 0;
            if(n_need * 100u < n_busy * a.general_quorum_pct && !waited) { ++gen_wait; want_gen = true; continue; }
            gen_wait = 0; want_gen = false;
            r = W.step() ? 1 : 0;
        }
        if(r == 1) continue;
        in_walk = false;
        const uint64_t tf0 = prof ? __builtin_readcyclecounter() : 0;
        WpSlot& s = a.slots[si];
        uint32_t plen = 0, mi = 0;
        const int code = W.finish(&plen, s.path, &mi);
        s.code = code; s.path_len = plen; s.match_i = mi; s.steps = (uint32_t)(W.steps - steps0); s.leaf_steps = W.leaf_steps; s.max_front = (uint8_t)W.max_front;
        s.flags |= (uint8_t)kWpFmValid;
        if(code <= 0 && code > LRSC_WALK_ERR_CHILDREN && a.auto_dp && s.next == 0) {
            const uint32_t j = atomicAdd(a.n_dp_items, 1u);
            if(j < a.dp_items_cap) {
                WpDpItem d; d.q = (uint64_t)s.dpq; d.slot = si; d.lq = s.dp_lq; d.k = s.dp_k; d.total_freq = s.dp_total_freq;
                a.dp_items[j] = d;
            }
        }
        if(prof) t_finish += __builtin_readcyclecounter() - tf0;
    }
    if(prof && lane_id < a.n_lanes) {
        for(int j = 0; j < 8; ++j) atomicAdd(&a.prof[j], (unsigned long long)pr[j]);
        atomicAdd(&a.prof[8], (unsigned long long)t_refill);
        atomicAdd(&a.prof[9], (unsigned long long)t_finish);
        atomicAdd(&a.prof[10], (unsigned long long)(__builtin_readcyclecounter() - t_loop0));
        atomicAdd(&a.prof[11], (unsigned long long)W.steps);
        atomicAdd(&a.prof[12], (unsigned long long)n_fast);
    }
    flush_counters(a.ctr, W.n_rank, W.n_blk);
}

// ---------------------------------------------------------------------------------------
// the two-class schedule (wp.h): wp_fast_kernel steps single-leaf frontiers, wp_general_kernel everything else
// ---------------------------------------------------------------------------------------
constexpr uint32_t kWpCtxHeader = 64;

// One atomic per wavefront instead of one per lane: the lanes that `want` a ticket of the counter get consecutive ones
// (a hundred thousand lanes hammering six list counters serialise in the L2 otherwise).  Call with any subset of lanes active.
__device__ __forceinline__ uint32_t wave_claim(uint32_t* ctr, bool want)
{
    const uint64_t m = __ballot(want);
    if(m == 0) return 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if(lane == leader) base = atomicAdd(ctr, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, (int)leader, 64);
    return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ void wp_push(WpSched* S, int which, uint32_t c, bool want = true)
{
    const uint32_t k = wave_claim(&S->list[which].n, want);
    if(want) S->list[which].items[k] = c;
}

// constants of the Walk object that do not depend on the walk
template <bool WIDE>
__device__ __forceinline__ void wp_walk_consts(Walk<WIDE>& W, const FmIndexDev& fm, const WpArgs& a, const uint32_t* mtab)
{
    using P = typename Lay<WIDE>::pos_t;
    W.sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
    W.sR = strand_consts<P>(fm.strand[LRSC_BWT]);
    W.fm = &fm; W.mtab = mtab;
    W.seedSize = a.seed_size; W.minOverlap = a.min_overlap; W.maxLeaves = a.max_leaves;
    W.PBcoverage = a.pb_coverage; W.PacBioErrorRate = a.pacbio_error_rate; W.errorRate = 0.25; W.localK = 100;
    W.freqsOfKmerSize = a.freqs_of_kmer_size;
    W.n_rank = 0; W.n_blk = 0; W.steps = 0; W.leaf_steps = 0; W.max_front = 1; W.error = 0; W.cyc_setup = 0; W.cyc_loop = 0; W.prof = nullptr; W.profile = false;
    W.n_nxt = 0; W.ended = false;
}

// the per-walk part: tables of the prepared state, geometry, the context's dynamic regions
template <bool WIDE>
__device__ __forceinline__ void wp_walk_bind(Walk<WIDE>& W, const WpArgs& a, const WpSlot& s, uint8_t* dyn, uint32_t pathw, const WpStatic* H)
{
    using P = typename Lay<WIDE>::pos_t;
    wp_bind_static<WIDE>(W, a, s);
    W.n9f = H->n9f; W.n9r = H->n9r; W.tmask0 = H->tmask0; W.tmask1 = H->tmask1;
    W.maxOverlap = (uint32_t)s.k + 2;
    W.min_SA_threshold = a.pb_coverage > 60 ? (uint64_t)((a.pb_coverage / 60) * 3) : 3;
    // .cpp:55-58,78-79: double expressions truncated to size_t
    if((int32_t)s.gap > 100) W.maxIndelSize = (uint64_t)((int32_t)s.gap * 0.2); else W.maxIndelSize = 20;
    W.maxLength = (uint64_t)((1.2 * ((int32_t)s.gap + 10)) + (double)(2 * (uint64_t)s.k));
    W.minLength = (uint64_t)((0.8 * ((int32_t)s.gap - 20)) + (double)(2 * (uint64_t)s.k));
    const WpLaneLayout LL = wp_lane_layout(a.lbytes, pathw);
    W.leaf_small = reinterpret_cast<Leaf<P>*>(dyn + LL.leaves);
    W.rings = reinterpret_cast<double*>(dyn + LL.rings);
    W.results = reinterpret_cast<WalkResultRec*>(dyn + LL.results);
    W.paths = reinterpret_cast<uint32_t*>(dyn + LL.paths);
    W.pathw = pathw;
    W.rpaths = W.paths + (uint64_t)32 * pathw;
}

template <bool WIDE>
__device__ __forceinline__ void wp_ctx_load(Walk<WIDE>& W, const WpCtx& C)
{
    W.currentLength = C.currentLength; W.currentKmerSize = C.currentKmerSize;
    W.n_cur = C.n_cur; W.n_results = C.n_results; W.n_nxt = 0;
    W.ring_free = C.ring_free; W.path_free = C.path_free;
    W.steps = C.steps; W.leaf_steps = C.pad; W.ended = C.ended != 0; W.error = C.error;
    if(C.cur_is_small) { W.cur = W.leaf_small; W.nxt = W.leaf_small + 32; }
    else { W.cur = W.leaf_small + 32; W.nxt = W.leaf_small; }
}
template <bool WIDE>
__device__ __forceinline__ void wp_ctx_save(const Walk<WIDE>& W, WpCtx& C, uint32_t slot)
{
    C.slot = slot;
    C.currentLength = (uint32_t)W.currentLength; C.currentKmerSize = (uint32_t)W.currentKmerSize;
    C.n_cur = W.n_cur; C.n_results = W.n_results;
    C.ring_free = W.ring_free; C.path_free = W.path_free;
    C.steps = (uint32_t)W.steps; C.pad = W.leaf_steps; C.cur_is_small = W.cur == W.leaf_small ? 1u : 0u; C.ended = W.ended ? 1u : 0u; C.error = W.error;
}

// extendOverlap is over: findTheBestPath / the failure code, the slot's result, the DP stage's work item
template <bool WIDE>
__device__ __noinline__ void wp_walk_finish(Walk<WIDE>& W, const WpArgs& a, uint32_t si)
{
    WpSlot& s = a.slots[si];
    uint32_t plen = 0, mi = 0;
    const int code = W.finish(&plen, s.path, &mi);
    s.code = code; s.path_len = plen; s.match_i = mi; s.steps = (uint32_t)W.steps; s.leaf_steps = W.leaf_steps; s.max_front = (uint8_t)W.max_front;
    s.flags |= (uint8_t)kWpFmValid;
    if(code <= 0 && code > LRSC_WALK_ERR_CHILDREN && a.auto_dp && s.next == 0) {
        const uint32_t j = atomicAdd(a.n_dp_items, 1u);
        if(j < a.dp_items_cap) {
            WpDpItem d; d.q = (uint64_t)s.dpq; d.slot = si; d.lq = s.dp_lq; d.k = s.dp_k; d.total_freq = s.dp_total_freq;
            a.dp_items[j] = d;
        }
    }
}

template <bool WIDE>
__global__ __launch_bounds__(64, 2) void wp_fast_kernel(FmIndexDev fm, WpArgs a, WpSchedArgs sa)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    WpSched* S = sa.sched;
    const uint32_t odd = S->round & 1u;
    const int Fin = odd ? kWpFastB : kWpFastA, Fout = odd ? kWpFastA : kWpFastB;
    const int Gin = odd ? kWpGenB : kWpGenA;
    const int FRin = odd ? kWpFreeB : kWpFreeA, FRout = odd ? kWpFreeA : kWpFreeB;
    const uint32_t n_fin = S->list[Fin].n, n_free = S->list[FRin].n, n_fresh = S->n_fresh;
    Walk<WIDE> W;
    wp_walk_consts<WIDE>(W, fm, a, mtab);
    Leaf<P> L;
    uint32_t pw = 0;
    bool have = false;
    uint32_t c = 0, si = 0, budget = 0;
    while(true) {
        if(!have) {
            // lanes between walks wait for company: pulling a walk stalls the stepping lanes of the wavefront
            const uint32_t n_all = (uint32_t)__builtin_popcountll(__ballot(true));
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(__ballot(!have));
            if(n_idle * 100u < n_all * sa.quorum_pct) continue;
            bool fresh = false;
            // resumed walks first, then fresh walks onto free contexts
            const bool fin_open = S->list[Fin].cursor < n_fin;           // stale reads only cost an extra claim
            uint32_t i = fin_open ? wave_claim(&S->list[Fin].cursor, true) : n_fin;
            bool got = i < n_fin;
            if(got) c = S->list[Fin].items[i];
            {
                const uint32_t j = wave_claim(&S->list[FRin].cursor, !got);
                const bool got_ctx = !got && j < n_free;
                if(got_ctx) c = S->list[FRin].items[j];
                const uint32_t f = wave_claim(&S->fresh_cursor, got_ctx);
                bool got_fresh = got_ctx && f < n_fresh;
                // an entry that is not a walk to run (a DP request of a later round, a slot with a bad geometry) is done at once
                bool skip = false;
                if(got_fresh) {
                    si = sa.fresh[f];
                    skip = (a.reqs && a.reqs[f].kind != kWpReqFm) || (a.slots[si].flags & kWpGeomBad) != 0;
                }
                wp_push(S, FRout, c, got_ctx && (!got_fresh || skip));    // the context goes back unused
                (void)wave_claim(&S->finished, skip);
                if(skip) continue;
                if(got_fresh) { fresh = true; got = true; }
            }
            if(!got) break;                                               // nothing left for this lane in this launch
            uint8_t* base = sa.ctx_ws + (uint64_t)c * sa.ctx_bytes;
            WpCtx& C = *reinterpret_cast<WpCtx*>(base);
            if(!fresh) si = C.slot;
            const WpSlot& s = a.slots[si];
            const WpStatic* H = reinterpret_cast<const WpStatic*>(s.prep);
            wp_walk_bind<WIDE>(W, a, s, base + kWpCtxHeader, sa.ctx_pathw, H);
            if(fresh) {
                W.cur = W.leaf_small; W.nxt = W.leaf_small + 32;
                W.error = 0; W.steps = 0; W.leaf_steps = 0; W.ended = false;
                const P riv[4] = {(P)H->root[0], (P)H->root[1], (P)H->root[2], (P)H->root[3]};
                W.begin_root(riv);
            } else
                wp_ctx_load<WIDE>(W, C);
            W.enter_fast(L, pw);
            have = true;
            budget = sa.budget;
        }
        const int r = W.step_fast(L, pw);
        --budget;
        if(r == 1 && budget != 0) continue;
        have = false;
        WpCtx& C = *reinterpret_cast<WpCtx*>(sa.ctx_ws + (uint64_t)c * sa.ctx_bytes);
        if(r == 1) W.leave_fast(L);
        if(r != 0) wp_ctx_save<WIDE>(W, C, si);
        else wp_walk_finish<WIDE>(W, a, si);
        wp_push(S, Fout, c, r == 1);
        wp_push(S, Gin, c, r == 2);
        wp_push(S, FRout, c, r == 0);
        (void)wave_claim(&S->finished, r == 0);
    }
    flush_counters(a.ctr, W.n_rank, W.n_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(64, 2) void wp_general_kernel(FmIndexDev fm, WpArgs a, WpSchedArgs sa)
{
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    WpSched* S = sa.sched;
    const uint32_t odd = S->round & 1u;
    const int Fout = odd ? kWpFastA : kWpFastB;
    const int Gin = odd ? kWpGenB : kWpGenA, Gout = odd ? kWpGenA : kWpGenB;
    const int FRin = odd ? kWpFreeB : kWpFreeA, FRout = odd ? kWpFreeA : kWpFreeB;
    // contexts the fast kernel did not hand out stay free
    {
        const uint32_t n_free = S->list[FRin].n, cur = S->list[FRin].cursor;
        const uint32_t first = cur < n_free ? cur : n_free;
        for(uint32_t i = first + blockIdx.x * 64 + threadIdx.x; i < n_free; i += gridDim.x * 64) wp_push(S, FRout, S->list[FRin].items[i]);
    }
    const uint32_t n_gin = S->list[Gin].n;
    Walk<WIDE> W;
    wp_walk_consts<WIDE>(W, fm, a, mtab);
    bool have = false;
    uint32_t c = 0, si = 0, budget = 0;
    while(true) {
        if(!have) {
            const uint32_t n_all = (uint32_t)__builtin_popcountll(__ballot(true));
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(__ballot(!have));
            if(n_idle * 100u < n_all * sa.quorum_pct) continue;
            const uint32_t i = wave_claim(&S->list[Gin].cursor, true);
            if(i >= n_gin) break;
            c = S->list[Gin].items[i];
            uint8_t* base = sa.ctx_ws + (uint64_t)c * sa.ctx_bytes;
            const WpCtx& C = *reinterpret_cast<const WpCtx*>(base);
            si = C.slot;
            const WpSlot& s = a.slots[si];
            wp_walk_bind<WIDE>(W, a, s, base + kWpCtxHeader, sa.ctx_pathw, reinterpret_cast<const WpStatic*>(s.prep));
            wp_ctx_load<WIDE>(W, C);
            have = true;
            budget = sa.budget;
        }
        budget = budget > W.n_cur ? budget - W.n_cur : 0u;            // the budget counts leaf-steps: a wide frontier hands back sooner
        const bool go = W.step();
        if(go && !W.can_fast() && budget != 0) continue;
        have = false;
        WpCtx& C = *reinterpret_cast<WpCtx*>(sa.ctx_ws + (uint64_t)c * sa.ctx_bytes);
        if(!go) wp_walk_finish<WIDE>(W, a, si);
        else wp_ctx_save<WIDE>(W, C, si);
        const bool to_fast = go && W.can_fast();
        wp_push(S, FRout, c, !go);
        (void)wave_claim(&S->finished, !go);
        wp_push(S, Fout, c, to_fast);
        wp_push(S, Gout, c, go && !to_fast);
    }
    flush_counters(a.ctr, W.n_rank, W.n_blk);
}

// list rotation between the launches of a round: phase 0 after the fast kernel, phase 1 after the general kernel
__global__ void wp_sched_advance_kernel(WpSched* S, int phase)
{
    const uint32_t odd = S->round & 1u;
    if(phase == 0) {
        WpList& F = S->list[odd ? kWpFastB : kWpFastA];
        F.n = 0; F.cursor = 0;
    } else {
        WpList& G = S->list[odd ? kWpGenB : kWpGenA];
        G.n = 0; G.cursor = 0;
        WpList& R = S->list[odd ? kWpFreeB : kWpFreeA];
        R.n = 0; R.cursor = 0;
        S->round += 1;
    }
}

__global__ __launch_bounds__(256) void wp_sched_init_kernel(WpSchedArgs sa, uint32_t* storage, uint32_t n_fresh)
{
    WpSched* S = sa.sched;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if(gid == 0) {
        for(int l = 0; l < kWpLists; ++l) { S->list[l].items = storage + (uint64_t)l * sa.n_ctx; S->list[l].n = 0; S->list[l].cursor = 0; }
        S->list[kWpFreeA].n = sa.n_ctx;
        S->fresh_cursor = 0; S->n_fresh = n_fresh; S->finished = 0; S->round = 0;
    }
    uint32_t* free_a = storage + (uint64_t)kWpFreeA * sa.n_ctx;
    for(uint32_t i = gid; i < sa.n_ctx; i += gridDim.x * 256) free_a[i] = i;
}

// ---------------------------------------------------------------------------------------
// DP answers -> slots
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wp_dp_collect_kernel(WpArgs a, const WpDpItem* items)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if(i >= a.n_dp) return;
    WpSlot& s = a.slots[items[i].slot];
    const DpMsaOut m = a.dp_msa[i];
    s.dp_rows = m.n_rows; s.dp_cons_len = m.cons_len; s.dp_error = m.error;
    s.dp_cons = a.dp_cons + a.dp_reqs[i].cons_off;
    s.flags |= (uint8_t)kWpDpValid;
}

// ---------------------------------------------------------------------------------------
// stitch: initCorrect's chain over the finished walks (one wavefront per read, scalar control flow replicated in every lane,
// copies spread over the lanes)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void wp_stitch_kernel(WpArgs a)
{
    const uint32_t r = a.r0 + blockIdx.x, lane = threadIdx.x;
    const WpReadWork rw = a.work[r];
    WpRead& R = a.reads[r];
    if(R.started && R.state == kReadDone) return;
    const uint64_t rs = a.read_off[r];
    const uint32_t rlen = (uint32_t)(a.read_off[r + 1] - rs);
    const uint8_t* read = a.codes + rs;
    const uint32_t n_seeds = rw.n_seeds;
    const int32_t* seeds = a.seeds + seed_slab(rs, r, a.min_k) * kSeedInts;
    uint8_t* out = a.out_codes + rw.out_off;
    uint8_t* wl = a.walk_log ? a.walk_log + seed_slab(rs, r, a.min_k) : nullptr;
    uint32_t* piece_start = a.piece_start + rw.piece_off;

    int64_t correctedLen = 0, totalWalkNum = 0, highErrorNum = 0, exceedDepthNum = 0, exceedLeaveNum = 0, FMNum = 0, DPNum = 0, seedDis = 0;
    uint64_t steps = 0;
    uint32_t out_len = 0, n_pieces = 0, it = 1;
    int error = 0, next = 0, firstType = 0;
    int S_seedLen = 0, S_end = 0, S_endBest = 0, S_maxFixed = 0;
    bool S_isRepeat = false;
    uint32_t state = kReadDone;
    if(R.started) {
        correctedLen = R.c[1]; totalWalkNum = R.c[3]; highErrorNum = R.c[4]; exceedDepthNum = R.c[5]; exceedLeaveNum = R.c[6];
        FMNum = R.c[7]; DPNum = R.c[8]; seedDis = R.c[9]; steps = R.steps;
        out_len = R.out_len; n_pieces = R.n_pieces; it = R.it; next = R.next; firstType = R.first_type;
        S_seedLen = R.s_seed_len; S_end = R.s_end; S_endBest = R.s_end_best; S_maxFixed = R.s_max_fixed; S_isRepeat = R.s_is_repeat != 0;
    } else if(n_seeds >= 2) {
        // pieceVec.push_back(seedVec[0])
        if(lane == 0) piece_start[0] = 0;
        n_pieces = 1;
        for(int t = (int)lane; t < seeds[1]; t += 64) out[t] = read[seeds[0] + t];
        out_len = (uint32_t)seeds[1];
        S_seedLen = seeds[1]; S_end = seeds[0] + seeds[1] - 1; S_endBest = seeds[5]; S_maxFixed = seeds[2];
        S_isRepeat = (seeds[3] & 1) != 0;
    }
    auto request = [&](uint32_t si, uint32_t kind) {
        if(lane == 0) {
            const uint32_t j = atomicAdd(a.n_req_out, 1u);
            if(j < a.req_cap) { a.req_out[j].slot = si; a.req_out[j].kind = kind; }
        }
        state = kReadParked;
    };
    if(n_seeds >= 2)
    while(it < n_seeds && !error) {
        const int32_t* T = seeds + (uint64_t)(it + (uint32_t)next) * kSeedInts;
        const WpGeom g = wp_geometry(a, S_seedLen, S_end, S_endBest, S_isRepeat, T);
        if(g.bad) { error = g.lq >= 65535u ? LRSC_WALK_ERR_GEOMETRY : LRSC_WALK_ERR_GEOMETRY; break; }
        const int k = g.k, interval = g.interval, trg_len = g.trg_len;
        __threadfence_block();                                   // the characters other lanes appended are visible from here on
        uint64_t src_lo, src_hi;
        pack_chars(out + out_len - k, (uint32_t)k, src_lo, src_hi);       // source.seedStr.substr(seedLen - k)
        const uint64_t si = rw.slot_first + it - 1;
        WpSlot& s = a.slots[si];
        const bool fm_ok = (s.flags & kWpFmValid) && !(s.flags & kWpGeomBad) && s.k == (uint8_t)k && s.next == (uint8_t)next &&
                           s.src_lo == src_lo && s.src_hi == src_hi;
        if(!fm_ok) {
            // not computed for this source yet (a misprediction, or the inner `next` loop moved on): re-queue with the true identity
            if(lane == 0) {
                s.k = (uint8_t)k; s.next = (uint8_t)next; s.rtou = g.rtou ? 1 : 0; s.src_lo = src_lo; s.src_hi = src_hi;
                s.flags &= (uint8_t)~(kWpFmValid | kWpGeomBad);
                s.lq = g.lq; s.gap = (uint32_t)interval; s.trg_len = (uint32_t)trg_len; s.pathw = g.pathw;
                if(next == 0) { s.dp_lq = (uint32_t)(k + interval + g.T_len); s.dp_total_freq = S_maxFixed + T[2]; }
            }
            request((uint32_t)si, kWpReqFm);
            break;
        }
        const int code = s.code;
        if(code <= LRSC_WALK_ERR_CHILDREN) { error = code; break; }
        if(next == 0) firstType = code;
        if(code > 0) {
            // merged = path + target.substr(i + minOverlap); out = merged (un-reversed) minus its first k characters (:194-201)
            const uint32_t plen = s.path_len, mi = s.match_i;
            const uint32_t* best = s.path;
            const uint8_t* q = s.q;
            const uint32_t tail_from = mi + a.min_overlap;
            const uint32_t tlen = (uint32_t)trg_len - tail_from;
            const uint32_t M = plen + tlen;
            uint32_t appended = 0;
            if(!g.rtou) {
                appended = M - (uint32_t)k;
                if(out_len + appended > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                for(uint32_t j = (uint32_t)k + lane; j < M; j += 64)
                    out[out_len + j - k] = (uint8_t)(j < plen ? path_get(best, j) : q[k + interval + tail_from + (j - plen)]);
            } else {
                // revcomp(merged) + target.substr(k), minus the first k characters (:195-200)
                const uint32_t total = M + (uint32_t)(g.T_len - k);
                appended = total - (uint32_t)k;
                if(out_len + appended > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                for(uint32_t j = (uint32_t)k + lane; j < total; j += 64) {
                    uint8_t c;
                    if(j < M) {
                        const uint32_t m = M - 1 - j;                       // index into merged
                        c = (uint8_t)(3 - (m < plen ? path_get(best, m) : q[k + interval + tail_from + (m - plen)]));
                    } else
                        c = read[g.T_start + k + (j - M)];
                    out[out_len + j - k] = c;
                }
            }
            out_len += appended;
            correctedLen += appended;
            seedDis += interval;
            FMNum++;
            totalWalkNum++;
            steps += s.steps;
            S_seedLen += (int)appended;                                      // SeedFeature::append
            S_end = g.T_start + g.T_len - 1; S_endBest = T[5]; S_isRepeat = (T[3] & 1) != 0; S_maxFixed = T[2];
            it += (uint32_t)next + 1;
            next = 0;
            continue;
        }
        if(next + 1 < a.next_target && it + (uint32_t)next + 1 < n_seeds) { next++; continue; }
        if(firstType != -1 && firstType != -2 && firstType != -3) { error = LRSC_WALK_ERR_CODE; break; }
        const int32_t* T0 = seeds + (uint64_t)it * kSeedInts;               // target = *iterTarget
        const uint64_t s0i = rw.slot_first + it - 1;
        WpSlot& s0 = a.slots[s0i];
        int dp_outcome = 0;                                                  // 0: no DP (--nodp), 1: consensus appended, 2: DP failed
        if(!a.no_dp) {
            // correctByMSAlignment (:208-245) for (source, *iterTarget): its own k and source k-mer
            const int iv0 = T0[0] - S_end - 1;
            int k0 = (S_endBest < T0[4] ? S_endBest : T0[4]) - 2;
            if(S_isRepeat || (T0[3] & 1)) {
                k0 = S_seedLen < T0[1] ? S_seedLen : T0[1];
                k0 = k0 < a.start_kmer_len + 2 ? k0 : a.start_kmer_len + 2;
            }
            if(k0 < 1 || k0 > S_seedLen || k0 > T0[1] || k0 > (int)kMaxInitK || iv0 < 0 || (uint32_t)(k0 + iv0 + T0[1]) >= 65535u) { error = LRSC_WALK_ERR_GEOMETRY; break; }
            uint64_t d_lo, d_hi;
            pack_chars(out + out_len - k0, (uint32_t)k0, d_lo, d_hi);
            const bool dp_ok = (s0.flags & kWpDpValid) && s0.dp_k == (uint8_t)k0 && s0.dp_src_lo == d_lo && s0.dp_src_hi == d_hi;
            if(!dp_ok) {
                if(lane == 0) {
                    s0.dp_k = (uint8_t)k0; s0.dp_src_lo = d_lo; s0.dp_src_hi = d_hi; s0.flags &= (uint8_t)~kWpDpValid;
                    s0.dp_lq = (uint32_t)(k0 + iv0 + T0[1]); s0.dp_total_freq = S_maxFixed + T0[2];
                }
                request((uint32_t)s0i, kWpReqDp);
                break;
            }
            if(s0.dp_error) { error = LRSC_WALK_ERR_DP; break; }
            dp_outcome = 2;
            if(s0.dp_rows > 3) {
                if(s0.dp_cons_len < (uint32_t)k0) { error = LRSC_WALK_ERR_DP; break; }       // out.erase(0, k) would throw in the reference
                const uint32_t appended = s0.dp_cons_len - (uint32_t)k0;
                if(out_len + appended > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                const uint8_t* cons = s0.dp_cons;
                for(uint32_t j = lane; j < appended; j += 64) out[out_len + j] = cons[k0 + j];
                out_len += appended;
                correctedLen += appended;
                seedDis += iv0;
                DPNum++;
                S_seedLen += (int)appended;
                dp_outcome = 1;
            }
        }
        switch(firstType) {
            case -1: highErrorNum++; break;
            case -2: exceedDepthNum++; break;
            default: exceedLeaveNum++; break;
        }
        totalWalkNum++;
        if(wl && lane == 0) wl[it] = (uint8_t)((firstType + 4) | (dp_outcome != 1 ? 0x10 : 0));
        if(dp_outcome != 1) {
            if(a.split) {
                if(out_len + (uint32_t)T0[1] > rw.out_cap || n_pieces >= rw.piece_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                if(lane == 0) piece_start[n_pieces] = out_len;                   // pieceVec.push_back(target)
                n_pieces++;
                for(int t = (int)lane; t < T0[1]; t += 64) out[out_len + t] = read[T0[0] + t];
                out_len += (uint32_t)T0[1];
                S_seedLen = T0[1];
            } else {
                const int raw = (T0[0] + T0[1] - 1) - S_end;                     // readSeq.substr(source.seedEndPos + 1, target.seedEndPos - source.seedEndPos)
                if(out_len + (uint32_t)raw > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                for(int t = (int)lane; t < raw; t += 64) out[out_len + t] = read[S_end + 1 + t];
                out_len += (uint32_t)raw;
                S_seedLen += raw;
            }
            correctedLen += T0[1];
        }
        S_end = T0[0] + T0[1] - 1; S_endBest = T0[5]; S_isRepeat = (T0[3] & 1) != 0; S_maxFixed = T0[2];
        it += 1;
        next = 0;
    }
    if(lane == 0) {
        R.c[0] = rlen; R.c[1] = correctedLen; R.c[2] = a.seed_count[r]; R.c[3] = totalWalkNum; R.c[4] = highErrorNum;
        R.c[5] = exceedDepthNum; R.c[6] = exceedLeaveNum; R.c[7] = FMNum; R.c[8] = DPNum; R.c[9] = seedDis;
        R.steps = steps;
        R.n_pieces = n_pieces; R.out_len = out_len; R.merge = n_pieces != 0; R.error = error;
        R.state = error ? (uint32_t)kReadDone : state;
        R.it = it; R.next = next; R.first_type = firstType;
        R.s_seed_len = S_seedLen; R.s_end = S_end; R.s_end_best = S_endBest; R.s_max_fixed = S_maxFixed; R.s_is_repeat = S_isRepeat ? 1 : 0;
        R.started = 1;
    }
}

__global__ __launch_bounds__(256) void wp_gather_kernel(WpArgs a, const uint64_t* dst_off, char* dst)
{
    const uint32_t r = blockIdx.x;
    const uint8_t* src = a.out_codes + a.work[r].out_off;
    char* d = dst + dst_off[r];
    const uint32_t n = (uint32_t)(dst_off[r + 1] - dst_off[r]);            // 0 for a read the host gave up on (per-read status)
    for(uint32_t i = threadIdx.x; i < n; i += 256) d[i] = "ACGT"[src[i] & 3u];
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------
hipError_t launch_wp_bounds(const WpArgs& a, ReadPlan* plan, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(wp_bounds_kernel, dim3((a.n_reads + 255) / 256), dim3(256), 0, stream, a, plan);
    return hipGetLastError();
}

hipError_t launch_wp_plan(const WpArgs& a, hipStream_t stream)
{
    if(a.reqs) {
        if(a.n_list == 0) return hipSuccess;
        hipLaunchKernelGGL(wp_plan_requests_kernel, dim3((a.n_list + 255) / 256), dim3(256), 0, stream, a);
    } else {
        if(a.r1 <= a.r0) return hipSuccess;
        hipLaunchKernelGGL(wp_plan_kernel, dim3((a.r1 - a.r0 + 255) / 256), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_wp_materialize(const WpArgs& a, hipStream_t stream)
{
    if(a.n_list == 0) return hipSuccess;
    const unsigned nb = a.n_list < 262144u ? a.n_list : 262144u;
    hipLaunchKernelGGL(wp_materialize_kernel, dim3(nb), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_wp_prepare(const FmIndexDev& fm, const WpArgs& a, hipStream_t stream)
{
    if(a.n_list == 0) return hipSuccess;
    const unsigned nb = a.n_list < 262144u ? a.n_list : 262144u;
    if(fm.wide) hipLaunchKernelGGL(wp_prepare_kernel<true>, dim3(nb), dim3(64), 0, stream, fm, a);
    else        hipLaunchKernelGGL(wp_prepare_kernel<false>, dim3(nb), dim3(64), 0, stream, fm, a);
    return hipGetLastError();
}

hipError_t launch_wp_begin(const FmIndexDev& fm, const WpArgs& a, hipStream_t stream)
{
    if(a.n_list == 0) return hipSuccess;
    const unsigned nb = (a.n_list + 63) / 64;
    if(fm.wide) hipLaunchKernelGGL(wp_begin_kernel<true>, dim3(nb), dim3(64), 0, stream, fm, a);
    else        hipLaunchKernelGGL(wp_begin_kernel<false>, dim3(nb), dim3(64), 0, stream, fm, a);
    return hipGetLastError();
}

hipError_t launch_wp_extend(const FmIndexDev& fm, const WpArgs& a, hipStream_t stream)
{
    if(a.n_list == 0 || a.n_lanes == 0) return hipSuccess;
    const unsigned stride = a.lane_stride ? a.lane_stride : 1u;
    const unsigned nb = (unsigned)(((uint64_t)a.n_lanes * stride + 63) / 64);
    const size_t lds = stride == 64u && a.leaves_in_lds ? (size_t)(32u + kMaxChildren) * a.lbytes : 0u;
    if(fm.wide) hipLaunchKernelGGL(wp_extend_kernel<true>, dim3(nb), dim3(64), lds, stream, fm, a);
    else        hipLaunchKernelGGL(wp_extend_kernel<false>, dim3(nb), dim3(64), lds, stream, fm, a);
    return hipGetLastError();
}

hipError_t launch_wp_sched_init(const WpSchedArgs& sa, uint32_t* list_storage, uint32_t n_fresh, hipStream_t stream)
{
    hipLaunchKernelGGL(wp_sched_init_kernel, dim3(256), dim3(256), 0, stream, sa, list_storage, n_fresh);
    return hipGetLastError();
}

hipError_t launch_wp_sched_round(const FmIndexDev& fm, const WpArgs& a, const WpSchedArgs& sa, const WpSchedArgs& sg, uint32_t n_lanes_fast,
                                 uint32_t n_lanes_general, hipStream_t stream)
{
    const unsigned nf = (n_lanes_fast + 63) / 64, ng = (n_lanes_general + 63) / 64;
    if(fm.wide) hipLaunchKernelGGL(wp_fast_kernel<true>, dim3(nf), dim3(64), 0, stream, fm, a, sa);
    else        hipLaunchKernelGGL(wp_fast_kernel<false>, dim3(nf), dim3(64), 0, stream, fm, a, sa);
    hipLaunchKernelGGL(wp_sched_advance_kernel, dim3(1), dim3(1), 0, stream, sa.sched, 0);
    if(fm.wide) hipLaunchKernelGGL(wp_general_kernel<true>, dim3(ng), dim3(64), 0, stream, fm, a, sg);
    else        hipLaunchKernelGGL(wp_general_kernel<false>, dim3(ng), dim3(64), 0, stream, fm, a, sg);
    hipLaunchKernelGGL(wp_sched_advance_kernel, dim3(1), dim3(1), 0, stream, sa.sched, 1);
    return hipGetLastError();
}

hipError_t launch_wp_dp_collect(const WpArgs& a, const WpDpItem* items, hipStream_t stream)
{
    if(a.n_dp == 0) return hipSuccess;
    hipLaunchKernelGGL(wp_dp_collect_kernel, dim3((a.n_dp + 255) / 256), dim3(256), 0, stream, a, items);
    return hipGetLastError();
}

hipError_t launch_wp_stitch(const WpArgs& a, hipStream_t stream)
{
    if(a.r1 <= a.r0) return hipSuccess;
    hipLaunchKernelGGL(wp_stitch_kernel, dim3(a.r1 - a.r0), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_wp_gather(const WpArgs& a, const uint64_t* dst_off, char* dst, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(wp_gather_kernel, dim3(a.n_reads), dim3(256), 0, stream, a, dst_off, dst);
    return hipGetLastError();
}

static hipError_t grow_tmp(void** tmp, size_t* tmp_cap, size_t need)
{
    if(need <= *tmp_cap) return hipSuccess;
    if(*tmp) (void)hipFree(*tmp);
    *tmp = nullptr; *tmp_cap = 0;
    hipError_t e = hipMalloc(tmp, need);
    if(e == hipSuccess) *tmp_cap = need;
    return e;
}

hipError_t wp_scan(uint64_t* v, uint64_t n, void** tmp, size_t* tmp_cap, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    size_t need = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, v, v, (int64_t)n, stream);
    if(e != hipSuccess) return e;
    e = grow_tmp(tmp, tmp_cap, need);
    if(e != hipSuccess) return e;
    return hipcub::DeviceScan::ExclusiveSum(*tmp, need, v, v, (int64_t)n, stream);
}

hipError_t wp_sort_list(const uint32_t* keys, uint32_t* keys_tmp, uint32_t* list, uint32_t* list_tmp, uint32_t n, void** tmp, size_t* tmp_cap, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    size_t need = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, keys, keys_tmp, list_tmp, list, (int)n, 0, 32, stream);
    if(e != hipSuccess) return e;
    e = grow_tmp(tmp, tmp_cap, need);
    if(e != hipSuccess) return e;
    return hipcub::DeviceRadixSort::SortPairsDescending(*tmp, need, keys, keys_tmp, list_tmp, list, (int)n, 0, 32, stream);
}

} // namespace lrsc
