// wp_coop.hip -- an experiment (LRSC_WP_COOP=1), not the default: the one-walk-per-wavefront extension launch with run-ahead
// helper lanes.  Its own translation unit so that the default extension kernels of wp.hip are the binary the GPU suite has seen.
#include <hip/hip_runtime.h>

#include "walk_device.h"
#include "wp.h"

namespace lrsc {

// the per-walk tables of the prepared state (as wp.hip binds them)
template <bool WIDE>
__device__ __forceinline__ void coop_bind_static(Walk<WIDE>& W, const WpArgs& a, const WpSlot& s)
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

// ---------------------------------------------------------------------------------------
// extend, one walk per wavefront with run-ahead helpers (LRSC_WP_COOP=1; an experiment, not the default)
//
// Lane 0 owns the walk and runs exactly what wp_extend_kernel's owner lane runs.  Before a general step over a frontier of n >= 2
// leaves, lanes 1..n each take a private copy of one leaf and run the memory-bound front of the step on it alone (Walk::warm:
// refine, extension, seed support) -- nothing of that is kept; it only pulls the rank blocks, table entries and 9-mer chains that
// leaf needs into the caches, n leaves' dependent chains side by side, so that the owner's sequential pass finds them there.
// Results are the owner's alone: bit-exact by construction as long as a helper writes nothing shared (its cur / nxt are private slots
// of coop_ws; rings and paths are only read in Walk::warm).
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(64, 2) void wp_extend_coop_kernel(FmIndexDev fm, WpArgs a, uint8_t* coop_ws)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint32_t lane = threadIdx.x;
    const bool owner = lane == 0;
    const uint32_t wave = blockIdx.x;
    auto first_u32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto first_u64 = [&](uint64_t v) -> uint64_t { return ((uint64_t)first_u32((uint32_t)(v >> 32)) << 32) | first_u32((uint32_t)v); };
    Walk<WIDE> W;
    W.sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
    W.sR = strand_consts<P>(fm.strand[LRSC_BWT]);
    W.fm = &fm; W.mtab = mtab;
    W.seedSize = a.seed_size; W.minOverlap = a.min_overlap; W.maxLeaves = a.max_leaves;
    W.PBcoverage = a.pb_coverage; W.PacBioErrorRate = a.pacbio_error_rate; W.errorRate = 0.25; W.localK = 100;
    W.freqsOfKmerSize = a.freqs_of_kmer_size;
    const WpLaneLayout LL = wp_lane_layout(a.lbytes, a.lane_pathw);
    uint8_t* lws = a.lane_ws + (uint64_t)wave * a.lane_ws_bytes;                  // the walk's workspace: every lane sees it
    Leaf<P>* const leaf_base = reinterpret_cast<Leaf<P>*>(lws + LL.leaves);
    Leaf<P>* const mine = reinterpret_cast<Leaf<P>*>(coop_ws + ((uint64_t)wave * 64 + lane) * kWpCoopLeaves * sizeof(Leaf<P>));
    W.rings = reinterpret_cast<double*>(lws + LL.rings);
    W.results = reinterpret_cast<WalkResultRec*>(lws + LL.results);
    W.paths = reinterpret_cast<uint32_t*>(lws + LL.paths);
    W.pathw = a.lane_pathw;
    W.rpaths = W.paths + (uint64_t)32 * a.lane_pathw;
    W.n_rank = 0; W.n_blk = 0; W.steps = 0; W.leaf_steps = 0; W.error = 0; W.cyc_setup = 0; W.cyc_loop = 0; W.prof = nullptr; W.profile = false;
    W.n_cur = 0; W.n_nxt = 0; W.n_results = 0; W.ended = false; W.max_front = 1;
    const uint64_t min_SA = a.pb_coverage > 60 ? (uint64_t)((a.pb_coverage / 60) * 3) : 3;

    bool in_walk = false;                    // wavefront-uniform, like every decision of the loop below (they are lane 0's, broadcast)
    uint32_t si = 0;
    uint64_t steps0 = 0;
    Leaf<P> L;
    uint32_t pw = 0;
    bool fast = false;
    if(wave < a.n_lanes)
    while(true) {
        if(!in_walk) {
            uint32_t i = 0;
            if(owner) i = atomicAdd(a.queue, 1u);
            i = first_u32(i);
            if(i >= a.n_list) break;
            if(a.reqs && a.reqs[i].kind != kWpReqFm) continue;
            si = a.list ? a.list[i] : (uint32_t)a.slot_base + i;
            const WpSlot& s = a.slots[si];
            if(s.flags & kWpGeomBad) continue;
            coop_bind_static<WIDE>(W, a, s);
            const WpStatic* H = reinterpret_cast<const WpStatic*>(s.prep);
            W.n9f = H->n9f; W.n9r = H->n9r; W.tmask0 = H->tmask0; W.tmask1 = H->tmask1;
            W.maxOverlap = (uint32_t)s.k + 2;
            W.min_SA_threshold = min_SA;
            if((int32_t)s.gap > 100) W.maxIndelSize = (uint64_t)((int32_t)s.gap * 0.2); else W.maxIndelSize = 20;
            W.maxLength = (uint64_t)((1.2 * ((int32_t)s.gap + 10)) + (double)(2 * (uint64_t)s.k));
            W.minLength = (uint64_t)((0.8 * ((int32_t)s.gap - 20)) + (double)(2 * (uint64_t)s.k));
            W.error = 0;
            if(owner) {
                W.cur = leaf_base; W.nxt = leaf_base + 32; W.leaf_small = leaf_base;
                steps0 = W.steps; W.leaf_steps = 0; W.max_front = 1;
                const P riv[4] = {(P)H->root[0], (P)H->root[1], (P)H->root[2], (P)H->root[3]};
                W.begin_root(riv);
            }
            in_walk = true;
            fast = false;
        }
        int r = 2;
        if(owner) {
            if(!fast && W.can_fast()) { W.enter_fast(L, pw); fast = true; }
            if(fast) {
                r = W.step_fast(L, pw);
                if(r != 1) fast = false;
            }
        }
        r = (int)first_u32((uint32_t)r);
        if(r == 2) {
            const uint32_t nc = first_u32(W.n_cur);
            if(nc >= 2u && nc <= 32u) {
                const uint64_t curp = first_u64((uint64_t)(uintptr_t)W.cur);
                const uint64_t cl = first_u64(W.currentLength), ck = first_u64(W.currentKmerSize);
                __threadfence_block();                              // the owner's leaf stores, before the helpers' loads of them
                if(!owner && lane <= nc) {
                    Leaf<P> lf = reinterpret_cast<const Leaf<P>*>((uintptr_t)curp)[lane - 1];
                    // a helper only ever reads with what it was given: never follow a leaf that is not plainly a leaf of this index
                    const bool sane = (lf.flo > lf.fhi || lf.fhi < W.sF.n) && (lf.rlo > lf.rhi || lf.rhi < W.sR.n) && lf.ring < 32u && lf.path < 32u;
                    if(sane) {
                        mine[0] = lf;
                        W.cur = mine; W.nxt = mine + 2; W.leaf_small = mine;
                        W.n_cur = 1; W.n_nxt = 0; W.currentLength = cl; W.currentKmerSize = ck; W.error = 0; W.ended = false;
                        W.warm();
                    }
                }
            }
            if(owner) r = W.step() ? 1 : 0;
            r = (int)first_u32((uint32_t)r);
        }
        if(r == 1) continue;
        in_walk = false;
        if(owner) {
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
        }
    }
    // the statistics count the walk's own queries: the helpers' repeats of them are not rank queries of the algorithm
    flush_counters(a.ctr, owner ? W.n_rank : 0u, owner ? W.n_blk : 0u);
}

hipError_t launch_wp_extend_coop(const FmIndexDev& fm, const WpArgs& a, uint8_t* coop_ws, hipStream_t stream)
{
    if(a.n_list == 0 || a.n_lanes == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(wp_extend_coop_kernel<true>, dim3(a.n_lanes), dim3(64), 0, stream, fm, a, coop_ws);
    else        hipLaunchKernelGGL(wp_extend_coop_kernel<false>, dim3(a.n_lanes), dim3(64), 0, stream, fm, a, coop_ws);
    return hipGetLastError();
}

} // namespace lrsc
