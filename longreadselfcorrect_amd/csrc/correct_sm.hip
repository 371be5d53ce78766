// correct_sm.hip -- the persistent correction kernel, wavefront-convergent form: ONE LANE PER READ as before, but the
// whole wavefront runs one loop of "answer every lane's pending FM-index request, then sweep the per-lane state
// machine" (walk_sm.h), so that lanes at unrelated points of their chains share instructions.
// PacBioSelfCorrectionProcess::initCorrect / correctByFMExtension (PacBio/PacBioSelfCorrectionProcess.cpp:56-206) over
// LongReadSelfCorrectByOverlap (PacBio/LongReadCorrectByOverlap.cpp:17-878).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "correct_dev.h"
#include "rank_device.h"

// Per-wavefront LDS copies of the wave-uniform context (walk_sm.h reaches them through LRSC_A / LRSC_FM / LRSC_SF / LRSC_SR):
// every member of the launch arguments is then one LDS read away from any block of the sweep, with no pointer to chase.
#define LRSC_SM_KERNEL_TU 1
namespace lrsc {
__shared__ CorrectArgs g_sm_a;
__shared__ FmIndexDev g_sm_fm;
__shared__ unsigned char g_sm_strands[2 * sizeof(StrandC<uint64_t>)] __attribute__((aligned(16)));
template <class P> __device__ __forceinline__ const StrandC<P>& sm_strand_lds(int which)
{
    return reinterpret_cast<const StrandC<P>*>(g_sm_strands)[which];
}
} // namespace lrsc

#include "walk_sm.h"

#ifdef LRSC_SM_NOSINK
#define LRSC_SM_KERNEL_NAME correct_sm_kernel_nosink
#else
#define LRSC_SM_KERNEL_NAME correct_sm_kernel
#endif

namespace lrsc {

template <bool WIDE>
__global__ __launch_bounds__(64, 1) void LRSC_SM_KERNEL_NAME(const FmIndexDev fm_arg, const CorrectArgs a_arg)
{
    const FmIndexDev* fmp = &fm_arg;
    const CorrectArgs* ap = &a_arg;
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const CorrectArgs& a = *ap;
    const uint32_t stride = 64u / a.reads_per_wave;
    const uint32_t slot = blockIdx.x * a.reads_per_wave + threadIdx.x / stride;
    const bool owner = (threadIdx.x % stride) == 0 && slot < a.n_reads;
    // the four extension pairs of a lane's getFMIndexExtensions wait here between R-phases (element-major: conflict-free)
    __shared__ P ex_lds[16 * 64];
    const StrandC<P> sF = strand_consts<P>(fmp->strand[LRSC_RBWT]);
    const StrandC<P> sR = strand_consts<P>(fmp->strand[LRSC_BWT]);
    if(threadIdx.x == 0) {
        g_sm_a = a_arg;
        g_sm_fm = fm_arg;
        reinterpret_cast<StrandC<P>*>(g_sm_strands)[0] = sF;
        reinterpret_cast<StrandC<P>*>(g_sm_strands)[1] = sR;
    }
    __syncthreads();
    // The per-lane state object lives in LDS (~100-cycle accesses, conflict-free for same-member accesses of a wavefront up to a
    // 2-way bank overlap).  As a plain local it ends up in scratch memory -- the object is too large for the register file next to
    // the blocks' temporaries -- and every member access becomes a global-memory round trip: the kernel then spends its time
    // waiting on its own state (measured: ~60 ticks per instruction in every block, independent of the block's real memory work).
    __shared__ ReadSM<WIDE> Ls[64];
    ReadSM<WIDE>& L = Ls[threadIdx.x];
    L.pc = PC_DONE;
    L.req.kind = kReqNone;
    L.n_rank = 0; L.n_blk = 0; L.n_tab = 0; L.tkp = nullptr;
    if(owner) L.init(fmp, ap, &sF, &sR, a.order ? a.order[slot] : slot);
    const uint32_t quorum = a.setup_quorum_pct, gate_pct = a.step_gate_pct;
    P* const ex = ex_lds + threadIdx.x;
    const uint32_t ex_stride = 64u;
    uint32_t trace_pos = 1, slow_wait = 0;
    unsigned long long* const prof = a.prof;
    uint64_t p_r = 0, p_cls[4] = {0, 0, 0, 0}, p_n[4] = {0, 0, 0, 0};
    uint64_t tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    SmReq<P> res;
    res.a_lo = res.a_hi = res.b_lo = res.b_hi = 0;
    while(true) {
        const bool live = L.pc != PC_DONE;
        const unsigned long long live_mask = __ballot(live);
        if(live_mask == 0) break;
        // Three wave-level gates keep the long, memory-latency-bound blocks of the sweep from running for one lane at a time:
        //  * set-up: lanes between walks wait until a quorum of the live lanes is between walks, then build their queries and
        //    start PREP together;
        //  * begin: the chain construction (sort) + root start when no lane is in PREP any more;
        //  * step: lanes whose extension request is answered (and lanes ready to prune) wait until a quorum of the lanes that
        //    are inside a walk has arrived, then run the acceptance ladder / children / pruning / commit in the same sweep.
        const uint32_t n_all = (uint32_t)__builtin_popcountll(live_mask);
        const uint32_t n_need = (uint32_t)__builtin_popcountll(__ballot(live && L.wants_setup()));
        const bool setup_now = n_need * 100u >= n_all * quorum;
        const bool begin_now = __ballot(live && L.in_prep()) == 0;
        const uint32_t n_walk = (uint32_t)__builtin_popcountll(__ballot(live && L.in_walk()));
        const uint32_t n_gate = (uint32_t)__builtin_popcountll(__ballot(live && L.at_gate()));
        const bool gate_now = n_gate * 100u >= n_walk * gate_pct;
        // the slow gate: lanes with a wide frontier wait until a quarter of the walkers wait with them, or for 12 sweeps
        const uint32_t n_slow = (uint32_t)__builtin_popcountll(__ballot(live && L.at_slow_gate()));
        slow_wait = n_slow != 0 ? slow_wait + 1u : 0u;
        const bool slow_now = n_slow != 0 && (n_slow * 100u >= (n_walk + n_slow) * 25u || slow_wait >= a.slow_gate_sweeps);
        if(slow_now) slow_wait = 0;
        const bool have = L.req.kind != kReqNone;
        // profiling classes of this sweep (wave-uniform): 0 begin (sort + root), 1 between walks (stitch / next query), 2 step gate
        // open, 3 light (searches only)
        uint32_t cls = 3;
        if(prof) {
            const bool any_begin = begin_now && __ballot(live && L.pc == PC_BEGIN) != 0;
            const bool any_next = __ballot(live && (L.pc == PC_WALK_END || (L.pc == PC_NEXT && setup_now))) != 0;
            cls = any_begin ? 0u : any_next ? 1u : (gate_now && n_gate != 0) ? 2u : 3u;
        }
        const uint64_t t0 = prof ? __builtin_readcyclecounter() : 0;
        if(have) sm_answer<WIDE>(*fmp, sF, sR, mtab, L.req, res, ex, ex_stride, L.n_rank, L.n_blk, L.n_tab);
        const uint64_t t1 = prof ? __builtin_readcyclecounter() : 0;
        if(a.trace != nullptr && live && L.r == a.trace_read) sm_trace<P>(a.trace, a.trace_cap, trace_pos, L.pc, have, L.req, res);
        if(live) L.sweep(have, res, setup_now, begin_now, gate_now, slow_now, ex, ex_stride, prof ? tk : nullptr);
        if(prof) {
            const uint64_t t2 = __builtin_readcyclecounter();
            p_r += t1 - t0;
            p_cls[cls] += t2 - t1; p_n[cls] += 1;
        }
    }
    if(prof && threadIdx.x == 0) {
        unsigned long long* o = prof + (size_t)blockIdx.x * 32;
        o[0] = p_r; for(int c = 0; c < 4; ++c) { o[1 + c] = p_cls[c]; o[5 + c] = p_n[c]; }
    }
    if(prof) {
        // per-block ticks: the maximum over the lanes (a lane only counts the blocks it ran; the wave waits for the slowest)
        unsigned long long* o = prof + (size_t)blockIdx.x * 32 + 16;
        for(int c = 0; c < 12; ++c) { unsigned long long v = tk[c]; for(int sft = 32; sft > 0; sft >>= 1) { const unsigned long long w = __shfl_xor(v, sft, 64); v = w > v ? w : v; } if(threadIdx.x == 0) o[c] = v; }
    }
    if(a.trace != nullptr && owner && L.r == a.trace_read) a.trace[0] = trace_pos;
    flush_counters(a.ctr, L.n_rank, L.n_blk, L.n_tab);
}

#ifdef LRSC_SM_NOSINK
hipError_t launch_correct_sm_nosink(const FmIndexDev* d_fm,
#else
hipError_t launch_correct_sm(const FmIndexDev* d_fm,
#endif
                             const CorrectArgs* d_args, const CorrectArgs& a, bool wide, hipStream_t stream, const FmIndexDev& fm)
{
    if(a.n_reads == 0) return hipSuccess;
    if(a.reads_per_wave == 0 || a.reads_per_wave > 64 || (a.reads_per_wave & (a.reads_per_wave - 1))) return hipErrorInvalidValue;
    const unsigned nb = (a.n_reads + a.reads_per_wave - 1) / a.reads_per_wave;
    if(wide) hipLaunchKernelGGL((LRSC_SM_KERNEL_NAME<true>), dim3(nb), dim3(64), 0, stream, fm, a);
    else     hipLaunchKernelGGL((LRSC_SM_KERNEL_NAME<false>), dim3(nb), dim3(64), 0, stream, fm, a);
    return hipGetLastError();
}

} // namespace lrsc
