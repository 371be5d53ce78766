// correct_sm.hip -- the persistent correction kernel, wavefront-convergent form: ONE LANE PER READ as before, but the
// whole wavefront runs one loop of "answer every lane's pending FM-index request, then sweep the per-lane state
// machine" (walk_sm.h), so that lanes at unrelated points of their chains share instructions.
// PacBioSelfCorrectionProcess::initCorrect / correctByFMExtension (PacBio/PacBioSelfCorrectionProcess.cpp:56-206) over
// LongReadSelfCorrectByOverlap (PacBio/LongReadCorrectByOverlap.cpp:17-878).
#include <hip/hip_runtime.h>

#include "walk_sm.h"

namespace lrsc {

template <bool WIDE>
__global__ __launch_bounds__(64, 2) void correct_sm_kernel(const FmIndexDev* __restrict__ fmp, const CorrectArgs* __restrict__ ap)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const CorrectArgs& a = *ap;
    const uint32_t stride = 64u / a.reads_per_wave;
    const uint32_t slot = blockIdx.x * a.reads_per_wave + threadIdx.x / stride;
    const bool owner = (threadIdx.x % stride) == 0 && slot < a.n_reads;
    ReadSM<WIDE> L;
    L.pc = PC_DONE;
    L.req.kind = kReqNone;
    L.n_rank = 0; L.n_blk = 0; L.n_tab = 0;
    if(owner) L.init(fmp, ap, a.order ? a.order[slot] : slot);
    const StrandC<P> sF = strand_consts<P>(fmp->strand[LRSC_RBWT]);
    const StrandC<P> sR = strand_consts<P>(fmp->strand[LRSC_BWT]);
    const uint32_t quorum = a.setup_quorum_pct;
    SmReq<P> res;
    res.a_lo = res.a_hi = res.b_lo = res.b_hi = 0;
    while(true) {
        const bool live = L.pc != PC_DONE;
        const unsigned long long live_mask = __ballot(live);
        if(live_mask == 0) break;
        // setting up a walk (interval trees, root) is a long lane-serial stretch: lanes between walks wait until a quorum
        // of the wavefront's live lanes is between walks, then set up together
        const uint32_t n_all = (uint32_t)__builtin_popcountll(live_mask);
        const uint32_t n_need = (uint32_t)__builtin_popcountll(__ballot(live && L.wants_setup()));
        const bool setup_now = n_need * 100u >= n_all * quorum;
        const bool have = L.req.kind != kReqNone;
        if(have) sm_answer<WIDE>(*fmp, sF, sR, mtab, L.req, res, L.n_rank, L.n_blk, L.n_tab);
        if(live) L.sweep(have, res, setup_now);
    }
    flush_counters(a.ctr, L.n_rank, L.n_blk, L.n_tab);
}

hipError_t launch_correct_sm(const FmIndexDev* d_fm, const CorrectArgs* d_args, const CorrectArgs& a, bool wide, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    if(a.reads_per_wave == 0 || a.reads_per_wave > 64 || (a.reads_per_wave & (a.reads_per_wave - 1))) return hipErrorInvalidValue;
    const unsigned nb = (a.n_reads + a.reads_per_wave - 1) / a.reads_per_wave;
    if(wide) hipLaunchKernelGGL(correct_sm_kernel<true>, dim3(nb), dim3(64), 0, stream, d_fm, d_args);
    else     hipLaunchKernelGGL(correct_sm_kernel<false>, dim3(nb), dim3(64), 0, stream, d_fm, d_args);
    return hipGetLastError();
}

} // namespace lrsc
