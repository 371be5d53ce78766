// dp_retrieve.hip -- LongReadOverlap::retrieveStr (PacBio/LongReadOverlap.cpp:667-756) on the device.
//
//   dp_seed_kernel      lane per (request, direction): bi-interval of the source k-mer (query[0..k)) and of the
//                       reverse-complemented target k-mer (revcomp(query[Lq-k..))): fwd = reverse(w) in the rBWT,
//                       rvc = revcomp(w) in the BWT, each with findInterval's early exit (:681-682).
//                       Round 1 built this kernel `optnone`; the cause is pinned in rank_device.h (kSelectChainNote) and
//                       profiles/r03_compiler_finding/: the select chain over a rank block's four counters is mis-lowered when
//                       the compiler cannot bound the symbol code, which `c = q[..]; if(dir) c = 3u - c;` -- the form this loop
//                       had -- provokes.  rank_device.h picks the base count by the code's bits now (LRSC_PICK4); this loop also forms its
//                       character as `(q[..] ^ (dir ? 3 : 0)) & 3`.
//   dp_retrieve_kernel  lane per retrieved string: starts at one row of such an interval (at most `coverage` rows
//                       per interval, :685-687,:704-706) and LF-walks up to maxLength - k characters, stopping at '$'.
//                       The string is written in the orientation retrieveMatches aligns (:697-700,:716-719):
//                         fwd, forward seed:  w . c0 c1 ...                  rvc, forward seed:  w . ~c0 ~c1 ...
//                         fwd, RC seed:       ... ~c1 ~c0 . query tail       rvc, RC seed:       ... c1 c0 . query tail
//                       (forward-seed strings grow rightwards from the slot start, RC-seed strings leftwards from
//                       the slot end) together with the DpJob the alignment kernel consumes.
#include <hip/hip_runtime.h>

#include "dp_dev.h"
#include "rank_device.h"

namespace lrsc {

template <bool WIDE>
__global__ __launch_bounds__(256) void dp_seed_kernel(FmIndexDev fm, DpPipeArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(gid < (uint64_t)a.n_reqs * 2) {
        DpRequest* R = a.reqs + (gid >> 1);
        const uint32_t dir = (uint32_t)(gid & 1);
        const uint32_t k = R->k, lq = R->lq, coverage = R->coverage;
        const uint8_t* q = a.codes + R->q_off;
        const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
        const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
        WalkState<P> st = walk_init<P>();
        for(uint32_t s = 0; s < k; ++s) {
            if(st.fwd_broken && st.rvc_broken) break;
            // complement as xor + mask (range [0, 3] visible: see the note in the file header)
            const uint32_t c = ((uint32_t)q[dir == 0 ? s : lq - 1 - s] ^ (dir != 0 ? 3u : 0u)) & 3u;
            st = walk_step<WIDE>(sf, sr, c, k, st, mtab);
        }
        const bool fv = st.fwd.lo <= st.fwd.hi, rv = st.rvc.lo <= st.rvc.hi;
        const uint64_t nf = fv ? (uint64_t)(st.fwd.hi - st.fwd.lo) + 1 : 0, nr = rv ? (uint64_t)(st.rvc.hi - st.rvc.lo) + 1 : 0;
        R->row_lo[2 * dir] = st.fwd.lo;     R->cnt[2 * dir] = (uint32_t)(nf < coverage ? nf : coverage);
        R->row_lo[2 * dir + 1] = st.rvc.lo; R->cnt[2 * dir + 1] = (uint32_t)(nr < coverage ? nr : coverage);
        n_rank = st.n_rank; n_blk = st.n_blk;
    }
    flush_counters(a.ctr, n_rank, n_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void dp_retrieve_kernel(FmIndexDev fm, DpPipeArgs a)
{
    using L = Lay<WIDE>;
    using P = typename L::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(gid < a.n_jobs) {
        // owning request: last one with job_first <= gid
        uint32_t lo = 0, hi = a.n_reqs;
        while(hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if(a.reqs[mid].job_first <= gid) lo = mid; else hi = mid;
        }
        const DpRequest& R = a.reqs[lo];
        uint32_t t = (uint32_t)(gid - R.job_first), grp = 0;
        while(grp < 3 && t >= R.cnt[grp]) { t -= R.cnt[grp]; ++grp; }
        const uint32_t slot_i = (uint32_t)(gid - R.job_first);
        const bool rc = grp >= 2, second = (grp & 1) == 0;        // groups 0 / 2 walk the rBWT (strand index LRSC_RBWT)
        const StrandC<P> S = strand_consts<P>(fm.strand[second ? LRSC_RBWT : LRSC_BWT]);
        const uint8_t* q = a.codes + R.q_off;
        uint8_t* slot = a.strings + R.str_off + (uint64_t)slot_i * R.str_cap;
        const uint32_t k = R.k, cap = R.str_cap;
        const bool comp = grp == 1 || grp == 2;
        if(!rc) for(uint32_t i = 0; i < k; ++i) slot[i] = q[i];
        else    for(uint32_t i = 0; i < k; ++i) slot[cap - k + i] = q[R.lq - k + i];
        const uint32_t max_steps = R.max_len > k ? R.max_len - k : 0;
        P idx = (P)(R.row_lo[grp] + t);
        uint32_t len = 0;
        for(; len < max_steps; ++len) {
            const P b = idx / L::kSyms;
            const uint32_t off = (uint32_t)(idx - b * L::kSyms);
            typename L::Regs r;
            L::load(S.blocks, b, r);
            const uint32_t code = L::symbol(r, off);
            const bool flagged = L::flagged(r);
            if(code == 0 && flagged && dollars_in_c(S, (uint64_t)idx, (uint64_t)idx + 1) != 0) break;           // '$'
            const uint8_t out = (uint8_t)(comp ? 3u - code : code);
            if(!rc) slot[k + len] = out; else slot[cap - k - 1 - len] = out;
            uint64_t c = L::count(r, code, mtab + off * L::kRow);
            if(code == 0 && off != 0 && flagged) c -= dollars_in_c(S, (uint64_t)b * L::kSyms, (uint64_t)b * L::kSyms + off);
            idx = pred_of(S, code) + (P)c;
            n_rank += 1; n_blk += 1;
        }
        const uint32_t slen = k + len;
        DpJob J;
        J.s1_off = R.q_off; J.s1_len = R.lq;
        J.s2_off = R.str_off + (uint64_t)slot_i * cap + (rc ? cap - slen : 0);
        J.s2_len = slen;
        J.ops_off = R.ops_off + (uint64_t)slot_i * R.ops_cap;
        J.start1 = rc ? (int32_t)(R.lq - k) : 0;
        J.start2 = rc ? (int32_t)(slen - k) : 0;
        J.mode = rc ? 2u : 1u;
        J.req = lo;
        a.jobs[gid] = J;
    }
    flush_counters(a.ctr, n_rank, n_blk);
}

static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_dp_seeds(const FmIndexDev& fm, const DpPipeArgs& a, hipStream_t stream)
{
    if(a.n_reqs == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(dp_seed_kernel<true>, dim3(nblk((uint64_t)a.n_reqs * 2)), dim3(256), 0, stream, fm, a);
    else        hipLaunchKernelGGL(dp_seed_kernel<false>, dim3(nblk((uint64_t)a.n_reqs * 2)), dim3(256), 0, stream, fm, a);
    return hipGetLastError();
}

hipError_t launch_dp_retrieve(const FmIndexDev& fm, const DpPipeArgs& a, hipStream_t stream)
{
    if(a.n_jobs == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(dp_retrieve_kernel<true>, dim3(nblk(a.n_jobs)), dim3(256), 0, stream, fm, a);
    else        hipLaunchKernelGGL(dp_retrieve_kernel<false>, dim3(nblk(a.n_jobs)), dim3(256), 0, stream, fm, a);
    return hipGetLastError();
}

} // namespace lrsc
