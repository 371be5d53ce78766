// grid_quad.hip -- the k-mer feature grid with QUAD-COOPERATIVE block loads (Block32 indexes).
//
// tools/gather_bench.hip shows that once the gathered footprint is GB-sized (index + k-mer table), a lane
// fetching its own 64-byte block with four dwordx4 loads tops out near 30 G lines/s, while four lanes
// fetching one block with ONE coalesced 64-byte access sustain ~80 G lines/s (4x fewer lines and
// translations per wave-instruction).  So here a QUAD owns a read position: lane q loads 16-byte piece q of
// every rank block (piece 0 = counts, pieces 1..3 = both bit planes of 64 symbols each), counts its slice,
// and the four partial counts are summed with two DPP quad-permute adds.  All four lanes keep the (small)
// walk state redundantly; lane 0 of the quad writes the outputs.  Same semantics as kmer_grid_kernel in
// compact mode (frequency rows, validity mask, base-search counter).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "rank_device.h"

namespace lrsc {

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ uint32_t dpp_lane0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false); }  // quad_perm [0,0,0,0]
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) { v += dpp_xor1(v); v += dpp_xor2(v); return v; }

// this lane's share of "count of `code` among the first `off` symbols of the block" (+ the base count on lane 0)
__device__ __forceinline__ uint32_t quad_partial(const uint4& piece, uint32_t q, uint32_t code, uint32_t off)
{
    const uint32_t base = code == 0 ? (piece.x & ~kFlag32) : code == 1 ? piece.y : code == 2 ? piece.z : piece.w;
    const int32_t n = (int32_t)off - 64 * ((int32_t)q - 1);
    const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
    const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
    const uint32_t cnt = __builtin_popcount((piece.x ^ L) & (piece.z ^ H) & low_mask(n)) +
                         __builtin_popcount((piece.y ^ L) & (piece.w ^ H) & low_mask(n - 32));
    return q == 0 ? base : cnt;
}

__device__ __forceinline__ IvT<uint32_t> quad_update(const StrandC<uint32_t>& s, uint32_t code, IvT<uint32_t> iv, uint32_t q,
                                                     uint32_t& n_blk)
{
    const uint32_t pl = iv.lo, pu = iv.hi + 1;
    const uint32_t bl = pl / Block32::kSyms, bu = pu / Block32::kSyms;
    const uint32_t ol = pl - bl * Block32::kSyms, ou = pu - bu * Block32::kSyms;
    const uint4* base = reinterpret_cast<const uint4*>(s.blocks);
    const uint4 pa = base[(uint64_t)bl * 4 + q];
    uint4 pb = pa;
    if(bu != bl) pb = base[(uint64_t)bu * 4 + q];
    uint32_t ca = quad_sum(quad_partial(pa, q, code, ol));
    uint32_t cb = quad_sum(quad_partial(pb, q, code, ou));
    if(code == 0) {
        const bool fa = dpp_lane0(pa.x >> 31) != 0, fb = dpp_lane0(pb.x >> 31) != 0;
        if(fa && ol != 0) ca -= (uint32_t)dollars_in_c(s, (uint64_t)bl * Block32::kSyms, (uint64_t)bl * Block32::kSyms + ol);
        if(fb && ou != 0) cb -= (uint32_t)dollars_in_c(s, (uint64_t)bu * Block32::kSyms, (uint64_t)bu * Block32::kSyms + ou);
    }
    const uint32_t pbase = pred_of(s, code);
    IvT<uint32_t> out;
    out.lo = pbase + ca;
    out.hi = pbase + cb - 1;
    n_blk += (bl == bu) ? 1u : 2u;
    return out;
}

__global__ __launch_bounds__(256) void kmer_grid_quad_kernel(FmIndexDev fm, GridArgs a, DevCounters* ctr)
{
    using P = uint32_t;
    const uint32_t q = threadIdx.x & 3u;
    const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    uint32_t n_rank = 0, n_blk = 0, n_tab = 0;
    if(gid < a.total_bases) {
        uint32_t r = a.chunk_read[gid >> kChunkShift];
        while(a.read_off[r + 1] <= gid) ++r;
        const uint64_t end = a.read_off[r + 1];
        const uint32_t kmax = a.ks[a.n_k - 1];
        const uint32_t base_k = a.ks[0];
        const uint64_t remain64 = end - gid;
        const uint32_t avail = remain64 < kmax ? (uint32_t)remain64 : kmax;
        const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
        const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
        const uint8_t* w = a.codes + gid;

        WalkState<P> st = walk_init<P>();
        uint32_t slot = 0, vmask = 0;
        uint32_t next_k = a.ks[0];

        auto emit = [&](uint32_t j) {
            if(q != 0 || a.freq_index[j] < 0) return;
            const bool fake = st.size != a.ks[j];
            a.freq[(uint64_t)a.freq_index[j] * a.total_bases + gid] = fake ? -1 : (int32_t)(iv_freq(st.fwd) + iv_freq(st.rvc));
            if(st.fwd.lo <= st.fwd.hi && st.rvc.lo <= st.rvc.hi) vmask |= 1u << a.freq_index[j];
        };

        // table start (see kmer_grid_kernel): valid on both strands => no early exit happened
        uint32_t s0 = 0;
        {
            WalkState<P> ts = st;
            const uint32_t tk = table_start<false>(fm, [&](uint32_t t) { return (uint32_t)w[t]; }, avail, ts);
            if(tk >= base_k && !ts.fwd_broken && !ts.rvc_broken) {
                bool ok = true;
                for(uint32_t j = 0; j < a.n_k && a.ks[j] < tk; ++j) ok = ok && (a.freq_index[j] < 0);
                if(ok) {
                    st = ts; st.counted = base_k;
                    s0 = tk; n_tab = 1;
                    while(slot < a.n_k && a.ks[slot] < tk) ++slot;
                    next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu;
                    if(st.size == next_k) { emit(slot); ++slot; next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu; }
                }
            }
        }
        for(uint32_t s = s0; s < avail; ++s) {
            const uint32_t c = w[s];
            if(st.size == 0) {
                st.fwd = init_interval<P>(sf, c);
                st.rvc = init_interval<P>(sr, 3u - c);
                st.counted = 1;
                n_rank += 2;
            } else {
                const bool in_base = st.size < base_k;
                const bool do_f = !(in_base && st.fwd_broken);
                const bool do_r = !(in_base && st.rvc_broken);
                uint32_t bf = 0, br = 0;
                const IvT<P> nf = quad_update(sf, c, st.fwd, q, bf);
                const IvT<P> nr = quad_update(sr, 3u - c, st.rvc, q, br);
                if(do_f) {
                    st.fwd = nf;
                    st.counted += in_base ? 1u : 0u;
                    st.fwd_broken = in_base && (nf.lo > nf.hi);
                    n_rank += 2; n_blk += bf;
                }
                if(do_r) {
                    st.rvc = nr;
                    st.rvc_broken = in_base && (nr.lo > nr.hi);
                    n_rank += 2; n_blk += br;
                }
            }
            ++st.size;
            if(st.size == next_k) { emit(slot); ++slot; next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu; }
        }
        for(; slot < a.n_k; ++slot) emit(slot);
        if(q == 0) {
            if(a.base_counted) a.base_counted[gid] = (uint8_t)st.counted;
            if(a.valid_mask) a.valid_mask[gid] = (uint8_t)vmask;
        } else { n_rank = 0; n_blk = 0; n_tab = 0; }
    }
    flush_counters(ctr, n_rank, n_blk, n_tab);
}

hipError_t launch_kmer_grid_quad(const FmIndexDev& fm, const GridArgs& a, DevCounters* ctr, hipStream_t stream)
{
    if(a.total_bases == 0) return hipSuccess;
    if(fm.wide || !a.freq || a.out_iv || a.out_size || a.out_count || a.slot_iv) return hipErrorInvalidValue;
    const uint64_t lanes = a.total_bases * 4;
    hipLaunchKernelGGL(kmer_grid_quad_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, fm, a, ctr);
    return hipGetLastError();
}

} // namespace lrsc
