// grid_coop.hip -- the compact k-mer feature grid with LINE-COOPERATIVE rank-block loads (Block32 indexes).
//
// kmer_grid_kernel lets every lane fetch its own 64-byte rank block with four dwordx4 loads: each of those four
// instructions touches 64 different lines in 64 different pages.  Once the gathered footprint is GB-sized (rank blocks
// + the 17 GB 15-mer table) that is what bounds it: ~30 G lines/s, while the same lines fetched "four lanes = one
// line" go 2-3x faster (tools/gather_bench.hip).  grid_quad.hip tried that by giving four lanes ONE search, which
// quartered the searches in flight and lost.  Here every lane keeps its own search; only the LOAD is transposed:
//   round j (j = 0..3): lane L fetches piece (L & 3) of the block that lane (L >> 2) + 16 j wants  -> 16 whole lines
//                       per instruction; the 16 bytes go to LDS slot [j][L];
//   afterwards lane l reads its own four pieces back from slots [l >> 4][4 (l & 15) .. + 3].
// Per block that is 4 ds_bpermute + 4 global loads + 4 ds_write_b128 + 4 ds_read_b128 instead of 4 global loads.
// Same outputs as kmer_grid_kernel in compact mode (frequency rows, validity mask, base-search counter); waves that
// hold a position which cannot start from the k-mer table fall back to the per-lane walk.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "rank_device.h"

namespace lrsc {

namespace {
constexpr uint32_t kNone = 0xFFFFFFFFu;
using P = uint32_t;
using L32 = Lay<false>;

// lane l receives the 64-byte block `b` it asked for (kNone: nothing); X = this wavefront's 4 KB exchange area
__device__ __forceinline__ void coop_load(const void* __restrict__ blocks, uint32_t b, uint4* __restrict__ X, uint32_t lane, L32::Regs& r)
{
    const uint4* base = reinterpret_cast<const uint4*>(blocks);
#pragma unroll
    for(uint32_t j = 0; j < 4; ++j) {
        const uint32_t bb = (uint32_t)__shfl((int)b, (int)((lane >> 2) + 16u * j));
        uint4 v = make_uint4(0, 0, 0, 0);
        if(bb != kNone) v = base[(uint64_t)bb * 4 + (lane & 3u)];
        X[j * 64 + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t at = (lane >> 4) * 64 + 4 * (lane & 15u);
    r.q[0] = X[at]; r.q[1] = X[at + 1]; r.q[2] = X[at + 2]; r.q[3] = X[at + 3];
    __builtin_amdgcn_wave_barrier();
}

// BWTAlgorithms::updateInterval for the lanes with need == true; every lane of the wavefront takes part in the loads
__device__ __forceinline__ IvT<P> coop_update(const StrandC<P>& s, uint32_t code, IvT<P> iv, bool need, uint4* X, uint32_t lane,
                                              const uint32_t* __restrict__ mtab, uint32_t& n_blk)
{
    const P pl = iv.lo, pu = iv.hi + 1;
    const P bl = pl / L32::kSyms, bu = pu / L32::kSyms;
    const uint32_t ol = pl - bl * L32::kSyms, ou = pu - bu * L32::kSyms;
    L32::Regs ra, rb;
    coop_load(s.blocks, need ? bl : kNone, X, lane, ra);
    const bool two = need && bu != bl;
    rb = ra;
    if(__ballot(two) != 0) {
        L32::Regs r2;
        coop_load(s.blocks, two ? bu : kNone, X, lane, r2);
        if(two) rb = r2;
    }
    IvT<P> out = iv;
    if(need) {
        uint64_t ca, cb;
        if(!two) L32::count2(ra, code, mtab + ol * L32::kRow, mtab + ou * L32::kRow, ca, cb);
        else { ca = L32::count(ra, code, mtab + ol * L32::kRow); cb = L32::count(rb, code, mtab + ou * L32::kRow); }
        if(code == 0) {
            if(ol != 0 && L32::flagged(ra)) ca -= dollars_in_c(s, (uint64_t)bl * L32::kSyms, (uint64_t)bl * L32::kSyms + ol);
            if(ou != 0 && L32::flagged(rb)) cb -= dollars_in_c(s, (uint64_t)bu * L32::kSyms, (uint64_t)bu * L32::kSyms + ou);
        }
        const P pb = pred_of(s, code);
        out.lo = pb + (P)ca;
        out.hi = pb + (P)cb - 1;
        n_blk += two ? 2u : 1u;
    }
    return out;
}
} // namespace

__global__ __launch_bounds__(256) void kmer_grid_coop_kernel(FmIndexDev fm, GridArgs a, DevCounters* ctr)
{
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<false>::value];
    __shared__ __attribute__((aligned(16))) uint4 xch[4][256];
    init_mask_table<false>(mtab);
    const uint32_t lane = threadIdx.x & 63u;
    uint4* X = xch[threadIdx.x >> 6];
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = gid < a.total_bases;
    uint32_t n_rank = 0, n_blk = 0, n_tab = 0;

    const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
    const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
    const uint32_t kmax = a.ks[a.n_k - 1];
    const uint32_t base_k = a.ks[0];
    uint32_t avail = 0;
    const uint8_t* w = a.codes + (in ? gid : 0);
    if(in) {
        uint32_t r = a.chunk_read[gid >> kChunkShift];
        while(a.read_off[r + 1] <= gid) ++r;
        const uint64_t remain64 = a.read_off[r + 1] - gid;
        avail = remain64 < kmax ? (uint32_t)remain64 : kmax;
    }
    WalkState<P> st = walk_init<P>();
    uint32_t slot = 0, next_k = a.ks[0], vmask = 0;

    auto emit = [&](uint32_t j) {
        if(a.freq_index[j] < 0) return;
        const bool fake = st.size != a.ks[j];
        a.freq[(uint64_t)a.freq_index[j] * a.total_bases + gid] = fake ? -1 : (int32_t)(iv_freq(st.fwd) + iv_freq(st.rvc));
        if(st.fwd.lo <= st.fwd.hi && st.rvc.lo <= st.rvc.hi) vmask |= 1u << a.freq_index[j];
    };

    // table start (see kmer_grid_kernel): possible when the fwd strand's base_k-mer search ran to its end
    uint32_t s0 = 0;
    bool lean = false;
    if(in) {
        WalkState<P> ts = st;
        const uint32_t tk = table_start<false>(fm, [&](uint32_t t) { return (uint32_t)w[t]; }, avail, ts);
        if(tk >= base_k) {
            bool fwd_base_ok = !ts.fwd_broken;
            if(!fwd_base_ok && tk > base_k) {
                WalkState<P> tb = st;
                const uint32_t kb = table_start<false>(fm, [&](uint32_t t) { return (uint32_t)w[t]; }, base_k, tb);
                fwd_base_ok = kb == base_k && !tb.fwd_broken;
                n_tab += kb != 0;
            }
            bool ok = fwd_base_ok;
            for(uint32_t j = 0; j < a.n_k && a.ks[j] < tk; ++j) ok = ok && (a.freq_index[j] < 0);
            if(ok) {
                st = ts; st.counted = base_k; st.n_rank = 0; st.n_blk = 0;
                s0 = tk; n_tab += 1; lean = true;
                while(slot < a.n_k && a.ks[slot] < tk) ++slot;
                next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu;
                if(st.size == next_k) { emit(slot); ++slot; next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu; }
            }
        }
    }
    if(__ballot(in && !lean) == 0) {
        // every position of the wavefront continues from a table entry: cooperative loads, wave-uniform step loop
        uint32_t s_lo = in ? s0 : 0xFFFFFFFFu, s_hi = in ? avail : 0u;
#pragma unroll
        for(int o = 32; o > 0; o >>= 1) {
            const uint32_t x = (uint32_t)__shfl_xor((int)s_lo, o), y = (uint32_t)__shfl_xor((int)s_hi, o);
            s_lo = x < s_lo ? x : s_lo;
            s_hi = y > s_hi ? y : s_hi;
        }
        for(uint32_t s = s_lo; s < s_hi; ++s) {
            const bool act = in && s >= s0 && s < avail;
            const uint32_t c = act ? (uint32_t)w[s] : 0u;
            const bool nf = act && !st.fwd_broken, nr = act && !st.rvc_broken;
            const IvT<P> f2 = coop_update(sf, c, st.fwd, nf, X, lane, mtab, n_blk);
            const IvT<P> r2 = coop_update(sr, 3u - c, st.rvc, nr, X, lane, mtab, n_blk);
            if(nf) { st.fwd = f2; st.fwd_broken = f2.lo > f2.hi; n_rank += 2; }
            if(nr) { st.rvc = r2; st.rvc_broken = r2.lo > r2.hi; n_rank += 2; }
            if(act) {
                ++st.size;
                if(st.size == next_k) { emit(slot); ++slot; next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu; }
            }
        }
    } else if(in) {
        // per-lane walk (kmer_grid_kernel's loop)
        for(uint32_t s = s0; s < avail; ++s) {
            if(lean) {
                const uint32_t c = w[s];
                if(!st.fwd_broken) { st.fwd = update_interval<false>(sf, c, st.fwd, mtab, n_blk); st.fwd_broken = st.fwd.lo > st.fwd.hi; n_rank += 2; }
                if(!st.rvc_broken) { st.rvc = update_interval<false>(sr, 3u - c, st.rvc, mtab, n_blk); st.rvc_broken = st.rvc.lo > st.rvc.hi; n_rank += 2; }
                ++st.size;
            } else
                st = walk_step<false>(sf, sr, w[s], base_k, st, mtab);
            if(st.size == next_k) { emit(slot); ++slot; next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu; }
        }
        n_rank += st.n_rank; n_blk += st.n_blk;
    }
    if(in) {
        for(; slot < a.n_k; ++slot) emit(slot);
        if(a.base_counted) a.base_counted[gid] = (uint8_t)st.counted;
        if(a.valid_mask) a.valid_mask[gid] = (uint8_t)vmask;
    }
    flush_counters(ctr, n_rank, n_blk, n_tab);
}

hipError_t launch_kmer_grid_coop(const FmIndexDev& fm, const GridArgs& a, DevCounters* ctr, hipStream_t stream)
{
    if(a.total_bases == 0) return hipSuccess;
    if(fm.wide || !a.freq || a.out_iv || a.out_size || a.out_count || a.slot_iv) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kmer_grid_coop_kernel, dim3((unsigned)((a.total_bases + 255) / 256)), dim3(256), 0, stream, fm, a, ctr);
    return hipGetLastError();
}

} // namespace lrsc
