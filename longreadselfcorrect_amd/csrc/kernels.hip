// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the FM-index hot path.
//
// All of these are integer rank/select over a read-only multi-GB structure: the bound is
// gather bandwidth / latency of 64-byte blocks, not MFMA.  Design points (measured with
// tools/gather_bench.hip: random 64-B block gathers sustain ~145 G blocks/s on this chip, so the
// first version of this kernel at 42 G blocks/s was instruction-bound, not memory-bound):
//   * one Occ query == one 64-byte aligned block (4 x global_load_dwordx4 by one lane); the second
//     block of an updateInterval is only fetched when lower-1 and upper fall in different blocks;
//   * bit-plane blocks: popcount((lo ^ L) & (hi ^ H) & mask) per 32 symbols; the partial-word masks
//     come from a 6 KB LDS table indexed by the in-block offset (2 ds_reads instead of ~36 VALU);
//   * 32-bit position arithmetic when the index has < 2^31 symbols (Block32), 64-bit otherwise;
//   * the per-strand early exit of findInterval is predicated, not branched: every lane of a
//     wavefront executes the same instruction stream for every step of the k-mer walk;
//   * adjacent lanes own adjacent read positions: read bytes are coalesced, index blocks are
//     inherently random (BWT order), so the 64-byte block is the unit of traffic.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace lrsc {

// ---------------------------------------------------------------------------------------
// layouts
// ---------------------------------------------------------------------------------------
template <bool WIDE> struct Lay;

template <> struct Lay<false> {
    using pos_t = uint32_t;
    static constexpr uint32_t kSyms = Block32::kSyms;
    static constexpr uint32_t kWords = Block32::kWords;
    static constexpr uint32_t kRow = 8;                    // mask-table row stride (u32)
    struct Regs { uint4 q[4]; };                           // q0 = counts, q1..q3 = lo[6], hi[6]
    static __device__ __forceinline__ void load(const void* blocks, uint64_t b, Regs& r)
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const Block32*>(blocks) + b);
        r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    }
    static __device__ __forceinline__ bool flagged(const Regs& r) { return (r.q[0].x & kFlag32) != 0; }
    // symbols equal to `code` among the first `off` symbols of the block + the block's base count
    static __device__ __forceinline__ uint64_t count(const Regs& r, uint32_t code, const uint32_t* __restrict__ mrow)
    {
        const uint32_t base = code == 0 ? (r.q[0].x & ~kFlag32) : code == 1 ? r.q[0].y : code == 2 ? r.q[0].z : r.q[0].w;
        const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
        const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        const uint2 m1 = *reinterpret_cast<const uint2*>(mrow + 4);
        uint32_t c = base;
        c += __builtin_popcount((r.q[1].x ^ L) & (r.q[2].z ^ H) & m0.x);
        c += __builtin_popcount((r.q[1].y ^ L) & (r.q[2].w ^ H) & m0.y);
        c += __builtin_popcount((r.q[1].z ^ L) & (r.q[3].x ^ H) & m0.z);
        c += __builtin_popcount((r.q[1].w ^ L) & (r.q[3].y ^ H) & m0.w);
        c += __builtin_popcount((r.q[2].x ^ L) & (r.q[3].z ^ H) & m1.x);
        c += __builtin_popcount((r.q[2].y ^ L) & (r.q[3].w ^ H) & m1.y);
        return c;
    }
    static __device__ __forceinline__ uint32_t symbol(const Regs& r, uint32_t off)
    {
        const uint32_t lo[6] = {r.q[1].x, r.q[1].y, r.q[1].z, r.q[1].w, r.q[2].x, r.q[2].y};
        const uint32_t hi[6] = {r.q[2].z, r.q[2].w, r.q[3].x, r.q[3].y, r.q[3].z, r.q[3].w};
        uint32_t l = 0, h = 0;
#pragma unroll
        for(uint32_t i = 0; i < 6; ++i) { l = (off >> 5) == i ? lo[i] : l; h = (off >> 5) == i ? hi[i] : h; }
        return ((l >> (off & 31u)) & 1u) | (((h >> (off & 31u)) & 1u) << 1);
    }
};

template <> struct Lay<true> {
    using pos_t = uint64_t;
    static constexpr uint32_t kSyms = Block64::kSyms;
    static constexpr uint32_t kWords = Block64::kWords;
    static constexpr uint32_t kRow = 4;
    struct Regs { uint4 q[4]; };                           // q0,q1 = counts, q2 = lo[4], q3 = hi[4]
    static __device__ __forceinline__ void load(const void* blocks, uint64_t b, Regs& r)
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const Block64*>(blocks) + b);
        r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    }
    static __device__ __forceinline__ uint64_t u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
    static __device__ __forceinline__ bool flagged(const Regs& r) { return (r.q[0].y & 0x80000000u) != 0; }
    static __device__ __forceinline__ uint64_t count(const Regs& r, uint32_t code, const uint32_t* __restrict__ mrow)
    {
        const uint64_t base = code == 0 ? (u64(r.q[0].x, r.q[0].y) & ~kFlag64) : code == 1 ? u64(r.q[0].z, r.q[0].w)
                            : code == 2 ? u64(r.q[1].x, r.q[1].y) : u64(r.q[1].z, r.q[1].w);
        const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
        const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        uint32_t c = 0;
        c += __builtin_popcount((r.q[2].x ^ L) & (r.q[3].x ^ H) & m0.x);
        c += __builtin_popcount((r.q[2].y ^ L) & (r.q[3].y ^ H) & m0.y);
        c += __builtin_popcount((r.q[2].z ^ L) & (r.q[3].z ^ H) & m0.z);
        c += __builtin_popcount((r.q[2].w ^ L) & (r.q[3].w ^ H) & m0.w);
        return base + c;
    }
    static __device__ __forceinline__ uint32_t symbol(const Regs& r, uint32_t off)
    {
        const uint32_t lo[4] = {r.q[2].x, r.q[2].y, r.q[2].z, r.q[2].w};
        const uint32_t hi[4] = {r.q[3].x, r.q[3].y, r.q[3].z, r.q[3].w};
        uint32_t l = 0, h = 0;
#pragma unroll
        for(uint32_t i = 0; i < 4; ++i) { l = (off >> 5) == i ? lo[i] : l; h = (off >> 5) == i ? hi[i] : h; }
        return ((l >> (off & 31u)) & 1u) | (((h >> (off & 31u)) & 1u) << 1);
    }
};

// LDS table: row `off` holds the kWords partial-word masks for "the first off symbols of a block"
template <bool WIDE>
__device__ __forceinline__ void init_mask_table(uint32_t* tab)
{
    using L = Lay<WIDE>;
    for(uint32_t i = threadIdx.x; i < (L::kSyms + 1) * L::kRow; i += blockDim.x) {
        const uint32_t off = i / L::kRow, w = i % L::kRow;
        tab[i] = w < L::kWords ? low_mask((int32_t)off - 32 * (int32_t)w) : 0u;
    }
    __syncthreads();
}
template <bool WIDE> struct MaskTabSize { static constexpr uint32_t value = (Lay<WIDE>::kSyms + 1) * Lay<WIDE>::kRow; };

// Per-strand constants as plain scalars (wave-uniform, live in SGPRs).  Built from the kernel
// argument with constant member indices only: indexing the by-value argument struct dynamically
// makes the compiler copy it to scratch.
template <class P>
struct StrandC {
    const void* blocks;
    const uint64_t* dollars;
    uint64_t n_dollars;
    P c1, c2, c3, c4, n;      // C[A], C[C], C[G], C[T], N
};
template <class P>
__device__ __forceinline__ StrandC<P> strand_consts(const FmStrand& s)
{
    StrandC<P> c;
    c.blocks = s.blocks; c.dollars = s.dollars; c.n_dollars = s.n_dollars;
    c.c1 = (P)s.pred[1]; c.c2 = (P)s.pred[2]; c.c3 = (P)s.pred[3]; c.c4 = (P)s.pred[4]; c.n = (P)s.n_symbols;
    return c;
}
// C[code + 1] without a table: three predicated adds of uniform deltas
template <class P>
__device__ __forceinline__ P pred_of(const StrandC<P>& s, uint32_t code)
{
    P v = s.c1;
    v += code >= 1 ? (s.c2 - s.c1) : 0;
    v += code >= 2 ? (s.c3 - s.c2) : 0;
    v += code >= 3 ? (s.c4 - s.c3) : 0;
    return v;
}
// C[code + 2] (or N for T): upper end of the single-symbol interval
template <class P>
__device__ __forceinline__ P pred_next(const StrandC<P>& s, uint32_t code)
{
    P v = s.c2;
    v += code >= 1 ? (s.c3 - s.c2) : 0;
    v += code >= 2 ? (s.c4 - s.c3) : 0;
    v += code >= 3 ? (s.n - s.c4) : 0;
    return v;
}
template <class P>
__device__ __forceinline__ uint64_t dollars_in_c(const StrandC<P>& s, uint64_t lo, uint64_t hi)
{
    uint64_t a = 0, b = s.n_dollars;
    while(a < b) { const uint64_t m = (a + b) >> 1; if(s.dollars[m] < lo) a = m + 1; else b = m; }
    const uint64_t first = a;
    b = s.n_dollars;
    while(a < b) { const uint64_t m = (a + b) >> 1; if(s.dollars[m] < hi) a = m + 1; else b = m; }
    return a - first;
}

// Occ over the first p symbols (p = idx + 1, 0 <= p <= N): RLBWT::getOcc (RLBWT.h:121-140)
template <bool WIDE>
__device__ __forceinline__ uint64_t occ_prefix(const StrandC<typename Lay<WIDE>::pos_t>& s, uint32_t code,
                                               typename Lay<WIDE>::pos_t p, const uint32_t* __restrict__ mtab)
{
    using L = Lay<WIDE>;
    const typename L::pos_t b = p / L::kSyms;
    const uint32_t off = (uint32_t)(p - b * L::kSyms);
    typename L::Regs r;
    L::load(s.blocks, b, r);
    uint64_t c = L::count(r, code, mtab + off * L::kRow);
    if(code == 0 && off != 0 && L::flagged(r)) c -= dollars_in_c(s, (uint64_t)b * L::kSyms, (uint64_t)b * L::kSyms + off);
    return c;
}

template <class P> struct IvT { P lo, hi; };   // lower, upper; upper stored as-is (>= 0 always: pred >= 1)

// BWTAlgorithms::updateInterval (BWTAlgorithms.h:66-72) on one strand.  The second block is only
// loaded when the two rank positions straddle a block boundary.
template <bool WIDE>
__device__ __forceinline__ IvT<typename Lay<WIDE>::pos_t> update_interval(const StrandC<typename Lay<WIDE>::pos_t>& s, uint32_t code,
                                                                          IvT<typename Lay<WIDE>::pos_t> iv,
                                                                          const uint32_t* __restrict__ mtab, uint32_t& n_blk)
{
    using L = Lay<WIDE>;
    using P = typename L::pos_t;
    const P pl = iv.lo;            // (lower - 1) + 1
    const P pu = iv.hi + 1;        // upper + 1
    const P bl = pl / L::kSyms, bu = pu / L::kSyms;
    const uint32_t ol = (uint32_t)(pl - bl * L::kSyms), ou = (uint32_t)(pu - bu * L::kSyms);
    typename L::Regs ra, rb;
    L::load(s.blocks, bl, ra);
    rb = ra;
    if(bu != bl) L::load(s.blocks, bu, rb);
    uint64_t ca = L::count(ra, code, mtab + ol * L::kRow);
    uint64_t cb = L::count(rb, code, mtab + ou * L::kRow);
    if(code == 0) {
        if(ol != 0 && L::flagged(ra)) ca -= dollars_in_c(s, (uint64_t)bl * L::kSyms, (uint64_t)bl * L::kSyms + ol);
        if(ou != 0 && L::flagged(rb)) cb -= dollars_in_c(s, (uint64_t)bu * L::kSyms, (uint64_t)bu * L::kSyms + ou);
    }
    const P pb = pred_of(s, code);
    IvT<P> out;
    out.lo = pb + (P)ca;
    out.hi = pb + (P)cb - 1;
    n_blk += (bl == bu) ? 1u : 2u;
    return out;
}

// BWTAlgorithms::initInterval (BWTAlgorithms.h:136-140): Occ(b, N-1) is the symbol total.
template <class P>
__device__ __forceinline__ IvT<P> init_interval(const StrandC<P>& s, uint32_t code)
{
    IvT<P> iv;
    iv.lo = pred_of(s, code);
    iv.hi = pred_next(s, code) - 1;
    return iv;
}

__device__ __forceinline__ void flush_counters(DevCounters* ctr, uint32_t n_rank, uint32_t n_blk)
{
    if(ctr == nullptr) return;
    unsigned long long a = n_rank, b = n_blk;
#pragma unroll
    for(int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    if((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr->rank_queries, a);
        atomicAdd(&ctr->block_loads, b);
    }
}

// ---------------------------------------------------------------------------------------
// The k-mer walk shared by lrsc_find_kmers and lrsc_kmer_grid.
//
// A lane owns one start position and steps left-to-right through the read: the fwd interval
// is the backward search of reverse(w) in the rbwt, the rvc interval the backward search of
// revcomp(w) in the bwt, so both consume w[0], w[1], ... in order (BWTAlgorithms.cpp:32-38).
// Steps < base_k reproduce findInterval's early exit per strand (BWTAlgorithms.cpp:28): a strand
// that went invalid keeps its interval until step base_k; steps >= base_k are
// KmerFeature::expand (KmerFeature.h:92-99): always applied, no validity check.
// ---------------------------------------------------------------------------------------
template <class P>
struct WalkState {
    IvT<P> fwd, rvc;
    uint32_t size;          // bases consumed
    uint32_t counted;       // bases counted by the base search (fwd strand)
    uint32_t n_rank, n_blk; // accounting: Occ queries issued / rank blocks needed
    bool fwd_broken, rvc_broken;
};
template <class P>
__device__ __forceinline__ WalkState<P> walk_init()
{
    WalkState<P> st;
    st.size = 0; st.counted = 0; st.n_rank = 0; st.n_blk = 0; st.fwd_broken = false; st.rvc_broken = false;
    st.fwd.lo = st.fwd.hi = st.rvc.lo = st.rvc.hi = 0;
    return st;
}

template <bool WIDE>
__device__ __forceinline__ WalkState<typename Lay<WIDE>::pos_t>
walk_step(const StrandC<typename Lay<WIDE>::pos_t>& sf, const StrandC<typename Lay<WIDE>::pos_t>& sr, uint32_t c,
          uint32_t base_k, WalkState<typename Lay<WIDE>::pos_t> st, const uint32_t* __restrict__ mtab)
{
    using P = typename Lay<WIDE>::pos_t;
    if(st.size == 0) {
        st.fwd = init_interval<P>(sf, c);
        st.rvc = init_interval<P>(sr, 3u - c);
        st.counted = 1;
        st.n_rank += 2;   // initInterval's getOcc(b, N-1) per strand (served from C[] here)
    } else {
        const bool in_base = st.size < base_k;
        const bool do_f = !(in_base && st.fwd_broken);
        const bool do_r = !(in_base && st.rvc_broken);
        uint32_t bf = 0, br = 0;
        const IvT<P> nf = update_interval<WIDE>(sf, c, st.fwd, mtab, bf);
        const IvT<P> nr = update_interval<WIDE>(sr, 3u - c, st.rvc, mtab, br);
        if(do_f) {
            st.fwd = nf;
            st.counted += in_base ? 1u : 0u;
            st.fwd_broken = in_base && (nf.lo > nf.hi);
            st.n_rank += 2; st.n_blk += bf;
        }
        if(do_r) {
            st.rvc = nr;
            st.rvc_broken = in_base && (nr.lo > nr.hi);
            st.n_rank += 2; st.n_blk += br;
        }
    }
    ++st.size;
    return st;
}

template <class P> __device__ __forceinline__ int64_t iv_freq(const IvT<P>& iv) { return iv.lo <= iv.hi ? (int64_t)(iv.hi - iv.lo) + 1 : 0; }
template <class P> __device__ __forceinline__ lrsc_biinterval to_out(const IvT<P>& f, const IvT<P>& r)
{
    lrsc_biinterval o;
    o.fwd.lower = (int64_t)f.lo; o.fwd.upper = (int64_t)f.hi;
    o.rvc.lower = (int64_t)r.lo; o.rvc.upper = (int64_t)r.hi;
    return o;
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void rank_kernel(FmIndexDev fm, const lrsc_rank_query* __restrict__ q,
                                                   uint64_t n, uint64_t* __restrict__ out, DevCounters* ctr)
{
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(i < n) {
        const lrsc_rank_query qq = q[i];
        const uint32_t code = ((qq.base >> 1) & 3u) ^ (((qq.base >> 1) & 3u) >> 1);
        using P = typename Lay<WIDE>::pos_t;
        const StrandC<P> s0 = strand_consts<P>(fm.strand[0]);
        const StrandC<P> s1 = strand_consts<P>(fm.strand[1]);
        const uint64_t r0 = occ_prefix<WIDE>(s0, code, (P)(qq.idx + 1), mtab);
        const uint64_t r1 = occ_prefix<WIDE>(s1, code, (P)(qq.idx + 1), mtab);
        out[i] = (qq.strand & 1) ? r1 : r0;
        n_rank = 1; n_blk = 1;
    }
    flush_counters(ctr, n_rank, n_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void bwt_chars_kernel(FmIndexDev fm, int strand, const uint64_t* __restrict__ idx,
                                                        uint64_t n, char* __restrict__ out)
{
    using L = Lay<WIDE>;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= n) return;
    const StrandC<uint64_t> s = (strand & 1) ? strand_consts<uint64_t>(fm.strand[1]) : strand_consts<uint64_t>(fm.strand[0]);
    const uint64_t p = idx[i];
    const uint64_t b = p / L::kSyms;
    const uint32_t off = (uint32_t)(p - b * L::kSyms);
    typename L::Regs r;
    L::load(s.blocks, b, r);
    const uint32_t code = L::symbol(r, off);
    char ch = "ACGT"[code];
    if(code == 0 && L::flagged(r) && dollars_in_c(s, p, p + 1) != 0) ch = '$';
    out[i] = ch;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void find_kmers_kernel(FmIndexDev fm, const uint8_t* __restrict__ codes, uint32_t k,
                                                         uint64_t n, lrsc_biinterval* __restrict__ out, DevCounters* ctr)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(i < n) {
        const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
        const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
        WalkState<P> st = walk_init<P>();
        const uint8_t* w = codes + i * k;
        for(uint32_t s = 0; s < k; ++s) {
            if(st.fwd_broken && st.rvc_broken) break;
            st = walk_step<WIDE>(sf, sr, w[s], k, st, mtab);
        }
        out[i] = to_out(st.fwd, st.rvc);
        n_rank = st.n_rank; n_blk = st.n_blk;
    }
    flush_counters(ctr, n_rank, n_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void kmer_grid_kernel(FmIndexDev fm, GridArgs a, DevCounters* ctr)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(gid < a.total_bases) {
        // locate the read: coarse table + short forward scan
        uint32_t r = a.chunk_read[gid >> kChunkShift];
        while(a.read_off[r + 1] <= gid) ++r;
        const uint64_t end = a.read_off[r + 1];
        const uint32_t kmax = a.ks[a.n_k - 1];
        const uint32_t base_k = a.ks[0];
        const uint64_t remain64 = end - gid;
        const uint32_t avail = remain64 < kmax ? (uint32_t)remain64 : kmax;

        const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
        const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
        WalkState<P> st = walk_init<P>();
        uint32_t slot = 0;
        uint32_t next_k = a.ks[0];
        const uint8_t* w = a.codes + gid;

        auto emit = [&](uint32_t j) {
            const uint64_t rec = gid * a.n_k + j;
            if(a.out_iv) a.out_iv[rec] = to_out(st.fwd, st.rvc);
            if(a.out_size) a.out_size[rec] = (uint8_t)st.size;
            if(a.out_count) {
                // composition of the counted bases: w[0..counted) from the base search, then
                // w[base_k..size) from expand()
                uint32_t cnt[4] = {0, 0, 0, 0};
                for(uint32_t t = 0; t < st.size; ++t) {
                    const bool in = (t < base_k) ? (t < st.counted) : true;
                    if(in) {
                        const uint32_t c = w[t];
                        cnt[0] += (c == 0); cnt[1] += (c == 1); cnt[2] += (c == 2); cnt[3] += (c == 3);
                    }
                }
                reinterpret_cast<uchar4*>(a.out_count)[rec] =
                    make_uchar4((uint8_t)cnt[0], (uint8_t)cnt[1], (uint8_t)cnt[2], (uint8_t)cnt[3]);
            }
            if(a.freq) {
                const bool fake = st.size != a.ks[j];
                a.freq[(uint64_t)j * a.total_bases + gid] = fake ? -1 : (int32_t)(iv_freq(st.fwd) + iv_freq(st.rvc));
            }
            if(a.slot_iv) a.slot_iv[(uint64_t)j * a.total_bases + gid] = to_out(st.fwd, st.rvc);
        };

        for(uint32_t s = 0; s < avail; ++s) {
            st = walk_step<WIDE>(sf, sr, w[s], base_k, st, mtab);
            if(st.size == next_k) {
                emit(slot);
                ++slot;
                next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu;
            }
        }
        // slots the read end cut short keep the last state ("fake" k-mers, KmerFeature.h:62)
        for(; slot < a.n_k; ++slot) emit(slot);
        if(a.base_counted) a.base_counted[gid] = (uint8_t)st.counted;
        n_rank = st.n_rank; n_blk = st.n_blk;
    }
    flush_counters(ctr, n_rank, n_blk);
}

__global__ __launch_bounds__(256) void encode_kernel(const char* __restrict__ ascii, uint8_t* __restrict__ codes,
                                                     uint64_t n, int* bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= n) return;
    const uint8_t c = (uint8_t)ascii[i];
    const uint32_t x = (c >> 1) & 3u;
    const uint32_t code = x ^ (x >> 1);
    if(c != 'A' && c != 'C' && c != 'G' && c != 'T') *bad = 1;
    codes[i] = (uint8_t)code;
}

__global__ __launch_bounds__(256) void chunk_table_kernel(const uint64_t* __restrict__ read_off, uint32_t n_reads,
                                                          uint64_t total_bases, uint32_t* __restrict__ chunk_read)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n_chunks = (total_bases + (1ull << kChunkShift) - 1) >> kChunkShift;
    if(c >= n_chunks) return;
    const uint64_t pos = c << kChunkShift;
    // largest r with read_off[r] <= pos
    uint32_t lo = 0, hi = n_reads;   // invariant: read_off[lo] <= pos < read_off[hi] (read_off[n] = total > pos)
    while(hi - lo > 1) {
        const uint32_t m = lo + ((hi - lo) >> 1);
        if(read_off[m] <= pos) lo = m; else hi = m;
    }
    chunk_read[c] = lo;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_rank(const FmIndexDev& fm, const lrsc_rank_query* q, uint64_t n, uint64_t* out,
                       DevCounters* ctr, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(rank_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, q, n, out, ctr);
    else        hipLaunchKernelGGL(rank_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, q, n, out, ctr);
    return hipGetLastError();
}

hipError_t launch_bwt_chars(const FmIndexDev& fm, int strand, const uint64_t* idx, uint64_t n, char* out,
                            hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(bwt_chars_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, strand, idx, n, out);
    else        hipLaunchKernelGGL(bwt_chars_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, strand, idx, n, out);
    return hipGetLastError();
}

hipError_t launch_find_kmers(const FmIndexDev& fm, const uint8_t* kmer_codes, uint32_t k, uint64_t n,
                             lrsc_biinterval* out, DevCounters* ctr, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(find_kmers_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, kmer_codes, k, n, out, ctr);
    else        hipLaunchKernelGGL(find_kmers_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, kmer_codes, k, n, out, ctr);
    return hipGetLastError();
}

hipError_t launch_kmer_grid(const FmIndexDev& fm, const GridArgs& a, DevCounters* ctr, hipStream_t stream)
{
    if(a.total_bases == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(kmer_grid_kernel<true>, dim3(blocks_for(a.total_bases)), dim3(256), 0, stream, fm, a, ctr);
    else        hipLaunchKernelGGL(kmer_grid_kernel<false>, dim3(blocks_for(a.total_bases)), dim3(256), 0, stream, fm, a, ctr);
    return hipGetLastError();
}

hipError_t launch_encode(const char* ascii, uint8_t* codes, uint64_t n, int* bad, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    hipLaunchKernelGGL(encode_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, ascii, codes, n, bad);
    return hipGetLastError();
}

hipError_t launch_chunk_table(const uint64_t* read_off, uint32_t n_reads, uint64_t total_bases,
                              uint32_t* chunk_read, hipStream_t stream)
{
    const uint64_t n_chunks = (total_bases + (1ull << kChunkShift) - 1) >> kChunkShift;
    if(n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_table_kernel, dim3(blocks_for(n_chunks)), dim3(256), 0, stream, read_off, n_reads, total_bases, chunk_read);
    return hipGetLastError();
}

} // namespace lrsc
