// rank_device.h -- device-side FM-index primitives shared by every kernel file: block layouts,
// in-block rank (bit planes + LDS mask table), updateInterval / initInterval, the k-mer walk.
// Included only from .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

// The block's base count for a symbol code, picked from the four counters by the code's two BITS (a binary tree of selects) and not by
// an equality chain `code == 0 ? a : code == 1 ? b : code == 2 ? c : d`: the chain is mis-lowered by ROCm 7.2 -O3 when the compiler
// cannot bound the code (kSelectChainNote below); bit tests are total functions of any 32-bit value, so no switch can be formed.
#ifndef LRSC_PICK4
#define LRSC_PICK4(code, a, b, c, d) (((code) & 2u) ? (((code) & 1u) ? (d) : (c)) : (((code) & 1u) ? (b) : (a)))
#endif

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
    struct Regs { uint4 q[4]; };                           // q0 = counts, q[j] = {lo[2j-2], lo[2j-1], hi[2j-2], hi[2j-1]}
    static __host__ __device__ __forceinline__ void load(const void* blocks, uint64_t b, Regs& r)
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const Block32*>(blocks) + b);
        r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    }
    static __host__ __device__ __forceinline__ bool flagged(const Regs& r) { return (r.q[0].x & kFlag32) != 0; }
    // symbols equal to `code` among the first `off` symbols of the block + the block's base count
    static __host__ __device__ __forceinline__ uint64_t count(const Regs& r, uint32_t code, const uint32_t* __restrict__ mrow)
    {
        const uint32_t base = LRSC_PICK4(code, r.q[0].x & ~kFlag32, r.q[0].y, r.q[0].z, r.q[0].w);     // kSelectChainNote below
        const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
        const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        const uint2 m1 = *reinterpret_cast<const uint2*>(mrow + 4);
        uint32_t c = base;
        c += __builtin_popcount((r.q[1].x ^ L) & (r.q[1].z ^ H) & m0.x);
        c += __builtin_popcount((r.q[1].y ^ L) & (r.q[1].w ^ H) & m0.y);
        c += __builtin_popcount((r.q[2].x ^ L) & (r.q[2].z ^ H) & m0.z);
        c += __builtin_popcount((r.q[2].y ^ L) & (r.q[2].w ^ H) & m0.w);
        c += __builtin_popcount((r.q[3].x ^ L) & (r.q[3].z ^ H) & m1.x);
        c += __builtin_popcount((r.q[3].y ^ L) & (r.q[3].w ^ H) & m1.y);
        return c;
    }
    // the same for two prefix lengths of one block
    static __host__ __device__ __forceinline__ void count2(const Regs& r, uint32_t code, const uint32_t* __restrict__ mrow_a,
                                                  const uint32_t* __restrict__ mrow_b, uint64_t& ca, uint64_t& cb)
    {
        const uint32_t base = LRSC_PICK4(code, r.q[0].x & ~kFlag32, r.q[0].y, r.q[0].z, r.q[0].w);
        const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
        const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
        const uint32_t m[6] = {(r.q[1].x ^ L) & (r.q[1].z ^ H), (r.q[1].y ^ L) & (r.q[1].w ^ H), (r.q[2].x ^ L) & (r.q[2].z ^ H),
                               (r.q[2].y ^ L) & (r.q[2].w ^ H), (r.q[3].x ^ L) & (r.q[3].z ^ H), (r.q[3].y ^ L) & (r.q[3].w ^ H)};
        const uint4 a0 = *reinterpret_cast<const uint4*>(mrow_a);
        const uint2 a1 = *reinterpret_cast<const uint2*>(mrow_a + 4);
        const uint4 b0 = *reinterpret_cast<const uint4*>(mrow_b);
        const uint2 b1 = *reinterpret_cast<const uint2*>(mrow_b + 4);
        uint32_t x = base, y = base;
        x += __builtin_popcount(m[0] & a0.x); y += __builtin_popcount(m[0] & b0.x);
        x += __builtin_popcount(m[1] & a0.y); y += __builtin_popcount(m[1] & b0.y);
        x += __builtin_popcount(m[2] & a0.z); y += __builtin_popcount(m[2] & b0.z);
        x += __builtin_popcount(m[3] & a0.w); y += __builtin_popcount(m[3] & b0.w);
        x += __builtin_popcount(m[4] & a1.x); y += __builtin_popcount(m[4] & b1.x);
        x += __builtin_popcount(m[5] & a1.y); y += __builtin_popcount(m[5] & b1.y);
        ca = x; cb = y;
    }
    static __host__ __device__ __forceinline__ uint32_t symbol(const Regs& r, uint32_t off)
    {
        const uint32_t lo[6] = {r.q[1].x, r.q[1].y, r.q[2].x, r.q[2].y, r.q[3].x, r.q[3].y};
        const uint32_t hi[6] = {r.q[1].z, r.q[1].w, r.q[2].z, r.q[2].w, r.q[3].z, r.q[3].w};
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
    static __host__ __device__ __forceinline__ void load(const void* blocks, uint64_t b, Regs& r)
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const Block64*>(blocks) + b);
        r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    }
    static __host__ __device__ __forceinline__ uint64_t u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
    static __host__ __device__ __forceinline__ bool flagged(const Regs& r) { return (r.q[0].y & 0x80000000u) != 0; }
    static __host__ __device__ __forceinline__ uint64_t count(const Regs& r, uint32_t code, const uint32_t* __restrict__ mrow)
    {
        const uint64_t base = LRSC_PICK4(code, u64(r.q[0].x, r.q[0].y) & ~kFlag64, u64(r.q[0].z, r.q[0].w), u64(r.q[1].x, r.q[1].y), u64(r.q[1].z, r.q[1].w));
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
    static __host__ __device__ __forceinline__ void count2(const Regs& r, uint32_t code, const uint32_t* __restrict__ mrow_a,
                                                  const uint32_t* __restrict__ mrow_b, uint64_t& ca, uint64_t& cb)
    {
        const uint64_t base = LRSC_PICK4(code, u64(r.q[0].x, r.q[0].y) & ~kFlag64, u64(r.q[0].z, r.q[0].w), u64(r.q[1].x, r.q[1].y), u64(r.q[1].z, r.q[1].w));
        const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
        const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
        const uint32_t m[4] = {(r.q[2].x ^ L) & (r.q[3].x ^ H), (r.q[2].y ^ L) & (r.q[3].y ^ H), (r.q[2].z ^ L) & (r.q[3].z ^ H),
                               (r.q[2].w ^ L) & (r.q[3].w ^ H)};
        const uint4 a0 = *reinterpret_cast<const uint4*>(mrow_a);
        const uint4 b0 = *reinterpret_cast<const uint4*>(mrow_b);
        uint32_t x = 0, y = 0;
        x += __builtin_popcount(m[0] & a0.x); y += __builtin_popcount(m[0] & b0.x);
        x += __builtin_popcount(m[1] & a0.y); y += __builtin_popcount(m[1] & b0.y);
        x += __builtin_popcount(m[2] & a0.z); y += __builtin_popcount(m[2] & b0.z);
        x += __builtin_popcount(m[3] & a0.w); y += __builtin_popcount(m[3] & b0.w);
        ca = base + x; cb = base + y;
    }
    static __host__ __device__ __forceinline__ uint32_t symbol(const Regs& r, uint32_t off)
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
// the same table filled by one thread (host-side emulation harness under tests/, single-lane kernels)
template <bool WIDE>
__host__ __device__ inline void fill_mask_table_serial(uint32_t* tab)
{
    using L = Lay<WIDE>;
    for(uint32_t i = 0; i < (L::kSyms + 1) * L::kRow; ++i) {
        const uint32_t off = i / L::kRow, w = i % L::kRow;
        tab[i] = w < L::kWords ? low_mask((int32_t)off - 32 * (int32_t)w) : 0u;
    }
}
template <bool WIDE> struct MaskTabSize { static constexpr uint32_t value = (Lay<WIDE>::kSyms + 1) * Lay<WIDE>::kRow; };

// Per-strand constants as plain scalars (wave-uniform, live in SGPRs).  Built from the kernel
// argument with constant member indices only: indexing the by-value argument struct dynamically
// makes the compiler copy it to scratch.
template <class P>
struct StrandC {
    const void* blocks;
    const uint64_t* dollars;
    const uint32_t* dollar_dir;
    uint64_t n_dollars;
    uint32_t syms_per_group;  // symbols of one '$'-directory group
    P c1, c2, c3, c4, n;      // C[A], C[C], C[G], C[T], N
};
template <class P>
__host__ __device__ __forceinline__ StrandC<P> strand_consts(const FmStrand& s)
{
    StrandC<P> c;
    c.blocks = s.blocks; c.dollars = s.dollars; c.dollar_dir = s.dollar_dir; c.n_dollars = s.n_dollars;
    c.syms_per_group = (uint32_t)s.dollar_group_syms;
    c.c1 = (P)s.pred[1]; c.c2 = (P)s.pred[2]; c.c3 = (P)s.pred[3]; c.c4 = (P)s.pred[4]; c.n = (P)s.n_symbols;
    return c;
}
// C[code + 1] without a table: three predicated adds of uniform deltas
template <class P>
__host__ __device__ __forceinline__ P pred_of(const StrandC<P>& s, uint32_t code)
{
    P v = s.c1;
    v += code >= 1 ? (s.c2 - s.c1) : 0;
    v += code >= 2 ? (s.c3 - s.c2) : 0;
    v += code >= 3 ? (s.c4 - s.c3) : 0;
    return v;
}
// C[code + 2] (or N for T): upper end of the single-symbol interval
template <class P>
__host__ __device__ __forceinline__ P pred_next(const StrandC<P>& s, uint32_t code)
{
    P v = s.c2;
    v += code >= 1 ? (s.c3 - s.c2) : 0;
    v += code >= 2 ? (s.c4 - s.c3) : 0;
    v += code >= 3 ? (s.n - s.c4) : 0;
    return v;
}
// '$' rows in [lo, hi), lo and hi inside one rank block: the directory entry of the block's group gives the first list entry that
// can matter; a group holds a fraction of a '$' on average (one per read in >= 1024 symbols), so the scan is one or two entries
template <class P>
__host__ __device__ __forceinline__ uint64_t dollars_in_c(const StrandC<P>& s, uint64_t lo, uint64_t hi)
{
    uint64_t j = s.dollar_dir[lo / s.syms_per_group];
    uint64_t n = 0;
    for(; j < s.n_dollars; ++j) {
        const uint64_t d = s.dollars[j];
        if(d >= hi) break;
        n += d >= lo ? 1u : 0u;
    }
    return n;
}

// Occ over the first p symbols (p = idx + 1, 0 <= p <= N): RLBWT::getOcc (RLBWT.h:121-140)
template <bool WIDE>
__host__ __device__ __forceinline__ uint64_t occ_prefix(const StrandC<typename Lay<WIDE>::pos_t>& s, uint32_t code,
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
// FAST2: take the shared-match-words path when both rank positions fall in one block.  It pays where few lanes of a
// wavefront are active (the correction kernel); in the position-parallel grid kernel some lane nearly always straddles
// a block boundary, both branches would run, and the branch-free form is cheaper.
template <bool WIDE, bool FAST2 = true>
__host__ __device__ __forceinline__ IvT<typename Lay<WIDE>::pos_t> update_interval(const StrandC<typename Lay<WIDE>::pos_t>& s, uint32_t code,
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
    uint64_t ca, cb;
    if(FAST2 && bu == bl) {
        // both rank positions in one block (the rule once an interval is small): match words once, two masked popcounts
        L::count2(ra, code, mtab + ol * L::kRow, mtab + ou * L::kRow, ca, cb);
        rb = ra;
    } else {
        rb = ra;
        if(bu != bl) L::load(s.blocks, bu, rb);
        ca = L::count(ra, code, mtab + ol * L::kRow);
        cb = L::count(rb, code, mtab + ou * L::kRow);
    }
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

// ---------------------------------------------------------------------------------------
// kSelectChainNote -- the compiler finding of rounds 1-2, closed in round 3 (profiles/r03_compiler_finding/README.md,
// tools/repro_complement/repro.hip).  Lay::count / count2 used to pick the block's base count with an equality chain over the
// symbol code.  When the compiler cannot bound the code to [0, 3] (a caller formed it as `3u - byte`), ROCm 7.2 -O3 lowers such a
// chain as a switch and leaves the code-3 arm without its `v_mov base, cnt[3]`: intervals that consumed a 'T' come back off by
// a constant.  The chain is gone: LRSC_PICK4 selects by the code's two bits.  The variants below fetch the base count by its own
// dword load at block + 4 * code instead (no select at all; the load hits the line that is being fetched anyway); they date from
// the time the cause was not pinned and stay where they are measured no slower.  Counts never reach the flag bit (N < 2^31 for
// Block32, < 2^63 for Block64), so the flag is masked off for every code.
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__host__ __device__ __forceinline__ uint64_t block_base(const void* blocks, uint64_t b, uint32_t code)
{
    if(WIDE) return reinterpret_cast<const uint64_t*>(reinterpret_cast<const Block64*>(blocks) + b)[code & 3u] & ~kFlag64;
    return reinterpret_cast<const uint32_t*>(reinterpret_cast<const Block32*>(blocks) + b)[code & 3u] & ~kFlag32;
}
// symbols equal to `code` among the first `off` symbols of the block (mask row given), without the base count
template <bool WIDE>
__host__ __device__ __forceinline__ uint32_t block_popc(const typename Lay<WIDE>::Regs& r, uint32_t code, const uint32_t* __restrict__ mrow)
{
    const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu;
    const uint32_t H = (code & 2u) ? 0u : 0xFFFFFFFFu;
    uint32_t c = 0;
    if(WIDE) {
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        c += __builtin_popcount((r.q[2].x ^ L) & (r.q[3].x ^ H) & m0.x);
        c += __builtin_popcount((r.q[2].y ^ L) & (r.q[3].y ^ H) & m0.y);
        c += __builtin_popcount((r.q[2].z ^ L) & (r.q[3].z ^ H) & m0.z);
        c += __builtin_popcount((r.q[2].w ^ L) & (r.q[3].w ^ H) & m0.w);
    } else {
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        const uint2 m1 = *reinterpret_cast<const uint2*>(mrow + 4);
        c += __builtin_popcount((r.q[1].x ^ L) & (r.q[1].z ^ H) & m0.x);
        c += __builtin_popcount((r.q[1].y ^ L) & (r.q[1].w ^ H) & m0.y);
        c += __builtin_popcount((r.q[2].x ^ L) & (r.q[2].z ^ H) & m0.z);
        c += __builtin_popcount((r.q[2].y ^ L) & (r.q[2].w ^ H) & m0.w);
        c += __builtin_popcount((r.q[3].x ^ L) & (r.q[3].z ^ H) & m1.x);
        c += __builtin_popcount((r.q[3].y ^ L) & (r.q[3].w ^ H) & m1.y);
    }
    return c;
}
template <bool WIDE>
__host__ __device__ __forceinline__ IvT<typename Lay<WIDE>::pos_t> update_interval_b(const StrandC<typename Lay<WIDE>::pos_t>& s, uint32_t code,
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
    L::load(s.blocks, bu, rb);                     // same line as ra in the common case: an L1 hit
    const uint64_t base_a = block_base<WIDE>(s.blocks, bl, code);
    const uint64_t base_b = block_base<WIDE>(s.blocks, bu, code);
    uint64_t ca = base_a + block_popc<WIDE>(ra, code, mtab + ol * L::kRow);
    uint64_t cb = base_b + block_popc<WIDE>(rb, code, mtab + ou * L::kRow);
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

// updateInterval for all four codes of one strand at once (getFMIndexExtensions, LongReadCorrectByOverlap.cpp:687-698):
// the four Occ(c, lo - 1) come from one block and the four Occ(c, hi) from one block, whatever c is.
template <bool WIDE>
__host__ __device__ __forceinline__ void block_popc4(const typename Lay<WIDE>::Regs& r, const uint32_t* __restrict__ mrow, uint32_t c[4])
{
    uint32_t lo[6], hi[6], m[6];
    if(WIDE) {
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        lo[0] = r.q[2].x; lo[1] = r.q[2].y; lo[2] = r.q[2].z; lo[3] = r.q[2].w; lo[4] = 0; lo[5] = 0;
        hi[0] = r.q[3].x; hi[1] = r.q[3].y; hi[2] = r.q[3].z; hi[3] = r.q[3].w; hi[4] = 0; hi[5] = 0;
        m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = 0; m[5] = 0;
    } else {
        const uint4 m0 = *reinterpret_cast<const uint4*>(mrow);
        const uint2 m1 = *reinterpret_cast<const uint2*>(mrow + 4);
        lo[0] = r.q[1].x; lo[1] = r.q[1].y; lo[2] = r.q[2].x; lo[3] = r.q[2].y; lo[4] = r.q[3].x; lo[5] = r.q[3].y;
        hi[0] = r.q[1].z; hi[1] = r.q[1].w; hi[2] = r.q[2].z; hi[3] = r.q[2].w; hi[4] = r.q[3].z; hi[5] = r.q[3].w;
        m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y;
    }
    c[0] = c[1] = c[2] = c[3] = 0;
#pragma unroll
    for(int w = 0; w < (WIDE ? 4 : 6); ++w) {
        const uint32_t l = lo[w], h = hi[w], mk = m[w];
        c[0] += __builtin_popcount(~l & ~h & mk);
        c[1] += __builtin_popcount(l & ~h & mk);
        c[2] += __builtin_popcount(~l & h & mk);
        c[3] += __builtin_popcount(l & h & mk);
    }
}
// updateInterval for all four bases at once: the two rank positions' blocks are loaded once, every symbol class is counted from
// the same registers.  SKIP2: do not load the second block when both positions share one (pays where few lanes are active).
template <bool WIDE, bool SKIP2 = false>
__host__ __device__ __forceinline__ void update_interval_all(const StrandC<typename Lay<WIDE>::pos_t>& s, IvT<typename Lay<WIDE>::pos_t> iv,
                                                             const uint32_t* __restrict__ mtab, IvT<typename Lay<WIDE>::pos_t> out[4], uint32_t& n_blk)
{
    using L = Lay<WIDE>;
    using P = typename L::pos_t;
    const P pl = iv.lo, pu = iv.hi + 1;
    const P bl = pl / L::kSyms, bu = pu / L::kSyms;
    const uint32_t ol = (uint32_t)(pl - bl * L::kSyms), ou = (uint32_t)(pu - bu * L::kSyms);
    typename L::Regs ra, rb;
    L::load(s.blocks, bl, ra);
    if(SKIP2) { rb = ra; if(bu != bl) L::load(s.blocks, bu, rb); }
    else L::load(s.blocks, bu, rb);
    uint32_t pa[4], pb4[4];
    block_popc4<WIDE>(ra, mtab + ol * L::kRow, pa);
    block_popc4<WIDE>(rb, mtab + ou * L::kRow, pb4);
    uint64_t ba[4], bb[4];
    if(WIDE) {
        ba[0] = Lay<true>::u64(ra.q[0].x, ra.q[0].y) & ~kFlag64; ba[1] = Lay<true>::u64(ra.q[0].z, ra.q[0].w); ba[2] = Lay<true>::u64(ra.q[1].x, ra.q[1].y); ba[3] = Lay<true>::u64(ra.q[1].z, ra.q[1].w);
        bb[0] = Lay<true>::u64(rb.q[0].x, rb.q[0].y) & ~kFlag64; bb[1] = Lay<true>::u64(rb.q[0].z, rb.q[0].w); bb[2] = Lay<true>::u64(rb.q[1].x, rb.q[1].y); bb[3] = Lay<true>::u64(rb.q[1].z, rb.q[1].w);
    } else {
        ba[0] = ra.q[0].x & ~kFlag32; ba[1] = ra.q[0].y; ba[2] = ra.q[0].z; ba[3] = ra.q[0].w;
        bb[0] = rb.q[0].x & ~kFlag32; bb[1] = rb.q[0].y; bb[2] = rb.q[0].z; bb[3] = rb.q[0].w;
    }
    uint64_t ca0 = ba[0] + pa[0], cb0 = bb[0] + pb4[0];
    if(ol != 0 && L::flagged(ra)) ca0 -= dollars_in_c(s, (uint64_t)bl * L::kSyms, (uint64_t)bl * L::kSyms + ol);
    if(ou != 0 && L::flagged(rb)) cb0 -= dollars_in_c(s, (uint64_t)bu * L::kSyms, (uint64_t)bu * L::kSyms + ou);
    out[0].lo = s.c1 + (P)ca0;               out[0].hi = s.c1 + (P)cb0 - 1;
    out[1].lo = s.c2 + (P)(ba[1] + pa[1]);   out[1].hi = s.c2 + (P)(bb[1] + pb4[1]) - 1;
    out[2].lo = s.c3 + (P)(ba[2] + pa[2]);   out[2].hi = s.c3 + (P)(bb[2] + pb4[2]) - 1;
    out[3].lo = s.c4 + (P)(ba[3] + pa[3]);   out[3].hi = s.c4 + (P)(bb[3] + pb4[3]) - 1;
    n_blk += (bl == bu) ? 1u : 2u;
}

// two updateIntervals (one per strand, possibly different codes) with all eight piece loads of each side issued before
// anything waits: one memory round trip instead of two
template <bool WIDE>
__host__ __device__ __forceinline__ void update_pair_b(const StrandC<typename Lay<WIDE>::pos_t>& sa, uint32_t ca, IvT<typename Lay<WIDE>::pos_t> a,
                                                       const StrandC<typename Lay<WIDE>::pos_t>& sb, uint32_t cb, IvT<typename Lay<WIDE>::pos_t> b,
                                                       const uint32_t* __restrict__ mtab, IvT<typename Lay<WIDE>::pos_t>& oa,
                                                       IvT<typename Lay<WIDE>::pos_t>& ob, uint32_t& n_blk_a, uint32_t& n_blk_b)
{
    using L = Lay<WIDE>;
    using P = typename L::pos_t;
    const P pla = a.lo, pua = a.hi + 1, plb = b.lo, pub = b.hi + 1;
    const P bla = pla / L::kSyms, bua = pua / L::kSyms, blb = plb / L::kSyms, bub = pub / L::kSyms;
    const uint32_t ola = (uint32_t)(pla - bla * L::kSyms), oua = (uint32_t)(pua - bua * L::kSyms);
    const uint32_t olb = (uint32_t)(plb - blb * L::kSyms), oub = (uint32_t)(pub - bub * L::kSyms);
    typename L::Regs raa, rba, rab, rbb;
    L::load(sa.blocks, bla, raa);
    L::load(sa.blocks, bua, rba);
    L::load(sb.blocks, blb, rab);
    L::load(sb.blocks, bub, rbb);
    const uint64_t base_aa = block_base<WIDE>(sa.blocks, bla, ca), base_ba = block_base<WIDE>(sa.blocks, bua, ca);
    const uint64_t base_ab = block_base<WIDE>(sb.blocks, blb, cb), base_bb = block_base<WIDE>(sb.blocks, bub, cb);
    uint64_t caa = base_aa + block_popc<WIDE>(raa, ca, mtab + ola * L::kRow);
    uint64_t cba = base_ba + block_popc<WIDE>(rba, ca, mtab + oua * L::kRow);
    uint64_t cab = base_ab + block_popc<WIDE>(rab, cb, mtab + olb * L::kRow);
    uint64_t cbb = base_bb + block_popc<WIDE>(rbb, cb, mtab + oub * L::kRow);
    if(ca == 0) {
        if(ola != 0 && L::flagged(raa)) caa -= dollars_in_c(sa, (uint64_t)bla * L::kSyms, (uint64_t)bla * L::kSyms + ola);
        if(oua != 0 && L::flagged(rba)) cba -= dollars_in_c(sa, (uint64_t)bua * L::kSyms, (uint64_t)bua * L::kSyms + oua);
    }
    if(cb == 0) {
        if(olb != 0 && L::flagged(rab)) cab -= dollars_in_c(sb, (uint64_t)blb * L::kSyms, (uint64_t)blb * L::kSyms + olb);
        if(oub != 0 && L::flagged(rbb)) cbb -= dollars_in_c(sb, (uint64_t)bub * L::kSyms, (uint64_t)bub * L::kSyms + oub);
    }
    const P pa = pred_of(sa, ca), pb = pred_of(sb, cb);
    oa.lo = pa + (P)caa; oa.hi = pa + (P)cba - 1;
    ob.lo = pb + (P)cab; ob.hi = pb + (P)cbb - 1;
    n_blk_a = (bla == bua) ? 1u : 2u;
    n_blk_b = (blb == bub) ? 1u : 2u;
}

// BWTAlgorithms::initInterval (BWTAlgorithms.h:136-140): Occ(b, N-1) is the symbol total.
template <class P>
__host__ __device__ __forceinline__ IvT<P> init_interval(const StrandC<P>& s, uint32_t code)
{
    IvT<P> iv;
    iv.lo = pred_of(s, code);
    iv.hi = pred_next(s, code) - 1;
    return iv;
}

__device__ __forceinline__ void flush_counters(DevCounters* ctr, uint32_t n_rank, uint32_t n_blk, uint32_t n_tab = 0)
{
    if(ctr == nullptr) return;
    unsigned long long a = n_rank, b = n_blk, c = n_tab;
#pragma unroll
    for(int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
        c += __shfl_down(c, o, 64);
    }
    if((threadIdx.x & 63) == 0) {
        DevCounters* shard = ctr + (blockIdx.x & (kCtrShards - 1));
        if(a) atomicAdd(&shard->rank_queries, a);
        if(b) atomicAdd(&shard->block_loads, b);
        if(c) atomicAdd(&shard->table_loads, c);
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
__host__ __device__ __forceinline__ WalkState<P> walk_init()
{
    WalkState<P> st;
    st.size = 0; st.counted = 0; st.n_rank = 0; st.n_blk = 0; st.fwd_broken = false; st.rvc_broken = false;
    st.fwd.lo = st.fwd.hi = st.rvc.lo = st.rvc.hi = 0;
    return st;
}

template <bool WIDE>
__host__ __device__ __forceinline__ WalkState<typename Lay<WIDE>::pos_t>
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

// ---------------------------------------------------------------------------------------
// k-mer table start: the state of a findInterval-semantics walk (walk_step with base_k = inf) after its
// first k characters, from one 16-byte load.  get(t) returns the t-th character code of the sequence.
// Returns 0 (and leaves st untouched) when no table of size <= max_k exists.
// ---------------------------------------------------------------------------------------
template <bool WIDE, class Get>
__host__ __device__ __forceinline__ uint32_t table_start(const FmIndexDev& fm, Get get, uint32_t max_k,
                                                WalkState<typename Lay<WIDE>::pos_t>& st)
{
    using P = typename Lay<WIDE>::pos_t;
    int best = -1;
    if(fm.ktab[0].k != 0 && fm.ktab[0].k <= max_k) best = 0;
    if(fm.ktab[1].k != 0 && fm.ktab[1].k <= max_k) best = 1;
    if(fm.ktab[2].k != 0 && fm.ktab[2].k <= max_k) best = 2;
    if(fm.ktab[3].k != 0 && fm.ktab[3].k <= max_k) best = 3;
    if(fm.ktab[4].k != 0 && fm.ktab[4].k <= max_k) best = 4;
    if(best < 0) return 0;
    const uint32_t k = best == 0 ? fm.ktab[0].k : best == 1 ? fm.ktab[1].k : best == 2 ? fm.ktab[2].k : best == 3 ? fm.ktab[3].k : fm.ktab[4].k;
    const void* tabv = best == 0 ? fm.ktab[0].entries : best == 1 ? fm.ktab[1].entries
                     : best == 2 ? fm.ktab[2].entries : best == 3 ? fm.ktab[3].entries : fm.ktab[4].entries;
    uint32_t code = 0;
    for(uint32_t t = 0; t < k; ++t) code = (code << 2) | get(t);
    // narrow indexes: 4 x u32 per entry; wide (Block64) indexes: 4 x u64
    struct { P x, y, z, w; } e;
    if(WIDE) {
        const uint4* tab = reinterpret_cast<const uint4*>(tabv) + (uint64_t)code * 2;
        const uint4 a = tab[0], b = tab[1];
        e.x = (P)(((uint64_t)a.y << 32) | a.x); e.y = (P)(((uint64_t)a.w << 32) | a.z);
        e.z = (P)(((uint64_t)b.y << 32) | b.x); e.w = (P)(((uint64_t)b.w << 32) | b.z);
    } else {
        const uint4 a = reinterpret_cast<const uint4*>(tabv)[code];
        e.x = (P)a.x; e.y = (P)a.y; e.z = (P)a.z; e.w = (P)a.w;
    }
    st.fwd.lo = e.x; st.fwd.hi = e.y; st.rvc.lo = e.z; st.rvc.hi = e.w;
    st.fwd_broken = e.x > e.y;
    st.rvc_broken = e.z > e.w;
    st.size = k;
    st.counted = k;          // only meaningful when !fwd_broken (callers that need the exact count fall back)
    return k;
}

template <class P> __host__ __device__ __forceinline__ int64_t iv_freq(const IvT<P>& iv) { return iv.lo <= iv.hi ? (int64_t)(iv.hi - iv.lo) + 1 : 0; }
template <class P> __device__ __forceinline__ lrsc_biinterval to_out(const IvT<P>& f, const IvT<P>& r)
{
    lrsc_biinterval o;
    o.fwd.lower = (int64_t)f.lo; o.fwd.upper = (int64_t)f.hi;
    o.rvc.lower = (int64_t)r.lo; o.rvc.upper = (int64_t)r.hi;
    return o;
}


} // namespace lrsc
