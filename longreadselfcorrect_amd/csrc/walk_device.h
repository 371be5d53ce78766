// walk_device.h -- the device-side LongReadSelfCorrectByOverlap state machine (Leaf, Walk) shared by the
// per-walk kernel (extend.hip) and the walk-parallel correction flow (wp.hip).
// Reference: PacBio/LongReadCorrectByOverlap.cpp:17-878, FMIndexWalk/SAINode.cpp:166-189.
#pragma once
#include <hip/hip_runtime.h>

#include "extend.h"
#include "introsort_emul.h"
#include "rank_device.h"

// The big sequential pieces of the walk are calls, not inline copies (code size; register budget per piece).  An experiment build can
// define this empty (-DLRSC_WALK_NOINLINE=) to let a kernel's launch bounds govern the whole call tree.
#ifndef LRSC_WALK_NOINLINE
#define LRSC_WALK_NOINLINE __noinline__
#endif
// Device code in the product.  tests/host_walk compiles this header with LRSC_WALK_FN = __host__ __device__ to run the very same
// walk on the CPU against the oracle (test infrastructure: nothing in the product calls the host versions).
#ifndef LRSC_WALK_FN
#define LRSC_WALK_FN __device__
#endif

namespace lrsc {

constexpr uint64_t kNoKey = ~0ull;

// The bulk part of the constructor for ONE offset i of m_query (LongReadCorrectByOverlap.cpp:82-94,127-152):
// bi-intervals of the 5-mer, the idmer (9-mer) and, inside the target seed, the minOverlap-mer starting at i,
// each with findInterval's early exit; k-mer tables short-cut whole searches when a table of that size exists.
template <bool WIDE>
LRSC_WALK_FN __forceinline__ void prepare_offset(const FmIndexDev& fm, const StrandC<typename Lay<WIDE>::pos_t>& sf,
                                               const StrandC<typename Lay<WIDE>::pos_t>& sr, const uint32_t* __restrict__ mtab,
                                               const uint8_t* __restrict__ q, uint32_t i, uint32_t Lq, uint32_t trg0,
                                               uint32_t seedk, uint32_t mink, SortItem* it9f, SortItem* it9r, uint8_t* flags5,
                                               typename Lay<WIDE>::pos_t* term, uint32_t& n_rank, uint32_t& n_blk)
{
    using P = typename Lay<WIDE>::pos_t;
    const bool want_term = i >= trg0 && i + mink <= Lq;
    uint32_t kmax = 0;
    if(i + 5 <= Lq) kmax = 5;
    if(i + seedk <= Lq) kmax = seedk;
    if(want_term) kmax = mink > kmax ? mink : kmax;
    WalkState<P> st = walk_init<P>();
    for(uint32_t s = 0; s < kmax;) {
        const uint32_t next_emit = s < 5 ? 5u : s < seedk ? seedk : mink;
        bool jumped = false;
        if(next_emit <= kmax) {
            WalkState<P> ts = walk_init<P>();
            const uint32_t tk = table_start<WIDE>(fm, [&](uint32_t t) { return (uint32_t)q[i + t]; }, next_emit, ts);
            if(tk == next_emit) { n_rank += st.n_rank; n_blk += st.n_blk; st = ts; st.n_rank = 0; st.n_blk = 0; s = tk; jumped = true; }
        }
        if(!jumped) { st = walk_step<WIDE>(sf, sr, q[i + s], 1u << 30, st, mtab); ++s; }
        if(st.size == 5) flags5[i] = (uint8_t)((st.fwd.lo <= st.fwd.hi ? 1 : 0) | (st.rvc.lo <= st.rvc.hi ? 2 : 0));
        if(st.size == seedk) {
            // pad carries the idmer's 2-bit code through the sort: the hit chains compare it instead of re-reading m_query
            uint32_t code = 0;
            for(uint32_t t = 0; t < seedk; ++t) code = (code << 2) | q[i + t];
            it9f[i].key = st.fwd.lo <= st.fwd.hi ? (uint64_t)st.fwd.lo : kNoKey; it9f[i].val = i; it9f[i].pad = code;
            it9r[i].key = st.rvc.lo <= st.rvc.hi ? (uint64_t)st.rvc.lo : kNoKey; it9r[i].val = i; it9r[i].pad = code;
        }
        if(st.size == mink && want_term) {
            P* t = term + (uint64_t)(i - trg0) * 4;
            t[0] = st.fwd.lo; t[1] = st.fwd.hi; t[2] = st.rvc.lo; t[3] = st.rvc.hi;
        }
    }
    n_rank += st.n_rank; n_blk += st.n_blk;
}

// The same for every offset of m_query at once, when k-mer tables of exactly the three emitted sizes exist (the normal
// configuration: 5, idmer = 9, minOverlap = 13): every emit is one table entry, so a rolling 2-bit window supplies the three
// table indexes from ONE character load per offset (the generic path re-reads 5 + 9 + 13 characters per offset).
template <bool WIDE>
LRSC_WALK_FN __forceinline__ int ktab_of(const FmIndexDev& fm, uint32_t k)
{
    int t = -1;
    if(fm.ktab[0].k == k) t = 0;
    if(fm.ktab[1].k == k) t = 1;
    if(fm.ktab[2].k == k) t = 2;
    if(fm.ktab[3].k == k) t = 3;
    if(fm.ktab[4].k == k) t = 4;
    return t;
}
template <bool WIDE>
LRSC_WALK_FN __forceinline__ void ktab_entry(const FmIndexDev& fm, int t, uint32_t code, typename Lay<WIDE>::pos_t e[4])
{
    using P = typename Lay<WIDE>::pos_t;
    const void* tabv = t == 0 ? fm.ktab[0].entries : t == 1 ? fm.ktab[1].entries : t == 2 ? fm.ktab[2].entries
                     : t == 3 ? fm.ktab[3].entries : fm.ktab[4].entries;
    if(WIDE) {
        const uint4* tp = reinterpret_cast<const uint4*>(tabv) + (uint64_t)code * 2;
        const uint4 a = tp[0], b = tp[1];
        e[0] = (P)(((uint64_t)a.y << 32) | a.x); e[1] = (P)(((uint64_t)a.w << 32) | a.z);
        e[2] = (P)(((uint64_t)b.y << 32) | b.x); e[3] = (P)(((uint64_t)b.w << 32) | b.z);
    } else {
        const uint4 a = reinterpret_cast<const uint4*>(tabv)[code];
        e[0] = (P)a.x; e[1] = (P)a.y; e[2] = (P)a.z; e[3] = (P)a.w;
    }
}
template <bool WIDE>
LRSC_WALK_FN LRSC_WALK_NOINLINE bool prepare_all_from_tables(const FmIndexDev& fm, const uint8_t* __restrict__ q, uint32_t Lq, uint32_t trg0, uint32_t seedk,
                                                     uint32_t mink, SortItem* it9f, SortItem* it9r, uint8_t* flags5, typename Lay<WIDE>::pos_t* term)
{
    using P = typename Lay<WIDE>::pos_t;
    if(seedk <= 5 || mink <= seedk || mink > 16) return false;
    const int t5 = ktab_of<WIDE>(fm, 5), t9 = ktab_of<WIDE>(fm, seedk), tm = ktab_of<WIDE>(fm, mink);
    if(t5 < 0 || t9 < 0 || tm < 0) return false;
    const uint32_t m5 = (1u << 10) - 1u, m9 = (1u << (2 * seedk)) - 1u, mm = mink >= 16 ? 0xFFFFFFFFu : (1u << (2 * mink)) - 1u;
    // window = characters [i, i + mink) of m_query, newest in the low bits (zeros past the end)
    uint32_t win = 0;
    for(uint32_t t = 0; t + 1 < mink; ++t) win = (win << 2) | (t < Lq ? (uint32_t)q[t] : 0u);
    for(uint32_t i = 0; i < Lq; ++i) {
        win = ((win << 2) | (i + mink - 1 < Lq ? (uint32_t)q[i + mink - 1] : 0u)) & mm;
        P e[4];
        if(i + 5 <= Lq) {
            ktab_entry<WIDE>(fm, t5, (win >> (2 * (mink - 5))) & m5, e);
            flags5[i] = (uint8_t)((e[0] <= e[1] ? 1 : 0) | (e[2] <= e[3] ? 2 : 0));
        }
        if(i + seedk <= Lq) {
            const uint32_t code = (win >> (2 * (mink - seedk))) & m9;
            ktab_entry<WIDE>(fm, t9, code, e);
            it9f[i].key = e[0] <= e[1] ? (uint64_t)e[0] : kNoKey; it9f[i].val = i; it9f[i].pad = code;
            it9r[i].key = e[2] <= e[3] ? (uint64_t)e[2] : kNoKey; it9r[i].val = i; it9r[i].pad = code;
        }
        if(i >= trg0 && i + mink <= Lq) {
            ktab_entry<WIDE>(fm, tm, win, e);
            P* t = term + (uint64_t)(i - trg0) * 4;
            t[0] = e[0]; t[1] = e[1]; t[2] = e[2]; t[3] = e[3];
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// 2. the walk
// ---------------------------------------------------------------------------------------
template <class P>
struct alignas(16) Leaf {             // SAIOverlapNode3 + leafInfo, flattened; 16-byte aligned: leaf copies are dwordx4 moves
    P flo, fhi, rlo, rhi;             // fwdInterval (rbwt), rvcInterval (bwt)
    P tflo, tfhi, trlo, trhi;         // SelectFreqsOfrange's per-leaf scratch intervals (maxKmerArray)
    uint64_t suf_lo, suf_hi;          // last 64 characters of the path, 2 bits each, newest in the low bits
    uint64_t lastSeedIdx, lastOverlapLen, totalSeeds, currOverlapLen, numOfErrors, queryOverlapLen;
    double numRedeemSeed, localErr, globalErr;
    uint32_t hist_size;               // GlobalErrorRateRecord.size()
    int32_t lastSeedIdxOffset;
    int32_t res_first, res_second;    // resultindex
    int32_t kmerFrequency;            // leafInfo::kmerFrequency
    int32_t tmpFreq;                  // SelectFreqsOfrange: FMidx::kmerFrequency
    uint32_t tailLetter, tailLetterCount;
    uint32_t path_len;
    uint16_t ring, path;              // slot ids (materialised leaves only)
    uint16_t parent;                  // children: index of the parent in cur[]
    uint8_t ext, alive;               // children: extension code; survival flag
};

template <class P> __host__ __device__ __forceinline__ int64_t isize(P lo, P hi) { return (int64_t)hi - (int64_t)lo + 1; }

// character `t` (0 = oldest) of the suffix of length l of a leaf's path
template <class P> __host__ __device__ __forceinline__ uint32_t suf_char(const Leaf<P>& lf, uint32_t l, uint32_t t)
{
    const uint32_t back = l - 1 - t;                     // distance from the newest character
    return back < 32 ? (uint32_t)(lf.suf_lo >> (2 * back)) & 3u : (uint32_t)(lf.suf_hi >> (2 * (back - 32))) & 3u;
}
template <class P> __host__ __device__ __forceinline__ void suf_push(Leaf<P>& lf, uint32_t c)
{
    lf.suf_hi = (lf.suf_hi << 2) | (lf.suf_lo >> 62);
    lf.suf_lo = (lf.suf_lo << 2) | c;
}

__host__ __device__ __forceinline__ uint32_t path_get(const uint32_t* p, uint32_t i) { return (p[i >> 4] >> (2 * (i & 15))) & 3u; }
__host__ __device__ __forceinline__ void path_set(uint32_t* p, uint32_t i, uint32_t c)
{
    const uint32_t sh = 2 * (i & 15);
    p[i >> 4] = (p[i >> 4] & ~(3u << sh)) | (c << sh);
}

template <bool WIDE>
struct Walk {
    using P = typename Lay<WIDE>::pos_t;
    // index
    StrandC<P> sF, sR;
    const FmIndexDev* fm;
    const uint32_t* mtab;
    // inputs
    const uint8_t* q;                 // m_query codes
    uint32_t Lq, initk, path_len, trg_len;
    int32_t dis;
    // parameters
    uint32_t seedSize, minOverlap, maxOverlap, maxLeaves;
    uint64_t min_SA_threshold;
    uint64_t PBcoverage;
    double PacBioErrorRate, errorRate;
    uint64_t localK;                  // m_localSimilarlykmerSize (100)
    const double* freqsOfKmerSize;    // [101]
    // derived
    uint64_t maxIndelSize, maxLength, minLength, currentLength, currentKmerSize;
    // workspace
    SortItem *it9f, *it9r;
    uint32_t n9f, n9r;
    uint16_t *next9f, *next9r, *head9f, *head9r;      // chains in sorted order, 256 hash buckets
    uint16_t *next5, *head5;                          // 5-mer chains by code (1024 heads), flags5 tells strand validity
    const uint8_t* flags5;
    const P* term;
    uint32_t n_term;
    uint64_t tmask0, tmask1;          // 128-bit filter over the target seed's minOverlap-mers (hash of their last <= 16 characters)
    Leaf<P>* cur;  uint32_t n_cur;
    Leaf<P>* leaf_small;              // the 32-slot leaf buffer (the other one holds kMaxChildren); cur / nxt swap between them
    Leaf<P>* nxt;  uint32_t n_nxt;
    double* rings;                    // [32][100]
    uint32_t* paths;  uint32_t pathw; // [32][pathw]
    uint32_t* rpaths;                 // [kMaxResults][pathw]
    WalkResultRec* results; uint32_t n_results;
    uint32_t ring_free, path_free;    // bit s: ring / path slot s is free (slot 0 belongs to the root's lineage)
    uint32_t n_rank, n_blk;
    uint64_t alive_lo, alive_hi;      // PrunedBySeedSupport: bit c = child c survived (kMaxChildren = 128)
    uint32_t has_child;               // ... bit i = leaf i of cur[] has a surviving child
    uint32_t n_highfreq;              // children of the last attempToExtend whose k-mer frequency is above isInsufficientFreqs' threshold
    uint64_t steps;
    uint32_t leaf_steps;              // frontier leaves summed over the steps (profiling)
    uint32_t max_front;               // widest frontier of the walk (profiling)
    uint64_t cyc_setup, cyc_loop;     // profiling: ticks spent building the trees/root and in the extension loop
    uint64_t* prof;                   // profiling: ReadOut::cyc_step of the read, or nullptr
    LRSC_WALK_FN __forceinline__ uint64_t tick() const { return prof ? __builtin_readcyclecounter() : 0; }
    LRSC_WALK_FN __forceinline__ void tock(int k, uint64_t t0) { if(prof) prof[k] += __builtin_readcyclecounter() - t0; }
    int error;

    LRSC_WALK_FN __forceinline__ IvT<P> upd(const StrandC<P>& s, uint32_t c, IvT<P> iv) { n_rank += 2; return update_interval<WIDE>(s, c, iv, mtab, n_blk); }

    // findInterval of the leaf's suffix of length l on both strands (initialRootNode / refineSAInterval):
    // fwd = reverse(kmer) in the rbwt, rvc = revcomp(kmer) in the bwt; both consume kmer[0], kmer[1], ... in order
    LRSC_WALK_FN __forceinline__ void find_suffix_v(uint64_t slo, uint64_t shi, uint32_t l, P& flo, P& fhi, P& rlo, P& rhi)
    {
        auto ch = [&](uint32_t t) -> uint32_t {
            const uint32_t back = l - 1 - t;
            return back < 32 ? (uint32_t)(slo >> (2 * back)) & 3u : (uint32_t)(shi >> (2 * (back - 32))) & 3u;
        };
        WalkState<P> st = walk_init<P>();
        const uint32_t t0 = table_start<WIDE>(*fm, ch, l, st);
        for(uint32_t t = t0; t < l; ++t) {
            if(st.fwd_broken && st.rvc_broken) break;
            st = walk_step<WIDE>(sF, sR, ch(t), 1u << 30, st, mtab);
        }
        n_rank += st.n_rank; n_blk += st.n_blk;
        flo = st.fwd.lo; fhi = st.fwd.hi; rlo = st.rvc.lo; rhi = st.rvc.hi;
    }
    LRSC_WALK_FN LRSC_WALK_NOINLINE void find_suffix(Leaf<P>& lf, uint32_t l)
    {
        P a, b, c, d;
        find_suffix_v(lf.suf_lo, lf.suf_hi, l, a, b, c, d);
        lf.flo = a; lf.fhi = b; lf.rlo = c; lf.rhi = d;
    }

    // refineSAInterval (.cpp:355-369).  The leaves' searches are independent: four run side by side, so that a dependent rank step
    // of one leaf waits together with those of three others (a lane walks its frontier leaf by leaf otherwise)
    LRSC_WALK_FN LRSC_WALK_NOINLINE void refineSAInterval(Leaf<P>* leaves, uint32_t n, uint64_t newKmerSize)
    {
        const uint32_t l = (uint32_t)newKmerSize;
        for(uint32_t i0 = 0; i0 < n; i0 += 4) {
            const uint32_t g = n - i0 < 4 ? n - i0 : 4;
            uint64_t slo[4], shi[4];
            WalkState<P> st[4];
            uint32_t t0[4];
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) {
                const uint32_t i = i0 + (j < g ? j : 0);
                slo[j] = leaves[i].suf_lo; shi[j] = leaves[i].suf_hi;
            }
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) {
                st[j] = walk_init<P>();
                const uint64_t a = slo[j], b = shi[j];
                t0[j] = table_start<WIDE>(*fm, [&](uint32_t t) -> uint32_t {
                    const uint32_t back = l - 1 - t;
                    return back < 32 ? (uint32_t)(a >> (2 * back)) & 3u : (uint32_t)(b >> (2 * (back - 32))) & 3u; }, l, st[j]);
            }
            // every chain starts from the same table size (or from nothing): one shared loop counter
            for(uint32_t t = t0[0]; t < l; ++t) {
                bool any = false;
#pragma unroll
                for(uint32_t j = 0; j < 4; ++j) {
                    if(j < g && !(st[j].fwd_broken && st[j].rvc_broken)) {
                        const uint32_t back = l - 1 - t;
                        const uint32_t c = back < 32 ? (uint32_t)(slo[j] >> (2 * back)) & 3u : (uint32_t)(shi[j] >> (2 * (back - 32))) & 3u;
                        st[j] = walk_step<WIDE>(sF, sR, c, 1u << 30, st[j], mtab);
                        any = true;
                    }
                }
                if(!any) break;
            }
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) {
                if(j < g) {
                    Leaf<P>& lf = leaves[i0 + j];
                    n_rank += st[j].n_rank; n_blk += st[j].n_blk;
                    lf.flo = st[j].fwd.lo; lf.fhi = st[j].fwd.hi; lf.rlo = st[j].rvc.lo; lf.rhi = st[j].rvc.hi;
                }
            }
        }
        currentKmerSize = newKmerSize;
    }

    // ---- SelectFreqsOfrange (.cpp:281-331) ---------------------------------------------------------
    // startkmer = last Lw chars of the suffix of length U; Fwd = findInterval(BWT, startkmer) walks it from
    // its last character backwards; Rvc = findInterval(RBWT, complement(startkmer)) likewise.
    // The pair is findBiInterval of x = revcomp(startkmer) with the strands' roles swapped (f walks the bwt with c, r the
    // rbwt with 3 - c, both from the k-mer's last character backwards), so a k-mer table entry of x's first characters
    // -- same early-exit semantics per strand -- replaces that many dependent rank steps.
    LRSC_WALK_FN __forceinline__ void select_first(uint64_t slo, uint64_t shi, uint32_t U, uint32_t Lw, IvT<P>& f, IvT<P>& r)
    {
        auto sc = [&](uint32_t t) -> uint32_t {                 // suf_char(lf, U, t)
            const uint32_t back = U - 1 - t;
            return back < 32 ? (uint32_t)(slo >> (2 * back)) & 3u : (uint32_t)(shi >> (2 * (back - 32))) & 3u;
        };
        bool fb = false, rb = false;
        uint32_t t1 = 1;
        {
            WalkState<P> ts = walk_init<P>();
            const uint32_t tk = table_start<WIDE>(*fm, [&](uint32_t t) { return 3u - sc(U - 1 - t); }, Lw, ts);
            if(tk != 0) {
                f = ts.rvc; r = ts.fwd; fb = ts.rvc_broken; rb = ts.fwd_broken; t1 = tk;
            } else {
                f = init_interval<P>(sR, sc(U - 1));
                r = init_interval<P>(sF, 3u - sc(U - 1));
                n_rank += 2;
            }
        }
        for(uint32_t t = t1; t < Lw; ++t) {
            if(fb && rb) break;
            const uint32_t c = sc(U - 1 - t);
            if(!fb) { f = upd(sR, c, f); fb = f.lo > f.hi; }
            if(!rb) { r = upd(sF, 3u - c, r); rb = r.lo > r.hi; }
            if(fb && rb) break;
        }
    }
    LRSC_WALK_FN __forceinline__ void select_next(uint64_t slo, uint64_t shi, uint32_t U, uint32_t t, IvT<P>& f, IvT<P>& r)
    {
        const uint32_t back = U - 1 - t;
        const uint32_t b = back < 32 ? (uint32_t)(slo >> (2 * back)) & 3u : (uint32_t)(shi >> (2 * (back - 32))) & 3u;
        f = upd(sR, b, f);                           // no validity check here (.cpp:317-318)
        r = upd(sF, 3u - b, r);
    }
    LRSC_WALK_FN LRSC_WALK_NOINLINE uint64_t SelectFreqsOfrange(uint64_t LowerBound, uint64_t UpperBound, Leaf<P>* leaves, uint32_t n)
    {
        int tempmaxfmfreqs = 0;
        const uint32_t U = (uint32_t)UpperBound, Lw = (uint32_t)LowerBound;
        for(uint32_t j = 0; j < n; ++j) {
            Leaf<P>& lf = leaves[j];
            IvT<P> f, r;
            select_first(lf.suf_lo, lf.suf_hi, U, Lw, f, r);
            lf.tflo = f.lo; lf.tfhi = f.hi; lf.trlo = r.lo; lf.trhi = r.hi;
            lf.tmpFreq = (int)(isize(f.lo, f.hi) + isize(r.lo, r.hi));
            if(lf.tmpFreq > tempmaxfmfreqs) tempmaxfmfreqs = lf.tmpFreq;
        }
        if(tempmaxfmfreqs - (int)freqsOfKmerSize[LowerBound] < 5) return LowerBound;

        for(uint64_t i = 1; i <= UpperBound - LowerBound; i++) {
            tempmaxfmfreqs = 0;
            for(uint32_t j = 0; j < n; ++j) {
                Leaf<P>& lf = leaves[j];
                IvT<P> f{lf.tflo, lf.tfhi}, r{lf.trlo, lf.trhi};
                select_next(lf.suf_lo, lf.suf_hi, U, (uint32_t)(UpperBound - LowerBound - i), f, r);
                lf.tflo = f.lo; lf.tfhi = f.hi; lf.trlo = r.lo; lf.trhi = r.hi;
                lf.tmpFreq = (int)(isize(f.lo, f.hi) + isize(r.lo, r.hi));
                if(lf.tmpFreq > tempmaxfmfreqs) tempmaxfmfreqs = lf.tmpFreq;
            }
            if(tempmaxfmfreqs - (int)freqsOfKmerSize[LowerBound + i] < 5) return LowerBound + i;
        }
        return UpperBound;
    }
    // the same for a frontier of one leaf, on values (the leaf's tf* scratch fields are write-only outside this function)
    LRSC_WALK_FN __forceinline__ uint64_t SelectFreqsOfrange1(uint64_t LowerBound, uint64_t UpperBound, uint64_t slo, uint64_t shi)
    {
        const uint32_t U = (uint32_t)UpperBound, Lw = (uint32_t)LowerBound;
        IvT<P> f, r;
        select_first(slo, shi, U, Lw, f, r);
        int freq = (int)(isize(f.lo, f.hi) + isize(r.lo, r.hi));
        int tempmaxfmfreqs = freq > 0 ? freq : 0;
        if(tempmaxfmfreqs - (int)freqsOfKmerSize[LowerBound] < 5) return LowerBound;
        for(uint64_t i = 1; i <= UpperBound - LowerBound; i++) {
            select_next(slo, shi, U, (uint32_t)(UpperBound - LowerBound - i), f, r);
            freq = (int)(isize(f.lo, f.hi) + isize(r.lo, r.hi));
            tempmaxfmfreqs = freq > 0 ? freq : 0;
            if(tempmaxfmfreqs - (int)freqsOfKmerSize[LowerBound + i] < 5) return LowerBound + i;
        }
        return UpperBound;
    }

    // isInsufficientFreqs (.cpp:334-352) over the children attempToExtend has just made: it counted those above the threshold
    LRSC_WALK_FN bool isInsufficientFreqs(uint64_t highfreqscount, uint32_t n)
    {
        if(highfreqscount == 0) return true;
        else if(highfreqscount <= 2 && n >= 5) return true;
        else if(highfreqscount <= 1 && n >= 3) return true;
        return false;
    }

    // ---- ismatchedbykmer (.cpp:787-821): any 5-mer hit of the extended path within the indel window --------
    LRSC_WALK_FN bool ismatchedbykmer(uint32_t code5, bool fvalid, bool rvalid)
    {
        const uint64_t startSeedIdx = (uint64_t)(((int)currentLength - (int)maxIndelSize) > 0 ? ((int)currentLength - (int)maxIndelSize) : 0);
        const uint64_t largeSeedIdx = currentLength + maxIndelSize;
        for(uint32_t j = head5[code5]; j != 0xFFFFu; j = next5[j]) {
            if(j >= startSeedIdx && j <= largeSeedIdx) {
                const uint32_t fl = flags5[j];
                if((fvalid && (fl & 1)) || (rvalid && (fl & 2))) return true;
            }
        }
        return false;
    }

    // ---- getFMIndexExtensions (.cpp:667-784): returns a bit mask of accepted bases, fills ext[] -------------
    struct Ext { IvT<P> f, r; int freq; };
    LRSC_WALK_FN __forceinline__ uint32_t getFMIndexExtensions(const Leaf<P>& lf, Ext ext[4], uint64_t& totalcount_out)
    {
        return getFMIndexExtensions_v(lf.flo, lf.fhi, lf.rlo, lf.rhi, lf.tailLetterCount, lf.suf_lo, ext, totalcount_out);
    }
    LRSC_WALK_FN __forceinline__ uint32_t getFMIndexExtensions_v(P lf_flo, P lf_fhi, P lf_rlo, P lf_rhi, uint32_t lf_tailLetterCount, uint64_t lf_suf_lo,
                                                               Ext ext[4], uint64_t& totalcount_out)
    {
        struct { P flo, fhi, rlo, rhi; uint32_t tailLetterCount; uint64_t suf_lo; } lf{lf_flo, lf_fhi, lf_rlo, lf_rhi, lf_tailLetterCount, lf_suf_lo};
        const uint64_t IntervalSizeCutoff = min_SA_threshold;
        uint64_t totalcount = 0;
        int maxfreqsofleave = 0;
        const bool fv = lf.flo <= lf.fhi, rv = lf.rlo <= lf.rhi;
        // all four bases of a strand from one pair of block loads (the reference issues eight updateIntervals, .cpp:685-700)
        IvT<P> fo[4], ro[4];
        if(fv) { update_interval_all<WIDE, true>(sF, IvT<P>{lf.flo, lf.fhi}, mtab, fo, n_blk); n_rank += 8; }
        if(rv) { update_interval_all<WIDE, true>(sR, IvT<P>{lf.rlo, lf.rhi}, mtab, ro, n_blk); n_rank += 8; }
#pragma unroll
        for(uint32_t b = 0; b < 4; ++b) {
            IvT<P> fp{lf.flo, lf.fhi}, rp{lf.rlo, lf.rhi};
            if(fv) fp = fo[b];
            if(rv) rp = ro[3u - b];
            ext[b].f = fp; ext[b].r = rp;
            ext[b].freq = (int)(isize(fp.lo, fp.hi) + isize(rp.lo, rp.hi));
            totalcount += (uint64_t)(int64_t)ext[b].freq;
            if(ext[b].freq > maxfreqsofleave) maxfreqsofleave = ext[b].freq;
        }
        totalcount_out = totalcount;
        uint32_t mask = 0;
        const bool isHomopolymer = lf.tailLetterCount >= 3;
        for(uint32_t b = 0; b < 4; ++b) {
            const uint64_t kmerFreq = (uint64_t)(int64_t)ext[b].freq;
            const double kmerRatioNotPass = 2;
            double kmerRatioCutoff = 0;
            const double kmerRatio = (double)kmerFreq / (double)maxfreqsofleave;
            const bool efv = ext[b].f.lo <= ext[b].f.hi, erv = ext[b].r.lo <= ext[b].r.hi;
            const uint32_t code5 = (uint32_t)(((lf.suf_lo << 2) | b) & 0x3FFu);
            const bool isFreqPass = kmerFreq >= IntervalSizeCutoff;
            const bool isLowCoverage = totalcount >= IntervalSizeCutoff + 2;
            const bool isRepeat = maxfreqsofleave > 100;
            const bool isHighlyRepeat = maxfreqsofleave > 150;
            const bool isLowlyRepeat = maxfreqsofleave > 50;
            // the 5-mer test (a walk over a chain in global memory) only enters the ladder for repeats: evaluated lazily, it has no
            // side effects (.cpp:716-733 computes it for every base)
            const bool isMatchedBy5mer = isLowlyRepeat && ismatchedbykmer(code5, efv, erv);
            if(isMatchedBy5mer && isHighlyRepeat) kmerRatioCutoff = 0.125;
            else if(isMatchedBy5mer && isLowlyRepeat) kmerRatioCutoff = 0.2;
            else if(isFreqPass) kmerRatioCutoff = 0.25;
            else if(isLowCoverage) kmerRatioCutoff = 0.6;
            else kmerRatioCutoff = kmerRatioNotPass;
            if(isHomopolymer && isRepeat) kmerRatioCutoff = kmerRatioCutoff > 0.3 ? kmerRatioCutoff : 0.3;
            else if(isHomopolymer) kmerRatioCutoff = kmerRatioCutoff > 0.6 ? kmerRatioCutoff : 0.6;
            if(kmerRatio >= kmerRatioCutoff) mask |= 1u << b;
        }
        return mask;
    }

    LRSC_WALK_FN void free_leaf_slots(const Leaf<P>& lf)
    {
        ring_free |= 1u << lf.ring;
        path_free |= 1u << lf.path;
    }

    // ---- attempToExtend (.cpp:373-465) + updateLeaves (:468-488) -------------------------------------------
    LRSC_WALK_FN void attempToExtend()
    {
        double minimumErrorRate = 1;
        for(uint32_t i = 0; i < n_cur; i += 4) {                 // four loads in flight, same comparisons in the same order
            double e[4];
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) e[j] = cur[i + j < n_cur ? i + j : i].localErr;
#pragma unroll
            for(uint32_t j = 0; j < 4; ++j) if(i + j < n_cur && e[j] < minimumErrorRate) minimumErrorRate = e[j];
        }
        // trim leaves whose error rate relative to the best one is high
        uint32_t w = 0;
        for(uint32_t i = 0; i < n_cur; ++i) {
            const double errorRateDiff = cur[i].localErr - minimumErrorRate;
            if((errorRateDiff > 0.05 && currentLength > localK / 2) || (errorRateDiff > 0.1 && currentLength > 15)) {
                free_leaf_slots(cur[i]);
                continue;
            }
            if(w != i) cur[w] = cur[i];
            ++w;
        }
        n_cur = w;

        n_highfreq = 0;
        const int highfreqThreshold = PBcoverage > 60 ? (int)((uint64_t)(PBcoverage / 60) * 3) : 3;
        for(uint32_t i = 0; i < n_cur; ++i) {
            const Leaf<P> par = cur[i];                          // one burst of loads: the leaf in registers
            Ext ext[4];
            uint32_t mask = 0;
            int count = 0;
            while(count < 2) {
                if(count == 1 && !(par.localErr == minimumErrorRate && n_cur > 1)) break;
                uint64_t tc;
                const uint64_t tg = tick();
                mask = getFMIndexExtensions_v(par.flo, par.fhi, par.rlo, par.rhi, par.tailLetterCount, par.suf_lo, ext, tc);
                tock(3, tg);
                if(mask != 0) break;
                min_SA_threshold--;
                count++;
            }
            min_SA_threshold += (uint64_t)count;
            if(mask == 0) continue;
            // updateLeaves: children in base order; leafInfo(child) fields (LongReadCorrectByOverlap.h:172-203)
            for(uint32_t b = 0; b < 4; ++b) {
                if(!(mask & (1u << b))) continue;
                if(n_nxt >= kMaxChildren) { error = LRSC_WALK_ERR_CHILDREN; return; }
                Leaf<P> ch = par;                                  // createChild copies the node state (SAINode.cpp:166-189)
                ch.flo = ext[b].f.lo; ch.fhi = ext[b].f.hi; ch.rlo = ext[b].r.lo; ch.rhi = ext[b].r.hi;
                ch.kmerFrequency = ext[b].freq;
                ch.currOverlapLen++;
                ch.queryOverlapLen++;
                if(par.tailLetter == b) ch.tailLetterCount = par.tailLetterCount + 1;
                else { ch.tailLetter = b; ch.tailLetterCount = 1; }
                suf_push(ch, b);
                ch.parent = (uint16_t)i;
                ch.ext = (uint8_t)b;
                ch.alive = 1;
                if(ch.kmerFrequency > highfreqThreshold) n_highfreq++;      // for isInsufficientFreqs, which looks at exactly these values
                nxt[n_nxt++] = ch;
            }
        }
    }

    // ---- extendLeaves (.cpp:239-278) ------------------------------------------------------------------------
    LRSC_WALK_FN void extendLeaves()
    {
        n_nxt = 0;
        uint64_t t = tick();
        if(currentKmerSize > maxOverlap) refineSAInterval(cur, n_cur, maxOverlap);
        tock(1, t);
        t = tick();
        attempToExtend();
        tock(2, t);
        if(error) return;
        if(n_nxt == 0) {                                    // level 1: reduce the k-mer size
            const uint64_t LowerBound = (currentKmerSize - 2) > minOverlap ? (currentKmerSize - 2) : minOverlap;
            const uint64_t ReduceSize = SelectFreqsOfrange(LowerBound, currentKmerSize, cur, n_cur);
            refineSAInterval(cur, n_cur, ReduceSize);
            attempToExtend();
            if(error) return;
            if(n_nxt == 0) {                                // level 2: reduce the threshold
                min_SA_threshold--;
                attempToExtend();
                min_SA_threshold++;
                if(error) return;
            }
        }
        if(n_nxt != 0) {
            currentLength++;
            currentKmerSize++;
            if(isInsufficientFreqs(n_highfreq, n_nxt)) {    // frequencies are low: relax the k-mer size
                const uint64_t LowerBound = (currentKmerSize - 2) > minOverlap ? (currentKmerSize - 2) : minOverlap;
                const uint64_t ReduceSize = SelectFreqsOfrange(LowerBound, currentKmerSize, nxt, n_nxt);
                refineSAInterval(nxt, n_nxt, ReduceSize);
            }
        }
    }

    // ---- isSupportedByNewSeed (.cpp:566-635) ------------------------------------------------------------------
    LRSC_WALK_FN LRSC_WALK_NOINLINE bool isSupportedByNewSeed(Leaf<P>& nd, uint64_t smallSeedIdx, uint64_t largeSeedIdx)
    {
        return seed_support_core(nd, smallSeedIdx, largeSeedIdx);
    }
    LRSC_WALK_FN __forceinline__ bool seed_support_core(Leaf<P>& nd, uint64_t smallSeedIdx, uint64_t largeSeedIdx)
    {
        const uint64_t seedIdxOffset = nd.lastOverlapLen < currentLength - seedSize ? (uint64_t)seedSize : currentLength - nd.lastOverlapLen;
        const uint64_t cand = nd.lastSeedIdx + seedIdxOffset;
        const uint64_t startSeedIdx = smallSeedIdx > cand ? smallSeedIdx : cand;
        bool isNewSeedFound = false;
        const bool fv = nd.flo <= nd.fhi, rv = nd.rlo <= nd.rhi;
        const uint32_t mask9 = seedSize >= 16 ? 0xFFFFFFFFu : ((1u << (2 * seedSize)) - 1u);
        const uint32_t code9 = (uint32_t)nd.suf_lo & mask9;
        const uint32_t hb = (code9 ^ (code9 >> 9)) & 255u;
        // fwd / rvc hit lists in post-sort order: entries of this k-mer chained in the bucket
        uint32_t jf = fv ? head9f[hb] : 0xFFFFu;
        uint32_t jr = rv ? head9r[hb] : 0xFFFFu;
        auto advance = [&](uint32_t j, const SortItem* it, const uint16_t* nx) -> uint32_t {
            while(j != 0xFFFFu && it[j].pad != code9) j = nx[j];
            return j;
        };
        jf = advance(jf, it9f, next9f);
        jr = advance(jr, it9r, next9r);
        int minIdxDiff = 10000;
        const uint64_t currSeedIdx = currentLength - seedSize;
        while(jf != 0xFFFFu || jr != 0xFFFFu) {
            const uint64_t vf = jf != 0xFFFFu ? it9f[jf].val : 0, vr = jr != 0xFFFFu ? it9r[jr].val : 0;
            if(fv && jf != 0xFFFFu && vf >= startSeedIdx && vf <= largeSeedIdx) {
                const int d = abs((int)vf - (int)currSeedIdx);
                if(d < minIdxDiff) { nd.lastSeedIdx = vf; nd.queryOverlapLen = vf + seedSize; minIdxDiff = d; }
                nd.lastOverlapLen = currentLength;
                nd.currOverlapLen = currentLength;
                isNewSeedFound = true;
            } else if(rv && jr != 0xFFFFu && vr >= startSeedIdx && vr <= largeSeedIdx) {
                const int d = abs((int)currSeedIdx - (int)vr);
                if(d < minIdxDiff) { nd.lastSeedIdx = vr; nd.queryOverlapLen = vr + seedSize; minIdxDiff = d; }
                nd.lastOverlapLen = currentLength;
                nd.currOverlapLen = currentLength;
                isNewSeedFound = true;
            }
            if(jf != 0xFFFFu) jf = advance(next9f[jf], it9f, next9f);
            if(jr != 0xFFFFu) jr = advance(next9r[jr], it9r, next9r);
        }
        if(isNewSeedFound) nd.totalSeeds++;
        return isNewSeedFound;
    }

    // seedSize-mer code of m_query at offset i (first character in the high bits)
    LRSC_WALK_FN __forceinline__ uint32_t kmer_code(uint32_t i) const
    {
        uint32_t c = 0;
        for(uint32_t t = 0; t < seedSize; ++t) c = (c << 2) | q[i + t];
        return c;
    }

    // ---- computeErrorRate (.cpp:638-664) ------------------------------------------------------------------
    LRSC_WALK_FN double computeErrorRate(Leaf<P>& nd, const double* parent_ring)
    {
        double matchedLen = (double)nd.totalSeeds + seedSize - 1;
        matchedLen += nd.numRedeemSeed;
        const double totalLen = (double)nd.currOverlapLen;
        const double unmatchedLen = totalLen - matchedLen;
        double currErrorRate = unmatchedLen / totalLen;
        nd.globalErr = currErrorRate;
        const uint32_t totalsize = nd.hist_size + 1;          // after the push_back
        nd.hist_size = totalsize;
        if(totalsize >= localK) {
            const double old = parent_ring[(totalsize - localK) % 100];
            currErrorRate = (currErrorRate * totalLen - old * (totalLen - localK)) / localK;
        }
        nd.localErr = currErrorRate;
        return currErrorRate;
    }

    // ---- PrunedBySeedSupport (.cpp:491-563) ------------------------------------------------------------------
    template <bool INLINE>
    LRSC_WALK_FN __forceinline__ void prune_leaf(Leaf<P>& leaf, const double* pring, uint64_t currSeedIdx, uint64_t smallSeedIdx, uint64_t largeSeedIdx)
    {
        bool isNewSeedFound = false;
        if(currentLength - leaf.lastOverlapLen > seedSize || currentLength - leaf.lastOverlapLen <= 1) {
            const uint64_t preSeedIdx = leaf.lastSeedIdx;
            isNewSeedFound = INLINE ? seed_support_core(leaf, smallSeedIdx, largeSeedIdx) : isSupportedByNewSeed(leaf, smallSeedIdx, largeSeedIdx);
            if(isNewSeedFound) {
                if(currSeedIdx + (uint64_t)(int64_t)leaf.lastSeedIdxOffset - preSeedIdx > seedSize)
                    leaf.numRedeemSeed += (seedSize - 1) * PacBioErrorRate;
                leaf.lastSeedIdxOffset = (int)leaf.lastSeedIdx - (int)currSeedIdx;
            } else {
                const uint64_t v = currSeedIdx + (uint64_t)(int64_t)leaf.lastSeedIdxOffset - leaf.lastSeedIdx;
                if(v % seedSize == 1) leaf.numOfErrors++;
                else if(v > (uint64_t)seedSize - 1) leaf.numRedeemSeed += 1 - PacBioErrorRate;
            }
        } else
            leaf.numRedeemSeed += 1 - PacBioErrorRate;
        const double currErrorRate = computeErrorRate(leaf, pring);
        if(currErrorRate > errorRate) leaf.alive = 0;
    }
    LRSC_WALK_FN void PrunedBySeedSupport()
    {
        const uint64_t currSeedIdx = currentLength - seedSize;
        const uint64_t indelOffset = seedSize + maxIndelSize;
        const uint64_t smallSeedIdx = currSeedIdx <= indelOffset ? 0 : currSeedIdx - indelOffset;
        const uint64_t largeSeedIdx = (currSeedIdx + indelOffset) >= (Lq - seedSize) ? (Lq - seedSize) : currSeedIdx + indelOffset;
        alive_lo = 0; alive_hi = 0; has_child = 0;
        for(uint32_t c = 0; c < n_nxt; ++c) {
            Leaf<P> leaf = nxt[c];                               // in registers for the whole evaluation, one burst each way
            // createChild copied the parent's ring id into the child: cur[leaf.parent].ring == leaf.ring until the commit
            prune_leaf<true>(leaf, rings + (uint64_t)leaf.ring * 100, currSeedIdx, smallSeedIdx, largeSeedIdx);
            nxt[c] = leaf;
            if(leaf.alive) { if(c < 64) alive_lo |= 1ull << c; else alive_hi |= 1ull << (c - 64); has_child |= 1u << leaf.parent; }
        }
    }

    // ---- isTerminated for one leaf (.cpp:825-878); path given as (words, len) + optional extra char ------------
    LRSC_WALK_FN LRSC_WALK_NOINLINE void terminated_leaf(Leaf<P>& lf, const uint32_t* pw, uint32_t plen, int extra)
    {
        const bool fvalid = lf.flo <= lf.fhi, rvalid = lf.rlo <= lf.rhi;
        // A non-empty interval of a k-mer K lies inside the interval of a k-mer w with |w| <= |K| only if K ends with w (fwd strand:
        // reverse(w) is a prefix of reverse(K); rvc strand: revcomp(w) is a prefix of revcomp(K)).  So if no target minOverlap-mer
        // hashes like the leaf's last characters, the scan below cannot hit: skip its loads.
        if(currentKmerSize >= minOverlap) {
            const uint32_t L = minOverlap < 16 ? (uint32_t)minOverlap : 16u;
            const uint32_t code = (uint32_t)(lf.suf_lo & (L >= 16 ? 0xFFFFFFFFull : ((1ull << (2 * L)) - 1ull)));
            const uint32_t h = (code * 0x9E3779B1u) >> 25;
            if((((h < 64 ? tmask0 : tmask1) >> (h & 63u)) & 1ull) == 0) return;
        }
        const uint64_t i0 = (uint64_t)(lf.res_second > 0 ? lf.res_second : 0);
        int hit = -1;
        for(uint64_t i = i0; i <= (uint64_t)trg_len - (int)minOverlap; i++) {
            const P* t = term + i * 4;
            const bool isFwdTerminated = fvalid && lf.flo >= t[0] && lf.fhi <= t[1];
            const bool isRvcTerminated = rvalid && lf.rlo >= t[2] && lf.rhi <= t[3];
            if(isFwdTerminated || isRvcTerminated) {
                hit = (int)i;
                if(lf.res_first == -1) {
                    if(n_results >= kMaxResults) { error = LRSC_WALK_ERR_RESULTS; return; }
                    ++n_results;
                    lf.res_first = (int)n_results;
                }
                lf.res_second = (int)i;
            }
        }
        if(hit < 0) return;
        // results.at(first - 1) = STresult: thread = getFullString() (+ target.substr(i + minOverlap) appended at the end)
        WalkResultRec& r = results[lf.res_first - 1];
        r.error_rate = lf.globalErr;
        r.match_i = (uint32_t)hit;
        uint32_t* dst = rpaths + (uint64_t)(lf.res_first - 1) * pathw;
        const uint32_t nw = (plen + 15) >> 4;
        for(uint32_t k = 0; k < nw; ++k) dst[k] = pw[k];
        uint32_t len = plen;
        if(extra >= 0) { path_set(dst, len, (uint32_t)extra); ++len; }
        r.path_len = len;
    }

    // ---- one walk = begin() + step() until it returns false + finish() ------------------------------------------
    // (split so that the persistent kernel can keep all lanes of a wavefront in the step loop: a lane whose walk
    //  ended sets up its next one while the others wait, instead of idling until the longest walk of the wave ends)
    LRSC_WALK_FN int run(uint32_t* out_len, uint32_t* out_words, uint32_t* out_match_i)
    {
        begin();
        while(step()) {}
        return finish(out_len, out_words, out_match_i);
    }

    bool ended;                        // the frontier overflowed maxLeaves: the loop ends after this isTerminated
    bool profile = false;              // per-step tick counters on (LRSC_CORRECT_PROFILE)

    // The constructor's per-walk tables that stay fixed during the walk (.cpp:90-94,127-152 after the bulk look-ups of
    // prepare_offset): the interval "trees" as sorted k-mer chains, the 5-mer chains, the isTerminated filter.
    LRSC_WALK_FN LRSC_WALK_NOINLINE void begin_static()
    {
        // --- interval "trees": compact the valid 9-mer entries (emplace_back order), introsort, chain by k-mer ---
        auto build9 = [&](SortItem* it, uint32_t n_all, uint16_t* head, uint16_t* next) -> uint32_t {
            uint32_t n = 0;
            for(uint32_t i = 0; i < n_all; ++i)
                if(it[i].key != kNoKey) { if(n != i) it[n] = it[i]; ++n; }
            introsort(it, (int64_t)n);
            for(uint32_t b = 0; b < 256; ++b) head[b] = 0xFFFFu;
            // append in sorted order: chains keep the post-sort order of equal keys
            uint16_t tail[256];
            for(uint32_t j = 0; j < n; ++j) {
                const uint32_t code = it[j].pad;
                const uint32_t hb = (code ^ (code >> 9)) & 255u;
                next[j] = 0xFFFFu;
                if(head[hb] == 0xFFFFu) head[hb] = (uint16_t)j; else next[tail[hb]] = (uint16_t)j;
                tail[hb] = (uint16_t)j;
            }
            return n;
        };
        const uint32_t n9_all = Lq >= seedSize ? Lq - seedSize + 1 : 0;
        n9f = build9(it9f, n9_all, head9f, next9f);
        n9r = build9(it9r, n9_all, head9r, next9r);
        for(uint32_t c = 0; c < 1024; ++c) head5[c] = 0xFFFFu;
        const uint32_t n5 = Lq >= 5 ? Lq - 5 + 1 : 0;
        for(uint32_t i = n5; i-- > 0;) {                       // prepend while walking backwards: ascending chains
            if(flags5[i] == 0) continue;
            uint32_t code = 0;
            for(uint32_t t = 0; t < 5; ++t) code = (code << 2) | q[i + t];
            next5[i] = head5[code];
            head5[code] = (uint16_t)i;
        }

        // filter for isTerminated: a leaf can only be contained in the interval of a target k-mer it ends with
        {
            tmask0 = 0; tmask1 = 0;
            const uint32_t L = minOverlap < 16 ? (uint32_t)minOverlap : 16u;
            const uint32_t trg0 = initk + path_len;
            for(uint32_t i = 0; i < n_term; ++i) {
                uint32_t code = 0;
                const uint8_t* p = q + trg0 + i + (uint32_t)minOverlap - L;
                for(uint32_t t = 0; t < L; ++t) code = (code << 2) | p[t];
                const uint32_t h = (code * 0x9E3779B1u) >> 25;
                if(h < 64) tmask0 |= 1ull << h; else tmask1 |= 1ull << (h - 64);
            }
        }
    }

    // --- root (initialRootNode, .cpp:108-124; leafInfo ctor, .h:156-171).  root_iv: the root k-mer's bi-interval {fwd.lo,
    //     fwd.hi, rvc.lo, rvc.hi} when a preparation pass has already searched it, nullptr to search it here ---
    LRSC_WALK_FN LRSC_WALK_NOINLINE void begin_root(const P* root_iv)
    {
        ring_free = 0xFFFFFFFEu; path_free = 0xFFFFFFFEu;
        Leaf<P>& root = cur[0];
        root.suf_lo = 0; root.suf_hi = 0;
        for(uint32_t t = 0; t < initk; ++t) suf_push(root, q[t]);
        if(root_iv) { root.flo = root_iv[0]; root.fhi = root_iv[1]; root.rlo = root_iv[2]; root.rhi = root_iv[3]; }
        else find_suffix(root, initk);
        root.lastOverlapLen = root.currOverlapLen = root.queryOverlapLen = initk;
        currentLength = currentKmerSize = initk;
        root.lastSeedIdx = (uint64_t)initk - seedSize;
        root.totalSeeds = (uint64_t)initk - seedSize + 1;
        root.numOfErrors = 0;
        root.numRedeemSeed = 0;
        root.localErr = 0; root.globalErr = 0; root.hist_size = 1;
        root.lastSeedIdxOffset = 0;
        root.res_first = -1; root.res_second = -1;
        root.kmerFrequency = (int)(isize(root.flo, root.fhi) + isize(root.rlo, root.rhi));
        root.tailLetter = q[initk - 1];
        root.tailLetterCount = 0;
        for(uint32_t t = initk; t-- > 0;) { if(q[t] == root.tailLetter) root.tailLetterCount++; else break; }
        root.ring = 0; root.path = 0; root.parent = 0; root.ext = 0; root.alive = 1;
        root.path_len = initk;
        rings[0] = 0.0;
        for(uint32_t t = 0; t < initk; ++t) path_set(paths, t, q[t]);
        n_cur = 1; n_nxt = 0; n_results = 0;
        ended = false;
    }

    LRSC_WALK_FN LRSC_WALK_NOINLINE void begin()
    {
        const uint64_t t_run0 = __builtin_readcyclecounter();
        begin_static();
        begin_root(nullptr);
        cyc_setup += __builtin_readcyclecounter() - t_run0;
    }

    // ---------------------------------------------------------------------------------------------------------
    // Single-leaf fast path.  While the frontier is ONE leaf (most steps: 1.36 leaves per step on average) the leaf lives in
    // registers (`L`) between steps, and a step touches memory only for what the algorithm itself reads: rank blocks, k-mer table
    // entries, the query's 9-mer chains, one error-history ring slot, one path word.  Same arithmetic in the same order as the
    // general step (extendLeaves / PrunedBySeedSupport / the commit in step_body); a step that is not of the simple kind -- no
    // accepted base, more than one -- is handed to the general code BEFORE it has had any effect.
    // ---------------------------------------------------------------------------------------------------------
    LRSC_WALK_FN __forceinline__ bool can_fast() const { return !ended && !error && n_cur == 1; }
    LRSC_WALK_FN __forceinline__ void enter_fast(Leaf<P>& L, uint32_t& pw)
    {
        L = cur[0];
        pw = (L.path_len & 15u) ? paths[(uint64_t)L.path * pathw + (L.path_len >> 4)] : 0u;
    }
    LRSC_WALK_FN __forceinline__ void leave_fast(const Leaf<P>& L) { cur[0] = L; }
    // 1: step done, the frontier is still the one leaf in L;  0: extendOverlap's loop is over (L written back, finish() decides);
    // 2: not a simple step -- nothing has happened, L written back: run the general step()
    LRSC_WALK_FN __forceinline__ int step_fast(Leaf<P>& L, uint32_t& pw)
    {
        if(!(currentLength <= maxLength)) { leave_fast(L); return 0; }
        // attempToExtend's trimming never fires for one leaf unless its local error rate is >= 1 (the minimum starts at 1): general code
        if(!(L.localErr < 1.0)) { leave_fast(L); return 2; }
        const uint32_t nr0 = n_rank, nb0 = n_blk;
        // --- extendLeaves (.cpp:239-278) ---
        uint64_t ck = currentKmerSize;
        P flo = L.flo, fhi = L.fhi, rlo = L.rlo, rhi = L.rhi;
        if(ck > maxOverlap) { find_suffix_v(L.suf_lo, L.suf_hi, (uint32_t)maxOverlap, flo, fhi, rlo, rhi); ck = maxOverlap; }
        Ext ext[4];
        uint64_t tc;
        const uint32_t mask = getFMIndexExtensions_v(flo, fhi, rlo, rhi, L.tailLetterCount, L.suf_lo, ext, tc);
        if(mask == 0 || (mask & (mask - 1u)) != 0) { n_rank = nr0; n_blk = nb0; leave_fast(L); return 2; }
        const uint32_t b = (uint32_t)__builtin_ctz(mask);
        // updateLeaves: the one child (createChild + leafInfo ctor)
        L.flo = ext[b].f.lo; L.fhi = ext[b].f.hi; L.rlo = ext[b].r.lo; L.rhi = ext[b].r.hi;
        L.kmerFrequency = ext[b].freq;
        L.currOverlapLen++;
        L.queryOverlapLen++;
        if(L.tailLetter == b) L.tailLetterCount = L.tailLetterCount + 1;
        else { L.tailLetter = b; L.tailLetterCount = 1; }
        suf_push(L, b);
        L.parent = 0; L.ext = (uint8_t)b; L.alive = 1;
        currentLength++;
        ck++;
        {   // isInsufficientFreqs for one leaf: no k-mer above the threshold
            const int highfreqThreshold = PBcoverage > 60 ? (int)((uint64_t)(PBcoverage / 60) * 3) : 3;
            if(!(L.kmerFrequency > highfreqThreshold)) {
                const uint64_t LowerBound = (ck - 2) > minOverlap ? (ck - 2) : minOverlap;
                const uint64_t ReduceSize = SelectFreqsOfrange1(LowerBound, ck, L.suf_lo, L.suf_hi);
                find_suffix_v(L.suf_lo, L.suf_hi, (uint32_t)ReduceSize, L.flo, L.fhi, L.rlo, L.rhi);
                ck = ReduceSize;
            }
        }
        currentKmerSize = ck;
        // --- PrunedBySeedSupport ---
        {
            const uint64_t currSeedIdx = currentLength - seedSize;
            const uint64_t indelOffset = seedSize + maxIndelSize;
            const uint64_t smallSeedIdx = currSeedIdx <= indelOffset ? 0 : currSeedIdx - indelOffset;
            const uint64_t largeSeedIdx = (currSeedIdx + indelOffset) >= (Lq - seedSize) ? (Lq - seedSize) : currSeedIdx + indelOffset;
            prune_leaf<true>(L, rings + (uint64_t)L.ring * 100, currSeedIdx, smallSeedIdx, largeSeedIdx);
        }
        ++steps; ++leaf_steps;
        if(!L.alive) { leave_fast(L); n_cur = 0; return 0; }         // the frontier is empty: "high error"
        // --- commit: the child takes over its parent's ring and path in place ---
        rings[(uint64_t)L.ring * 100 + (L.hist_size - 1) % 100] = L.globalErr;     // GlobalErrorRateRecord.push_back
        {
            const uint32_t sh = 2 * (L.path_len & 15u);
            if(sh == 0) pw = 0;
            pw = (pw & ~(3u << sh)) | (b << sh);
            paths[(uint64_t)L.path * pathw + (L.path_len >> 4)] = pw;
            L.path_len++;
        }
        // --- isTerminated ---
        if(currentLength >= minLength) {
            bool may_hit = true;
            if(currentKmerSize >= minOverlap) {
                const uint32_t Lm = minOverlap < 16 ? (uint32_t)minOverlap : 16u;
                const uint32_t code = (uint32_t)(L.suf_lo & (Lm >= 16 ? 0xFFFFFFFFull : ((1ull << (2 * Lm)) - 1ull)));
                const uint32_t h = (code * 0x9E3779B1u) >> 25;
                may_hit = (((h < 64 ? tmask0 : tmask1) >> (h & 63u)) & 1ull) != 0;
            }
            if(may_hit) {
                leave_fast(L);
                terminated_leaf(cur[0], paths + (uint64_t)L.path * pathw, L.path_len, -1);
                if(error) return 0;
                L.res_first = cur[0].res_first; L.res_second = cur[0].res_second;
            }
        }
        return 1;
    }

    // The memory-bound front of a step (extension + seed support) on whatever frontier `cur` points at, nothing committed: a helper
    // lane of wp_extend_coop_kernel runs it on a private copy of ONE leaf ahead of the owner's real step, to have the rank blocks,
    // table entries and 9-mer chains that leaf needs in the caches.  Writes: cur / nxt (private there) and this object's scalars.
    LRSC_WALK_FN void warm()
    {
        extendLeaves();
        if(!error) PrunedBySeedSupport();
    }

    // one iteration of extendOverlap's loop (.cpp:155-211); false when the loop is over (or on an internal error)
    LRSC_WALK_FN bool step()
    {
        if(ended || error || !(n_cur != 0 && n_cur <= maxLeaves && currentLength <= maxLength)) return false;
        leaf_steps += n_cur;
        if(n_cur > max_front) max_front = n_cur;
        if(profile) {                                    // two s_memtime round trips per step are not free: only when asked for
            const uint64_t t_step0 = __builtin_readcyclecounter();
            step_body();
            cyc_loop += __builtin_readcyclecounter() - t_step0;
        } else
            step_body();
        return true;
    }

    LRSC_WALK_FN void step_body()
    {
        {
            uint64_t t = tick();
            extendLeaves();
            tock(0, t);
            if(error) return;
            t = tick();
            PrunedBySeedSupport();
            tock(4, t);
            t = tick();
            const uint32_t survivors = (uint32_t)(__builtin_popcountll(alive_lo) + __builtin_popcountll(alive_hi));
            auto is_alive = [&](uint32_t c) -> bool { return ((c < 64 ? alive_lo >> c : alive_hi >> (c - 64)) & 1ull) != 0; };
            ++steps;
            if(survivors > maxLeaves) {
                // the frontier overflows: the loop ends after this isTerminated, no leaf state is needed any more
                if(currentLength >= minLength)
                    for(uint32_t c = 0; c < n_nxt; ++c) {
                        if(!is_alive(c)) continue;
                        const Leaf<P>& par = cur[nxt[c].parent];
                        terminated_leaf(nxt[c], paths + (uint64_t)par.path * pathw, par.path_len, (int)nxt[c].ext);
                        if(error) return;
                    }
                n_cur = survivors;
                ended = true;
                return;
            }
            // materialise the survivors: the first surviving child of a parent takes over its ring and path
            // in place (SAINode::extend), further ones get copies (createChild).  A child still carries its parent's
            // ring / path ids and path length (createChild copied them), so the parent need not be read again.
            // has_child (from PrunedBySeedSupport): bit i = leaf i of cur[] has a surviving child (n_cur <= 32)
            for(uint32_t i = 0; i < n_cur; ++i) if(!((has_child >> i) & 1u)) free_leaf_slots(cur[i]);
            const bool want_term = currentLength >= minLength;
            uint32_t seen = 0, w = 0;
            for(uint32_t c = 0; c < n_nxt; ++c) {
                if(!is_alive(c)) continue;
                Leaf<P> ch = nxt[c];                             // registers: one burst in, one out
                if((seen >> ch.parent) & 1u) {
                    // a further child of this parent: copies of the ring and of the path as they are BEFORE this step's appends
                    // (the in-place child of the parent has only written slots the copies do not read: see below)
                    const uint16_t pr = ch.ring, pp = ch.path;
                    ch.ring = (uint16_t)__builtin_ctz(ring_free); ring_free &= ring_free - 1u;     // survivors <= 32 slots: never empty here
                    ch.path = (uint16_t)__builtin_ctz(path_free); path_free &= path_free - 1u;
                    const double* src = rings + (uint64_t)pr * 100;
                    double* dst = rings + (uint64_t)ch.ring * 100;
                    const uint32_t own = (ch.hist_size - 1) % 100;                 // the slot this child overwrites anyway
                    for(uint32_t k = 0; k < 100; ++k) if(k != own) dst[k] = src[k];
                    const uint32_t* ps = paths + (uint64_t)pp * pathw;
                    uint32_t* pd = paths + (uint64_t)ch.path * pathw;
                    const uint32_t nw = (ch.path_len + 16) >> 4;
                    for(uint32_t k = 0; k < nw; ++k) pd[k] = ps[k];
                }
                seen |= 1u << ch.parent;
                rings[(uint64_t)ch.ring * 100 + (ch.hist_size - 1) % 100] = ch.globalErr;     // GlobalErrorRateRecord.push_back
                path_set(paths + (uint64_t)ch.path * pathw, ch.path_len, ch.ext);
                ch.path_len++;
                nxt[w] = ch;                                       // compaction
                ++w;
            }
            (void)want_term;
            // m_leaves = newLeaves: the two leaf buffers trade places when the old `cur` region (32 or kMaxChildren slots) can take
            // the next step's children (4 per survivor); otherwise the survivors are copied down as before
            if(4u * w <= (cur == leaf_small ? 32u : kMaxChildren)) { Leaf<P>* t2 = cur; cur = nxt; nxt = t2; }
            else for(uint32_t i = 0; i < w; ++i) cur[i] = nxt[i];
            n_cur = w;
            tock(5, t);
            t = tick();
            if(currentLength >= minLength)
                for(uint32_t i = 0; i < n_cur; ++i) {
                    terminated_leaf(cur[i], paths + (uint64_t)cur[i].path * pathw, cur[i].path_len, -1);
                    if(error) return;
                }
            tock(6, t);
        }
    }

    LRSC_WALK_FN LRSC_WALK_NOINLINE int finish(uint32_t* out_len, uint32_t* out_words, uint32_t* out_match_i)
    {
        if(error) return error;
        // --- findTheBestPath (.cpp:214-236) ---
        if(n_results > 0) {
            double minErrorRate = 1;
            int best = -1;
            for(uint32_t i = 0; i < n_results; ++i)
                if(results[i].error_rate < minErrorRate) { minErrorRate = results[i].error_rate; best = (int)i; }
            if(best < 0) return -4;
            *out_len = results[best].path_len;
            *out_match_i = results[best].match_i;
            const uint32_t* src = rpaths + (uint64_t)best * pathw;
            const uint32_t nw = (results[best].path_len + 15) >> 4;
            for(uint32_t k = 0; k < nw; ++k) out_words[k] = src[k];
            return 1;
        }
        if(n_cur == 0) return -1;                    // high error
        else if(currentLength > maxLength) return -2;   // exceed search depth
        else if(n_cur > maxLeaves) return -3;        // too much repeats
        return -4;
    }
};


} // namespace lrsc
