// walk_sm.h -- the per-read correction chain (PacBioSelfCorrectionProcess::initCorrect / correctByFMExtension,
// PacBio/PacBioSelfCorrectionProcess.cpp:56-206) and the seed-to-seed FM-extension it drives
// (LongReadSelfCorrectByOverlap, PacBio/LongReadCorrectByOverlap.cpp:17-878) as an explicit per-lane STATE MACHINE.
//
// Why a state machine: one lane owns one read, and the reads of a wavefront are at unrelated points of their chains.
// Written as nested loops, every lane sits at a different program counter and the wavefront executes the lanes one
// after the other (1.4 of 64 lanes active per instruction in round 1).  Here the control flow of the whole wavefront
// is ONE loop:
//
//      loop:  R-phase   every lane that has a pending FM-index request (a pair of updateIntervals, or a k-mer
//                        interval-table look-up) gets it answered -- all lanes in the same instructions
//             sweep     a fixed sequence of `if(pc == X)` blocks; a lane runs the blocks its state selects until it
//                        has the next request (lanes in the same state share the instructions)
//
// so that divergence is in data, not in control.  All rank queries of the path -- refineSAInterval / initialRootNode
// (FS), SelectFreqsOfrange (SF), getFMIndexExtensions (EXT), the constructor's per-offset searches (PREP) -- go through
// the R-phase; everything else is the reference's bookkeeping, restated once per state.
//
// The file is host/device neutral (no wavefront intrinsics): correct_sm.hip drives it on the GPU, the emulation
// harness under tests/host_emul drives the very same code lane by lane on the CPU (test infrastructure only).
#pragma once
#include <hip/hip_runtime.h>

#include "correct_dev.h"
#include "walk_device.h"
#include "ws_layout.h"

namespace lrsc {

#define LRSC_SM __host__ __device__ inline
// (LRSC_SM_NI marks the block-level pieces of the sweep; they are inlined like the rest so that the address space of the lane's
//  state object -- LDS in the kernel -- is known at every access.)
#define LRSC_SM_NI __host__ __device__ inline


// Byte-string helpers for the stitching code.  A plain `for(t) d[t] = s[t]` over uint8_t pointers must assume d aliases
// everything: every iteration waits for its own load (a global-memory round trip per byte).  These move 16 bytes per
// round trip: the loads of a batch are independent.
LRSC_SM void copy_codes(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, uint32_t n)
{
    uint32_t t = 0;
    for(; t + 16 <= n; t += 16) {
        uint8_t v[16];
#pragma unroll
        for(int j = 0; j < 16; ++j) v[j] = s[t + j];
#pragma unroll
        for(int j = 0; j < 16; ++j) d[t + j] = v[j];
    }
    for(; t < n; ++t) d[t] = s[t];
}
// d[t] = 3 - s_last[-t], t = 0..n-1 (reverse complement, reading backwards from s_last)
LRSC_SM void copy_codes_rc(uint8_t* __restrict__ d, const uint8_t* __restrict__ s_last, uint32_t n)
{
    uint32_t t = 0;
    for(; t + 16 <= n; t += 16) {
        uint8_t v[16];
#pragma unroll
        for(int j = 0; j < 16; ++j) v[j] = *(s_last - (int64_t)(t + j));
#pragma unroll
        for(int j = 0; j < 16; ++j) d[t + j] = (uint8_t)(3 - v[j]);
    }
    for(; t < n; ++t) d[t] = (uint8_t)(3 - *(s_last - (int64_t)t));
}
// d[t] = path character (from + t), t = 0..n-1 (2-bit packed words)
LRSC_SM void unpack_path(uint8_t* __restrict__ d, const uint32_t* __restrict__ words, uint32_t from, uint32_t n)
{
    uint32_t t = 0;
    while(t < n) {
        const uint32_t j = from + t;
        const uint32_t w = words[j >> 4];
        const uint32_t in_word = 16u - (j & 15u);
        const uint32_t m = in_word < n - t ? in_word : n - t;
        for(uint32_t u = 0; u < m; ++u) d[t + u] = (uint8_t)((w >> (2 * ((j + u) & 15u))) & 3u);
        t += m;
    }
}
// d[t] = 3 - path character (last - t), t = 0..n-1
LRSC_SM void unpack_path_rc(uint8_t* __restrict__ d, const uint32_t* __restrict__ words, uint32_t last, uint32_t n)
{
    uint32_t t = 0;
    while(t < n) {
        const uint32_t j = last - t;
        const uint32_t w = words[j >> 4];
        const uint32_t in_word = (j & 15u) + 1u;
        const uint32_t m = in_word < n - t ? in_word : n - t;
        for(uint32_t u = 0; u < m; ++u) d[t + u] = (uint8_t)(3u - ((w >> (2 * ((j - u) & 15u))) & 3u));
        t += m;
    }
}

// Where the wave-uniform context lives.  In the kernel translation unit (correct_sm.hip defines LRSC_SM_KERNEL_TU and the
// file-scope __shared__ objects before including this header) the launch arguments, the index description and the strand
// constants are per-wavefront LDS copies with a statically known address space; everywhere else (the CPU harness) they are
// reached through the pointers kept in the state object.
#if defined(__HIP_DEVICE_COMPILE__) && defined(LRSC_SM_KERNEL_TU)
#define LRSC_A (g_sm_a)
#define LRSC_FM (g_sm_fm)
#define LRSC_SF (sm_strand_lds<P>(0))
#define LRSC_SR (sm_strand_lds<P>(1))
template <class T> __device__ __forceinline__ T* sm_global(T* p) { return p; }   // (address-space hints did not survive; accesses stay flat)
#else
#define LRSC_A (*A)
#define LRSC_FM (*fm)
#define LRSC_SF (*sFp)
#define LRSC_SR (*sRp)
template <class T> __host__ __device__ inline T* sm_global(T* p) { return p; }
#endif

enum : uint32_t { kReqNone = 0, kReqRank = 1, kReqTab = 2, kReqExt = 3 };
enum : uint32_t { kReqDoA = 1u, kReqDoB = 2u, kReqSwap = 4u };

// A pending FM-index request of one lane.
//   kReqRank: a = updateInterval(X, ca, a) if DoA, b = updateInterval(Y, cb, b) if DoB, with (X, Y) = (rBWT, BWT), or
//             (BWT, rBWT) when Swap is set (SelectFreqsOfrange searches the k-mer itself, not its reverse).
//   kReqTab:  entry `code` of k-mer table `tab` -> a = {fwd.lo, fwd.hi}, b = {rvc.lo, rvc.hi}
//   kReqExt:  getFMIndexExtensions' rank queries for all four bases at once: for base x, a_x = updateInterval(rBWT, x, a)
//             if DoA and b_x = updateInterval(BWT, 3 - x, b) if DoB.  The four bases of a strand read the SAME two rank
//             blocks (Occ(., lo - 1) and Occ(., hi)), so this costs the memory traffic of one kReqRank; the sixteen
//             results go straight to the lane's `ex` slots.
template <class P>
struct SmReq {
    uint32_t kind, flags;
    P a_lo, a_hi, b_lo, b_hi;
    uint32_t ca, cb;
    uint32_t tab, code;
};

enum : uint32_t {
    // order = order of the blocks in sweep(); forward transitions are taken in the same sweep
    PC_FS = 1,            // find_suffix over a leaf list: waiting for a table entry or a step
    PC_SF,                // SelectFreqsOfrange
    PC_ROOT_DONE,
    PC_EXT,               // getFMIndexExtensions of leaf att_i: waiting for the R-phase
    PC_EXT_READY,         // ... answered: waits for the wavefront's step gate, then the acceptance ladder + children
    PC_ATT_DONE,
    PC_AFTER_SF_A,
    PC_POST,
    PC_AFTER_SF_B,
    PC_PRUNE,
    PC_PRUNE_SLOW,        // a frontier with several leaves / children: the general commit, run behind its own (rarer) gate
    PC_STEP_ENTRY,
    PC_WALK_END,
    PC_NEXT,              // between walks: next target / yield / done
    PC_PREP,              // constructor: per-offset searches
    PC_BEGIN,             // interval "trees" + root
    PC_ATT_ENTRY,
    PC_ATT_LEAF,
    PC_FINAL,
    PC_DONE
};

// debugging aid: one 14-word record per sweep of the traced read (the same on the device and in the CPU harness)
template <class P>
__host__ __device__ inline void sm_trace(uint32_t* trace, uint32_t cap, uint32_t& pos, uint32_t pc, bool have, const SmReq<P>& rq, const SmReq<P>& res)
{
    if(trace == nullptr || pos + 14 > cap) return;
    uint32_t* t = trace + pos;
    t[0] = pc; t[1] = have ? rq.kind : 0u; t[2] = rq.flags; t[3] = rq.ca | (rq.cb << 8); t[4] = rq.tab; t[5] = rq.code;
    t[6] = (uint32_t)rq.a_lo; t[7] = (uint32_t)rq.a_hi; t[8] = (uint32_t)rq.b_lo; t[9] = (uint32_t)rq.b_hi;
    t[10] = (uint32_t)res.a_lo; t[11] = (uint32_t)res.a_hi; t[12] = (uint32_t)res.b_lo; t[13] = (uint32_t)res.b_hi;
    pos += 14;
}

// The R-phase for one lane: answers rq into res (a_lo, a_hi, b_lo, b_hi).  All lanes of a wavefront call this together.
template <bool WIDE>
__host__ __device__ __forceinline__ void sm_answer(const FmIndexDev& fm, const StrandC<typename Lay<WIDE>::pos_t>& sF,
                                                   const StrandC<typename Lay<WIDE>::pos_t>& sR, const uint32_t* mtab,
                                                   const SmReq<typename Lay<WIDE>::pos_t>& rq, SmReq<typename Lay<WIDE>::pos_t>& res,
                                                   typename Lay<WIDE>::pos_t* ex, uint32_t ex_stride,
                                                   uint32_t& n_rank, uint32_t& n_blk, uint32_t& n_tab)
{
    using P = typename Lay<WIDE>::pos_t;
    res.a_lo = rq.a_lo; res.a_hi = rq.a_hi; res.b_lo = rq.b_lo; res.b_hi = rq.b_hi;
    if(rq.kind == kReqTab) {
        const void* tabv = rq.tab == 0 ? fm.ktab[0].entries : rq.tab == 1 ? fm.ktab[1].entries : rq.tab == 2 ? fm.ktab[2].entries
                         : rq.tab == 3 ? fm.ktab[3].entries : fm.ktab[4].entries;
        if(WIDE) {
            const uint4* t = reinterpret_cast<const uint4*>(tabv) + (uint64_t)rq.code * 2;
            const uint4 a = t[0], b = t[1];
            res.a_lo = (P)(((uint64_t)a.y << 32) | a.x); res.a_hi = (P)(((uint64_t)a.w << 32) | a.z);
            res.b_lo = (P)(((uint64_t)b.y << 32) | b.x); res.b_hi = (P)(((uint64_t)b.w << 32) | b.z);
        } else {
            const uint4 a = reinterpret_cast<const uint4*>(tabv)[rq.code];
            res.a_lo = (P)a.x; res.a_hi = (P)a.y; res.b_lo = (P)a.z; res.b_hi = (P)a.w;
        }
        n_tab += 1;
    } else if(rq.kind == kReqRank) {
        const bool swap = (rq.flags & kReqSwap) != 0;
        // per-lane strand choice: (rBWT, BWT) or swapped
        StrandC<P> SA, SB;
        SA.blocks = swap ? sR.blocks : sF.blocks;   SB.blocks = swap ? sF.blocks : sR.blocks;
        SA.dollars = swap ? sR.dollars : sF.dollars; SB.dollars = swap ? sF.dollars : sR.dollars;
        SA.n_dollars = swap ? sR.n_dollars : sF.n_dollars; SB.n_dollars = swap ? sF.n_dollars : sR.n_dollars;
        SA.dollar_dir = swap ? sR.dollar_dir : sF.dollar_dir; SB.dollar_dir = swap ? sF.dollar_dir : sR.dollar_dir;
        SA.syms_per_group = sF.syms_per_group; SB.syms_per_group = sF.syms_per_group;
        SA.c1 = swap ? sR.c1 : sF.c1; SA.c2 = swap ? sR.c2 : sF.c2; SA.c3 = swap ? sR.c3 : sF.c3; SA.c4 = swap ? sR.c4 : sF.c4; SA.n = swap ? sR.n : sF.n;
        SB.c1 = swap ? sF.c1 : sR.c1; SB.c2 = swap ? sF.c2 : sR.c2; SB.c3 = swap ? sF.c3 : sR.c3; SB.c4 = swap ? sF.c4 : sR.c4; SB.n = swap ? sF.n : sR.n;
        // a side that is not asked for searches the empty interval [0, -1] of block 0 and its answer is dropped: no branch,
        // all loads of both sides in flight together
        const bool doa = (rq.flags & kReqDoA) != 0, dob = (rq.flags & kReqDoB) != 0;
        IvT<P> oa, ob;
        uint32_t nba = 0, nbb = 0;
        update_pair_b<WIDE>(SA, rq.ca & 3u, IvT<P>{doa ? rq.a_lo : (P)0, doa ? rq.a_hi : (P)0}, SB, rq.cb & 3u,
                            IvT<P>{dob ? rq.b_lo : (P)0, dob ? rq.b_hi : (P)0}, mtab, oa, ob, nba, nbb);
        if(doa) { res.a_lo = oa.lo; res.a_hi = oa.hi; n_rank += 2; n_blk += nba; }
        if(dob) { res.b_lo = ob.lo; res.b_hi = ob.hi; n_rank += 2; n_blk += nbb; }
    } else if(rq.kind == kReqExt) {
        // element (x * 4 + f) of the lane's ex slots: f = 0,1 the rBWT pair of base x, f = 2,3 the BWT pair of base x
        IvT<P> fo[4], ro[4];
        for(uint32_t x = 0; x < 4; ++x) { fo[x] = IvT<P>{rq.a_lo, rq.a_hi}; ro[x] = IvT<P>{rq.b_lo, rq.b_hi}; }
        if(rq.flags & kReqDoA) { update_interval_all<WIDE>(sF, IvT<P>{rq.a_lo, rq.a_hi}, mtab, fo, n_blk); n_rank += 8; }
        if(rq.flags & kReqDoB) { update_interval_all<WIDE>(sR, IvT<P>{rq.b_lo, rq.b_hi}, mtab, ro, n_blk); n_rank += 8; }
        for(uint32_t x = 0; x < 4; ++x) {
            ex[(x * 4u + 0u) * ex_stride] = fo[x].lo; ex[(x * 4u + 1u) * ex_stride] = fo[x].hi;
            ex[(x * 4u + 2u) * ex_stride] = ro[3u - x].lo; ex[(x * 4u + 3u) * ex_stride] = ro[3u - x].hi;      // BWT strand: code 3 - x
        }
    }
}

template <bool WIDE>
struct ReadSM {
    using P = typename Lay<WIDE>::pos_t;
    using LeafT = Leaf<P>;

    // ---- wave-uniform context (pointers into kernel arguments) ----
    const FmIndexDev* __restrict__ fm;
    const CorrectArgs* __restrict__ A;
    const StrandC<P>* sFp;            // rBWT / BWT constants (the caller's wave-uniform copies)
    const StrandC<P>* sRp;

    // ---- the read ----
    uint32_t r;
    uint64_t rs;                      // first base of the read in LRSC_A.codes
    uint32_t n_seeds;
    uint8_t* ws;                      // the read's workspace
    uint32_t lq_max, pathw;           // what its variable-size regions are laid out for (ws_layout.h)
    // chain state (pieceVec.back()'s SeedFeature fields + iterTarget); the integer counters live in LRSC_A.out[r].c[]
    int32_t S_seedLen, S_end, S_endBest, S_maxFixed;
    bool S_isRepeat;
    uint32_t it;
    int32_t next, firstType;
    uint32_t out_len, n_pieces;
    int32_t error;
    uint32_t state;               // kReadDone / kReadParked / kReadYield
    uint32_t walks_here;
    uint32_t steps, steps0;
    // current walk geometry
    int32_t T_start, T_len, interval, k, trg_len;
    bool T_isRepeat, rtou;
    uint32_t Lq, initk;
    uint32_t min_SA_threshold, maxIndelSize, maxLength, minLength, currentLength, currentKmerSize;
    uint32_t maxOverlap;
    uint32_t n_cur, n_nxt, n_results;
    uint32_t slot_free;           // free ring/path slots (bit mask; a leaf's ring and path slot ids are always equal)
    bool ended;
    // ---- state machine ----
    uint32_t pc;
    SmReq<P> req;
    // one register set for the three searches that are never active together (FS / SF / PREP)
    uint32_t m_list, m_n, m_j, m_len, m_t, m_ret;
    bool m_start, m_fb, m_rb, m_flag;          // m_flag: FS "set currentKmerSize", PREP "inside the target seed"
    P m_flo, m_fhi, m_rlo, m_rhi;
    uint64_t m_suf_lo, m_suf_hi;
    // SF only
    uint32_t sf_i, sf_phase, sf_LB, sf_UB, sf_result;
    int32_t sf_max;
    // ATT / EXT
    uint32_t att_no, att_i;
    double att_minErr;
    // accounting
    uint32_t n_rank, n_blk, n_tab;
    uint64_t* tkp;                // profiling: fine-grained tick accumulators (nullptr normally)
    uint32_t lds_pad[3];          // the kernel keeps one ReadSM per lane in LDS: an odd dword count keeps same-member accesses conflict-free

    // ---- workspace views (ws_layout.h: fixed-size regions at constant offsets, the rest from lq_max / pathw) ----
    static constexpr uint32_t kLB = (uint32_t)sizeof(Leaf<P>);
    LRSC_SM uint8_t* wsg() const { return sm_global(ws); }
    LRSC_SM WsVar var() const { return ws_var_offsets(kLB, (uint32_t)sizeof(P), lq_max, LRSC_A.seed_size, pathw); }
    LRSC_SM uint32_t n9cap() const { return (lq_max > 16u ? lq_max : 16u) - LRSC_A.seed_size + 1u; }
    LRSC_SM uint32_t lqcap() const { return lq_max > 16u ? lq_max : 16u; }
    LRSC_SM SortItem* it9f() const { return reinterpret_cast<SortItem*>(wsg() + ws_var_base(kLB)); }
    LRSC_SM SortItem* it9r() const { return reinterpret_cast<SortItem*>(wsg() + ws_var_base(kLB)) + n9cap(); }
    LRSC_SM P* term() const { return reinterpret_cast<P*>(wsg() + ws_var_base(kLB) + n9cap() * 32u); }
    LRSC_SM uint32_t o_paths() const { return ws_al16(ws_var_base(kLB) + n9cap() * 32u + lqcap() * 4u * (uint32_t)sizeof(P)); }
    LRSC_SM uint32_t* paths() const { return reinterpret_cast<uint32_t*>(wsg() + o_paths()); }
    LRSC_SM uint32_t* rpaths() const { return paths() + 32u * pathw; }
    LRSC_SM uint32_t* best() const { return paths() + (32u + kMaxResults) * pathw; }
    LRSC_SM uint32_t o_next9f() const { return o_paths() + (32u + kMaxResults + 1u) * pathw * 4u; }
    LRSC_SM uint16_t* next9f() const { return reinterpret_cast<uint16_t*>(wsg() + o_next9f()); }
    LRSC_SM uint16_t* next9r() const { return next9f() + n9cap(); }
    // per query offset: 5-mer code | (fwd interval valid) << 10 | (rvc interval valid) << 11   (PREP writes it)
    LRSC_SM uint16_t* c5() const { return next9f() + 2u * n9cap(); }
    LRSC_SM uint8_t* flags5() const { return wsg() + o_next9f() + 4u * n9cap() + 2u * (lqcap() - 4u); }
    LRSC_SM uint8_t* q() const { return flags5() + (lqcap() - 4u); }
    LRSC_SM uint8_t* dpq() const { return q() + lqcap(); }
    LRSC_SM uint16_t* head9f() const { return reinterpret_cast<uint16_t*>(wsg() + ws_fixed_head9(kLB)); }
    LRSC_SM uint16_t* head9r() const { return head9f() + 256; }
    LRSC_SM uint16_t* head5() const { return reinterpret_cast<uint16_t*>(wsg() + ws_fixed_head5(kLB)); }
    LRSC_SM LeafT* cur() const { return reinterpret_cast<LeafT*>(wsg()); }
    LRSC_SM LeafT* nxt() const { return reinterpret_cast<LeafT*>(wsg()) + 32; }
    LRSC_SM LeafT* leaves(uint32_t list) const { return list ? nxt() : cur(); }
    LRSC_SM double* rings() const { return reinterpret_cast<double*>(wsg() + ws_fixed_rings(kLB)); }
    LRSC_SM WalkResultRec* results() const { return reinterpret_cast<WalkResultRec*>(wsg() + ws_fixed_results(kLB)); }
    // read-level views, recomputed where needed (used between walks only)
    LRSC_SM const uint8_t* read() const { return sm_global(LRSC_A.codes) + rs; }
    LRSC_SM const int32_t* seeds() const { return sm_global(LRSC_A.seeds) + seed_slab(rs, r, LRSC_A.min_k) * kSeedInts; }
    LRSC_SM uint8_t* out() const { return sm_global(LRSC_A.out_codes) + sm_global(LRSC_A.work)[r].out_off; }
    LRSC_SM uint32_t* piece_start() const { return sm_global(LRSC_A.piece_start) + sm_global(LRSC_A.work)[r].piece_off; }
    LRSC_SM uint32_t out_cap() const { return sm_global(LRSC_A.work)[r].out_cap; }
    LRSC_SM uint8_t* walk_log() const { return LRSC_A.walk_log ? LRSC_A.walk_log + seed_slab(rs, r, LRSC_A.min_k) : nullptr; }
    LRSC_SM int64_t& ctr(int j) const { return sm_global(LRSC_A.out)[r].c[j]; }
    LRSC_SM const StrandC<P>& sF() const { return LRSC_SF; }
    LRSC_SM const StrandC<P>& sR() const { return LRSC_SR; }
    // the lane's extension slots are passed down from the kernel (an LDS pointer the compiler can see), not read back from the
    // state object: a pointer loaded from memory is generic, and generic (flat) accesses to LDS are slow and serialising
    static LRSC_SM P& exv(P* ex, uint32_t ex_stride, uint32_t b, uint32_t f) { return ex[(b * 4u + f) * ex_stride]; }

    LRSC_SM uint32_t seedSize() const { return LRSC_A.seed_size; }
    LRSC_SM uint32_t minOverlap() const { return LRSC_A.min_overlap; }

    // ---- request helpers ----------------------------------------------------------------------------------
    LRSC_SM void req_rank(P alo, P ahi, uint32_t ca, bool doa, P blo, P bhi, uint32_t cb, bool dob, bool swap)
    {
        req.kind = kReqRank;
        req.flags = (doa ? kReqDoA : 0u) | (dob ? kReqDoB : 0u) | (swap ? kReqSwap : 0u);
        req.a_lo = alo; req.a_hi = ahi; req.b_lo = blo; req.b_hi = bhi; req.ca = ca; req.cb = cb;
    }
    LRSC_SM void req_tab(uint32_t tab, uint32_t code) { req.kind = kReqTab; req.tab = tab; req.code = code; }

    // largest table with k <= max_k (-1: none)
    LRSC_SM int best_table(uint32_t max_k) const
    {
        int b = -1;
        if(LRSC_FM.ktab[0].k != 0 && LRSC_FM.ktab[0].k <= max_k) b = 0;
        if(LRSC_FM.ktab[1].k != 0 && LRSC_FM.ktab[1].k <= max_k) b = 1;
        if(LRSC_FM.ktab[2].k != 0 && LRSC_FM.ktab[2].k <= max_k) b = 2;
        if(LRSC_FM.ktab[3].k != 0 && LRSC_FM.ktab[3].k <= max_k) b = 3;
        if(LRSC_FM.ktab[4].k != 0 && LRSC_FM.ktab[4].k <= max_k) b = 4;
        return b;
    }
    LRSC_SM uint32_t table_k(int t) const
    {
        return t == 0 ? LRSC_FM.ktab[0].k : t == 1 ? LRSC_FM.ktab[1].k : t == 2 ? LRSC_FM.ktab[2].k : t == 3 ? LRSC_FM.ktab[3].k : LRSC_FM.ktab[4].k;
    }
    // characters [t0, t0 + k) (0 = oldest) of the suffix of length l of a packed path, first character in the high bits
    static LRSC_SM uint32_t suf_code(uint64_t lo, uint64_t hi, uint32_t l, uint32_t t0, uint32_t k)
    {
        uint32_t code = 0;
        for(uint32_t t = 0; t < k; ++t) {
            const uint32_t back = l - 1 - (t0 + t);
            const uint32_t c = back < 32 ? (uint32_t)(lo >> (2 * back)) & 3u : (uint32_t)(hi >> (2 * (back - 32))) & 3u;
            code = (code << 2) | c;
        }
        return code;
    }
    static LRSC_SM uint32_t suf_at(uint64_t lo, uint64_t hi, uint32_t l, uint32_t t)
    {
        const uint32_t back = l - 1 - t;
        return back < 32 ? (uint32_t)(lo >> (2 * back)) & 3u : (uint32_t)(hi >> (2 * (back - 32))) & 3u;
    }

    // =========================================================================================================
    // FS: findInterval of the suffix of length m_len of every leaf of a list (refineSAInterval .cpp:355-369,
    // initialRootNode :108-124); fwd = reverse(kmer) in the rBWT, rvc = revcomp(kmer) in the BWT.
    // =========================================================================================================
    LRSC_SM void fs_begin(uint32_t list, uint32_t n, uint32_t len, uint32_t ret, bool set_k)
    {
        m_list = list; m_n = n; m_len = len; m_ret = ret; m_j = 0; m_start = true; m_flag = set_k;
        pc = PC_FS;
        fs_advance();
    }
    LRSC_SM_NI void fs_advance()
    {
        while(true) {
            if(m_start) {
                if(m_j >= m_n) {
                    if(m_flag) currentKmerSize = m_len;
                    pc = m_ret;
                    return;
                }
                const LeafT& lf = leaves(m_list)[m_j];
                m_suf_lo = lf.suf_lo; m_suf_hi = lf.suf_hi;
                m_start = false;
                m_fb = false; m_rb = false;
                const int tb = best_table(m_len);
                if(tb >= 0) {
                    const uint32_t tk = table_k(tb);
                    m_t = tk;
                    req_tab((uint32_t)tb, suf_code(m_suf_lo, m_suf_hi, m_len, 0, tk));
                    m_t |= 0x80000000u;                         // marks "table entry pending"
                    return;
                }
                // no table: the first character is initInterval on both strands
                const uint32_t c = suf_at(m_suf_lo, m_suf_hi, m_len, 0);
                const IvT<P> f = init_interval<P>(sF(), c), rr = init_interval<P>(sR(), 3u - c);
                m_flo = f.lo; m_fhi = f.hi; m_rlo = rr.lo; m_rhi = rr.hi;
                m_t = 1;
                n_rank += 2;
            }
            if(m_t < m_len && !(m_fb && m_rb)) {
                const uint32_t c = suf_at(m_suf_lo, m_suf_hi, m_len, m_t);
                req_rank(m_flo, m_fhi, c, !m_fb, m_rlo, m_rhi, 3u - c, !m_rb, false);
                return;
            }
            LeafT& lf = leaves(m_list)[m_j];
            lf.flo = m_flo; lf.fhi = m_fhi; lf.rlo = m_rlo; lf.rhi = m_rhi;
            ++m_j;
            m_start = true;
        }
    }
    LRSC_SM void fs_result(const SmReq<P>& res)
    {
        if(m_t & 0x80000000u) {                                 // table entry
            m_t &= 0x7FFFFFFFu;
            m_flo = res.a_lo; m_fhi = res.a_hi; m_rlo = res.b_lo; m_rhi = res.b_hi;
            m_fb = m_flo > m_fhi; m_rb = m_rlo > m_rhi;
        } else {
            if(!m_fb) { m_flo = res.a_lo; m_fhi = res.a_hi; m_fb = m_flo > m_fhi; }
            if(!m_rb) { m_rlo = res.b_lo; m_rhi = res.b_hi; m_rb = m_rlo > m_rhi; }
            ++m_t;
        }
        fs_advance();
    }

    // =========================================================================================================
    // SF: SelectFreqsOfrange (.cpp:281-331).  Phase 1 = findInterval(BWT, startkmer) / findInterval(RBWT,
    // complement(startkmer)) per leaf, startkmer = the last LB characters of the leaf's suffix, searched from its last
    // character backwards: exactly the k-mer table state of w[t] = 3 - x_t (x_0 = newest character), fwd <-> rvc swapped.
    // =========================================================================================================
    LRSC_SM void sf_begin(uint32_t list, uint32_t n, uint64_t LB, uint64_t UB, uint32_t ret)
    {
        m_list = list; m_n = n; sf_LB = (uint32_t)LB; sf_UB = (uint32_t)UB; m_ret = ret;
        sf_phase = 1; m_j = 0; m_start = true; sf_max = 0;
        pc = PC_SF;
        sf_advance();
    }
    LRSC_SM void sf_store_leaf()
    {
        LeafT& lf = leaves(m_list)[m_j];
        lf.tflo = m_flo; lf.tfhi = m_fhi; lf.trlo = m_rlo; lf.trhi = m_rhi;
        lf.tmpFreq = (int)(isize(m_flo, m_fhi) + isize(m_rlo, m_rhi));
        if(lf.tmpFreq > sf_max) sf_max = lf.tmpFreq;
    }
    LRSC_SM void sf_finish(uint64_t result) { sf_result = result; pc = m_ret; }
    LRSC_SM_NI void sf_advance()
    {
        const uint32_t U = sf_UB, Lw = sf_LB;
        while(true) {
            if(sf_phase == 1) {
                if(m_start) {
                    if(m_j >= m_n) {
                        if(sf_max - (int)LRSC_A.freqs_of_kmer_size[sf_LB] < 5) { sf_finish(sf_LB); return; }
                        if(sf_UB == sf_LB) { sf_finish(sf_UB); return; }
                        sf_phase = 2; sf_i = 1; m_j = 0; sf_max = 0;
                        continue;
                    }
                    const LeafT& lf = leaves(m_list)[m_j];
                    m_suf_lo = lf.suf_lo; m_suf_hi = lf.suf_hi;
                    m_start = false; m_fb = false; m_rb = false;
                    const int tb = best_table(Lw);
                    if(tb >= 0) {
                        // w[t] = 3 - x_t, x_t = character at distance t from the newest one
                        const uint32_t tk = table_k(tb);
                        uint32_t code = 0;
                        for(uint32_t t = 0; t < tk; ++t) {
                            const uint32_t x = t < 32 ? (uint32_t)(m_suf_lo >> (2 * t)) & 3u : (uint32_t)(m_suf_hi >> (2 * (t - 32))) & 3u;
                            code = (code << 2) | (3u - x);
                        }
                        m_t = tk | 0x80000000u;
                        req_tab((uint32_t)tb, code);
                        return;
                    }
                    const uint32_t x0 = (uint32_t)m_suf_lo & 3u;
                    const IvT<P> f = init_interval<P>(sR(), x0), rr = init_interval<P>(sF(), 3u - x0);
                    m_flo = f.lo; m_fhi = f.hi; m_rlo = rr.lo; m_rhi = rr.hi;
                    m_t = 1;
                    n_rank += 2;
                }
                if(m_t < Lw && !(m_fb && m_rb)) {
                    const uint32_t t = m_t;
                    const uint32_t c = t < 32 ? (uint32_t)(m_suf_lo >> (2 * t)) & 3u : (uint32_t)(m_suf_hi >> (2 * (t - 32))) & 3u;
                    req_rank(m_flo, m_fhi, c, !m_fb, m_rlo, m_rhi, 3u - c, !m_rb, true);
                    return;
                }
                sf_store_leaf();
                ++m_j;
                m_start = true;
            } else {
                // phase 2: extend every leaf's pair by one more (older) character, no validity check (.cpp:317-318)
                if(m_j >= m_n) {
                    if(sf_max - (int)LRSC_A.freqs_of_kmer_size[sf_LB + sf_i] < 5) { sf_finish(sf_LB + sf_i); return; }
                    ++sf_i;
                    if(sf_i > sf_UB - sf_LB) { sf_finish(sf_UB); return; }
                    m_j = 0; sf_max = 0;
                    continue;
                }
                const LeafT& lf = leaves(m_list)[m_j];
                const uint32_t b = suf_char(lf, U, (uint32_t)(sf_UB - sf_LB - sf_i));
                m_flo = lf.tflo; m_fhi = lf.tfhi; m_rlo = lf.trlo; m_rhi = lf.trhi;
                req_rank(m_flo, m_fhi, b, true, m_rlo, m_rhi, 3u - b, true, true);
                return;
            }
        }
    }
    LRSC_SM void sf_result_in(const SmReq<P>& res)
    {
        if(sf_phase == 1) {
            if(m_t & 0x80000000u) {
                // table entry of w: fwd = rBWT state (our r), rvc = BWT state (our f)
                m_t &= 0x7FFFFFFFu;
                m_rlo = res.a_lo; m_rhi = res.a_hi; m_flo = res.b_lo; m_fhi = res.b_hi;
                m_fb = m_flo > m_fhi; m_rb = m_rlo > m_rhi;
            } else {
                if(!m_fb) { m_flo = res.a_lo; m_fhi = res.a_hi; m_fb = m_flo > m_fhi; }
                if(!m_rb) { m_rlo = res.b_lo; m_rhi = res.b_hi; m_rb = m_rlo > m_rhi; }
                ++m_t;
            }
        } else {
            m_flo = res.a_lo; m_fhi = res.a_hi; m_rlo = res.b_lo; m_rhi = res.b_hi;
            sf_store_leaf();
            ++m_j;
        }
        sf_advance();
    }

    // =========================================================================================================
    // rank-free pieces of the walk (restated from walk_device.h's Walk, which the per-walk kernels keep using)
    // =========================================================================================================
    LRSC_SM bool isInsufficientFreqs(const LeafT* lv, uint32_t n) const             // .cpp:334-352
    {
        uint64_t highfreqscount = 0;
        const uint64_t cov = LRSC_A.pb_coverage;
        for(uint32_t i = 0; i < n; ++i) {
            const int highfreqThreshold = cov > 60 ? (int)((uint64_t)(cov / 60) * 3) : 3;
            if(lv[i].kmerFrequency > highfreqThreshold) highfreqscount++;
        }
        if(highfreqscount == 0) return true;
        else if(highfreqscount <= 2 && n >= 5) return true;
        else if(highfreqscount <= 1 && n >= 3) return true;
        return false;
    }

    // ismatchedbykmer (.cpp:787-821) for the four extension bases at once: bit x of the result = some offset j of the query
    // inside the indel window carries the 5-mer (last four path characters + x) with a valid interval on a strand whose
    // extended interval is valid too.  The reference asks two interval trees; here the window of per-offset codes is read
    // in independent loads (no chain to follow).
    LRSC_SM_NI uint32_t matched_by_5mer(uint32_t suf8, uint32_t fvalid4, uint32_t rvalid4) const
    {
        const uint32_t startSeedIdx = (uint32_t)(((int)currentLength - (int)maxIndelSize) > 0 ? ((int)currentLength - (int)maxIndelSize) : 0);
        uint32_t largeSeedIdx = currentLength + maxIndelSize;
        const uint32_t n5 = Lq >= 5 ? Lq - 5 + 1 : 0;
        if(n5 == 0 || startSeedIdx >= n5) return 0;
        if(largeSeedIdx > n5 - 1) largeSeedIdx = n5 - 1;
        const uint16_t* cc = c5();
        const uint32_t pre = (suf8 & 0xFFu) << 2;                 // the 5-mer code without its last character
        uint32_t mask = 0;
        // whole 16-byte chunks (8 entries) that cover the window: one wide load each (the array starts 16-byte aligned + even)
        const uintptr_t base = reinterpret_cast<uintptr_t>(cc);
        const uintptr_t a0 = (base + 2u * startSeedIdx) & ~(uintptr_t)15, a1 = base + 2u * largeSeedIdx;
        for(uintptr_t a = a0; a <= a1; a += 64) {
            uint4 v[4];
#pragma unroll
            for(uint32_t u = 0; u < 4; ++u) v[u] = a + 16u * u <= a1 ? *reinterpret_cast<const uint4*>(a + 16u * u) : uint4{0, 0, 0, 0};
#pragma unroll
            for(uint32_t u = 0; u < 4; ++u) {
                const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for(uint32_t h = 0; h < 8; ++h) {
                    const uintptr_t ea = a + 16u * u + 2u * h;
                    const uint32_t e = (w[h >> 1] >> (16u * (h & 1u))) & 0xFFFFu;
                    if(ea < base + 2u * startSeedIdx || ea > a1) continue;
                    if(((e ^ pre) & 0x3FCu) != 0 || (e & 0xC00u) == 0) continue;
                    const uint32_t x = e & 3u;
                    const bool ok = (((fvalid4 >> x) & 1u) && (e & 0x400u)) || (((rvalid4 >> x) & 1u) && (e & 0x800u));
                    if(ok) mask |= 1u << x;
                }
            }
        }
        return mask;
    }

    // the acceptance ladder of getFMIndexExtensions (.cpp:700-784) over the four extension pairs in ex_*
    LRSC_SM_NI uint32_t ext_mask(const LeafT& lf, int* freq, P* ex, uint32_t ex_stride) const
    {
        const uint64_t IntervalSizeCutoff = min_SA_threshold;
        uint64_t totalcount = 0;
        int maxfreqsofleave = 0;
        for(uint32_t b = 0; b < 4; ++b) {
            freq[b] = (int)(isize(exv(ex, ex_stride, b, 0), exv(ex, ex_stride, b, 1)) + isize(exv(ex, ex_stride, b, 2), exv(ex, ex_stride, b, 3)));
            totalcount += (uint64_t)(int64_t)freq[b];
            if(freq[b] > maxfreqsofleave) maxfreqsofleave = freq[b];
        }
        uint32_t mask = 0;
        const bool isHomopolymer = lf.tailLetterCount >= 3;
        uint32_t fvalid4 = 0, rvalid4 = 0;
        for(uint32_t b = 0; b < 4; ++b) {
            if(exv(ex, ex_stride, b, 0) <= exv(ex, ex_stride, b, 1)) fvalid4 |= 1u << b;
            if(exv(ex, ex_stride, b, 2) <= exv(ex, ex_stride, b, 3)) rvalid4 |= 1u << b;
        }
        // only repeats (maxfreqsofleave > 50) look at the 5-mer test in the ladder below: skip the window scan otherwise
        const uint32_t matched4 = maxfreqsofleave > 50 ? matched_by_5mer((uint32_t)lf.suf_lo, fvalid4, rvalid4) : 0u;
        for(uint32_t b = 0; b < 4; ++b) {
            const uint64_t kmerFreq = (uint64_t)(int64_t)freq[b];
            const double kmerRatioNotPass = 2;
            double kmerRatioCutoff = 0;
            const double kmerRatio = (double)kmerFreq / (double)maxfreqsofleave;
            const bool isMatchedBy5mer = ((matched4 >> b) & 1u) != 0;
            const bool isFreqPass = kmerFreq >= IntervalSizeCutoff;
            const bool isLowCoverage = totalcount >= IntervalSizeCutoff + 2;
            const bool isRepeat = maxfreqsofleave > 100;
            const bool isHighlyRepeat = maxfreqsofleave > 150;
            const bool isLowlyRepeat = maxfreqsofleave > 50;
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

    LRSC_SM void free_leaf_slots(const LeafT& lf) { slot_free |= 1u << lf.ring; }
    LRSC_SM uint32_t alloc_slot()
    {
        uint32_t s = 0;
        while(s < 31 && !((slot_free >> s) & 1u)) ++s;          // a free slot always exists: <= 32 leaves share 32 slots
        slot_free &= ~(1u << s);
        return s;
    }

    LRSC_SM uint32_t kmer_code(uint32_t i) const
    {
        const uint8_t* qq = q();
        uint32_t c = 0;
        for(uint32_t t = 0; t < seedSize(); ++t) c = (c << 2) | qq[i + t];
        return c;
    }

    LRSC_SM_NI bool isSupportedByNewSeed(LeafT& nd, uint64_t smallSeedIdx, uint64_t largeSeedIdx)     // .cpp:566-635
    {
        const uint32_t seedSz = seedSize();
        const uint64_t seedIdxOffset = nd.lastOverlapLen < currentLength - seedSz ? (uint64_t)seedSz : currentLength - nd.lastOverlapLen;
        const uint64_t cand = nd.lastSeedIdx + seedIdxOffset;
        const uint64_t startSeedIdx = smallSeedIdx > cand ? smallSeedIdx : cand;
        bool isNewSeedFound = false;
        const bool fv = nd.flo <= nd.fhi, rv = nd.rlo <= nd.rhi;
        const uint32_t mask9 = seedSz >= 16 ? 0xFFFFFFFFu : ((1u << (2 * seedSz)) - 1u);
        const uint32_t code9 = (uint32_t)nd.suf_lo & mask9;
        const uint32_t hb = (code9 ^ (code9 >> 9)) & 255u;
        const SortItem* i9f = it9f(); const SortItem* i9r = it9r();
        const uint16_t* nf = next9f(); const uint16_t* nr = next9r();
        uint32_t jf = fv ? head9f()[hb] : 0xFFFFu;
        uint32_t jr = rv ? head9r()[hb] : 0xFFFFu;
        while(jf != 0xFFFFu && i9f[jf].pad != code9) jf = nf[jf];
        while(jr != 0xFFFFu && i9r[jr].pad != code9) jr = nr[jr];
        int minIdxDiff = 10000;
        const uint64_t currSeedIdx = currentLength - seedSz;
        while(jf != 0xFFFFu || jr != 0xFFFFu) {
            const uint64_t vf = jf != 0xFFFFu ? i9f[jf].val : 0, vr = jr != 0xFFFFu ? i9r[jr].val : 0;
            if(fv && jf != 0xFFFFu && vf >= startSeedIdx && vf <= largeSeedIdx) {
                const int d = abs((int)vf - (int)currSeedIdx);
                if(d < minIdxDiff) { nd.lastSeedIdx = vf; nd.queryOverlapLen = vf + seedSz; minIdxDiff = d; }
                nd.lastOverlapLen = currentLength;
                nd.currOverlapLen = currentLength;
                isNewSeedFound = true;
            } else if(rv && jr != 0xFFFFu && vr >= startSeedIdx && vr <= largeSeedIdx) {
                const int d = abs((int)currSeedIdx - (int)vr);
                if(d < minIdxDiff) { nd.lastSeedIdx = vr; nd.queryOverlapLen = vr + seedSz; minIdxDiff = d; }
                nd.lastOverlapLen = currentLength;
                nd.currOverlapLen = currentLength;
                isNewSeedFound = true;
            }
            if(jf != 0xFFFFu) { jf = nf[jf]; while(jf != 0xFFFFu && i9f[jf].pad != code9) jf = nf[jf]; }
            if(jr != 0xFFFFu) { jr = nr[jr]; while(jr != 0xFFFFu && i9r[jr].pad != code9) jr = nr[jr]; }
        }
        if(isNewSeedFound) nd.totalSeeds++;
        return isNewSeedFound;
    }

    LRSC_SM double computeErrorRate(LeafT& nd, const double* parent_ring) const                  // .cpp:638-664
    {
        const uint64_t localK = 100;
        double matchedLen = (double)nd.totalSeeds + seedSize() - 1;
        matchedLen += nd.numRedeemSeed;
        const double totalLen = (double)nd.currOverlapLen;
        const double unmatchedLen = totalLen - matchedLen;
        double currErrorRate = unmatchedLen / totalLen;
        nd.globalErr = currErrorRate;
        const uint32_t totalsize = nd.hist_size + 1;
        nd.hist_size = totalsize;
        if(totalsize >= localK) {
            const double old = parent_ring[(totalsize - localK) % 100];
            currErrorRate = (currErrorRate * totalLen - old * (totalLen - localK)) / localK;
        }
        nd.localErr = currErrorRate;
        return currErrorRate;
    }

    // PrunedBySeedSupport (.cpp:491-563) for one child held in registers.  A child still carries its parent's ring slot id
    // (createChild copies it; the commit assigns the final one), so the parent's error history is found without reading the parent.
    LRSC_SM void prune_child(LeafT& leaf)
    {
        const uint32_t seedSz = seedSize();
        const double PacBioErrorRate = LRSC_A.pacbio_error_rate;
        const uint64_t currSeedIdx = currentLength - seedSz;
        const uint64_t indelOffset = seedSz + maxIndelSize;
        const uint64_t smallSeedIdx = currSeedIdx <= indelOffset ? 0 : currSeedIdx - indelOffset;
        const uint64_t largeSeedIdx = (currSeedIdx + indelOffset) >= (Lq - seedSz) ? (Lq - seedSz) : currSeedIdx + indelOffset;
        bool isNewSeedFound = false;
        if(currentLength - leaf.lastOverlapLen > seedSz || currentLength - leaf.lastOverlapLen <= 1) {
            const uint64_t preSeedIdx = leaf.lastSeedIdx;
            isNewSeedFound = isSupportedByNewSeed(leaf, smallSeedIdx, largeSeedIdx);
            if(isNewSeedFound) {
                if(currSeedIdx + (uint64_t)(int64_t)leaf.lastSeedIdxOffset - preSeedIdx > seedSz)
                    leaf.numRedeemSeed += (seedSz - 1) * PacBioErrorRate;
                leaf.lastSeedIdxOffset = (int)leaf.lastSeedIdx - (int)currSeedIdx;
            } else {
                const uint64_t v = currSeedIdx + (uint64_t)(int64_t)leaf.lastSeedIdxOffset - leaf.lastSeedIdx;
                if(v % seedSz == 1) leaf.numOfErrors++;
                else if(v > (uint64_t)seedSz - 1) leaf.numRedeemSeed += 1 - PacBioErrorRate;
            }
        } else
            leaf.numRedeemSeed += 1 - PacBioErrorRate;
        const double* pring = rings() + (uint64_t)leaf.ring * 100;
        const double currErrorRate = computeErrorRate(leaf, pring);
        if(currErrorRate > 0.25) leaf.alive = 0;
    }

    LRSC_SM_NI void terminated_leaf(LeafT& lf, const uint32_t* pw, uint32_t plen, int extra)         // .cpp:825-878
    {
        const bool fvalid = lf.flo <= lf.fhi, rvalid = lf.rlo <= lf.rhi;
        const uint64_t i0 = (uint64_t)(lf.res_second > 0 ? lf.res_second : 0);
        int hit = -1;
        const P* tm = term();
        for(uint64_t i = i0; i <= (uint64_t)trg_len - (int)minOverlap(); i++) {
            const P* t = tm + i * 4;
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
        WalkResultRec& rr = results()[lf.res_first - 1];
        rr.error_rate = lf.globalErr;
        rr.match_i = (uint32_t)hit;
        uint32_t* dst = rpaths() + (uint64_t)(lf.res_first - 1) * pathw;
        const uint32_t nw = (plen + 15) >> 4;
        for(uint32_t kk = 0; kk < nw; ++kk) dst[kk] = pw[kk];
        uint32_t len = plen;
        if(extra >= 0) { path_set(dst, len, (uint32_t)extra); ++len; }
        rr.path_len = len;
    }

    // interval "trees" (IntervalTree.cpp:4-48 -> k-mer chains in std::sort's order) + the root's bookkeeping
    LRSC_SM_NI void begin_walk()
    {
        const uint32_t seedSz = seedSize();
        const uint8_t* qq = q();
        const uint32_t n9_all = Lq >= seedSz ? Lq - seedSz + 1 : 0;
        for(int strand = 0; strand < 2; ++strand) {
            SortItem* itx = strand ? it9r() : it9f();
            uint16_t* head = strand ? head9r() : head9f();
            uint16_t* nextp = strand ? next9r() : next9f();
            uint32_t n = 0;
            for(uint32_t i = 0; i < n9_all; ++i)
                if(itx[i].key != kNoKey) { if(n != i) itx[n] = itx[i]; ++n; }
            introsort(itx, (int64_t)n);
            for(uint32_t b = 0; b < 256; ++b) head[b] = 0xFFFFu;
            for(uint32_t j = n; j-- > 0;) {                       // prepend walking backwards: chains keep the post-sort order
                const uint32_t code = itx[j].pad;                  // the idmer's 2-bit code, stored by PREP
                const uint32_t hb = (code ^ (code >> 9)) & 255u;
                nextp[j] = head[hb];
                head[hb] = (uint16_t)j;
            }
        }
        // root (initialRootNode, .cpp:108-124; leafInfo ctor, .h:156-171)
        slot_free = 0xFFFFFFFEu;
        LeafT& root = cur()[0];
        root.suf_lo = 0; root.suf_hi = 0;
        for(uint32_t t = 0; t < initk; ++t) suf_push(root, qq[t]);
        root.lastOverlapLen = root.currOverlapLen = root.queryOverlapLen = initk;
        currentLength = currentKmerSize = initk;
        root.lastSeedIdx = (uint64_t)initk - seedSz;
        root.totalSeeds = (uint64_t)initk - seedSz + 1;
        root.numOfErrors = 0;
        root.numRedeemSeed = 0;
        root.localErr = 0; root.globalErr = 0; root.hist_size = 1;
        root.lastSeedIdxOffset = 0;
        root.res_first = -1; root.res_second = -1;
        root.tailLetter = qq[initk - 1];
        root.tailLetterCount = 0;
        for(uint32_t t = initk; t-- > 0;) { if(qq[t] == root.tailLetter) root.tailLetterCount++; else break; }
        root.ring = 0; root.path = 0; root.parent = 0; root.ext = 0; root.alive = 1;
        root.path_len = initk;
        rings()[0] = 0.0;
        uint32_t* p0 = paths();
        for(uint32_t t = 0; t < initk; ++t) path_set(p0, t, qq[t]);
        n_cur = 1; n_nxt = 0; n_results = 0;
        ended = false;
        fs_begin(0, 1, initk, PC_ROOT_DONE, false);
    }

    // attempToExtend's prologue (.cpp:373-398): drop leaves whose local error rate is far from the best one
    LRSC_SM_NI void att_entry()
    {
        const uint64_t localK = 100;
        LeafT* cu = cur();
        double minimumErrorRate = 1;
        for(uint32_t i = 0; i < n_cur; ++i)
            if(cu[i].localErr < minimumErrorRate) minimumErrorRate = cu[i].localErr;
        uint32_t w = 0;
        for(uint32_t i = 0; i < n_cur; ++i) {
            const double errorRateDiff = cu[i].localErr - minimumErrorRate;
            if((errorRateDiff > 0.05 && currentLength > localK / 2) || (errorRateDiff > 0.1 && currentLength > 15)) {
                free_leaf_slots(cu[i]);
                continue;
            }
            if(w != i) cu[w] = cu[i];
            ++w;
        }
        n_cur = w;
        att_minErr = minimumErrorRate;
        att_i = 0;
        pc = PC_ATT_LEAF;
    }
    // the one request of leaf att_i's getFMIndexExtensions (or the end of attempToExtend)
    LRSC_SM void att_leaf()
    {
        if(att_i >= n_cur) { pc = PC_ATT_DONE; return; }
        const LeafT& lf = cur()[att_i];
        const bool fv = lf.flo <= lf.fhi, rv = lf.rlo <= lf.rhi;
        req.kind = kReqExt;
        req.flags = (fv ? kReqDoA : 0u) | (rv ? kReqDoB : 0u);
        req.a_lo = lf.flo; req.a_hi = lf.fhi; req.b_lo = lf.rlo; req.b_hi = lf.rhi;
        pc = PC_EXT;
    }
    // all four extension pairs of leaf att_i are in ex: the acceptance ladder, at most twice (second time with the threshold
    // lowered by one, only for the best leaf of a multi-leaf frontier: .cpp:403-421), then updateLeaves (:468-488)
    LRSC_SM_NI void ext_eval(P* ex, uint32_t ex_stride)
    {
#if defined(__HIP_DEVICE_COMPILE__)
#define LRSC_SM_T2() (tkp ? (uint64_t)__builtin_readcyclecounter() : 0ull)
#else
#define LRSC_SM_T2() 0ull
#endif
        uint64_t t_a = LRSC_SM_T2();
        LeafT par = cur()[att_i];                                  // one struct load; everything below works on registers
        if(tkp) { const uint64_t t_b = LRSC_SM_T2(); tkp[9] += t_b - t_a; t_a = t_b; }
        int freq[4];
        uint32_t mask = 0;
        int count = 0;
        while(count < 2) {
            if(count == 1 && !(par.localErr == att_minErr && n_cur > 1)) break;
            mask = ext_mask(par, freq, ex, ex_stride);
            if(mask != 0) break;
            min_SA_threshold--;
            count++;
        }
        min_SA_threshold += (uint32_t)count;
        if(tkp) { const uint64_t t_b = LRSC_SM_T2(); tkp[10] += t_b - t_a; t_a = t_b; }
        // updateLeaves: children in base order.  The loop runs over the lane's accepted bases (one, mostly), not over the four
        // bases: every iteration is a whole-struct store for the wavefront, whoever takes part
        LeafT* nx = nxt();
        for(uint32_t rest = mask; rest != 0; rest &= rest - 1u) {
            uint32_t bb = 0;
            while(!((rest >> bb) & 1u)) ++bb;
            if(n_nxt >= kMaxChildren) { error = LRSC_WALK_ERR_CHILDREN; pc = PC_WALK_END; return; }
            LeafT ch = par;                                        // createChild copies the node state (SAINode.cpp:166-189)
            ch.flo = exv(ex, ex_stride, bb, 0); ch.fhi = exv(ex, ex_stride, bb, 1); ch.rlo = exv(ex, ex_stride, bb, 2); ch.rhi = exv(ex, ex_stride, bb, 3);
            ch.kmerFrequency = bb == 0 ? freq[0] : bb == 1 ? freq[1] : bb == 2 ? freq[2] : freq[3];
            ch.currOverlapLen++;
            ch.queryOverlapLen++;
            if(par.tailLetter == bb) ch.tailLetterCount = par.tailLetterCount + 1;
            else { ch.tailLetter = bb; ch.tailLetterCount = 1; }
            suf_push(ch, bb);
            ch.parent = (uint16_t)att_i;
            ch.ext = (uint8_t)bb;
            ch.alive = 1;
            nx[n_nxt++] = ch;
        }
        if(tkp) { const uint64_t t_b = LRSC_SM_T2(); tkp[11] += t_b - t_a; t_a = t_b; }
        ++att_i;
        pc = att_i >= n_cur ? PC_ATT_DONE : PC_ATT_LEAF;          // ATT_DONE's block comes later in this same sweep
    }

    // extendLeaves' control flow after an attempToExtend (.cpp:239-278)
    LRSC_SM_NI void att_done()
    {
        if(n_nxt == 0) {
            if(att_no == 1) {                                   // level 1: reduce the k-mer size
                const uint64_t LowerBound = (currentKmerSize - 2) > minOverlap() ? (currentKmerSize - 2) : minOverlap();
                sf_begin(0, n_cur, LowerBound, currentKmerSize, PC_AFTER_SF_A);
                return;
            }
            if(att_no == 2) {                                   // level 2: reduce the threshold
                min_SA_threshold--;
                att_no = 3;
                pc = PC_ATT_ENTRY;
                return;
            }
        }
        if(att_no == 3) min_SA_threshold++;
        pc = PC_POST;
    }
    LRSC_SM_NI void post()
    {
        if(n_nxt != 0) {
            currentLength++;
            currentKmerSize++;
            if(isInsufficientFreqs(nxt(), n_nxt)) {
                const uint64_t LowerBound = (currentKmerSize - 2) > minOverlap() ? (currentKmerSize - 2) : minOverlap();
                sf_begin(1, n_nxt, LowerBound, currentKmerSize, PC_AFTER_SF_B);
                return;
            }
        }
        pc = PC_PRUNE;
    }

    // PrunedBySeedSupport + the rest of one extendOverlap iteration (.cpp:155-211)
    // the common shape (one leaf, one accepted base) is committed at the step gate; anything wider waits for the slow gate: the
    // general commit is several times the instructions, and a wavefront pays a block's instructions however few lanes run it
    LRSC_SM bool prune_is_simple() const { return n_nxt == 1 && n_cur == 1 && LRSC_A.max_leaves >= 1; }
    LRSC_SM_NI void prune_and_commit()
    {
        LeafT* nx = nxt(); LeafT* cu = cur();
        if(prune_is_simple()) {
            // the common shape (one leaf, one accepted base): the child stays in registers from the pruning to the commit
            LeafT ch = nx[0];
            prune_child(ch);
            ++steps;
            if(!ch.alive) {
                free_leaf_slots(cu[0]);
                n_cur = 0;
                pc = PC_STEP_ENTRY;
                return;
            }
            // it takes over the parent's ring and path in place (SAINode::extend)
            rings()[(uint64_t)ch.ring * 100 + (ch.hist_size - 1) % 100] = ch.globalErr;     // GlobalErrorRateRecord.push_back
            uint32_t* pw = paths() + (uint64_t)ch.path * pathw;
            path_set(pw, ch.path_len, ch.ext);
            ch.path_len++;
            if(currentLength >= minLength) {
                terminated_leaf(ch, pw, ch.path_len, -1);
                if(error) { pc = PC_WALK_END; return; }
            }
            cu[0] = ch;
            n_cur = 1;
            pc = PC_STEP_ENTRY;
            return;
        }
        for(uint32_t c = 0; c < n_nxt; ++c) {
            LeafT leaf = nx[c];
            prune_child(leaf);
            nx[c] = leaf;
        }
        const uint32_t pathw = this->pathw;
        uint32_t survivors = 0;
        for(uint32_t c = 0; c < n_nxt; ++c) survivors += nx[c].alive;
        ++steps;
        if(survivors > LRSC_A.max_leaves) {
            if(currentLength >= minLength)
                for(uint32_t c = 0; c < n_nxt; ++c) {
                    if(!nx[c].alive) continue;
                    const LeafT& par = cu[nx[c].parent];
                    terminated_leaf(nx[c], paths() + (uint64_t)par.path * pathw, par.path_len, (int)nx[c].ext);
                    if(error) { pc = PC_WALK_END; return; }
                }
            n_cur = survivors;
            ended = true;
            pc = PC_STEP_ENTRY;
            return;
        }
        uint32_t has_child = 0;
        for(uint32_t c = 0; c < n_nxt; ++c) if(nx[c].alive) has_child |= 1u << nx[c].parent;
        for(uint32_t i = 0; i < n_cur; ++i) if(!((has_child >> i) & 1u)) free_leaf_slots(cu[i]);
        // copies first (they read the parent's buffers before the in-place child appends to them)
        uint32_t seen = 0;
        for(uint32_t c = 0; c < n_nxt; ++c) {
            LeafT& ch = nx[c];
            if(!ch.alive) continue;
            const LeafT& par = cu[ch.parent];
            if(!((seen >> ch.parent) & 1u)) { seen |= 1u << ch.parent; ch.ring = par.ring; ch.path = par.path; continue; }
            const uint32_t s = alloc_slot();
            ch.ring = (uint16_t)s; ch.path = (uint16_t)s;
            const double* src = rings() + (uint64_t)par.ring * 100;
            double* dst = rings() + (uint64_t)ch.ring * 100;
            for(uint32_t kk = 0; kk < 100; ++kk) dst[kk] = src[kk];
            const uint32_t* ps = paths() + (uint64_t)par.path * pathw;
            uint32_t* pd = paths() + (uint64_t)ch.path * pathw;
            const uint32_t nw = (par.path_len + 16) >> 4;
            for(uint32_t kk = 0; kk < nw; ++kk) pd[kk] = ps[kk];
        }
        uint32_t w = 0;
        for(uint32_t c = 0; c < n_nxt; ++c) {
            LeafT& ch = nx[c];
            if(!ch.alive) continue;
            rings()[(uint64_t)ch.ring * 100 + (ch.hist_size - 1) % 100] = ch.globalErr;     // GlobalErrorRateRecord.push_back
            path_set(paths() + (uint64_t)ch.path * pathw, ch.path_len, ch.ext);
            ch.path_len++;
            cu[w++] = ch;                                          // m_leaves = newLeaves (w <= 32 here)
        }
        n_cur = w;
        if(currentLength >= minLength)
            for(uint32_t i = 0; i < n_cur; ++i) {
                terminated_leaf(cu[i], paths() + (uint64_t)cu[i].path * pathw, cu[i].path_len, -1);
                if(error) { pc = PC_WALK_END; return; }
            }
        pc = PC_STEP_ENTRY;
    }

    // loop condition of extendOverlap + the head of extendLeaves
    LRSC_SM_NI void step_entry()
    {
        if(ended || error || !(n_cur != 0 && n_cur <= LRSC_A.max_leaves && currentLength <= maxLength)) { pc = PC_WALK_END; return; }
        n_nxt = 0;
        att_no = 1;
        if(currentKmerSize > maxOverlap) { fs_begin(0, n_cur, maxOverlap, PC_ATT_ENTRY, true); return; }
        pc = PC_ATT_ENTRY;
    }

    LRSC_SM_NI int finish_walk(uint32_t* out_len_, uint32_t* out_words, uint32_t* out_match_i)      // findTheBestPath (.cpp:214-236)
    {
        if(error) return error;
        if(n_results > 0) {
            const WalkResultRec* rs = results();
            double minErrorRate = 1;
            int bestI = -1;
            for(uint32_t i = 0; i < n_results; ++i)
                if(rs[i].error_rate < minErrorRate) { minErrorRate = rs[i].error_rate; bestI = (int)i; }
            if(bestI < 0) return -4;
            *out_len_ = rs[bestI].path_len;
            *out_match_i = rs[bestI].match_i;
            const uint32_t* src = rpaths() + (uint64_t)bestI * pathw;
            const uint32_t nw = (rs[bestI].path_len + 15) >> 4;
            for(uint32_t kk = 0; kk < nw; ++kk) out_words[kk] = src[kk];
            return 1;
        }
        if(n_cur == 0) return -1;
        else if(currentLength > maxLength) return -2;
        else if(n_cur > LRSC_A.max_leaves) return -3;
        return -4;
    }

    // =========================================================================================================
    // PREP: the constructor's per-offset searches (.cpp:82-94,127-152): bi-intervals of the 5-mer, the idmer and,
    // inside the target seed, the minOverlap-mer starting at every offset of m_query
    // =========================================================================================================
    LRSC_SM_NI void prep_emit()
    {
        const uint32_t i = m_j, s = m_t;
        const bool fval = m_flo <= m_fhi, rval = m_rlo <= m_rhi;
        if(s == 5) {
            const uint8_t* qq = q();
            uint32_t code = 0;
            for(uint32_t t = 0; t < 5; ++t) code = (code << 2) | qq[i + t];
            c5()[i] = (uint16_t)(code | (fval ? 0x400u : 0u) | (rval ? 0x800u : 0u));
        }
        if(s == seedSize()) {
            SortItem* a = it9f() + i; SortItem* b = it9r() + i;
            const uint32_t code = kmer_code(i);
            a->key = fval ? (uint64_t)m_flo : kNoKey; a->val = i; a->pad = code;
            b->key = rval ? (uint64_t)m_rlo : kNoKey; b->val = i; b->pad = code;
        }
        if(s == minOverlap() && m_flag) {
            P* t = term() + (uint64_t)(i - (uint32_t)(k + interval)) * 4;
            t[0] = m_flo; t[1] = m_fhi; t[2] = m_rlo; t[3] = m_rhi;
        }
    }
    // table index holding exactly k-mers of size kk (-1: none)
    LRSC_SM int table_of(uint32_t kk) const
    {
        const int t = best_table(kk);
        return (t >= 0 && table_k(t) == kk) ? t : -1;
    }
    // one entry of table t: {fwd.lo, fwd.hi, rvc.lo, rvc.hi}
    LRSC_SM void table_entry(int t, uint32_t code, P e[4]) const
    {
        const void* tabv = t == 0 ? LRSC_FM.ktab[0].entries : t == 1 ? LRSC_FM.ktab[1].entries : t == 2 ? LRSC_FM.ktab[2].entries
                         : t == 3 ? LRSC_FM.ktab[3].entries : LRSC_FM.ktab[4].entries;
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
    // PREP when tables of exactly the three sizes exist (the normal configuration: 5, idmer = 9, minOverlap = 13): every emit
    // of an offset is one table entry, nothing chains, so kPrepBatch offsets are answered per sweep with all look-ups in flight.
    static constexpr uint32_t kPrepBatch = 16;
    LRSC_SM_NI bool prep_fast()
    {
        const uint32_t seedk = seedSize(), mink = minOverlap();
        if(seedk <= 5 || mink <= seedk || mink > 16) return false;
        const int t5 = table_of(5), t9 = table_of(seedk), t13 = table_of(mink);
        if(t5 < 0 || t9 < 0 || t13 < 0) return false;
        const uint8_t* qq = q();
        const uint32_t trg0 = (uint32_t)(k + interval);
        const uint32_t i0 = m_j;
        // the characters the batch's k-mers cover, fetched once (independent byte loads), as one 2-bit string
        uint64_t bits = 0;                                        // character i0 + t at bits [2t, 2t + 2)
        {
            const uint32_t nch = i0 + kPrepBatch + mink - 1 <= Lq ? kPrepBatch + mink - 1 : (Lq > i0 ? Lq - i0 : 0u);
            uint8_t ch[kPrepBatch + 15];
#pragma unroll
            for(uint32_t t = 0; t < kPrepBatch + 15; ++t) ch[t] = t < nch ? qq[i0 + t] : (uint8_t)0;
#pragma unroll
            for(uint32_t t = 0; t < kPrepBatch + 15; ++t) bits |= (uint64_t)(ch[t] & 3u) << (2 * t);
        }
        // codes of the kk-mers at offsets i0 .. i0 + 15 (first character in the high bits = the table index), rolling
        uint32_t c5s[kPrepBatch], c9s[kPrepBatch], c13s[kPrepBatch];
        {
            uint32_t c5v = 0, c9v = 0, c13v = 0;
            const uint32_t m5 = (1u << 10) - 1u, m9 = (1u << (2 * seedk)) - 1u, m13 = mink >= 16 ? 0xFFFFFFFFu : (1u << (2 * mink)) - 1u;
            for(uint32_t t = 0; t + 1 < 5; ++t) c5v = (c5v << 2) | (uint32_t)((bits >> (2 * t)) & 3u);
            for(uint32_t t = 0; t + 1 < seedk; ++t) c9v = (c9v << 2) | (uint32_t)((bits >> (2 * t)) & 3u);
            for(uint32_t t = 0; t + 1 < mink; ++t) c13v = (c13v << 2) | (uint32_t)((bits >> (2 * t)) & 3u);
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u) {
                c5v = ((c5v << 2) | (uint32_t)((bits >> (2 * (u + 4))) & 3u)) & m5;
                c9v = ((c9v << 2) | (uint32_t)((bits >> (2 * (u + seedk - 1))) & 3u)) & m9;
                c13v = ((c13v << 2) | (uint32_t)((bits >> (2 * (u + mink - 1))) & 3u)) & m13;
                c5s[u] = c5v; c9s[u] = c9v; c13s[u] = c13v;
            }
        }
        // pass 1: 5-mers -> per-offset code + strand-validity flags
        {
            P e[kPrepBatch][4];
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u) if(i0 + u + 5 <= Lq) table_entry(t5, c5s[u], e[u]);
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u)
                if(i0 + u + 5 <= Lq) { c5()[i0 + u] = (uint16_t)(c5s[u] | (e[u][0] <= e[u][1] ? 0x400u : 0u) | (e[u][2] <= e[u][3] ? 0x800u : 0u)); n_tab += 1; }
        }
        // pass 2: idmers -> the two sort arrays
        {
            P e[kPrepBatch][4];
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u) if(i0 + u + seedk <= Lq) table_entry(t9, c9s[u], e[u]);
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u)
                if(i0 + u + seedk <= Lq) {
                    const uint32_t i = i0 + u, code = c9s[u];
                    SortItem* a = it9f() + i; SortItem* b = it9r() + i;
                    a->key = e[u][0] <= e[u][1] ? (uint64_t)e[u][0] : kNoKey; a->val = i; a->pad = code;
                    b->key = e[u][2] <= e[u][3] ? (uint64_t)e[u][2] : kNoKey; b->val = i; b->pad = code;
                    n_tab += 1;
                }
        }
        // pass 3: minOverlap-mers inside the target seed -> terminal intervals
        if(i0 + kPrepBatch > trg0) {
            P e[kPrepBatch][4];
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u) if(i0 + u >= trg0 && i0 + u + mink <= Lq) table_entry(t13, c13s[u], e[u]);
#pragma unroll
            for(uint32_t u = 0; u < kPrepBatch; ++u)
                if(i0 + u >= trg0 && i0 + u + mink <= Lq) {
                    P* t = term() + (uint64_t)(i0 + u - trg0) * 4;
                    t[0] = e[u][0]; t[1] = e[u][1]; t[2] = e[u][2]; t[3] = e[u][3];
                    n_tab += 1;
                }
        }
        m_j += kPrepBatch;
        if(m_j >= Lq) pc = PC_BEGIN;
        return true;
    }
    LRSC_SM_NI void prep_advance()
    {
        if(prep_fast()) return;
        const uint32_t seedk = seedSize(), mink = minOverlap();
        const uint8_t* qq = q();
        const uint32_t trg0 = (uint32_t)(k + interval);
        while(true) {
            if(m_start) {
                if(m_j >= Lq) { pc = PC_BEGIN; return; }
                const uint32_t i = m_j;
                m_flag = i >= trg0 && i + mink <= Lq;
                uint32_t kmax = 0;
                if(i + 5 <= Lq) kmax = 5;
                if(i + seedk <= Lq) kmax = seedk;
                if(m_flag) kmax = mink > kmax ? mink : kmax;
                m_len = kmax;
                m_t = 0; m_fb = false; m_rb = false;
                m_flo = m_fhi = m_rlo = m_rhi = 0;
                m_start = false;
            }
            if(m_t >= m_len) { ++m_j; m_start = true; continue; }
            const uint32_t s = m_t;
            const uint32_t next_emit = s < 5 ? 5u : s < seedk ? seedk : mink;
            if(next_emit <= m_len) {
                const int tb = best_table(next_emit);
                if(tb >= 0 && table_k(tb) == next_emit) {
                    uint32_t code = 0;
                    for(uint32_t t = 0; t < next_emit; ++t) code = (code << 2) | qq[m_j + t];
                    m_t = next_emit | 0x80000000u;
                    req_tab((uint32_t)tb, code);
                    return;
                }
            }
            const uint32_t c = qq[m_j + s];
            if(s == 0) {
                const IvT<P> f = init_interval<P>(sF(), c), rr = init_interval<P>(sR(), 3u - c);
                m_flo = f.lo; m_fhi = f.hi; m_rlo = rr.lo; m_rhi = rr.hi;
                m_t = 1;
                n_rank += 2;
                prep_emit();
                continue;
            }
            if(m_fb && m_rb) { ++m_t; prep_emit(); continue; }     // both strands dead: nothing can change any more
            req_rank(m_flo, m_fhi, c, !m_fb, m_rlo, m_rhi, 3u - c, !m_rb, false);
            return;
        }
    }
    LRSC_SM void prep_result(const SmReq<P>& res)
    {
        if(m_t & 0x80000000u) {
            m_t &= 0x7FFFFFFFu;
            m_flo = res.a_lo; m_fhi = res.a_hi; m_rlo = res.b_lo; m_rhi = res.b_hi;
            m_fb = m_flo > m_fhi; m_rb = m_rlo > m_rhi;
        } else {
            if(!m_fb) { m_flo = res.a_lo; m_fhi = res.a_hi; m_fb = m_flo > m_fhi; }
            if(!m_rb) { m_rlo = res.b_lo; m_rhi = res.b_hi; m_rb = m_rlo > m_rhi; }
            ++m_t;
        }
        prep_emit();
        prep_advance();
    }

    // =========================================================================================================
    // the chain: PacBioSelfCorrectionProcess::initCorrect (:56-157) / correctByFMExtension (:159-206)
    // =========================================================================================================
    LRSC_SM void load_source(const int32_t* T0)
    {
        S_end = T0[0] + T0[1] - 1; S_endBest = T0[5]; S_isRepeat = (T0[3] & 1) != 0; S_maxFixed = T0[2];
    }

    // set up at kernel start (fresh read, or a read that yielded / was parked in an earlier launch)
    LRSC_SM_NI void init(const FmIndexDev* fm_, const CorrectArgs* a_, const StrandC<P>* sF_, const StrandC<P>* sR_, uint32_t read_index)
    {
        fm = fm_; A = a_; sFp = sF_; sRp = sR_; r = read_index;
        const ReadWork& rw = LRSC_A.work[r];
        ReadOut& R = LRSC_A.out[r];
        rs = LRSC_A.read_off[r];
        n_seeds = LRSC_A.seed_count[r];
        lq_max = rw.lq_max; pathw = rw.pathw;
        ws = LRSC_A.workspace + rw.ws_off;
        const uint8_t* read = this->read();
        const int32_t* seeds = this->seeds();
        uint8_t* out = this->out();
        uint32_t* piece_start = this->piece_start();
        req.kind = kReqNone;
        n_rank = 0; n_blk = 0; n_tab = 0;
        error = 0; state = kReadDone; walks_here = 0; next = 0; firstType = 0;
        out_len = 0; n_pieces = 0; steps = 0;
        S_seedLen = 0; S_end = 0; S_endBest = 0; S_maxFixed = 0; S_isRepeat = false; it = 1;
        const bool resume = LRSC_A.resume != 0;
        if(resume) {
            out_len = R.out_len; n_pieces = R.n_pieces; steps = (uint32_t)R.steps;
        } else {
            for(int j = 0; j < 10; ++j) R.c[j] = 0;
            R.cyc[0] = 0; R.cyc[1] = 0; R.cyc[2] = 0; R.cyc[3] = 0; R.steps = 0;
        }
        steps0 = steps;
        if(!(n_seeds >= 2) || lq_max == 0 || (resume && R.state == kReadDone)) { pc = PC_FINAL; return; }     // lq_max == 0: skipped by the host (capacity)
        if(!resume) {
            // pieceVec.push_back(seedVec[0])
            piece_start[n_pieces++] = 0;
            copy_codes(out + out_len, read + seeds[0], (uint32_t)seeds[1]); out_len += (uint32_t)seeds[1];
            S_seedLen = seeds[1];
            load_source(seeds);
            it = 1;
        } else {
            S_seedLen = R.s_seed_len; S_end = R.s_end; S_endBest = R.s_end_best; S_maxFixed = R.s_max_fixed; S_isRepeat = R.s_is_repeat != 0;
            it = R.it;
            if(R.state == kReadParked) {
                // correctByMSAlignment's tail (:237-244) with the DP stage's answer for target = *iterTarget
                const uint32_t di = LRSC_A.dp_index[r];
                const DpMsaOut m = LRSC_A.dp_msa[di];
                const int32_t* T0 = seeds + (uint64_t)it * kSeedInts;
                if(m.error) error = LRSC_WALK_ERR_DP;
                else if(m.n_rows > 3) {
                    const uint8_t* cons = LRSC_A.dp_cons + LRSC_A.dp_reqs[di].cons_off;
                    if(m.cons_len < R.dp_k) error = LRSC_WALK_ERR_DP;              // out.erase(0, k) would throw in the reference
                    else {
                        const uint32_t appended = m.cons_len - R.dp_k;
                        if(out_len + appended > out_cap()) error = LRSC_WALK_ERR_OUTPUT;
                        else {
                            copy_codes(out + out_len, cons + R.dp_k, appended);
                            out_len += appended;
                            ctr(1) += appended;
                            ctr(9) += T0[0] - S_end - 1;
                            ctr(8)++;
                            S_seedLen += (int)appended;
                        }
                    }
                } else if(LRSC_A.split) {
                    if(uint8_t* wl = walk_log()) wl[it] |= 0x10;
                    if(out_len + (uint32_t)T0[1] > out_cap()) error = LRSC_WALK_ERR_OUTPUT;
                    else {
                        piece_start[n_pieces++] = out_len;
                        { copy_codes(out + out_len, read + T0[0], (uint32_t)T0[1]); out_len += (uint32_t)T0[1]; }
                        S_seedLen = T0[1];
                        ctr(1) += T0[1];
                    }
                } else {
                    if(uint8_t* wl = walk_log()) wl[it] |= 0x10;
                    const int raw = (T0[0] + T0[1] - 1) - S_end;
                    if(out_len + (uint32_t)raw > out_cap()) error = LRSC_WALK_ERR_OUTPUT;
                    else {
                        { copy_codes(out + out_len, read + S_end + 1, (uint32_t)raw); out_len += (uint32_t)raw; }
                        S_seedLen += raw;
                        ctr(1) += T0[1];
                    }
                }
                load_source(T0);
                it += 1;
            }
        }
        pc = PC_NEXT;
    }

    // true while the lane waits for its wavefront's set-up quorum
    LRSC_SM bool wants_setup() const { return pc == PC_NEXT; }

    // between walks: end of the chain, budget, or the next walk's geometry + m_query (:163-184)
    LRSC_SM_NI void next_walk(bool setup_now)
    {
        if(!(it < n_seeds) || error) { pc = PC_FINAL; return; }
        if(next == 0 && LRSC_A.max_walks != 0 && (walks_here >= LRSC_A.max_walks || steps - steps0 >= LRSC_A.max_steps)) {
            state = kReadYield; pc = PC_FINAL; return;
        }
        if(!setup_now) return;
        ++walks_here;
        const uint8_t* read = this->read();
        const int32_t* seeds = this->seeds();
        const uint8_t* out = this->out();
        const int32_t* T = seeds + (uint64_t)(it + (uint32_t)next) * kSeedInts;
        T_start = T[0]; T_len = T[1];
        T_isRepeat = (T[3] & 1) != 0;
        interval = T_start - S_end - 1;
        k = (S_endBest < T[4] ? S_endBest : T[4]) - 2;                     // min(source.endBest, target.startBest) - 2
        if(S_isRepeat || T_isRepeat) {
            k = S_seedLen < T_len ? S_seedLen : T_len;
            k = k < LRSC_A.start_kmer_len + 2 ? k : LRSC_A.start_kmer_len + 2;
        }
        rtou = S_isRepeat && !T_isRepeat;
        trg_len = rtou ? k : T_len;
        if(k < (int)LRSC_A.seed_size || k > (int)kMaxInitK || k > S_seedLen || interval < 0 || trg_len < (int)LRSC_A.min_overlap ||
           (uint32_t)(k + interval + trg_len) > lq_max) { error = LRSC_WALK_ERR_GEOMETRY; pc = PC_FINAL; return; }
        Lq = (uint32_t)(k + interval + trg_len);
        uint8_t* qq = q();
        const uint8_t* tail = out + out_len - k;                           // source.seedStr.substr(seedLen - k)
        if(!rtou) {
            copy_codes(qq, tail, (uint32_t)k);
            copy_codes(qq + k, read + S_end + 1, (uint32_t)interval);
            copy_codes(qq + k + interval, read + T_start, (uint32_t)T_len);
        } else {
            // src <-> trg swapped and everything reverse-complemented (:176-184)
            copy_codes_rc(qq, read + T_start + k - 1, (uint32_t)k);
            copy_codes_rc(qq + k, read + S_end + interval, (uint32_t)interval);
            copy_codes_rc(qq + k + interval, tail + k - 1, (uint32_t)k);
        }
        initk = (uint32_t)k;
        maxOverlap = (uint32_t)k + 2;
        const int min_SA = LRSC_A.pb_coverage > 60 ? (int)((LRSC_A.pb_coverage / 60) * 3) : 3;
        min_SA_threshold = (uint64_t)min_SA;
        if(interval > 100) maxIndelSize = (uint64_t)(interval * 0.2); else maxIndelSize = 20;
        maxLength = (uint64_t)((1.2 * (interval + 10)) + (double)(2 * (uint64_t)k));
        minLength = (uint64_t)((0.8 * (interval - 20)) + (double)(2 * (uint64_t)k));
        m_j = 0; m_start = true;
        pc = PC_PREP;
    }

    // the walk is over: stitch its result or fall back (:185-206, :119-149)
    LRSC_SM_NI void walk_end()
    {
        uint32_t plen = 0, mi = 0;
        uint32_t* bestw = best();
        const int code = finish_walk(&plen, bestw, &mi);
        if(code <= LRSC_WALK_ERR_CHILDREN) { error = code; pc = PC_FINAL; return; }
        if(next == 0) firstType = code;
        const uint8_t* read = this->read();
        const int32_t* seeds = this->seeds();
        uint8_t* out = this->out();
        uint32_t* piece_start = this->piece_start();
        const int32_t* T = seeds + (uint64_t)(it + (uint32_t)next) * kSeedInts;
        const uint8_t* qq = q();
        if(code > 0) {
            // merged = path + target.substr(i + minOverlap); out = merged (un-reversed) minus its first k characters
            const uint32_t tail_from = mi + LRSC_A.min_overlap;
            const uint32_t tlen = (uint32_t)trg_len - tail_from;
            const uint32_t M = plen + tlen;
            uint32_t appended = 0;
            if(!rtou) {
                appended = M - (uint32_t)k;
                if(out_len + appended > out_cap()) { error = LRSC_WALK_ERR_OUTPUT; pc = PC_FINAL; return; }
                // merged[k .. M): the path from character k on, then the rest of the target
                uint8_t* d = out + out_len;
                const uint32_t from_path = plen > (uint32_t)k ? plen - (uint32_t)k : 0u;
                unpack_path(d, bestw, (uint32_t)k, from_path);
                copy_codes(d + from_path, qq + k + interval + tail_from + ((uint32_t)k > plen ? (uint32_t)k - plen : 0u), appended - from_path);
            } else {
                // revcomp(merged) + target.substr(k), minus the first k characters (:195-200)
                const uint32_t total = M + (uint32_t)(T_len - k);
                appended = total - (uint32_t)k;
                if(out_len + appended > out_cap()) { error = LRSC_WALK_ERR_OUTPUT; pc = PC_FINAL; return; }
                uint8_t* d = out + out_len;
                // j = k .. M-1 reads merged[m], m = M-1-j = M-1-k .. 0: first the target tail backwards (m >= plen), then the path backwards
                const uint32_t n_rc = M > (uint32_t)k ? M - (uint32_t)k : 0u;
                const uint32_t m_first = M - 1u - (uint32_t)k;                                  // valid when n_rc > 0
                const uint32_t n_tail = n_rc == 0 ? 0u : (m_first >= plen ? (m_first - plen + 1u < n_rc ? m_first - plen + 1u : n_rc) : 0u);
                if(n_tail) copy_codes_rc(d, qq + k + interval + tail_from + (m_first - plen), n_tail);
                if(n_rc > n_tail) unpack_path_rc(d + n_tail, bestw, m_first - n_tail, n_rc - n_tail);
                copy_codes(d + n_rc, read + T_start + k, total - M);
            }
            out_len += appended;
            ctr(1) += appended;
            ctr(9) += interval;
            ctr(7)++;
            ctr(3)++;
            S_seedLen += (int)appended;                                      // SeedFeature::append
            S_end = T_start + T_len - 1; S_endBest = T[5]; S_isRepeat = T_isRepeat; S_maxFixed = T[2];
            it += (uint32_t)next + 1;
            next = 0;
            pc = PC_NEXT;
            return;
        }
        if(next + 1 < LRSC_A.next_target && it + (uint32_t)next + 1 < n_seeds) { next++; pc = PC_NEXT; return; }
        switch(firstType) {
            case -1: ctr(4)++; break;
            case -2: ctr(5)++; break;
            case -3: ctr(6)++; break;
            default: error = LRSC_WALK_ERR_CODE; break;
        }
        if(error) { pc = PC_FINAL; return; }
        ctr(3)++;
        if(uint8_t* wl = walk_log()) wl[it] = (uint8_t)((firstType + 4) | (LRSC_A.no_dp ? 0x10 : 0));
        const int32_t* T0 = seeds + (uint64_t)it * kSeedInts;               // target = *iterTarget
        if(!LRSC_A.no_dp) {
            // correctByMSAlignment (:208-236): park the read with its query = src k-mer + raw segment + target seed
            ReadOut& R = LRSC_A.out[r];
            const int iv0 = T0[0] - S_end - 1;
            int k0 = (S_endBest < T0[4] ? S_endBest : T0[4]) - 2;
            if(S_isRepeat || (T0[3] & 1)) {
                k0 = S_seedLen < T0[1] ? S_seedLen : T0[1];
                k0 = k0 < LRSC_A.start_kmer_len + 2 ? k0 : LRSC_A.start_kmer_len + 2;
            }
            if(k0 < 1 || k0 > S_seedLen || k0 > T0[1] || iv0 < 0 || (uint32_t)(k0 + iv0 + T0[1]) > lq_max) { error = LRSC_WALK_ERR_GEOMETRY; pc = PC_FINAL; return; }
            uint8_t* dq = dpq();
            const uint8_t* tl = out + out_len - k0;
            copy_codes(dq, tl, (uint32_t)k0);
            copy_codes(dq + k0, read + S_end + 1, (uint32_t)iv0);
            copy_codes(dq + k0 + iv0, read + T0[0], (uint32_t)T0[1]);
            R.dp_k = (uint32_t)k0; R.dp_lq = (uint32_t)(k0 + iv0 + T0[1]);
            R.dp_total_freq = (int64_t)S_maxFixed + (int64_t)T0[2];
            state = kReadParked;
            pc = PC_FINAL;
            return;
        }
        if(LRSC_A.split) {
            if(out_len + (uint32_t)T0[1] > out_cap()) { error = LRSC_WALK_ERR_OUTPUT; pc = PC_FINAL; return; }
            piece_start[n_pieces++] = out_len;                               // pieceVec.push_back(target)
            { copy_codes(out + out_len, read + T0[0], (uint32_t)T0[1]); out_len += (uint32_t)T0[1]; }
            S_seedLen = T0[1];
        } else {
            const int raw = (T0[0] + T0[1] - 1) - S_end;                     // readSeq.substr(source.seedEndPos + 1, ...)
            if(out_len + (uint32_t)raw > out_cap()) { error = LRSC_WALK_ERR_OUTPUT; pc = PC_FINAL; return; }
            { copy_codes(out + out_len, read + S_end + 1, (uint32_t)raw); out_len += (uint32_t)raw; }
            S_seedLen += raw;
        }
        ctr(1) += T0[1];
        load_source(T0);
        it += 1;
        next = 0;
        pc = PC_NEXT;
    }

    LRSC_SM_NI void finalize()
    {
        ReadOut& R = LRSC_A.out[r];
        R.steps = steps;
        R.it = it; R.s_seed_len = S_seedLen; R.s_end = S_end; R.s_end_best = S_endBest; R.s_max_fixed = S_maxFixed;
        R.s_is_repeat = S_isRepeat ? 1 : 0;
        R.c[0] = (int64_t)(LRSC_A.read_off[r + 1] - rs); R.c[2] = n_seeds;
        R.n_pieces = n_pieces; R.out_len = out_len; R.merge = n_pieces != 0; R.error = error;
        R.state = error ? kReadDone : state;
        pc = PC_DONE;
    }

    // =========================================================================================================
    // one sweep: consume the answered request (if any), then run forward through the blocks until the next request
    // =========================================================================================================
    // lanes inside the extension loop of a walk (the population the step gate is a quorum of)
    LRSC_SM bool in_walk() const { return ((pc >= PC_FS && pc <= PC_STEP_ENTRY) || pc == PC_ATT_ENTRY || pc == PC_ATT_LEAF) && pc != PC_PRUNE_SLOW; }
    // ... of which: waiting at the gate in front of the memory-heavy blocks (acceptance ladder + children, pruning + commit)
    LRSC_SM bool at_gate() const { return pc == PC_EXT_READY || pc == PC_PRUNE || pc == PC_ATT_ENTRY; }
    LRSC_SM bool at_slow_gate() const { return pc == PC_PRUNE_SLOW; }
    LRSC_SM bool in_prep() const { return pc == PC_PREP; }

    // One sweep: consume the answered request (if any), then run forward through the blocks until the next request.
    //   setup_now  the set-up quorum is met: lanes between walks build their next m_query and start PREP
    //   begin_now  no lane of the wavefront is in PREP any more: lanes waiting with a finished PREP build their chains
    //              (sort) and root together
    //   slow_now   the slow gate is open (general commits of wide frontiers, batched over several steps)
    //   gate_now   the step gate is open: lanes at the gate run the memory-heavy blocks together, so that their
    //              dependent workspace accesses overlap instead of each lane paying its own latency in its own sweep
    // tk: optional per-block tick accumulators (profiling build of the kernel passes them; nullptr otherwise)
#if defined(__HIP_DEVICE_COMPILE__)
#define LRSC_SM_TICK() (tk ? (uint64_t)__builtin_readcyclecounter() : 0ull)
#else
#define LRSC_SM_TICK() 0ull
#endif
#define LRSC_SM_ACC(i) do { if(tk) { const uint64_t t_now = LRSC_SM_TICK(); tk[i] += t_now - t_prev; t_prev = t_now; } } while(0)
    LRSC_SM void sweep(bool have_result, const SmReq<P>& res, bool setup_now, bool begin_now, bool gate_now, bool slow_now, P* ex_, uint32_t ex_stride_, uint64_t* tk = nullptr)
    {
        uint64_t t_prev = LRSC_SM_TICK();
        tkp = tk;
        req.kind = kReqNone;
        if(have_result) {
            if(pc == PC_FS) fs_result(res);
            else if(pc == PC_SF) sf_result_in(res);
            else if(pc == PC_EXT) pc = PC_EXT_READY;
            else if(pc == PC_PREP) prep_result(res);
        }
        if(pc == PC_ROOT_DONE) {
            LeafT& root = cur()[0];
            root.kmerFrequency = (int)(isize(root.flo, root.fhi) + isize(root.rlo, root.rhi));
            pc = PC_STEP_ENTRY;
        }
        LRSC_SM_ACC(0);                                            // results of FS / SF / PREP requests
        if(pc == PC_EXT_READY && gate_now) ext_eval(ex_, ex_stride_);
        LRSC_SM_ACC(1);
        if(pc == PC_ATT_DONE) att_done();
        if(pc == PC_AFTER_SF_A) { att_no = 2; fs_begin(0, n_cur, (uint32_t)sf_result, PC_ATT_ENTRY, true); }
        if(pc == PC_POST) post();
        if(pc == PC_AFTER_SF_B) fs_begin(1, n_nxt, (uint32_t)sf_result, PC_PRUNE, true);
        LRSC_SM_ACC(2);
        if(pc == PC_PRUNE && gate_now && !prune_is_simple()) pc = PC_PRUNE_SLOW;
        if((pc == PC_PRUNE && gate_now) || (pc == PC_PRUNE_SLOW && slow_now)) prune_and_commit();
        LRSC_SM_ACC(3);
        if(pc == PC_STEP_ENTRY) step_entry();
        LRSC_SM_ACC(4);
        if(pc == PC_WALK_END) walk_end();
        if(pc == PC_NEXT) next_walk(setup_now);
        LRSC_SM_ACC(5);
        if(pc == PC_PREP && req.kind == kReqNone) prep_advance();
        LRSC_SM_ACC(6);
        if(pc == PC_BEGIN && begin_now) begin_walk();
        LRSC_SM_ACC(7);
        if(pc == PC_ATT_ENTRY && gate_now) att_entry();
        if(pc == PC_ATT_LEAF) att_leaf();
        if(pc == PC_FINAL) finalize();
        LRSC_SM_ACC(8);
    }
};

} // namespace lrsc
