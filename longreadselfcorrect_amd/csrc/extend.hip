// extend.hip -- seed-to-seed FM-extend on the device: LongReadSelfCorrectByOverlap
// (reference: PacBio/LongReadCorrectByOverlap.cpp:17-878, FMIndexWalk/SAINode.cpp:166-189,
//  PacBio/IntervalTree.cpp:4-91).
//
// Two kernels per batch of walks:
//   walk_prepare_kernel  one lane per (walk, query offset): the bulk, perfectly parallel part of the
//                        constructor -- bi-intervals of every 5-mer, 9-mer and (inside the target seed)
//                        13-mer of m_query = beginning k-mer + raw read segment + target seed
//                        (.cpp:82-94,127-152).  ~60 % of the Occ queries of a walk.
//   walk_extend_kernel   one lane per walk: root interval, the interval "trees", then the
//                        extendOverlap loop (.cpp:155-211) with its <= 32-leaf frontier.
//
// The reference's four IntervalTree<size_t> objects are replaced by flat per-k-mer chains:
//   * a leaf interval is contained in a tree entry iff the entry's k-mer is the leaf's current suffix
//     k-mer (intervals of distinct equal-length k-mers are disjoint), so a query is an equality
//     look-up of the 5-/9-mer code, not an interval search;
//   * for the 5-mer trees only "is there a hit in [start, large]" is observable (ismatchedbykmer);
//   * for the 9-mer trees the ORDER of the hits is observable (isSupportedByNewSeed walks the fwd and
//     rvc hit lists in lock-step and the first minimum wins): the hits come back in the order
//     std::sort left the equal keys in, so the entries are sorted with an exact re-implementation of
//     libstdc++'s introsort (introsort_emul.h) and chained in that order.
// All double arithmetic is evaluated in the reference's order; the file is built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "walk_device.h"

namespace lrsc {

constexpr uint32_t kWalksPerWave = 64;

// ---------------------------------------------------------------------------------------
// 1. prepare
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void walk_prepare_kernel(FmIndexDev fm, ExtendArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt_rank = 0, cnt_blk = 0;
    if(gid < a.total_q) {
        uint32_t w = a.chunk_walk[gid >> kChunkShift];
        while(a.q_off[w + 1] <= gid) ++w;
        const WalkWork ww = a.work[w];
        const uint32_t i = (uint32_t)(gid - a.q_off[w]);          // offset inside m_query
        uint8_t* ws = a.workspace + ww.ws_off;
        const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
        const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
        prepare_offset<WIDE>(fm, sf, sr, mtab, a.codes + ww.codes_off, i, ww.lq, ww.initk + ww.path_len, a.seed_size, a.min_overlap,
                             reinterpret_cast<SortItem*>(ws + ww.o_item9f), reinterpret_cast<SortItem*>(ws + ww.o_item9r),
                             ws + ww.o_flags5, reinterpret_cast<P*>(ws + ww.o_term), cnt_rank, cnt_blk);
    }
    flush_counters(a.ctr, cnt_rank, cnt_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(64, 2) void walk_extend_kernel(FmIndexDev fm, ExtendArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint32_t slot = blockIdx.x * kWalksPerWave + threadIdx.x / (64 / kWalksPerWave);
    const bool owner = (threadIdx.x % (64 / kWalksPerWave)) == 0;
    uint32_t n_rank = 0, n_blk = 0;
    if(owner && slot < a.n_walks) {
        const uint32_t w = a.order ? a.order[slot] : slot;
        const WalkWork ww = a.work[w];
        uint8_t* ws = a.workspace + ww.ws_off;
        Walk<WIDE> W;
        W.sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
        W.sR = strand_consts<P>(fm.strand[LRSC_BWT]);
        W.fm = &fm;
        W.mtab = mtab;
        W.q = a.codes + ww.codes_off;
        W.Lq = ww.lq; W.initk = ww.initk; W.path_len = ww.path_len; W.trg_len = ww.trg_len; W.dis = ww.dis;
        W.seedSize = a.seed_size; W.minOverlap = a.min_overlap; W.maxOverlap = ww.max_overlap; W.maxLeaves = a.max_leaves;
        W.min_SA_threshold = ww.min_sa;
        W.PBcoverage = a.pb_coverage; W.PacBioErrorRate = a.pacbio_error_rate; W.errorRate = 0.25; W.localK = 100;
        W.freqsOfKmerSize = a.freqs_of_kmer_size;
        // .cpp:55-58,78-79: double expressions truncated to size_t
        if(ww.dis > 100) W.maxIndelSize = (uint64_t)(ww.dis * 0.2); else W.maxIndelSize = 20;
        W.maxLength = (uint64_t)((1.2 * (ww.dis + 10)) + (double)(2 * (uint64_t)ww.initk));
        W.minLength = (uint64_t)((0.8 * (ww.dis - 20)) + (double)(2 * (uint64_t)ww.initk));
        W.it9f = reinterpret_cast<SortItem*>(ws + ww.o_item9f);
        W.it9r = reinterpret_cast<SortItem*>(ws + ww.o_item9r);
        W.next9f = reinterpret_cast<uint16_t*>(ws + ww.o_next9f);
        W.next9r = reinterpret_cast<uint16_t*>(ws + ww.o_next9r);
        W.head9f = reinterpret_cast<uint16_t*>(ws + ww.o_head9);
        W.head9r = W.head9f + 256;
        W.head5 = reinterpret_cast<uint16_t*>(ws + ww.o_head5);
        W.next5 = reinterpret_cast<uint16_t*>(ws + ww.o_next5);
        W.flags5 = ws + ww.o_flags5;
        W.term = reinterpret_cast<const P*>(ws + ww.o_term);
        W.n_term = ww.trg_len >= a.min_overlap ? ww.trg_len - a.min_overlap + 1 : 0;
        W.cur = reinterpret_cast<Leaf<P>*>(ws + ww.o_leaves);
        W.nxt = W.cur + 32;
        W.leaf_small = W.cur;
        W.rings = reinterpret_cast<double*>(ws + ww.o_rings);
        W.paths = reinterpret_cast<uint32_t*>(ws + ww.o_paths);
        W.pathw = ww.pathw;
        W.rpaths = W.paths + (uint64_t)32 * ww.pathw;
        W.results = reinterpret_cast<WalkResultRec*>(ws + ww.o_results);
        W.n_rank = 0; W.n_blk = 0; W.steps = 0; W.error = 0; W.cyc_setup = 0; W.cyc_loop = 0; W.prof = nullptr; W.profile = false;

        WalkOut& o = a.out[w];
        uint32_t len = 0, mi = 0;
        const int code = W.run(&len, reinterpret_cast<uint32_t*>(a.out_paths + ww.out_off), &mi);
        o.code = code; o.path_len = len; o.match_i = mi; o.steps = (uint32_t)W.steps;
        n_rank = W.n_rank; n_blk = W.n_blk;
    }
    flush_counters(a.ctr, n_rank, n_blk);
}

// ---------------------------------------------------------------------------------------
static inline unsigned nblk256(uint64_t n) { return (unsigned)((n + 255) / 256); }

size_t leaf_bytes(bool wide) { return wide ? sizeof(Leaf<uint64_t>) : sizeof(Leaf<uint32_t>); }

hipError_t launch_walk_prepare(const FmIndexDev& fm, const ExtendArgs& a, hipStream_t stream)
{
    if(a.total_q == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(walk_prepare_kernel<true>, dim3(nblk256(a.total_q)), dim3(256), 0, stream, fm, a);
    else        hipLaunchKernelGGL(walk_prepare_kernel<false>, dim3(nblk256(a.total_q)), dim3(256), 0, stream, fm, a);
    return hipGetLastError();
}

hipError_t launch_walk_extend(const FmIndexDev& fm, const ExtendArgs& a, hipStream_t stream)
{
    if(a.n_walks == 0) return hipSuccess;
    const unsigned nb = (a.n_walks + kWalksPerWave - 1) / kWalksPerWave;
    if(fm.wide) hipLaunchKernelGGL(walk_extend_kernel<true>, dim3(nb), dim3(64), 0, stream, fm, a);
    else        hipLaunchKernelGGL(walk_extend_kernel<false>, dim3(nb), dim3(64), 0, stream, fm, a);
    return hipGetLastError();
}

} // namespace lrsc
