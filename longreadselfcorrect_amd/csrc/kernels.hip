// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the FM-index hot path.
//
// All of these are integer rank/select over a read-only multi-GB structure: the bound is
// HBM/L2 gather bandwidth and latency, not MFMA.  Design points:
//   * one Occ query == one 64-byte aligned block load (4 x global_load_dwordx4 by one lane);
//     a backward-search step issues its 4 block loads (2 strands x {lower-1, upper})
//     back-to-back so every lane keeps 4 lines in flight, i.e. 256 lines per wavefront;
//   * no LDS and < 128 VGPRs so that >= 16 wavefronts per CU stay resident to cover the
//     ~2 us loaded-HBM latency (Little: ~770 lines in flight per CU saturate 6.3 TB/s);
//   * adjacent lanes own adjacent read positions: read bytes are coalesced, index blocks are
//     inherently random (BWT order), so the 64-byte block is the unit of traffic.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace lrsc {

// ---------------------------------------------------------------------------------------
// block load + in-block rank
// ---------------------------------------------------------------------------------------
template <bool WIDE> struct Lay;

template <> struct Lay<false> {
    static constexpr uint32_t kSyms = Block32::kSyms;
    struct Regs { uint4 q[4]; };   // q[0] = counts, q[1..3] = 192 symbols
    static __device__ __forceinline__ void load(const void* blocks, uint64_t b, Regs& r)
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const Block32*>(blocks) + b);
        r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    }
    static __device__ __forceinline__ uint64_t count(const Regs& r, uint32_t code, uint32_t off, bool& flagged)
    {
        const uint32_t c0 = r.q[0].x;
        flagged = (c0 & kFlag32) != 0;
        const uint32_t base = code == 0 ? (c0 & ~kFlag32) : code == 1 ? r.q[0].y : code == 2 ? r.q[0].z : r.q[0].w;
        const uint32_t w[12] = {r.q[1].x, r.q[1].y, r.q[1].z, r.q[1].w, r.q[2].x, r.q[2].y,
                                r.q[2].z, r.q[2].w, r.q[3].x, r.q[3].y, r.q[3].z, r.q[3].w};
        return (uint64_t)base + inblock32(w, code, off);
    }
    static __device__ __forceinline__ uint32_t symbol(const Regs& r, uint32_t off)
    {
        const uint32_t w[12] = {r.q[1].x, r.q[1].y, r.q[1].z, r.q[1].w, r.q[2].x, r.q[2].y,
                                r.q[2].z, r.q[2].w, r.q[3].x, r.q[3].y, r.q[3].z, r.q[3].w};
        uint32_t v = 0;
#pragma unroll
        for(uint32_t i = 0; i < 12; ++i) v = (off / 16 == i) ? w[i] : v;
        return (v >> (2 * (off % 16))) & 3u;
    }
    static __device__ __forceinline__ bool flagged(const Regs& r) { return (r.q[0].x & kFlag32) != 0; }
};

template <> struct Lay<true> {
    static constexpr uint32_t kSyms = Block64::kSyms;
    struct Regs { uint4 q[4]; };   // q[0..1] = counts, q[2..3] = 128 symbols
    static __device__ __forceinline__ void load(const void* blocks, uint64_t b, Regs& r)
    {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const Block64*>(blocks) + b);
        r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    }
    static __device__ __forceinline__ uint64_t u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
    static __device__ __forceinline__ uint64_t count(const Regs& r, uint32_t code, uint32_t off, bool& flagged)
    {
        const uint64_t c0 = u64(r.q[0].x, r.q[0].y);
        flagged = (c0 & kFlag64) != 0;
        const uint64_t base = code == 0 ? (c0 & ~kFlag64) : code == 1 ? u64(r.q[0].z, r.q[0].w)
                            : code == 2 ? u64(r.q[1].x, r.q[1].y) : u64(r.q[1].z, r.q[1].w);
        const uint64_t w[4] = {u64(r.q[2].x, r.q[2].y), u64(r.q[2].z, r.q[2].w),
                               u64(r.q[3].x, r.q[3].y), u64(r.q[3].z, r.q[3].w)};
        return base + inblock64(w, code, off);
    }
    static __device__ __forceinline__ uint32_t symbol(const Regs& r, uint32_t off)
    {
        const uint64_t w[4] = {u64(r.q[2].x, r.q[2].y), u64(r.q[2].z, r.q[2].w),
                               u64(r.q[3].x, r.q[3].y), u64(r.q[3].z, r.q[3].w)};
        uint64_t v = 0;
#pragma unroll
        for(uint32_t i = 0; i < 4; ++i) v = (off / 32 == i) ? w[i] : v;
        return (uint32_t)(v >> (2 * (off % 32))) & 3u;
    }
    static __device__ __forceinline__ bool flagged(const Regs& r) { return (u64(r.q[0].x, r.q[0].y) & kFlag64) != 0; }
};

__device__ __forceinline__ uint64_t pred_of(const FmStrand& s, uint32_t code)
{
    return code == 0 ? s.pred[1] : code == 1 ? s.pred[2] : code == 2 ? s.pred[3] : s.pred[4];
}

// Occ over the first p symbols (p = idx + 1, 0 <= p <= N): RLBWT::getOcc (RLBWT.h:121-140)
template <bool WIDE>
__device__ __forceinline__ uint64_t occ_prefix(const FmStrand& s, uint32_t code, uint64_t p)
{
    using L = Lay<WIDE>;
    const uint64_t b = p / L::kSyms;
    const uint32_t off = (uint32_t)(p - b * L::kSyms);
    typename L::Regs r;
    L::load(s.blocks, b, r);
    bool flagged;
    uint64_t c = L::count(r, code, off, flagged);
    if(code == 0 && flagged && off != 0) c -= dollars_in(s, b * L::kSyms, b * L::kSyms + off);
    return c;
}

struct Iv { int64_t lo, hi; };

// BWTAlgorithms::updateInterval (BWTAlgorithms.h:66-72) on one strand.
// Both block loads are issued before either is consumed.
template <bool WIDE>
__device__ __forceinline__ void update_interval(const FmStrand& s, uint32_t code, Iv& iv,
                                                uint32_t& n_rank, uint32_t& n_blk)
{
    using L = Lay<WIDE>;
    const uint64_t pl = (uint64_t)iv.lo;            // (lower - 1) + 1
    const uint64_t pu = (uint64_t)iv.hi + 1;        // upper + 1
    const uint64_t bl = pl / L::kSyms, bu = pu / L::kSyms;
    const uint32_t ol = (uint32_t)(pl - bl * L::kSyms), ou = (uint32_t)(pu - bu * L::kSyms);
    typename L::Regs ra, rb;
    L::load(s.blocks, bl, ra);
    L::load(s.blocks, bu, rb);
    bool fa, fb;
    uint64_t ca = L::count(ra, code, ol, fa);
    uint64_t cb = L::count(rb, code, ou, fb);
    if(code == 0) {
        if(fa && ol != 0) ca -= dollars_in(s, bl * L::kSyms, bl * L::kSyms + ol);
        if(fb && ou != 0) cb -= dollars_in(s, bu * L::kSyms, bu * L::kSyms + ou);
    }
    const uint64_t pb = pred_of(s, code);
    iv.lo = (int64_t)(pb + ca);
    iv.hi = (int64_t)(pb + cb) - 1;
    n_rank += 2;
    n_blk += (bl == bu) ? 1u : 2u;
}

// BWTAlgorithms::initInterval (BWTAlgorithms.h:136-140): Occ(b, N-1) is the symbol total.
__device__ __forceinline__ void init_interval(const FmStrand& s, uint32_t code, Iv& iv)
{
    const uint64_t lo = pred_of(s, code);
    const uint64_t next = code == 3 ? s.n_symbols : pred_of(s, code + 1);
    iv.lo = (int64_t)lo;
    iv.hi = (int64_t)next - 1;
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
// Steps < base_k reproduce findInterval's early exit per strand (BWTAlgorithms.cpp:28);
// steps >= base_k are KmerFeature::expand (KmerFeature.h:92-99): no validity check.
// ---------------------------------------------------------------------------------------
struct WalkState {
    Iv fwd, rvc;
    uint32_t size;          // bases consumed
    uint32_t counted;       // bases counted by the base search (fwd strand)
    bool fwd_broken, rvc_broken;
};

template <bool WIDE>
__device__ __forceinline__ void walk_step(const FmIndexDev& fm, uint32_t c, uint32_t base_k, WalkState& st,
                                          uint32_t& n_rank, uint32_t& n_blk)
{
    const FmStrand& sf = fm.strand[LRSC_RBWT];
    const FmStrand& sr = fm.strand[LRSC_BWT];
    if(st.size == 0) {
        init_interval(sf, c, st.fwd);
        init_interval(sr, 3u - c, st.rvc);
        st.counted = 1;
        n_rank += 2;   // initInterval's getOcc(b, N-1) per strand (served from pred[] here)
    } else if(st.size >= base_k) {
        update_interval<WIDE>(sf, c, st.fwd, n_rank, n_blk);
        update_interval<WIDE>(sr, 3u - c, st.rvc, n_rank, n_blk);
    } else {
        if(!st.fwd_broken) {
            ++st.counted;
            update_interval<WIDE>(sf, c, st.fwd, n_rank, n_blk);
            st.fwd_broken = st.fwd.lo > st.fwd.hi;
        }
        if(!st.rvc_broken) {
            update_interval<WIDE>(sr, 3u - c, st.rvc, n_rank, n_blk);
            st.rvc_broken = st.rvc.lo > st.rvc.hi;
        }
    }
    ++st.size;
}

__device__ __forceinline__ int64_t iv_freq(const Iv& iv) { return iv.lo <= iv.hi ? iv.hi - iv.lo + 1 : 0; }

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void rank_kernel(FmIndexDev fm, const lrsc_rank_query* __restrict__ q,
                                                   uint64_t n, uint64_t* __restrict__ out, DevCounters* ctr)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(i < n) {
        const lrsc_rank_query qq = q[i];
        const uint32_t code = ((qq.base >> 1) & 3u) ^ (((qq.base >> 1) & 3u) >> 1);
        out[i] = occ_prefix<WIDE>(fm.strand[qq.strand & 1], code, (uint64_t)(qq.idx + 1));
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
    const FmStrand& s = fm.strand[strand & 1];
    const uint64_t p = idx[i];
    const uint64_t b = p / L::kSyms;
    const uint32_t off = (uint32_t)(p - b * L::kSyms);
    typename L::Regs r;
    L::load(s.blocks, b, r);
    const uint32_t code = L::symbol(r, off);
    char ch = "ACGT"[code];
    if(code == 0 && L::flagged(r) && dollars_in(s, p, p + 1) != 0) ch = '$';
    out[i] = ch;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void find_kmers_kernel(FmIndexDev fm, const uint8_t* __restrict__ codes, uint32_t k,
                                                         uint64_t n, lrsc_biinterval* __restrict__ out, DevCounters* ctr)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(i < n) {
        WalkState st;
        st.size = 0; st.counted = 0; st.fwd_broken = false; st.rvc_broken = false;
        st.fwd.lo = st.fwd.hi = st.rvc.lo = st.rvc.hi = 0;
        const uint8_t* w = codes + i * k;
        for(uint32_t s = 0; s < k; ++s) {
            if(st.fwd_broken && st.rvc_broken) break;
            walk_step<WIDE>(fm, w[s], k, st, n_rank, n_blk);
        }
        lrsc_biinterval o;
        o.fwd.lower = st.fwd.lo; o.fwd.upper = st.fwd.hi;
        o.rvc.lower = st.rvc.lo; o.rvc.upper = st.rvc.hi;
        out[i] = o;
    }
    flush_counters(ctr, n_rank, n_blk);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void kmer_grid_kernel(FmIndexDev fm, GridArgs a, DevCounters* ctr)
{
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

        WalkState st;
        st.size = 0; st.counted = 0; st.fwd_broken = false; st.rvc_broken = false;
        st.fwd.lo = st.fwd.hi = st.rvc.lo = st.rvc.hi = 0;
        uint32_t slot = 0;
        const uint8_t* w = a.codes + gid;

        auto emit = [&](uint32_t j) {
            const uint64_t rec = gid * a.n_k + j;
            if(a.out_iv) {
                lrsc_biinterval o;
                o.fwd.lower = st.fwd.lo; o.fwd.upper = st.fwd.hi;
                o.rvc.lower = st.rvc.lo; o.rvc.upper = st.rvc.hi;
                a.out_iv[rec] = o;
            }
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
                uchar4 v = make_uchar4((uint8_t)cnt[0], (uint8_t)cnt[1], (uint8_t)cnt[2], (uint8_t)cnt[3]);
                reinterpret_cast<uchar4*>(a.out_count)[rec] = v;
            }
            if(a.freq) {
                const bool fake = st.size != a.ks[j];
                a.freq[(uint64_t)j * a.total_bases + gid] = fake ? -1 : (int32_t)(iv_freq(st.fwd) + iv_freq(st.rvc));
            }
            if(a.slot_iv) {
                lrsc_biinterval o;
                o.fwd.lower = st.fwd.lo; o.fwd.upper = st.fwd.hi;
                o.rvc.lower = st.rvc.lo; o.rvc.upper = st.rvc.hi;
                a.slot_iv[(uint64_t)j * a.total_bases + gid] = o;
            }
        };

        for(uint32_t s = 0; s < avail; ++s) {
            walk_step<WIDE>(fm, w[s], base_k, st, n_rank, n_blk);
            if(st.size == a.ks[slot]) { emit(slot); ++slot; }
        }
        // slots the read end cut short keep the last state ("fake" k-mers, KmerFeature.h:62)
        for(; slot < a.n_k; ++slot) emit(slot);
        if(a.base_counted) a.base_counted[gid] = (uint8_t)st.counted;
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
