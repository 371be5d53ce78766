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

#include <algorithm>

#include "kernels.h"
#include "rank_device.h"

namespace lrsc {

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

// One lane per BWT row: BWT[idx] and Occ(BWT[idx], idx - 1) come from the SAME rank block (prefix length idx
// lives in block idx / kSyms), so an LF step is one 64-byte load.
template <bool WIDE>
__global__ __launch_bounds__(256) void lf_walk_kernel(FmIndexDev fm, const LfJob* __restrict__ jobs, uint64_t n,
                                                      uint8_t* __restrict__ out, uint32_t* __restrict__ out_len, DevCounters* ctr)
{
    using L = Lay<WIDE>;
    using P = typename L::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0;
    if(i < n) {
        const LfJob job = jobs[i];
        const StrandC<P> s0 = strand_consts<P>(fm.strand[0]);
        const StrandC<P> s1 = strand_consts<P>(fm.strand[1]);
        const bool second = (job.strand & 1) != 0;
        uint8_t* dst = out + job.out_off;
        P idx = (P)job.row;
        uint32_t len = 0;
        for(; len < job.max_steps; ++len) {
            const P b = idx / L::kSyms;
            const uint32_t off = (uint32_t)(idx - b * L::kSyms);
            typename L::Regs r;
            L::load(second ? s1.blocks : s0.blocks, b, r);
            const uint32_t code = L::symbol(r, off);
            const bool flagged = L::flagged(r);
            if(code == 0 && flagged && dollars_in_c(second ? s1 : s0, (uint64_t)idx, (uint64_t)idx + 1) != 0) break;   // '$'
            dst[len] = (uint8_t)code;
            uint64_t c = L::count(r, code, mtab + off * L::kRow);
            if(code == 0 && off != 0 && flagged)
                c -= dollars_in_c(second ? s1 : s0, (uint64_t)b * L::kSyms, (uint64_t)b * L::kSyms + off);
            idx = (second ? pred_of(s1, code) : pred_of(s0, code)) + (P)c;
            n_rank += 1; n_blk += 1;
        }
        out_len[i] = len;
    }
    flush_counters(ctr, n_rank, n_blk);
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
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void kmer_grid_kernel(FmIndexDev fm, GridArgs a, DevCounters* ctr)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n_rank = 0, n_blk = 0, n_tab = 0;
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
        uint32_t vmask = 0;
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
            if(a.freq && a.freq_index[j] >= 0) {
                const bool fake = st.size != a.ks[j];
                a.freq[(uint64_t)a.freq_index[j] * a.total_bases + gid] = fake ? -1 : (int32_t)(iv_freq(st.fwd) + iv_freq(st.rvc));
                if(st.fwd.lo <= st.fwd.hi && st.rvc.lo <= st.rvc.hi) vmask |= 1u << a.freq_index[j];
            }
            if(a.slot_iv) a.slot_iv[(uint64_t)j * a.total_bases + gid] = to_out(st.fwd, st.rvc);
        };

        // Compact mode: only frequency / validity / the base-search counter are emitted.  An empty interval stays empty
        // under updateInterval (Occ(c, lo-1) == Occ(c, hi) when hi == lo-1) and contributes frequency 0, so (a) the table
        // entry -- findInterval semantics, a strand frozen at its first empty interval -- is as good as the reference's
        // chained expand() state whenever the 5-mer base search itself ran to the end on the fwd strand (then the base
        // counter is base_k), and (b) a strand that went empty needs no further Occ queries at all.
        uint32_t s0 = 0;
        bool lean = false;
        if(!a.out_iv && !a.out_size && !a.out_count && !a.slot_iv) {
            WalkState<P> ts = st;
            const uint32_t tk = table_start<WIDE>(fm, [&](uint32_t t) { return (uint32_t)w[t]; }, avail, ts);
            if(tk >= base_k) {
                bool fwd_base_ok = !ts.fwd_broken;              // valid after tk >= base_k steps => valid after base_k
                if(!fwd_base_ok && tk > base_k) {
                    WalkState<P> tb = st;
                    const uint32_t kb = table_start<WIDE>(fm, [&](uint32_t t) { return (uint32_t)w[t]; }, base_k, tb);
                    fwd_base_ok = kb == base_k && !tb.fwd_broken;
                    n_tab += kb != 0;
                }
                // a stored slot below the table size (freq_index >= 0) must have its own table of exactly that size
                // (the 15-mers when the search starts from the 16-mer table): check, else fall back
                bool ok = fwd_base_ok;
                for(uint32_t j = 0; j < a.n_k && a.ks[j] < tk; ++j)
                    if(a.freq_index[j] >= 0)
                        ok = ok && (fm.ktab[0].k == a.ks[j] || fm.ktab[1].k == a.ks[j] || fm.ktab[2].k == a.ks[j] || fm.ktab[3].k == a.ks[j]);
                if(ok) {
                    for(uint32_t j = 0; j < a.n_k && a.ks[j] < tk; ++j) {
                        if(a.freq_index[j] < 0) continue;
                        WalkState<P> te = walk_init<P>();
                        (void)table_start<WIDE>(fm, [&](uint32_t t) { return (uint32_t)w[t]; }, a.ks[j], te);
                        st = te; st.counted = base_k;
                        emit(j);
                        n_tab += 1;
                    }
                    st = ts; st.counted = base_k; st.n_rank = 0; st.n_blk = 0;
                    s0 = tk; n_tab += 1; lean = true;
                    while(slot < a.n_k && a.ks[slot] < tk) ++slot;
                    next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu;
                    if(st.size == next_k) { emit(slot); ++slot; next_k = slot < a.n_k ? a.ks[slot] : 0xFFFFFFFFu; }
                }
            }
        }
        for(uint32_t s = s0; s < avail; ++s) {
            if(lean) {
                const uint32_t c = w[s];
                if(!st.fwd_broken) {
                    st.fwd = update_interval<WIDE, false>(sf, c, st.fwd, mtab, st.n_blk);
                    st.fwd_broken = st.fwd.lo > st.fwd.hi;
                    st.n_rank += 2;
                }
                if(!st.rvc_broken) {
                    st.rvc = update_interval<WIDE, false>(sr, 3u - c, st.rvc, mtab, st.n_blk);
                    st.rvc_broken = st.rvc.lo > st.rvc.hi;
                    st.n_rank += 2;
                }
                ++st.size;
            } else
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
        if(a.valid_mask) a.valid_mask[gid] = (uint8_t)vmask;
        n_rank = st.n_rank; n_blk = st.n_blk;
    }
    flush_counters(ctr, n_rank, n_blk, n_tab);
}

// every k-mer's findBiInterval (with early exit) -> table entry (4 x u32 for narrow indexes, 4 x u64 for wide ones).  A smaller
// table that is already built (prev_k < k) supplies the state after the first prev_k characters.
template <bool WIDE>
__global__ __launch_bounds__(256) void ktab_build_kernel(FmIndexDev fm, uint32_t k, uint4* __restrict__ entries, uint32_t prev_k,
                                                         const uint4* __restrict__ prev)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
    const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
    const uint64_t n_codes = 1ull << (2 * k);
    for(uint64_t code = (uint64_t)blockIdx.x * 256 + threadIdx.x; code < n_codes; code += (uint64_t)gridDim.x * 256) {
        WalkState<P> st = walk_init<P>();
        if(prev_k != 0) {
            const uint64_t pc = code >> (2 * (k - prev_k));
            if(WIDE) {
                const uint4 a = prev[pc * 2], b = prev[pc * 2 + 1];
                st.fwd.lo = (P)(((uint64_t)a.y << 32) | a.x); st.fwd.hi = (P)(((uint64_t)a.w << 32) | a.z);
                st.rvc.lo = (P)(((uint64_t)b.y << 32) | b.x); st.rvc.hi = (P)(((uint64_t)b.w << 32) | b.z);
            } else {
                const uint4 e = prev[pc];
                st.fwd.lo = (P)e.x; st.fwd.hi = (P)e.y; st.rvc.lo = (P)e.z; st.rvc.hi = (P)e.w;
            }
            st.fwd_broken = st.fwd.lo > st.fwd.hi; st.rvc_broken = st.rvc.lo > st.rvc.hi;
            st.size = prev_k;
        }
        for(uint32_t t = prev_k; t < k; ++t) {
            if(st.fwd_broken && st.rvc_broken) break;
            const uint32_t c = (uint32_t)(code >> (2 * (k - 1 - t))) & 3u;
            st = walk_step<WIDE>(sf, sr, c, 1u << 30, st, mtab);
        }
        if(WIDE) {
            const uint64_t fl = (uint64_t)st.fwd.lo, fh = (uint64_t)st.fwd.hi, rl = (uint64_t)st.rvc.lo, rh = (uint64_t)st.rvc.hi;
            entries[code * 2] = make_uint4((uint32_t)fl, (uint32_t)(fl >> 32), (uint32_t)fh, (uint32_t)(fh >> 32));
            entries[code * 2 + 1] = make_uint4((uint32_t)rl, (uint32_t)(rl >> 32), (uint32_t)rh, (uint32_t)(rh >> 32));
        } else
            entries[code] = make_uint4((uint32_t)st.fwd.lo, (uint32_t)st.fwd.hi, (uint32_t)st.rvc.lo, (uint32_t)st.rvc.hi);
    }
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

hipError_t launch_lf_walk(const FmIndexDev& fm, const LfJob* jobs, uint64_t n, uint8_t* out, uint32_t* out_len,
                          DevCounters* ctr, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    if(fm.wide) hipLaunchKernelGGL(lf_walk_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, jobs, n, out, out_len, ctr);
    else        hipLaunchKernelGGL(lf_walk_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, jobs, n, out, out_len, ctr);
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

hipError_t launch_ktab_build(const FmIndexDev& fm, uint32_t k, void* entries, uint32_t prev_k, const void* prev, hipStream_t stream)
{
    if(k == 0 || k > 16 || prev_k >= k) return hipErrorInvalidValue;
    const uint64_t n = std::min<uint64_t>(1ull << (2 * k), 1ull << 30);          // grid-stride above 2^30 threads
    if(fm.wide) hipLaunchKernelGGL(ktab_build_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, k, reinterpret_cast<uint4*>(entries), prev_k,
                                   reinterpret_cast<const uint4*>(prev));
    else        hipLaunchKernelGGL(ktab_build_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, stream, fm, k, reinterpret_cast<uint4*>(entries), prev_k,
                                   reinterpret_cast<const uint4*>(prev));
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
