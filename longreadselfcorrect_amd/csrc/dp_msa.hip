// dp_msa.hip -- MultipleAlignment::addOverlap / _addSequence / calculateBaseConsensus
// (Thirdparty/multiple_alignment.cpp:208-393,517-594, as driven by LongReadOverlap::buildMultipleAlignment,
// PacBio/LongReadOverlap.cpp:17-55) on the device: one WAVEFRONT per correctByMSAlignment call, state in LDS.
//
// The reference keeps every row's padded string; the consensus only needs, per column, how many rows show
// A / C / G / T / '-', plus the padded base row.  So the state here is
//   T[]            the base row's padded_sequence (codes 0-3, kGap),
//   cnt[col]       five 16-bit counters per global column,
//   lead[e],size[e] leading_columns and padded length of every other row (what insertGapBeforeColumn looks at),
// and the operations are the reference's, step for step, including its quirks: incoming rows are aligned against
// the base row only; an insertion opens a new column unless the base row already has a gap column *at the
// current template position*; leading_columns of the base row is read once per incoming row (:306), so an
// insertion in front of base 0 moves the template cursor past a base without consuming a cigar op.
// Rows are added in retrieval order (source-seed strings, then target-seed strings).
//
// One incoming row is added in a few wavefront-wide passes instead of one dependent step per column (a.row_batch,
// the default).  What makes that possible: in the coordinates the base row has when the row starts, every cigar
// op's effect is known from prefix counts over the cigar alone --
//   * an M/D op consumes the next base of the base row; the b-th base sits at bpos[b];
//   * a run of r I ops in front of base b first fills the G = bpos[b] - bpos[b-1] - 1 gap columns already there
//     (op i < G lands in column bpos[b-1] + 1 + i), the others each open a new column in front of bpos[b];
//   * gap columns of the base row the cigar walks over without an I show '-' in the new row;
// and insertGapBeforeColumn's effect on every other row (leading_columns += 1 when the column is at or before its
// start, one more padded symbol when strictly inside) only depends on where the new column is in those OLD
// coordinates, because all earlier insertions of the same row lie at or before it.  So: one pass over the cigar
// (prefix counts by ballot), the row's symbols scattered into a per-column byte array, one shift of T / cnt / that
// array by "insertions at or before me", the new columns written, the row's symbols counted in.  Rows that meet the
// reference's corner cases -- leading insertions in front of base 0 (the cursor quirk above), insertions behind the
// last base, more than kMaxIns new columns -- take the step-by-step walk, which stays in the kernel.
#include <hip/hip_runtime.h>

#include "dp_dev.h"

namespace lrsc {

namespace {
constexpr uint32_t kGap = 4;
// per-column counters of A C G T '-' : five 12-bit fields of one 64-bit word (at most 4095 rows)
using Cnt = unsigned long long;
constexpr uint32_t kCntBits = 12;
constexpr uint32_t kMaxIns = 256;         // new columns one row may open on the batched path
constexpr uint32_t kUnset = 0xFEu;
__device__ __forceinline__ uint32_t cnt_get(Cnt v, uint32_t sym) { return (uint32_t)(v >> (kCntBits * sym)) & ((1u << kCntBits) - 1u); }

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
} // namespace

uint32_t dp_msa_lds_bytes(uint32_t w_cols, uint32_t lq, uint32_t str_cap, uint32_t ops_cap, uint32_t n_str)
{
    const uint32_t W = w_cols, E = n_str + 2;
    uint32_t o = 0;
    o += W * (uint32_t)sizeof(Cnt);        // cnt
    o += E * 8;                            // lead, size
    o += (lq + 2) * 4;                     // bpos
    o += kMaxIns * 4;                      // insx
    o += kMaxIns * 2;                      // insg
    o += kMaxIns;                          // inss
    o += (W + 3) & ~3u;                    // T
    o += (W + 3) & ~3u;                    // Rw
    o += (str_cap + 7) & ~3u;              // S (staged by dwords from the aligned address below the string)
    o += (ops_cap + 3) & ~3u;              // ops
    return o;
}

// GLOBAL = false: state in LDS (dynamic, a.lds_bytes); true: in a global workspace slot per workgroup, for the few
// pile-ups too wide for 160 KB.
template <bool GLOBAL>
__global__ __launch_bounds__(64) void dp_msa_kernel(DpPipeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_lds[];
    uint8_t* smem = GLOBAL ? a.msa_ws + (uint64_t)blockIdx.x * a.lds_bytes : smem_lds;
    const uint32_t lane = lane_id();
    const uint32_t n_work = a.req_list ? a.n_list : a.n_reqs;
    for(uint32_t wi = blockIdx.x; wi < n_work;) {
        const uint32_t rq = a.req_list ? a.req_list[wi] : wi;
        // the next request: round-robin, or (work_ctr) whichever is next when this wavefront is done -- the lists come biggest first
        uint32_t wi_next = wi + gridDim.x;
        if(a.work_ctr) {
            if(lane == 0) wi_next = gridDim.x + atomicAdd(a.work_ctr, 1u);
            wi_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)wi_next);
        }
        const DpRequest R = a.reqs[rq];
        const uint64_t t_req0 = __builtin_readcyclecounter();
        uint64_t t_stage = 0, t_ins = 0;
        uint32_t n_ins = 0, n_walk = 0;
        const uint32_t W = R.w_cols, E = R.n_str + 2;
        Cnt* cnt = reinterpret_cast<Cnt*>(smem);
        uint32_t* lead = reinterpret_cast<uint32_t*>(smem + W * sizeof(Cnt));
        uint32_t* size = lead + E;
        uint32_t* bpos = size + E;                              // padded position of the base row's bases from match[0].start on
        uint32_t* insx = bpos + (R.lq + 2);                     // new columns of the current row: position in old coordinates,
        uint16_t* insg = reinterpret_cast<uint16_t*>(insx + kMaxIns);   // rows showing a gap there,
        uint8_t* inss = reinterpret_cast<uint8_t*>(insg + kMaxIns);     // the row's own symbol
        uint8_t* T = inss + kMaxIns;
        uint8_t* Rw = T + ((W + 3) & ~3u);                      // the current row's symbol per base-row position
        uint8_t* Sbase = Rw + ((W + 3) & ~3u);
        uint8_t* ops = Sbase + ((R.str_cap + 7) & ~3u);              // the cigar as the alignment kernel wrote it: last op first
        __syncthreads();
        const uint8_t* q = a.codes + R.q_off;
        for(uint32_t c = lane; c < W; c += 64) {
            Cnt z = 0;
            if(c < R.lq) { T[c] = q[c]; z = 1ull << (kCntBits * q[c]); }
            cnt[c] = z;
        }
        __syncthreads();
        uint32_t lead_b = 0, size_b = R.lq, n_el = 0, n_rows = 1;
        bool overflow = false;

        DpAlignOut A_next{};
        DpJob J_next{};
        if(R.n_str) { A_next = a.align[R.job_first]; J_next = a.jobs[R.job_first]; }
        for(uint32_t s = 0; s < R.n_str && !overflow; ++s) {
            const DpAlignOut A = A_next;
            const DpJob J = J_next;
            if(s + 1 < R.n_str) { A_next = a.align[R.job_first + s + 1]; J_next = a.jobs[R.job_first + s + 1]; }   // in flight during this row
            if(!A.accept) continue;
            ++n_rows;
            const uint64_t t_s0 = __builtin_readcyclecounter();
            __syncthreads();
            // staged four bytes per lane; the string from the aligned address at or below its start
            const uint32_t so = (uint32_t)(J.s2_off & 3u);
            const uint8_t* S = Sbase + so;
            {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(a.strings + (J.s2_off - so));
                uint32_t* dst = reinterpret_cast<uint32_t*>(Sbase);
                for(uint32_t i = lane, n = (so + J.s2_len + 3) >> 2; i < n; i += 64) dst[i] = src[i];
                const uint32_t* osrc = reinterpret_cast<const uint32_t*>(a.ops + J.ops_off);
                uint32_t* odst = reinterpret_cast<uint32_t*>(ops);
                for(uint32_t i = lane, n = (A.n_ops + 3) >> 2; i < n; i += 64) odst[i] = osrc[i];
            }
            __syncthreads();
            const uint32_t op_last = A.n_ops - 1;                         // forward op j = ops[op_last - j]
            t_stage += __builtin_readcyclecounter() - t_s0;

            // getPaddedPositionOfBase(match[0].start): index in T of the match0_start-th non-gap symbol
            uint32_t ti = 0;
            {
                uint32_t seen = 0;
                const uint32_t want = (uint32_t)A.m0s;
                for(uint32_t base = 0; base < size_b; base += 64) {
                    const uint32_t i = base + lane;
                    const bool ng = i < size_b && T[i] != kGap;
                    const uint64_t m = __ballot(ng);
                    const uint32_t nn = (uint32_t)__builtin_popcountll(m);
                    if(seen + nn > want) {
                        const uint64_t hit = __ballot(ng && mbcnt(m) == want - seen);
                        ti = base + (uint32_t)__builtin_ctzll(hit);
                        break;
                    }
                    seen += nn;
                }
            }
            const uint32_t tl = lead_b;                         // template_leading, read once (:306)
            const uint32_t il = ti + tl;                        // incoming_leading
            bool batched = false;
            if(a.row_batch && A.n_ops > 0 && !(ti == 0 && ops[op_last] == 'I')) {
                const uint32_t ti0 = ti, m1s = (uint32_t)A.m1s;
                // bases from ti0 on, and the row's default symbol per position: '-' over a gap column, unset over a base
                uint32_t nbr = 0;
                for(uint32_t base = ti0; base < size_b; base += 64) {
                    const uint32_t i = base + lane;
                    const bool in = i < size_b;
                    const bool ng = in && T[i] != kGap;
                    const uint64_t m = __ballot(ng);
                    if(ng) { const uint32_t b = nbr + mbcnt(m); if(b < R.lq + 1) bpos[b] = i; }
                    if(in) Rw[i] = ng ? (uint8_t)kUnset : (uint8_t)kGap;
                    nbr += (uint32_t)__builtin_popcountll(m);
                }
                bool bad = nbr > R.lq + 1;
                if(lane == 0 && !bad) bpos[nbr] = size_b;
                __syncthreads();
                uint32_t md_tot = 0, mi_tot = 0, st_tot = 0, run_carry = 0, last_y = 0;
                for(uint32_t cb = 0; cb < A.n_ops && !bad; cb += 64) {
                    const uint32_t j = cb + lane;
                    const bool valid = j < A.n_ops;
                    const uint32_t op = valid ? (uint32_t)ops[op_last - j] : 0u;
                    const bool isM = op == 'M', isD = op == 'D', isI = op == 'I';
                    const uint64_t mMD = __ballot(isM || isD), mMI = __ballot(isM || isI);
                    const uint32_t brel = md_tot + mbcnt(mMD);
                    const uint32_t inc = m1s + mi_tot + mbcnt(mMI);
                    const uint64_t lower = mMD & ((1ull << lane) - 1ull);
                    const uint32_t run_start = lower ? cb + 64u - (uint32_t)__builtin_clzll(lower) : run_carry;
                    const uint32_t i_rank = j - run_start;
                    bool lane_bad = valid && !(isM || isD || isI);
                    uint32_t y = 0, G = 0;
                    bool st = false, fill = false;
                    if((isM || isD) && brel >= nbr) lane_bad = true;
                    if(isI && brel >= nbr) lane_bad = true;                // behind the last base: the walk's job
                    if(!lane_bad && valid) {
                        const uint32_t pb = bpos[brel];
                        if(isI) {
                            if(brel > 0) { const uint32_t pp = bpos[brel - 1]; G = pb - pp - 1; y = pp + 1 + i_rank; }
                            fill = i_rank < G; st = !fill;
                            if(st) y = pb - 1;                              // last position consumed if the cigar ends here
                        } else y = pb;
                    }
                    const uint64_t mS = __ballot(st);
                    const uint32_t k = st_tot + mbcnt(mS);
                    if(st && k >= kMaxIns) lane_bad = true;
                    if(__ballot(lane_bad)) { bad = true; break; }
                    const uint32_t sym = (isM || isI) ? (inc < J.s2_len ? (uint32_t)S[inc] : 0u) : kGap;
                    if(isM || isD || fill) Rw[y] = (uint8_t)sym;
                    if(st) { insx[k] = y + 1; inss[k] = (uint8_t)sym; }
                    md_tot += (uint32_t)__builtin_popcountll(mMD);
                    mi_tot += (uint32_t)__builtin_popcountll(mMI);
                    st_tot += (uint32_t)__builtin_popcountll(mS);
                    if(mMD) run_carry = cb + 64u - (uint32_t)__builtin_clzll(mMD);
                    if(cb + 64 >= A.n_ops) last_y = (uint32_t)__shfl((int)y, (int)((A.n_ops - 1) & 63u));
                }
                __syncthreads();
                if(!bad) {
                    batched = true;
                    const uint32_t nin = st_tot;
                    if(nin) {
                        const uint64_t t_i0 = __builtin_readcyclecounter();
                        n_ins += nin;
                        if(size_b + nin > W || lead_b + size_b + nin > W) { overflow = true; break; }
                        // rows showing a gap in each new column (strictly inside, old coordinates) ...
                        for(uint32_t k0 = 0; k0 < nin; k0 += 64) {
                            const uint32_t k = k0 + lane;
                            const uint32_t xg = k < nin ? insx[k] + tl : 0u;
                            uint32_t g = 0;
                            for(uint32_t e = 0; e < n_el; ++e) { const uint32_t le = lead[e], se = size[e]; g += (le < xg && xg - le < se) ? 1u : 0u; }
                            if(k < nin) insg[k] = (uint16_t)g;
                        }
                        __syncthreads();
                        // ... then every row takes all of them at once
                        for(uint32_t e0 = 0; e0 < n_el; e0 += 64) {
                            const uint32_t e = e0 + lane;
                            if(e < n_el) {
                                const uint32_t le = lead[e], se = size[e];
                                uint32_t before = 0, inside = 0;
                                for(uint32_t k = 0; k < nin; ++k) {
                                    const uint32_t xg = insx[k] + tl;
                                    before += xg <= le ? 1u : 0u;
                                    inside += (le < xg && xg - le < se) ? 1u : 0u;
                                }
                                lead[e] = le + before; size[e] = se + inside;
                            }
                        }
                        // T, cnt and the row's symbols move right by the number of new columns at or before them
                        const uint32_t x0 = insx[0];
                        for(uint32_t hi = size_b; hi > x0;) {
                            const uint32_t n = hi - x0 < 64 ? hi - x0 : 64;
                            const uint32_t yy = hi - 1 - lane;
                            Cnt v = 0; uint8_t t = 0, r = 0; uint32_t sh = 0;
                            if(lane < n) {
                                v = cnt[yy + tl]; t = T[yy]; r = Rw[yy];
                                uint32_t lo = 0, hi2 = nin;                  // upper_bound(insx, yy)
                                while(lo < hi2) { const uint32_t mid = (lo + hi2) >> 1; if(insx[mid] <= yy) lo = mid + 1; else hi2 = mid; }
                                sh = lo;
                            }
                            __syncthreads();
                            if(lane < n && sh) { cnt[yy + sh + tl] = v; T[yy + sh] = t; Rw[yy + sh] = r; }
                            __syncthreads();
                            hi -= n;
                        }
                        for(uint32_t k0 = 0; k0 < nin; k0 += 64) {
                            const uint32_t k = k0 + lane;
                            if(k < nin) {
                                const uint32_t np = insx[k] + k;
                                T[np] = (uint8_t)kGap; Rw[np] = inss[k];
                                cnt[np + tl] = (Cnt)((uint32_t)insg[k] + 1u) << (kCntBits * kGap);
                            }
                        }
                        size_b += nin;
                        __syncthreads();
                        t_ins += __builtin_readcyclecounter() - t_i0;
                    }
                    const uint32_t nout = last_y + 1 - ti0 + nin;
                    if(nout > W) { overflow = true; break; }
                    for(uint32_t z0 = 0; z0 < nout; z0 += 64) {
                        const uint32_t z = z0 + lane;
                        if(z < nout) {
                            const uint32_t sym = Rw[ti0 + z];
                            if(sym <= kGap && ti0 + z + tl < W) cnt[ti0 + z + tl] += 1ull << (kCntBits * sym);
                        }
                    }
                    __syncthreads();
                    if(n_el >= E) { overflow = true; break; }
                    if(lane == 0) { lead[n_el] = il; size[n_el] = nout; }
                    ++n_el;
                    __syncthreads();
                }
            }
            if(batched) continue;
            ++n_walk;
            // The cigar walk is one dependent step per column.  To keep a step at register speed the next 64 bytes of the
            // base row, of the cigar and of the incoming string sit one per lane in a register and are picked with
            // v_readlane; the output symbols are collected the same way and counted into their columns 64 at a time.
            uint32_t inc = (uint32_t)A.m1s, cig = 0, nout = 0;
            uint32_t tb = 0xFFFFFF00u, ob = 0xFFFFFF00u, sb = 0xFFFFFF00u, Tw = 0, Ow = 0, Sw = 0, acc = 0;
            auto getT = [&](uint32_t i) -> uint32_t {
                if(i - tb >= 64u) { tb = i; const uint32_t j = i + lane; Tw = j < size_b ? (uint32_t)T[j] : 0xFFu; }   // past the end: the string's NUL, not a gap
                return (uint32_t)__builtin_amdgcn_readlane((int)Tw, (int)(i - tb));
            };
            auto getO = [&](uint32_t i) -> uint32_t {
                if(i - ob >= 64u) { ob = i; const uint32_t j = i + lane; Ow = j < A.n_ops ? (uint32_t)ops[op_last - j] : 0u; }
                return (uint32_t)__builtin_amdgcn_readlane((int)Ow, (int)(i - ob));
            };
            auto getS = [&](uint32_t i) -> uint32_t {
                if(i - sb >= 64u) { sb = i; const uint32_t j = i + lane; Sw = j < J.s2_len ? (uint32_t)S[j] : 0u; }
                return (uint32_t)__builtin_amdgcn_readlane((int)Sw, (int)(i - sb));
            };
            auto flush = [&](uint32_t first, uint32_t n) {                  // outputs first .. first + n land in columns il + first ..
                const uint32_t col = il + first + lane;
                if(lane < n && col < W) cnt[col] += 1ull << (kCntBits * acc);
            };
            while(cig < A.n_ops) {
                ti = (uint32_t)__builtin_amdgcn_readfirstlane((int)ti);
                cig = (uint32_t)__builtin_amdgcn_readfirstlane((int)cig);
                inc = (uint32_t)__builtin_amdgcn_readfirstlane((int)inc);
                const uint32_t tsym = getT(ti);
                const uint32_t op = getO(cig);
                uint32_t sym;
                if(tsym == kGap) {
                    if(op == 'I') { sym = getS(inc); ++inc; ++cig; }
                    else sym = kGap;
                } else if(op == 'M') { sym = getS(inc); ++inc; ++cig; }
                else if(op == 'D') { sym = kGap; ++cig; }
                else {
                    // insertGapBeforeColumn(template_index + template_leading) on every row (:1205-1211, :165-180)
                    const uint64_t t_i0 = __builtin_readcyclecounter();
                    ++n_ins;
                    const uint32_t c = ti + tl;
                    uint32_t ngap = 0;
                    for(uint32_t e0 = 0; e0 < n_el; e0 += 64) {
                        const uint32_t e = e0 + lane;
                        bool inside = false;
                        if(e < n_el) {
                            const uint32_t le = lead[e], se = size[e];
                            if(c <= le) lead[e] = le + 1;
                            else if(c - le < se) { size[e] = se + 1; inside = true; }
                        }
                        ngap += (uint32_t)__builtin_popcountll(__ballot(inside));
                    }
                    const uint32_t end_b = lead_b + size_b;                  // tracked columns: [0, end_b)
                    bool base_inside = false;
                    if(c <= lead_b) lead_b += 1;
                    else if(c - lead_b < size_b) {
                        base_inside = true;
                        const uint32_t ip = c - lead_b;
                        if(size_b + 1 > W) { overflow = true; break; }
                        for(uint32_t hi = size_b; hi > ip;) {                // T.insert(ip, '-')
                            const uint32_t n = hi - ip < 64 ? hi - ip : 64;
                            const uint32_t i = hi - 1 - lane;
                            uint8_t v = 0;
                            if(lane < n) v = T[i];
                            __syncthreads();
                            if(lane < n) T[i + 1] = v;
                            __syncthreads();
                            hi -= n;
                        }
                        if(lane == 0) T[ip] = (uint8_t)kGap;
                        size_b += 1;
                        tb = 0xFFFFFF00u;                                    // the register window of T is stale
                    }
                    if(c < end_b) {                                          // columns at / after c move right
                        if(end_b + 1 > W) { overflow = true; break; }
                        for(uint32_t hi = end_b; hi > c;) {
                            const uint32_t n = hi - c < 64 ? hi - c : 64;
                            const uint32_t i = hi - 1 - lane;
                            Cnt v = 0;
                            if(lane < n) v = cnt[i];
                            __syncthreads();
                            if(lane < n) cnt[i + 1] = v;
                            __syncthreads();
                            hi -= n;
                        }
                    }
                    if(c < W && lane == 0) cnt[c] = (Cnt)(ngap + (base_inside ? 1u : 0u)) << (kCntBits * kGap);
                    __syncthreads();
                    t_ins += __builtin_readcyclecounter() - t_i0;
                    sym = getS(inc); ++inc; ++cig;
                }
                if(nout >= W) { overflow = true; break; }
                acc = lane == (nout & 63u) ? sym : acc;
                ++nout;
                if((nout & 63u) == 0) flush(nout - 64, 64);
                ++ti;
            }
            if(overflow) break;
            if((nout & 63u) != 0) flush(nout & ~63u, nout & 63u);
            __syncthreads();
            if(n_el >= E) { overflow = true; break; }
            if(lane == 0) { lead[n_el] = il; size[n_el] = nout; }
            ++n_el;
            __syncthreads();
        }

        // calculateBaseConsensus(min_call_coverage, -1) over the base row's columns
        uint32_t cons_len = 0;
        uint8_t* cons = a.cons + R.cons_off;
        if(!overflow) {
            for(uint32_t base = 0; base < size_b; base += 64) {
                const uint32_t i = base + lane;
                uint32_t sym = kGap;
                if(i < size_b && lead_b + i < W) {
                    const Cnt v = cnt[lead_b + i];
                    int max_count = -1; uint32_t max_symbol = 0;
#pragma unroll
                    for(uint32_t x = 0; x < 5; ++x)                         // "ACGT" then '-' ('N' never counted)
                        if((int)cnt_get(v, x) > max_count) { max_count = (int)cnt_get(v, x); max_symbol = x; }
                    const uint32_t base_symbol = T[i];
                    const int base_count = (int)cnt_get(v, base_symbol);
                    sym = (max_count >= base_count && base_count < R.min_call_coverage) ? max_symbol : base_symbol;
                }
                const bool keep = i < size_b && sym != kGap;
                const uint64_t m = __ballot(keep);
                if(keep) {
                    const uint32_t pos = cons_len + mbcnt(m);
                    if(pos < R.cons_cap) cons[pos] = (uint8_t)sym;
                }
                cons_len += (uint32_t)__builtin_popcountll(m);
            }
            if(cons_len > R.cons_cap) { overflow = true; cons_len = 0xFFFFFFFFu; }
        }
        if(lane == 0) {
            DpMsaOut o;
            o.n_rows = n_rows; o.cons_len = overflow ? 0 : cons_len; o.error = !overflow ? 0u : cons_len == 0xFFFFFFFFu ? 2u : 1u; o.pad = n_walk;
            o.kc_total = (uint32_t)((__builtin_readcyclecounter() - t_req0) >> 10); o.kc_stage = (uint32_t)(t_stage >> 10);
            o.kc_insert = (uint32_t)(t_ins >> 10); o.n_insert = n_ins;
            a.msa[rq] = o;
        }
        wi = wi_next;
    }
}

uint32_t dp_msa_waves(const DpPipeArgs& a, bool global)
{
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint32_t per_cu = global ? 8u : (a.lds_bytes ? (160u * 1024u) / a.lds_bytes : 16u);
    per_cu = per_cu < 1 ? 1 : per_cu > 32 ? 32 : per_cu;
    uint32_t n_waves = (uint32_t)cus * per_cu;
    const uint32_t n_work = a.req_list ? a.n_list : a.n_reqs;
    return n_waves > n_work ? n_work : n_waves;
}

hipError_t launch_dp_msa(const DpPipeArgs& a, hipStream_t stream)
{
    const uint32_t n_work = a.req_list ? a.n_list : a.n_reqs;
    if(n_work == 0) return hipSuccess;
    if(a.msa_ws) {                                            // global-workspace variant: a.lds_bytes = bytes per workgroup slot
        hipLaunchKernelGGL(dp_msa_kernel<true>, dim3(dp_msa_waves(a, true)), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    if(a.lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if(a.lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dp_msa_kernel<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds_bytes);
        if(e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(dp_msa_kernel<false>, dim3(dp_msa_waves(a, false)), dim3(64), a.lds_bytes, stream, a);
    return hipGetLastError();
}

} // namespace lrsc
