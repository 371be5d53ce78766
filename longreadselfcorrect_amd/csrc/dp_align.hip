// dp_align.hip -- Overlapper::extendMatch (Thirdparty/overlapper.cpp:421-701) on the device: banded
// semi-global DP (band 201 around the seed diagonal, scores +1 / -1 / -8 at the call site,
// LongReadOverlap.cpp:635-643) plus the homopolymer-aware traceback, one WAVEFRONT per alignment.
//
// Fill: columns (s1 = the query) are processed in order; the <= 255 cells of a band column live four per
// lane in registers.  The in-column dependency cell[j] = max(A[j], cell[j-1] + gap) is a prefix maximum of
// A[j] - gap*j, so a column costs one wave scan instead of a serial chain.  The reference's quirks are kept:
// the first computed row of a column ignores "up", the last computed row ignores "left" (:476,:506-512),
// never-written cells read as 0.
// Traceback: which neighbour the traceback takes at a cell (:604-661) depends only on that cell (its
// score, the three neighbour scores as _getBandedCellScore sees them, and the two homopolymer tests), so
// the fill stores the decision itself -- 2 bits per cell, one 64-byte line per column -- and the traceback
// just follows them, 16 columns per fetch.
#include <hip/hip_runtime.h>

#include "dp_dev.h"

namespace lrsc {

namespace {
constexpr int kNeg = -(1 << 29);
constexpr int kIntMin = -2147483647 - 1;
constexpr uint32_t kS2Pad = 264;

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// DPP moves (GFX9 encodings): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shl:1 = 0x130,
// wave_shr:1 = 0x138.  Lanes without a source (and rows masked out) receive `old`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xF, false); }

// inclusive prefix max over the wave (lane order): Kogge-Stone inside each row of 16, then two row broadcasts
__device__ __forceinline__ int wave_prefix_max(int x)
{
    x = imax(x, dpp<0x111>(kNeg, x));
    x = imax(x, dpp<0x112>(kNeg, x));
    x = imax(x, dpp<0x114>(kNeg, x));
    x = imax(x, dpp<0x118>(kNeg, x));
    x = imax(x, dpp<0x142, 0xA>(kNeg, x));
    x = imax(x, dpp<0x143, 0xC>(kNeg, x));
    return x;
}
__device__ __forceinline__ int lane_below(int old, int x) { return dpp<0x138>(old, x); }     // value of lane - 1
__device__ __forceinline__ int lane_above(int old, int x) { return dpp<0x130>(old, x); }     // value of lane + 1
} // namespace

template <bool GLOBAL>
__global__ __launch_bounds__(64) void dp_align_kernel(DpAlignArgs a)
{
    extern __shared__ uint8_t lds[];
    // the two sequences are staged in LDS, or -- for the few alignments beyond it -- in this wavefront's slice of a global workspace
    uint8_t* smem = GLOBAL ? a.seq_ws + (uint64_t)blockIdx.x * a.seq_ws_stride : lds;
    uint8_t* S1 = smem;                                        // s1 codes, S1[L1] = 4 (the string's NUL)
    // s2 codes at S2[0 .. L2), sentinel 4 in the kS2Pad bytes before and the 16 after: a lane's five characters
    // s2[j-1 .. j+3] are then two aligned dword reads for any band position
    uint8_t* S2buf = smem + ((a.max_s1 + 2 + 3) & ~3u);
    uint8_t* S2 = S2buf + kS2Pad;
    for(uint32_t i = threadIdx.x; i < kS2Pad; i += 64) S2buf[i] = 4;
    const uint32_t lane = threadIdx.x;
    uint8_t* trace = a.trace + (uint64_t)blockIdx.x * a.trace_stride;
    const int half = (int)a.band_width / 2;
    const int bw = 2 * half + 1;
    const int g = a.gap_penalty, MS = a.match_score, MX = a.mismatch_penalty;
    const int r0 = 4 * (int)lane;

    for(uint32_t job = blockIdx.x; job < a.n_jobs; job += gridDim.x) {
        const DpJob J = a.jobs[job];
        DpAlignOut o;
        o.m0s = 0; o.m0e = -1; o.m1s = 0; o.m1e = -1; o.score = -1; o.edit_distance = -1; o.total_columns = -1; o.n_ops = 0;
        o.accept = 0; o.skipped = 1; o.t_fill = 0; o.t_trace = 0;
        const uint64_t t_job0 = __builtin_readcyclecounter();
        if(J.s1_len == 0 || J.s2_len == 0) {
            if(lane == 0) a.out[job] = o;
            continue;
        }
        const int L1 = (int)J.s1_len, L2 = (int)J.s2_len;
        {
            // an alignment beyond the LDS stage belongs to the global-workspace launch (and only those do)
            // (classified by the owning request's capacities, as the host sized the two launches: a long query with a short retrieved
            //  string is still long)
            const bool is_long = a.reqs ? dp_align_stage_bytes(a.reqs[J.req].lq, a.reqs[J.req].str_cap) > a.lds_cap
                                        : dp_align_stage_bytes(J.s1_len, J.s2_len) > a.lds_cap;
            if(GLOBAL ? (a.only_long && !is_long) : is_long) continue;
        }
        __syncthreads();
        if(GLOBAL) __threadfence_block();
        for(int i = (int)lane; i <= L1; i += 64) S1[i] = i < L1 ? a.codes[J.s1_off + i] : (uint8_t)4;
        for(int j = (int)lane; j < L2 + 16; j += 64) S2[j] = j < L2 ? a.strings[J.s2_off + j] : (uint8_t)4;
        __syncthreads();
        if(GLOBAL) __threadfence_block();
        if(J.mode != 0 && L2 >= L1) {                              // identical sequence from the forward / backward extension
            const int shift = J.mode == 1 ? 0 : L2 - L1;
            bool diff = false;
            for(int i = (int)lane; i < L1; i += 64) diff = diff || S1[i] != S2[shift + i];
            if(__ballot(diff) == 0) {
                if(lane == 0) a.out[job] = o;
                continue;
            }
        }
        o.skipped = 0;

        const int origin = (J.start2 - J.start1 + 1) - (half + 1);
        const int num_rows = L2 + 1;
        int prev[4] = {0, 0, 0, 0};
        int best_row_val = kIntMin, best_row_i = 0;

        for(int i = 1; i <= L1; ++i) {
            const int jbase = origin + i;
            const int j_lo = jbase < 1 ? 1 : jbase;
            const int j_hi = jbase + bw > num_rows ? num_rows : jbase + bw;
            const bool skipcol = j_hi <= 0 || j_lo >= num_rows || j_lo >= j_hi;
            int cur[4] = {0, 0, 0, 0};
            uint32_t flags = 0;
            const int prev_next = lane_above(0, prev[0]);
            if(!skipcol) {
                const uint32_t c1 = S1[i - 1];
                const bool h1 = c1 == S1[i];
                uint32_t s2c[5];
                {
                    int idx = jbase + r0 - 1;                               // >= -kS2Pad whenever the column is computed
                    idx = idx > L2 + 8 ? L2 + 8 : idx;                      // rows past the end are out of range anyway
                    const uint32_t b = (uint32_t)(idx + (int)kS2Pad);
                    const uint32_t* p32 = reinterpret_cast<const uint32_t*>(S2buf + (b & ~3u));
                    const unsigned long long v = (((unsigned long long)p32[1] << 32) | p32[0]) >> (8u * (b & 3u));
#pragma unroll
                    for(int t = 0; t < 5; ++t) s2c[t] = (uint32_t)(v >> (8 * t)) & 0xFFu;
                }
                int diag[4], leftg[4], B[4];
                bool inr[4], left_in[4];
#pragma unroll
                for(int t = 0; t < 4; ++t) {
                    const int r = r0 + t, j = jbase + r;
                    inr[t] = j >= j_lo && j < j_hi;
                    diag[t] = prev[t] + (c1 == s2c[t] ? MS : MX);
                    left_in[t] = r + 1 < bw;
                    leftg[t] = (t < 3 ? prev[t + 1] : prev_next) + g;
                    int A;
                    if(j == j_lo) A = imax(left_in[t] ? leftg[t] : kNeg, diag[t]);
                    else if(j == j_hi - 1) A = diag[t];
                    else A = imax(diag[t], leftg[t]);
                    B[t] = inr[t] ? A - g * r : kNeg;
                }
                const int p0 = B[0], p1 = imax(p0, B[1]), p2 = imax(p1, B[2]), p3 = imax(p2, B[3]);
                const int incl = wave_prefix_max(p3);
                const int excl = lane_below(kNeg, incl);
                const int P[4] = {imax(p0, excl), imax(p1, excl), imax(p2, excl), imax(p3, excl)};
#pragma unroll
                for(int t = 0; t < 4; ++t) cur[t] = inr[t] ? P[t] + g * (r0 + t) : 0;
                const int below = lane_below(0, cur[3]);            // band row r0 - 1 of this column
#pragma unroll
                for(int t = 0; t < 4; ++t) {
                    const int r = r0 + t;
                    const int curr = cur[t];
                    const bool eq_diag = curr == diag[t];
                    const bool eq_up = r >= 1 && curr == (t > 0 ? cur[t - 1] : below) + g;
                    const bool eq_left = left_in[t] && curr == leftg[t];
                    const bool h2 = s2c[t] == s2c[t + 1];
                    uint32_t dir;                                   // 0 = M, 2 = I, 3 = D
                    if(h2) dir = eq_up ? 2u : eq_left ? 3u : 0u;
                    else if(h1) dir = eq_left ? 3u : eq_up ? 2u : 0u;
                    else dir = eq_diag ? 0u : eq_left ? 3u : 2u;
                    if(dir == 0u && c1 != s2c[t]) dir = 1u;         // M over a mismatch
                    flags |= inr[t] ? dir << (2 * t) : 0u;
                }
            }
            trace[(uint64_t)i * kDpTraceStride + lane] = (uint8_t)flags;
            // last row (j = L2) of this column, if it is inside the band
            {
                const int r = L2 - jbase;
                if(r >= r0 && r < r0 + 4 && r >= 0 && r < bw) {
                    const int v = cur[r - r0];
                    if(v > best_row_val) { best_row_val = v; best_row_i = i; }
                }
            }
#pragma unroll
            for(int t = 0; t < 4; ++t) prev[t] = cur[t];
        }

        // best of the last column (rows ascending, first maximum wins) and of the last row (columns ascending)
        int best_col_val = kIntMin, best_col_j = 0;
        {
            const int jbase = origin + L1;
#pragma unroll
            for(int t = 0; t < 4; ++t) {
                const int r = r0 + t, j = jbase + r;
                if(r < bw && j >= 1 && j <= L2 && prev[t] > best_col_val) { best_col_val = prev[t]; best_col_j = j; }
            }
        }
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) {
            const int ov = __shfl_xor(best_col_val, d), oj = __shfl_xor(best_col_j, d);
            if(ov > best_col_val || (ov == best_col_val && ov != kIntMin && oj < best_col_j)) { best_col_val = ov; best_col_j = oj; }
            const int rv = __shfl_xor(best_row_val, d), ri = __shfl_xor(best_row_i, d);
            if(rv > best_row_val || (rv == best_row_val && rv != kIntMin && ri < best_row_i)) { best_row_val = rv; best_row_i = ri; }
        }
        int ti, tj;
        if(best_col_val > best_row_val) { ti = L1; tj = best_col_j; o.score = best_col_val; }
        else { ti = best_row_i; tj = L2; o.score = best_row_val; }
        ti = __builtin_amdgcn_readfirstlane(ti);
        tj = __builtin_amdgcn_readfirstlane(tj);
        o.m0e = ti - 1; o.m1e = tj - 1;
        o.edit_distance = 0; o.total_columns = 0;

        // ---- traceback ----------------------------------------------------------------------------------------
        const uint64_t t_job1 = __builtin_readcyclecounter();
        __threadfence();
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        int wb = -1;                                                // first column of the 16-column window held in w0..w3
        uint32_t acc = 0, n_ops = 0;
        uint8_t* ops = a.ops + J.ops_off;
        bool bad = false;
        while(ti > 0 && tj > 0) {
            if((ti & ~15) != wb) {
                wb = ti & ~15;
                const uint32_t* src = reinterpret_cast<const uint32_t*>(trace + (uint64_t)(wb + (int)(lane >> 2)) * kDpTraceStride + (lane & 3u) * 16u);
                w0 = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w3 = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int r = tj - origin - ti;
            if(r < 0 || r >= bw) { bad = true; break; }
            const uint32_t d = ((uint32_t)r >> 4) & 3u;
            const uint32_t word = d == 0 ? w0 : d == 1 ? w1 : d == 2 ? w2 : w3;
            const uint32_t src_lane = ((uint32_t)(ti - wb) << 2) | ((uint32_t)r >> 6);
            const uint32_t code = ((uint32_t)__builtin_amdgcn_readlane((int)word, (int)src_lane) >> (2 * ((uint32_t)r & 15u))) & 3u;
            uint32_t op;
            if(code <= 1u) { op = 'M'; o.edit_distance += (int)code; --ti; --tj; }
            else if(code == 2u) { op = 'I'; o.edit_distance += 1; --tj; }
            else { op = 'D'; o.edit_distance += 1; --ti; }
            o.total_columns += 1;
            acc = lane == (n_ops & 63u) ? op : acc;
            ++n_ops;
            if((n_ops & 63u) == 0) ops[n_ops - 64 + lane] = (uint8_t)acc;
        }
        if((n_ops & 63u) != 0 && lane < (n_ops & 63u)) ops[(n_ops & ~63u) + lane] = (uint8_t)acc;
        o.m0s = ti; o.m1s = tj;
        o.n_ops = bad ? 0xFFFFFFFFu : n_ops;
        o.t_fill = (uint32_t)(t_job1 - t_job0); o.t_trace = (uint32_t)(__builtin_readcyclecounter() - t_job1);
        if(a.reqs && !bad) {
            const DpRequest& R = a.reqs[J.req];
            const bool bPassedOverlap = (uint64_t)(int64_t)o.total_columns >= (uint64_t)R.min_overlap;
            const double pid = (double)(o.total_columns - o.edit_distance) * 100.0f / o.total_columns;      // getPercentIdentity
            o.accept = (bPassedOverlap && pid / 100 >= R.min_identity) ? 1u : 0u;
        }
        if(lane == 0) a.out[job] = o;
    }
}

hipError_t launch_dp_align(const DpAlignArgs& a, uint32_t n_waves, hipStream_t stream)
{
    if(a.n_jobs == 0) return hipSuccess;
    if(a.band_width < 2 || (a.band_width / 2) * 2 + 1 > kDpMaxBand) return hipErrorInvalidValue;
    const size_t lds = dp_align_stage_bytes(a.max_s1, a.max_s2);
    if(n_waves > a.n_jobs) n_waves = a.n_jobs;
    if(a.seq_ws) {
        if(a.seq_ws_stride < lds) return hipErrorInvalidValue;
        hipLaunchKernelGGL(dp_align_kernel<true>, dim3(n_waves), dim3(64), 0, stream, a);
    } else {
        if(lds > kDpAlignLdsCap) return hipErrorInvalidValue;
        hipLaunchKernelGGL(dp_align_kernel<false>, dim3(n_waves), dim3(64), lds, stream, a);
    }
    return hipGetLastError();
}

} // namespace lrsc
