// bwt_build.hip -- multi-string BWT construction on the GPU (the `stride index` step that feeds
// the correction path; reference: StriDe/index.cpp:164-213 -> BWTCA::runRopebwt2,
// SuffixTools/BWTCARopebwt.cpp:160-247).
//
// The reference inserts reads one batch at a time into a rope (ropebwt2, BCR).  On an MI355X
// the whole suffix array of a read set fits in HBM, so the BWT is built by sorting:
//   1. text T = read_0 $ read_1 $ ... (3-bit codes $=0 A=1 C=2 G=3 T=4)
//   2. key(i) = first 21 symbols of suffix i, cut after its sentinel -> one stable LSD radix sort
//      (hipCUB DeviceRadixSort, 63 key bits) of (key, i)
//   3. groups of equal keys that did not reach a sentinel are refined with the next 21 symbols
//      (two stable sorts on the unresolved subset only), repeated until none is left
//   4. ties that contain the sentinel are ordered by text position == read order, which is
//      exactly ropebwt2's MR_SO_IO rule (BWTCARopebwt.cpp:167): stable sorting keeps it for free
//   5. BWT[j] = T[SA[j]-1]
// Output is byte-identical to the reference's .bwt/.rbwt payload (tests/test_bwt_build.py).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lrsc.h"

namespace lrsc {

static constexpr uint32_t kSymsPerKey = 21;

struct MaxOp {
    __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

// T[toff[r] + p] = code of base p of read r (reversed if rev), T[toff[r] + len] = 0
__global__ __launch_bounds__(256) void text_kernel(const char* __restrict__ ascii, const uint64_t* __restrict__ off,
                                                   uint32_t n_reads, uint64_t N, int rev, uint8_t* __restrict__ T, int* bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= N) return;
    // read r starts at off[r] + r in T: largest r with off[r] + r <= i
    uint32_t lo = 0, hi = n_reads;
    while(hi - lo > 1) {
        const uint32_t m = lo + ((hi - lo) >> 1);
        if(off[m] + m <= i) lo = m; else hi = m;
    }
    const uint64_t start = off[lo] + lo;
    const uint64_t len = off[lo + 1] - off[lo];
    const uint64_t p = i - start;
    uint8_t code = 0;
    if(p < len) {
        const uint8_t c = (uint8_t)ascii[off[lo] + (rev ? (len - 1 - p) : p)];
        const uint32_t x = (c >> 1) & 3u;
        code = (uint8_t)((x ^ (x >> 1)) + 1u);
        if(c != 'A' && c != 'C' && c != 'G' && c != 'T') *bad = 1;
    }
    T[i] = code;
}

__device__ __forceinline__ uint64_t pack_key(const uint8_t* __restrict__ T, uint64_t N, uint64_t pos)
{
    uint64_t key = 0;
    bool ended = false;
#pragma unroll
    for(uint32_t s = 0; s < kSymsPerKey; ++s) {
        uint32_t c = 0;
        if(!ended && pos + s < N) c = T[pos + s];
        if(c == 0) ended = true;
        key = (key << 3) | c;
    }
    return key;
}

__device__ __forceinline__ bool key_has_zero(uint64_t key)
{
    // any 3-bit field == 0 among the 21 fields
    const uint64_t m = 0x1249249249249249ull;                 // bit 0 of every field
    const uint64_t any = (key | (key >> 1) | (key >> 2)) & m;  // field != 0
    return any != m;
}

__global__ __launch_bounds__(256) void init_keys_kernel(const uint8_t* __restrict__ T, uint64_t N,
                                                        uint64_t* __restrict__ key, uint32_t* __restrict__ sa)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= N) return;
    key[i] = pack_key(T, N, i);
    sa[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void mark_kernel(const uint64_t* __restrict__ key, uint64_t N, uint8_t* __restrict__ bnd)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(j >= N) return;
    if(j == 0 || key[j] != key[j - 1]) bnd[j] = 1;
}

__global__ __launch_bounds__(256) void unresolved_kernel(const uint64_t* __restrict__ key, const uint8_t* __restrict__ bnd,
                                                         uint64_t N, uint8_t* __restrict__ flag, uint32_t* __restrict__ head)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(j >= N) return;
    const bool single = bnd[j] && (j + 1 == N || bnd[j + 1]);
    flag[j] = (!single && !key_has_zero(key[j])) ? 1 : 0;
    head[j] = bnd[j] ? (uint32_t)j : 0u;
}

__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* __restrict__ T, uint64_t N, uint64_t depth,
                                                     const uint32_t* __restrict__ U, uint32_t n_u,
                                                     const uint32_t* __restrict__ sa, const uint32_t* __restrict__ headscan,
                                                     uint64_t* __restrict__ key2, uint32_t* __restrict__ gid,
                                                     uint32_t* __restrict__ val, uint32_t* __restrict__ perm)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if(k >= n_u) return;
    const uint32_t j = U[k];
    const uint32_t s = sa[j];
    key2[k] = pack_key(T, N, (uint64_t)s + depth);
    gid[k] = headscan[j];
    val[k] = s;
    perm[k] = k;
}

__global__ __launch_bounds__(256) void permute_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm,
                                                          uint32_t n, uint32_t* __restrict__ dst)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if(k < n) dst[k] = src[perm[k]];
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t* __restrict__ p, uint32_t n)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if(k < n) p[k] = k;
}

// p2[k]: rank in sort-1 order of the element that belongs at U[k]; p1[rank]: its gathered index
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t* __restrict__ U, uint32_t n_u,
                                                      const uint32_t* __restrict__ p2, const uint32_t* __restrict__ p1,
                                                      const uint32_t* __restrict__ val, const uint64_t* __restrict__ key_sorted,
                                                      uint32_t* __restrict__ sa, uint64_t* __restrict__ key)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if(k >= n_u) return;
    const uint32_t j = U[k];
    const uint32_t r1 = p2[k];
    sa[j] = val[p1[r1]];
    key[j] = key_sorted[r1];
}

__global__ __launch_bounds__(256) void bwt_kernel(const uint8_t* __restrict__ T, const uint32_t* __restrict__ sa, uint64_t N,
                                                  uint8_t* __restrict__ bwt)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(j >= N) return;
    const uint32_t s = sa[j];
    bwt[j] = T[s == 0 ? N - 1 : (uint64_t)s - 1];
}

struct Dev {
    std::vector<void*> ptrs;
    ~Dev() { for(void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t n)
    {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
        if(e == hipSuccess) { ptrs.push_back(q); *p = static_cast<T*>(q); }
        return e;
    }
};

static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

#define BB_TRY(expr)                                                                 \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if(_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return LRSC_ERR_DEVICE; } \
    } while(0)

// Builds the BWT of the read set (or of the reversed reads) on `device`; bwt_out receives N codes
// 0..4 ($ACGT).  N = total bases + n_reads must be < 2^32.
int build_bwt_device(const char* reads, const uint64_t* off, uint32_t n_reads, int reverse_reads, int device,
                     std::vector<uint8_t>& bwt_out, uint32_t* rounds_out, std::string& err)
{
    const uint64_t total = off[n_reads];
    const uint64_t N = total + n_reads;
    if(N >= (1ull << 32)) { err = "GPU BWT builder currently handles < 2^32 symbols per strand"; return LRSC_ERR_UNSUPPORTED; }
    BB_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;   // default stream: this is a one-off setup step
    Dev d;
    char* d_ascii; uint64_t* d_off; uint8_t* d_T; int* d_bad;
    BB_TRY(d.alloc(&d_ascii, total));
    BB_TRY(d.alloc(&d_off, (size_t)n_reads + 1));
    BB_TRY(d.alloc(&d_T, N + 64));
    BB_TRY(d.alloc(&d_bad, 1));
    BB_TRY(hipMemset(d_bad, 0, sizeof(int)));
    BB_TRY(hipMemset(d_T + N, 0, 64));
    BB_TRY(hipMemcpy(d_ascii, reads, total, hipMemcpyHostToDevice));
    BB_TRY(hipMemcpy(d_off, off, ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(text_kernel, dim3(nblk(N)), dim3(256), 0, st, d_ascii, d_off, n_reads, N, reverse_reads, d_T, d_bad);
    BB_TRY(hipGetLastError());
    int bad = 0;
    BB_TRY(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
    if(bad) { err = "sequence contains a base other than A,C,G,T"; return LRSC_ERR_ARG; }

    uint64_t *d_key[2]; uint32_t *d_sa[2];
    BB_TRY(d.alloc(&d_key[0], N)); BB_TRY(d.alloc(&d_key[1], N));
    BB_TRY(d.alloc(&d_sa[0], N));  BB_TRY(d.alloc(&d_sa[1], N));
    hipLaunchKernelGGL(init_keys_kernel, dim3(nblk(N)), dim3(256), 0, st, d_T, N, d_key[0], d_sa[0]);
    BB_TRY(hipGetLastError());

    // 1. one big stable sort on the first 21 symbols
    hipcub::DoubleBuffer<uint64_t> kb(d_key[0], d_key[1]);
    hipcub::DoubleBuffer<uint32_t> vb(d_sa[0], d_sa[1]);
    size_t tmp_bytes = 0;
    BB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kb, vb, (int64_t)N, 0, 63, st));
    uint8_t* d_tmp; size_t tmp_cap = tmp_bytes;
    BB_TRY(d.alloc(&d_tmp, tmp_cap));
    BB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, kb, vb, (int64_t)N, 0, 63, st));
    uint64_t* key = kb.Current();
    uint32_t* sa = vb.Current();
    // the alternate buffers are free scratch from here on
    uint64_t* key_alt = kb.Alternate();
    uint32_t* sa_alt = vb.Alternate();

    uint8_t* d_bnd; uint8_t* d_flag; uint32_t* d_head; uint32_t* d_U; uint32_t* d_nsel;
    BB_TRY(d.alloc(&d_bnd, N)); BB_TRY(d.alloc(&d_flag, N)); BB_TRY(d.alloc(&d_head, N));
    BB_TRY(d.alloc(&d_U, N)); BB_TRY(d.alloc(&d_nsel, 1));
    BB_TRY(hipMemset(d_bnd, 0, N));

    // scratch sized for the first (largest) unresolved subset; allocated lazily
    uint64_t *u_key[2] = {nullptr, nullptr};
    uint32_t *u_gid[2] = {nullptr, nullptr}, *u_perm[2] = {nullptr, nullptr}, *u_p2[2] = {nullptr, nullptr}, *u_val = nullptr;
    uint32_t u_cap = 0;

    auto ensure_tmp = [&](size_t need) -> hipError_t {
        if(need <= tmp_cap) return hipSuccess;
        hipError_t e = d.alloc(&d_tmp, need);
        if(e == hipSuccess) tmp_cap = need;
        return e;
    };

    uint32_t rounds = 0;
    for(uint64_t depth = kSymsPerKey;; depth += kSymsPerKey) {
        hipLaunchKernelGGL(mark_kernel, dim3(nblk(N)), dim3(256), 0, st, key, N, d_bnd);
        hipLaunchKernelGGL(unresolved_kernel, dim3(nblk(N)), dim3(256), 0, st, key, d_bnd, N, d_flag, d_head);
        BB_TRY(hipGetLastError());
        // compact the unresolved SA positions
        size_t need = 0;
        hipcub::CountingInputIterator<uint32_t> iota(0);
        BB_TRY(hipcub::DeviceSelect::Flagged(nullptr, need, iota, d_flag, d_U, d_nsel, (int64_t)N, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceSelect::Flagged(d_tmp, need, iota, d_flag, d_U, d_nsel, (int64_t)N, st));
        uint32_t n_u = 0;
        BB_TRY(hipMemcpy(&n_u, d_nsel, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if(n_u == 0) break;
        ++rounds;
        if(depth > (1ull << 22)) { err = "BWT refinement did not converge"; return LRSC_ERR_UNSUPPORTED; }
        // group id of every position = position of the closest boundary at or before it
        BB_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, need, d_head, sa_alt, MaxOp(), (int64_t)N, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceScan::InclusiveScan(d_tmp, need, d_head, sa_alt, MaxOp(), (int64_t)N, st));
        if(n_u > u_cap) {
            u_cap = n_u;
            for(int b = 0; b < 2; ++b) {
                BB_TRY(d.alloc(&u_key[b], u_cap));
                BB_TRY(d.alloc(&u_gid[b], u_cap));
                BB_TRY(d.alloc(&u_perm[b], u_cap));
                BB_TRY(d.alloc(&u_p2[b], u_cap));
            }
            BB_TRY(d.alloc(&u_val, u_cap));
        }
        hipLaunchKernelGGL(gather_kernel, dim3(nblk(n_u)), dim3(256), 0, st, d_T, N, depth, d_U, n_u, sa, sa_alt,
                           u_key[0], u_gid[0], u_val, u_perm[0]);
        BB_TRY(hipGetLastError());
        // sort 1 (stable): by the next 21 symbols.  p1[k] = gathered index of the k-th smallest key.
        hipcub::DoubleBuffer<uint64_t> k2(u_key[0], u_key[1]);
        hipcub::DoubleBuffer<uint32_t> p1(u_perm[0], u_perm[1]);
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k2, p1, (int)n_u, 0, 63, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, need, k2, p1, (int)n_u, 0, 63, st));
        // sort 2 (stable): by group id, payload = rank in sort-1 order; inside a group the
        // sort-1 order (next 21 symbols, then text position) is preserved.
        hipLaunchKernelGGL(permute_u32_kernel, dim3(nblk(n_u)), dim3(256), 0, st, u_gid[0], p1.Current(), n_u, u_gid[1]);
        hipLaunchKernelGGL(iota_kernel, dim3(nblk(n_u)), dim3(256), 0, st, u_p2[0], n_u);
        BB_TRY(hipGetLastError());
        hipcub::DoubleBuffer<uint32_t> gd(u_gid[1], u_gid[0]);
        hipcub::DoubleBuffer<uint32_t> p2(u_p2[0], u_p2[1]);
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, gd, p2, (int)n_u, 0, 32, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, need, gd, p2, (int)n_u, 0, 32, st));
        // U is ascending and groups are contiguous, so the k-th element of the (group, key2) order
        // belongs at SA position U[k].
        hipLaunchKernelGGL(scatter_kernel, dim3(nblk(n_u)), dim3(256), 0, st, d_U, n_u, p2.Current(), p1.Current(), u_val,
                           k2.Current(), sa, key);
        BB_TRY(hipGetLastError());
    }
    (void)key_alt;

    uint8_t* d_bwt = reinterpret_cast<uint8_t*>(key_alt);   // N bytes fit in the idle key buffer
    hipLaunchKernelGGL(bwt_kernel, dim3(nblk(N)), dim3(256), 0, st, d_T, sa, N, d_bwt);
    BB_TRY(hipGetLastError());
    bwt_out.resize(N);
    BB_TRY(hipMemcpy(bwt_out.data(), d_bwt, N, hipMemcpyDeviceToHost));
    if(rounds_out) *rounds_out = rounds;
    return LRSC_OK;
}

} // namespace lrsc
