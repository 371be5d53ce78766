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
// Read sets above the per-job limit (2^30 suffixes) are cut into classes of leading symbols -- whose order is the order of the
// suffixes -- and the classes are sorted one group after the other; from 2^32 symbols on the positions are 64-bit.
// Output is byte-identical to the reference's .bwt/.rbwt payload (tests/test_bwt_build.py).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
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
    // grid-stride: a launch holds fewer than 2^32 threads, the text may not
    for(uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256) {
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

// ---- kernels of one sorting job: a set of n suffix start positions `sa[j]` (type P: u32 below 2^32 symbols, u64 above),
// ---- j < n < 2^31, handed over in ascending text order
template <class P>
__global__ __launch_bounds__(256) void init_keys_kernel(const uint8_t* __restrict__ T, uint64_t N, uint32_t n,
                                                        const P* __restrict__ sa, uint64_t* __restrict__ key)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if(j >= n) return;
    key[j] = pack_key(T, N, (uint64_t)sa[j]);
}

template <class P>
__global__ __launch_bounds__(256) void iota_pos_kernel(P* __restrict__ sa, uint32_t n)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if(j < n) sa[j] = (P)j;
}

__global__ __launch_bounds__(256) void mark_kernel(const uint64_t* __restrict__ key, uint32_t n, uint8_t* __restrict__ bnd)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if(j >= n) return;
    if(j == 0 || key[j] != key[j - 1]) bnd[j] = 1;
}

__global__ __launch_bounds__(256) void unresolved_kernel(const uint64_t* __restrict__ key, const uint8_t* __restrict__ bnd,
                                                         uint32_t n, uint8_t* __restrict__ flag, uint32_t* __restrict__ head)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if(j >= n) return;
    const bool single = bnd[j] && (j + 1 == n || bnd[j + 1]);
    flag[j] = (!single && !key_has_zero(key[j])) ? 1 : 0;
    head[j] = bnd[j] ? j : 0u;
}

template <class P>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* __restrict__ T, uint64_t N, uint64_t depth,
                                                     const uint32_t* __restrict__ U, uint32_t n_u,
                                                     const P* __restrict__ sa, const uint32_t* __restrict__ headscan,
                                                     uint64_t* __restrict__ key2, uint32_t* __restrict__ gid,
                                                     P* __restrict__ val, uint32_t* __restrict__ perm)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if(k >= n_u) return;
    const uint32_t j = U[k];
    const P s = sa[j];
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
template <class P>
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t* __restrict__ U, uint32_t n_u,
                                                      const uint32_t* __restrict__ p2, const uint32_t* __restrict__ p1,
                                                      const P* __restrict__ val, const uint64_t* __restrict__ key_sorted,
                                                      P* __restrict__ sa, uint64_t* __restrict__ key)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if(k >= n_u) return;
    const uint32_t j = U[k];
    const uint32_t r1 = p2[k];
    sa[j] = val[p1[r1]];
    key[j] = key_sorted[r1];
}

template <class P>
__global__ __launch_bounds__(256) void bwt_kernel(const uint8_t* __restrict__ T, const P* __restrict__ sa, uint64_t N, uint32_t n,
                                                  uint8_t* __restrict__ bwt)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if(j >= n) return;
    const uint64_t s = (uint64_t)sa[j];
    bwt[j] = T[s == 0 ? N - 1 : s - 1];
}

// ---- partition of the suffixes by their first `len` symbols (cut after a sentinel, like the sort keys): id = those symbols,
// ---- 3 bits each.  Ids order like the suffixes, so the BWT is the concatenation of the groups' BWTs in id order.
__device__ __forceinline__ uint32_t prefix_id(const uint8_t* __restrict__ T, uint64_t N, uint64_t pos, uint32_t len)
{
    uint32_t id = 0;
    bool ended = false;
    for(uint32_t s = 0; s < len; ++s) {
        uint32_t c = 0;
        if(!ended && pos + s < N) c = T[pos + s];
        if(c == 0) ended = true;
        id = (id << 3) | c;
    }
    return id;
}

__global__ __launch_bounds__(256) void prefix_hist_kernel(const uint8_t* __restrict__ T, uint64_t N, uint32_t len,
                                                          unsigned long long* __restrict__ hist)
{
    extern __shared__ uint32_t lh[];
    const uint32_t bins = 1u << (3 * len);
    for(uint32_t b = threadIdx.x; b < bins; b += 256) lh[b] = 0;
    __syncthreads();
    for(uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256) atomicAdd(&lh[prefix_id(T, N, i, len)], 1u);
    __syncthreads();
    for(uint32_t b = threadIdx.x; b < bins; b += 256)
        if(lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

template <class P>
struct InGroup {
    const uint8_t* T;
    uint64_t N;
    uint32_t len, lo, hi;
    __host__ __device__ __forceinline__ bool operator()(const P& pos) const
    {
        const uint32_t id = prefix_id(T, N, (uint64_t)pos, len);
        return id >= lo && id <= hi;
    }
};

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
    void release(void* p)
    {
        for(size_t i = 0; i < ptrs.size(); ++i) if(ptrs[i] == p) { (void)hipFree(p); ptrs.erase(ptrs.begin() + (long)i); return; }
    }
};

static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

#define BB_TRY(expr)                                                                 \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if(_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return LRSC_ERR_DEVICE; } \
    } while(0)

// Scratch of the sorting jobs, sized once for the largest job.
template <class P>
struct SortSpace {
    uint32_t cap = 0;
    uint64_t* key[2] = {nullptr, nullptr};
    P* sa[2] = {nullptr, nullptr};
    uint8_t *bnd = nullptr, *flag = nullptr;
    uint32_t *head = nullptr, *U = nullptr, *nsel = nullptr, *headscan = nullptr;
    uint8_t* tmp = nullptr;
    size_t tmp_cap = 0;
    // refinement subset
    uint32_t u_cap = 0;
    uint64_t* u_key[2] = {nullptr, nullptr};
    uint32_t *u_gid[2] = {nullptr, nullptr}, *u_perm[2] = {nullptr, nullptr}, *u_p2[2] = {nullptr, nullptr};
    P* u_val = nullptr;
};

// Sorts the n suffixes whose start positions are in sp.sa[0] (ascending text order) and writes their BWT symbols to d_bwt[0..n).
template <class P>
static int sort_job(Dev& d, SortSpace<P>& sp, const uint8_t* d_T, uint64_t N, uint32_t n, uint8_t* d_bwt, uint32_t& rounds, std::string& err)
{
    hipStream_t st = nullptr;
    if(n == 0) return LRSC_OK;
    auto ensure_tmp = [&](size_t need) -> hipError_t {
        if(need <= sp.tmp_cap) return hipSuccess;
        if(sp.tmp) d.release(sp.tmp);
        sp.tmp = nullptr; sp.tmp_cap = 0;
        hipError_t e = d.alloc(&sp.tmp, need);
        if(e == hipSuccess) sp.tmp_cap = need;
        return e;
    };
    hipLaunchKernelGGL(init_keys_kernel<P>, dim3(nblk(n)), dim3(256), 0, st, d_T, N, n, sp.sa[0], sp.key[0]);
    BB_TRY(hipGetLastError());
    // 1. one big stable sort on the first 21 symbols
    hipcub::DoubleBuffer<uint64_t> kb(sp.key[0], sp.key[1]);
    hipcub::DoubleBuffer<P> vb(sp.sa[0], sp.sa[1]);
    size_t need = 0;
    BB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, kb, vb, (int)n, 0, 63, st));
    BB_TRY(ensure_tmp(need));
    BB_TRY(hipcub::DeviceRadixSort::SortPairs(sp.tmp, need, kb, vb, (int)n, 0, 63, st));
    uint64_t* key = kb.Current();
    P* sa = vb.Current();
    BB_TRY(hipMemsetAsync(sp.bnd, 0, n, st));

    for(uint64_t depth = kSymsPerKey;; depth += kSymsPerKey) {
        hipLaunchKernelGGL(mark_kernel, dim3(nblk(n)), dim3(256), 0, st, key, n, sp.bnd);
        hipLaunchKernelGGL(unresolved_kernel, dim3(nblk(n)), dim3(256), 0, st, key, sp.bnd, n, sp.flag, sp.head);
        BB_TRY(hipGetLastError());
        // compact the unresolved SA positions
        hipcub::CountingInputIterator<uint32_t> iota(0);
        BB_TRY(hipcub::DeviceSelect::Flagged(nullptr, need, iota, sp.flag, sp.U, sp.nsel, (int)n, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceSelect::Flagged(sp.tmp, need, iota, sp.flag, sp.U, sp.nsel, (int)n, st));
        uint32_t n_u = 0;
        BB_TRY(hipMemcpy(&n_u, sp.nsel, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if(n_u == 0) break;
        ++rounds;
        if(depth > (1ull << 22)) { err = "BWT refinement did not converge"; return LRSC_ERR_UNSUPPORTED; }
        // group id of every position = position of the closest boundary at or before it
        BB_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, need, sp.head, sp.headscan, MaxOp(), (int)n, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceScan::InclusiveScan(sp.tmp, need, sp.head, sp.headscan, MaxOp(), (int)n, st));
        if(n_u > sp.u_cap) {
            sp.u_cap = n_u;
            for(int b = 0; b < 2; ++b) {
                BB_TRY(d.alloc(&sp.u_key[b], sp.u_cap));
                BB_TRY(d.alloc(&sp.u_gid[b], sp.u_cap));
                BB_TRY(d.alloc(&sp.u_perm[b], sp.u_cap));
                BB_TRY(d.alloc(&sp.u_p2[b], sp.u_cap));
            }
            BB_TRY(d.alloc(&sp.u_val, sp.u_cap));
        }
        hipLaunchKernelGGL(gather_kernel<P>, dim3(nblk(n_u)), dim3(256), 0, st, d_T, N, depth, sp.U, n_u, sa, sp.headscan,
                           sp.u_key[0], sp.u_gid[0], sp.u_val, sp.u_perm[0]);
        BB_TRY(hipGetLastError());
        // sort 1 (stable): by the next 21 symbols.  p1[k] = gathered index of the k-th smallest key.
        hipcub::DoubleBuffer<uint64_t> k2(sp.u_key[0], sp.u_key[1]);
        hipcub::DoubleBuffer<uint32_t> p1(sp.u_perm[0], sp.u_perm[1]);
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k2, p1, (int)n_u, 0, 63, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(sp.tmp, need, k2, p1, (int)n_u, 0, 63, st));
        // sort 2 (stable): by group id, payload = rank in sort-1 order; inside a group the
        // sort-1 order (next 21 symbols, then text position) is preserved.
        hipLaunchKernelGGL(permute_u32_kernel, dim3(nblk(n_u)), dim3(256), 0, st, sp.u_gid[0], p1.Current(), n_u, sp.u_gid[1]);
        hipLaunchKernelGGL(iota_kernel, dim3(nblk(n_u)), dim3(256), 0, st, sp.u_p2[0], n_u);
        BB_TRY(hipGetLastError());
        hipcub::DoubleBuffer<uint32_t> gd(sp.u_gid[1], sp.u_gid[0]);
        hipcub::DoubleBuffer<uint32_t> p2(sp.u_p2[0], sp.u_p2[1]);
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, gd, p2, (int)n_u, 0, 32, st));
        BB_TRY(ensure_tmp(need));
        BB_TRY(hipcub::DeviceRadixSort::SortPairs(sp.tmp, need, gd, p2, (int)n_u, 0, 32, st));
        // U is ascending and groups are contiguous, so the k-th element of the (group, key2) order
        // belongs at SA position U[k].
        hipLaunchKernelGGL(scatter_kernel<P>, dim3(nblk(n_u)), dim3(256), 0, st, sp.U, n_u, p2.Current(), p1.Current(), sp.u_val,
                           k2.Current(), sa, key);
        BB_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(bwt_kernel<P>, dim3(nblk(n)), dim3(256), 0, st, d_T, sa, N, n, d_bwt);
    BB_TRY(hipGetLastError());
    return LRSC_OK;
}

template <class P>
static int alloc_space(Dev& d, SortSpace<P>& sp, uint32_t cap, std::string& err)
{
    sp.cap = cap;
    for(int b = 0; b < 2; ++b) { BB_TRY(d.alloc(&sp.key[b], cap)); BB_TRY(d.alloc(&sp.sa[b], cap)); }
    BB_TRY(d.alloc(&sp.bnd, cap)); BB_TRY(d.alloc(&sp.flag, cap)); BB_TRY(d.alloc(&sp.head, cap));
    BB_TRY(d.alloc(&sp.U, cap)); BB_TRY(d.alloc(&sp.nsel, 1)); BB_TRY(d.alloc(&sp.headscan, cap));
    return LRSC_OK;
}

// Groups of consecutive prefix ids, each with at most `limit` suffixes, sorted one after the other.
template <class P>
static int build_grouped(Dev& d, const uint8_t* d_T, uint64_t N, uint64_t limit, uint8_t* d_bwt, uint32_t& rounds, uint32_t& n_groups,
                         std::string& err)
{
    hipStream_t st = nullptr;
    // the shortest prefix whose largest class fits
    uint32_t len = 0;
    std::vector<unsigned long long> hist;
    unsigned long long* d_hist = nullptr;
    BB_TRY(d.alloc(&d_hist, 1u << 15));
    for(len = 1; len <= 4; ++len) {
        const uint32_t bins = 1u << (3 * len);
        BB_TRY(hipMemset(d_hist, 0, bins * sizeof(unsigned long long)));
        hipLaunchKernelGGL(prefix_hist_kernel, dim3(4096), dim3(256), bins * sizeof(uint32_t), st, d_T, N, len, d_hist);
        BB_TRY(hipGetLastError());
        hist.resize(bins);
        BB_TRY(hipMemcpy(hist.data(), d_hist, bins * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for(unsigned long long h : hist) mx = std::max(mx, h);
        if(mx <= limit) break;
    }
    if(len > 4) { err = "BWT builder: a 4-symbol prefix class exceeds the per-job limit"; return LRSC_ERR_UNSUPPORTED; }
    struct Group { uint32_t lo, hi; uint64_t count; };
    std::vector<Group> groups;
    uint64_t largest = 0;
    for(uint32_t id = 0; id < hist.size(); ++id) {
        if(hist[id] == 0) continue;
        if(groups.empty() || groups.back().count + hist[id] > limit) groups.push_back(Group{id, id, 0});
        groups.back().hi = id;
        groups.back().count += hist[id];
        largest = std::max<uint64_t>(largest, groups.back().count);
    }
    SortSpace<P> sp;
    int rc = alloc_space(d, sp, (uint32_t)largest, err);
    if(rc != LRSC_OK) return rc;
    uint32_t* d_nsel = nullptr;
    BB_TRY(d.alloc(&d_nsel, 1));
    uint8_t* d_sel_tmp = nullptr;
    size_t sel_cap = 0;
    const uint64_t chunk = 1ull << 30;
    uint64_t out = 0;
    for(const Group& g : groups) {
        // the group's suffix starts, ascending: stream compaction over the text in chunks of 2^30 positions
        uint64_t got = 0;
        const InGroup<P> pred{d_T, N, len, g.lo, g.hi};
        for(uint64_t base = 0; base < N; base += chunk) {
            const uint64_t cnt = std::min<uint64_t>(chunk, N - base);
            hipcub::CountingInputIterator<P> first((P)base);
            size_t need = 0;
            BB_TRY(hipcub::DeviceSelect::If(nullptr, need, first, sp.sa[0] + got, d_nsel, (int)cnt, pred, st));
            if(need > sel_cap) {
                if(d_sel_tmp) d.release(d_sel_tmp);
                d_sel_tmp = nullptr;
                BB_TRY(d.alloc(&d_sel_tmp, need));
                sel_cap = need;
            }
            BB_TRY(hipcub::DeviceSelect::If(d_sel_tmp, need, first, sp.sa[0] + got, d_nsel, (int)cnt, pred, st));
            uint32_t n_sel = 0;
            BB_TRY(hipMemcpy(&n_sel, d_nsel, sizeof(uint32_t), hipMemcpyDeviceToHost));
            got += n_sel;
        }
        if(got != g.count) { err = "BWT builder: group size mismatch"; return LRSC_ERR_DEVICE; }
        rc = sort_job<P>(d, sp, d_T, N, (uint32_t)got, d_bwt + out, rounds, err);
        if(rc != LRSC_OK) return rc;
        out += got;
    }
    if(out != N) { err = "BWT builder: groups do not cover the text"; return LRSC_ERR_DEVICE; }
    n_groups = (uint32_t)groups.size();
    return LRSC_OK;
}

// Builds the BWT of the read set (or of the reversed reads) on `device`; bwt_out receives N codes 0..4 ($ACGT),
// N = total bases + n_reads.  Up to LRSC_BWT_JOB suffixes (default 2^30) are sorted per job: a read set below that is one job over
// all positions; a larger one is cut into groups of leading-symbol classes that are sorted one after the other (positions
// become 64-bit from 2^32 symbols on).  LRSC_BWT_WIDE_POS=1 forces 64-bit positions (tests).
int build_bwt_device(const char* reads, const uint64_t* off, uint32_t n_reads, int reverse_reads, int device,
                     std::vector<uint8_t>& bwt_out, uint32_t* rounds_out, std::string& err)
{
    const uint64_t total = off[n_reads];
    const uint64_t N = total + n_reads;
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
    hipLaunchKernelGGL(text_kernel, dim3((unsigned)std::min<uint64_t>((N + 255) / 256, 1u << 22)), dim3(256), 0, st, d_ascii, d_off, n_reads, N, reverse_reads, d_T, d_bad);
    BB_TRY(hipGetLastError());
    int bad = 0;
    BB_TRY(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
    if(bad) { err = "sequence contains a base other than A,C,G,T"; return LRSC_ERR_ARG; }
    d.release(d_ascii);

    uint64_t limit = 1ull << 30;
    if(const char* e = std::getenv("LRSC_BWT_JOB")) { const long long v = std::atoll(e); if(v >= 16 && v < (1ll << 31)) limit = (uint64_t)v; }
    const char* wp = std::getenv("LRSC_BWT_WIDE_POS");
    const bool wide_pos = N >= (1ull << 32) || (wp && std::atoi(wp) != 0);
    uint8_t* d_bwt;
    BB_TRY(d.alloc(&d_bwt, N));
    uint32_t rounds = 0, n_groups = 1;
    int rc;
    if(N <= limit) {
        // one job over every position
        if(wide_pos) {
            SortSpace<uint64_t> sp;
            rc = alloc_space(d, sp, (uint32_t)N, err);
            if(rc != LRSC_OK) return rc;
            hipLaunchKernelGGL(iota_pos_kernel<uint64_t>, dim3(nblk(N)), dim3(256), 0, st, sp.sa[0], (uint32_t)N);
            rc = sort_job<uint64_t>(d, sp, d_T, N, (uint32_t)N, d_bwt, rounds, err);
        } else {
            SortSpace<uint32_t> sp;
            rc = alloc_space(d, sp, (uint32_t)N, err);
            if(rc != LRSC_OK) return rc;
            hipLaunchKernelGGL(iota_pos_kernel<uint32_t>, dim3(nblk(N)), dim3(256), 0, st, sp.sa[0], (uint32_t)N);
            rc = sort_job<uint32_t>(d, sp, d_T, N, (uint32_t)N, d_bwt, rounds, err);
        }
    } else if(wide_pos) rc = build_grouped<uint64_t>(d, d_T, N, limit, d_bwt, rounds, n_groups, err);
    else rc = build_grouped<uint32_t>(d, d_T, N, limit, d_bwt, rounds, n_groups, err);
    if(rc != LRSC_OK) return rc;
    if(std::getenv("LRSC_BWT_PROFILE"))
        std::fprintf(stderr, "[lrsc] BWT of %llu symbols: %u job(s), %u refinement rounds, %d-bit positions\n", (unsigned long long)N, n_groups,
                     rounds, wide_pos ? 64 : 32);
    bwt_out.resize(N);
    BB_TRY(hipMemcpy(bwt_out.data(), d_bwt, N, hipMemcpyDeviceToHost));
    if(rounds_out) *rounds_out = rounds;
    return LRSC_OK;
}

} // namespace lrsc
