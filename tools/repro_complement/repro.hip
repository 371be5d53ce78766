// Reduced form of the compiler finding of rounds 1-2, self-contained.  One lane per query runs k backward-search
// steps over rank blocks in the product's Block32 layout: per step two in-block counts whose BASE is picked from the block's four
// counters by a select chain over the symbol code -- the code being a byte from the query, complemented for odd lanes:
//   VARIANT 0   c = q[i]; if(dir) c = 3u - c;          the compiler only knows c in [-252, 255]
//   VARIANT 1   c = (q[i] ^ (dir ? 3 : 0)) & 3          c in [0, 3] visible
//   VARIANT 2   as 0, then `c &= 3u` before the select chain
//   VARIANT 3   as 0, the base picked by the code's two bits (a binary tree of selects: what rank_device.h does now, LRSC_PICK4)
// ROCm 7.2, -O3, gfx950: variant 0 returns wrong intervals for every lane that consumed code 3 ('T'); 1 and 2 match the host
// evaluation of the same source, and so does 3.  ISA (profiles/r03_compiler_finding/): with the unbounded range the select chain is lowered as a
// switch (NodeBlock / LeafBlock, signed compares) and, for the SECOND count of a step, the default arm's `v_mov base, cnt[3]` lands
// in LeafBlock -- executed by the lanes with c <= 0 -- instead of the arm the code-3 lanes take, which is left without it: those
// lanes keep cnt[1] as their base.  With the range visible the chain becomes v_cndmask and no branch exists.
//   hipcc -O3 --offload-arch=gfx950 -DVARIANT=0 repro.hip -o repro0 && ./repro0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Blk { uint32_t cnt[4]; uint32_t w[12]; };                   // counts before the block + 192 symbols as bit planes
__host__ __device__ inline uint32_t count_in(const uint4 q[4], uint32_t code, uint32_t off)
{
#if VARIANT == 3
    const uint32_t base = (code & 2u) ? ((code & 1u) ? q[0].w : q[0].z) : ((code & 1u) ? q[0].y : (q[0].x & 0x7FFFFFFFu));   // bit tree
#else
    const uint32_t base = code == 0 ? (q[0].x & 0x7FFFFFFFu) : code == 1 ? q[0].y : code == 2 ? q[0].z : q[0].w;     // the select chain
#endif
    const uint32_t L = (code & 1u) ? 0u : 0xFFFFFFFFu, H = (code & 2u) ? 0u : 0xFFFFFFFFu;
    const uint32_t lo[6] = {q[1].x, q[1].y, q[2].x, q[2].y, q[3].x, q[3].y}, hi[6] = {q[1].z, q[1].w, q[2].z, q[2].w, q[3].z, q[3].w};
    uint32_t c = base;
    for(int i = 0; i < 6; ++i) {
        const int n = (int)off - 32 * i;
        const uint32_t m = n >= 32 ? 0xFFFFFFFFu : n <= 0 ? 0u : ((1u << n) - 1u);
        c += __builtin_popcount((lo[i] ^ L) & (hi[i] ^ H) & m);
    }
    return c;
}
__host__ __device__ inline void step(const Blk* blocks, const uint32_t* pred, uint32_t c, uint32_t& lo, uint32_t& hi)
{
    const uint32_t pl = lo, pu = hi + 1, bl = pl / 192, bu = pu / 192;
    uint4 ra[4], rb[4];
    for(int j = 0; j < 4; ++j) { ra[j] = reinterpret_cast<const uint4*>(blocks + bl)[j]; rb[j] = ra[j]; }
    if(bu != bl) for(int j = 0; j < 4; ++j) rb[j] = reinterpret_cast<const uint4*>(blocks + bu)[j];
    lo = pred[c] + count_in(ra, c, pl - bl * 192);
    hi = pred[c] + count_in(rb, c, pu - bu * 192) - 1;
}
__global__ void repro_kernel(const Blk* blocks, const uint32_t* pred, uint32_t n_sym, const uint8_t* q, uint32_t lq, uint32_t k, uint32_t n, uint32_t* out)
{
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if(g >= n) return;
    const uint32_t dir = g & 1u;
    uint32_t lo = 0, hi = n_sym - 1;
    for(uint32_t i = 0; i < k && lo <= hi; ++i) {
#if VARIANT == 1
        const uint32_t c = ((uint32_t)q[(g * 7 + (dir == 0 ? i : lq - 1 - i)) % lq] ^ (dir != 0 ? 3u : 0u)) & 3u;
#else
        uint32_t c = q[(g * 7 + (dir == 0 ? i : lq - 1 - i)) % lq];
        if(dir != 0) c = 3u - c;
#if VARIANT == 2
        c &= 3u;
#endif
#endif
        step(blocks, pred, c, lo, hi);
    }
    out[2 * g] = lo; out[2 * g + 1] = hi;
}
static uint32_t rnd(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main()
{
    const uint32_t N = 300000, lq = 4096, k = 19, n = 4096, nb = N / 192 + 1;
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    std::vector<Blk> blocks(nb);
    uint32_t cnt[4] = {0, 0, 0, 0};
    for(uint32_t p = 0; p < N; ++p) {
        const uint32_t b = p / 192, o = p % 192, c = rnd(seed) & 3u;
        if(o == 0) for(int j = 0; j < 4; ++j) blocks[b].cnt[j] = cnt[j];
        blocks[b].w[(o >> 6) * 4 + ((o >> 5) & 1)] |= (c & 1u) << (o & 31u);
        blocks[b].w[(o >> 6) * 4 + 2 + ((o >> 5) & 1)] |= (c >> 1) << (o & 31u);
        ++cnt[c];
    }
    const uint32_t pred[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
    std::vector<uint8_t> q(lq);
    for(auto& c : q) c = (uint8_t)(rnd(seed) & 3u);
    Blk* d_b; uint32_t *d_p, *d_out; uint8_t* d_q;
    hipMalloc((void**)&d_b, nb * sizeof(Blk)); hipMemcpy(d_b, blocks.data(), nb * sizeof(Blk), hipMemcpyHostToDevice);
    hipMalloc((void**)&d_p, 16); hipMemcpy(d_p, pred, 16, hipMemcpyHostToDevice);
    hipMalloc((void**)&d_q, lq); hipMemcpy(d_q, q.data(), lq, hipMemcpyHostToDevice);
    hipMalloc((void**)&d_out, n * 8);
    hipLaunchKernelGGL(repro_kernel, dim3(n / 64), dim3(64), 0, 0, d_b, d_p, N, d_q, lq, k, n, d_out);
    std::vector<uint32_t> got(2 * n);
    if(hipMemcpy(got.data(), d_out, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("device error\n"); return 2; }
    uint32_t bad = 0, bad_t = 0;
    for(uint32_t g = 0; g < n; ++g) {
        const uint32_t dir = g & 1u; uint32_t lo = 0, hi = N - 1; bool saw_t = false;
        for(uint32_t i = 0; i < k && lo <= hi; ++i) {
            const uint32_t c = ((uint32_t)q[(g * 7 + (dir == 0 ? i : lq - 1 - i)) % lq] ^ (dir != 0 ? 3u : 0u)) & 3u;
            saw_t = saw_t || c == 3;
            step(blocks.data(), pred, c, lo, hi);
        }
        if(got[2 * g] != lo || got[2 * g + 1] != hi) { ++bad; bad_t += saw_t; }
    }
    std::printf("VARIANT %d: %u of %u lanes differ from the host evaluation (%u of them consumed code 3)\n", VARIANT, bad, n, bad_t);
    return bad ? 1 : 0;
}
