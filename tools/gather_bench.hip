// tools/gather_bench.hip -- micro-benchmark: random 64-byte block gathers on MI355X.
// Variants: A lane loads its own 64 B (4 x dwordx4); B lane loads 16 B; C lane loads 32 B;
// D quad-cooperative (4 lanes x 16 B of one block); E lane loads 64 B, two independent blocks.
// Dependent chains of `steps` gathers per lane mimic backward search (next index from loaded data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while(0)

__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int VARIANT>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ tab, uint64_t nblk, int steps, uint64_t* out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    if(VARIANT == 3) {
        // quad-cooperative: 4 lanes share a query id
        const uint64_t q = gid >> 2; const uint32_t part = gid & 3;
        uint64_t idx = mix(q + 1) % nblk;
        for(int s = 0; s < steps; ++s) {
            const uint4 v = tab[idx * 4 + part];
            uint32_t x = v.x ^ v.y ^ v.z ^ v.w;
            x ^= __shfl_xor(x, 1, 64); x ^= __shfl_xor(x, 2, 64);
            acc += x;
            idx = mix(q + x + s) % nblk;
        }
    } else {
        uint64_t idx = mix(gid + 1) % nblk;
        uint64_t idx2 = mix(gid + 77) % nblk;
        for(int s = 0; s < steps; ++s) {
            uint32_t x = 0;
            if(VARIANT == 0) { const uint4 a = tab[idx*4], b = tab[idx*4+1], c = tab[idx*4+2], d = tab[idx*4+3];
                               x = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ b.x ^ c.y ^ d.z; }
            if(VARIANT == 1) { const uint4 a = tab[idx*4]; x = a.x ^ a.y ^ a.z ^ a.w; }
            if(VARIANT == 2) { const uint4 a = tab[idx*4], b = tab[idx*4+1]; x = a.x ^ b.y ^ a.z ^ b.w; }
            if(VARIANT == 4) { const uint4 a = tab[idx*4], b = tab[idx*4+1], c = tab[idx*4+2], d = tab[idx*4+3];
                               const uint4 e = tab[idx2*4], f = tab[idx2*4+1], g = tab[idx2*4+2], h = tab[idx2*4+3];
                               x = a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ h.w; idx2 = mix(gid + x + 3*s) % nblk; }
            acc += x;
            idx = mix(gid + x + s) % nblk;
        }
    }
    if(acc == 0x1234567) out[0] = acc;
}

template <int V> double run(const uint4* tab, uint64_t nblk, uint64_t lanes, int steps, uint64_t* out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather<V>, dim3(lanes / 256), dim3(256), 0, 0, tab, nblk, 2, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather<V>, dim3(lanes / 256), dim3(256), 0, 0, tab, nblk, steps, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    const uint64_t sizes_mb[] = {2, 24, 128, 700, 4096};
    uint64_t* out; CK(hipMalloc(&out, 8));
    for(uint64_t mb : sizes_mb) {
        const uint64_t bytes = mb << 20, nblk = bytes / 64;
        uint4* tab; CK(hipMalloc(&tab, bytes));
        CK(hipMemset(tab, 0x5A, bytes));
        const uint64_t lanes = 1ull << 24; const int steps = 32;
        const char* names[5] = {"lane 64B (4x dwordx4)", "lane 16B", "lane 32B", "quad-coop 64B", "lane 2 x 64B"};
        double ms[5] = {run<0>(tab, nblk, lanes, steps, out), run<1>(tab, nblk, lanes, steps, out), run<2>(tab, nblk, lanes, steps, out),
                        run<3>(tab, nblk, lanes, steps, out), run<4>(tab, nblk, lanes, steps, out)};
        for(int v = 0; v < 5; ++v) {
            double blocks = (double)lanes * steps * (v == 3 ? 0.25 : v == 4 ? 2.0 : 1.0);
            printf("table %5lu MB  %-22s %8.2f ms  %7.2f G blocks/s  %7.2f GB/s(64B lines)\n", (unsigned long)mb, names[v], ms[v],
                   blocks / ms[v] / 1e6, blocks * 64 / ms[v] / 1e6);
        }
        CK(hipFree(tab));
    }
    return 0;
}
