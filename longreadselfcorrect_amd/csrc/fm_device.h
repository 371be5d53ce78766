// fm_device.h -- HBM layout of the FM-index and the device-side rank primitive.
//
// The reference keeps one byte per run plus a 12-byte marker every 32 symbols and a
// 48-byte marker every 8192 (SuffixTools/RLBWT.h:105-140): one Occ query touches three
// cache lines through two dependent loads.  Here one Occ query touches exactly ONE
// 64-byte, 64-byte-aligned block:
//
//   Block32 (N < 2^31 symbols):  4 x u32 cumulative counts of A,C,G,T before the block
//                                + 12 x u32 = 192 symbols packed 2 bits each
//   Block64 (any N):             4 x u64 cumulative counts + 4 x u64 = 128 symbols
//
// Symbol codes are A=0 C=1 G=2 T=3 (rank - 1).  '$' rows (one per read, ~1e-4 of the
// BWT) are stored as code 0 and listed in a sorted side array; a block that contains a
// '$' has the top bit of its A counter set, and only an A query into such a block pays
// the side-array lookup.  Counts never include '$'.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LRSC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define LRSC_HD inline
#endif

namespace lrsc {

struct alignas(64) Block32 {
    static constexpr uint32_t kSyms = 192;
    uint32_t cnt[4];
    uint32_t bits[12];
};
struct alignas(64) Block64 {
    static constexpr uint32_t kSyms = 128;
    uint64_t cnt[4];
    uint64_t bits[4];
};
static_assert(sizeof(Block32) == 64, "Block32 must be one 64-byte line");
static_assert(sizeof(Block64) == 64, "Block64 must be one 64-byte line");

constexpr uint32_t kFlag32 = 0x80000000u;
constexpr uint64_t kFlag64 = 0x8000000000000000ull;

// One strand of the index as the kernels see it.
struct FmStrand {
    const void* blocks;          // Block32[] or Block64[]
    const uint64_t* dollars;     // sorted BWT positions holding '$'
    uint64_t n_dollars;
    uint64_t n_symbols;
    uint64_t n_blocks;
    uint64_t pred[5];            // C[$ACGT]
};
struct FmIndexDev {
    FmStrand strand[2];          // [LRSC_BWT], [LRSC_RBWT]
    uint32_t wide;               // 0 -> Block32, 1 -> Block64
};

// number of 2-bit symbols equal to `code` among the low `n` symbols of a 32-bit word (n <= 16)
LRSC_HD uint32_t match16(uint32_t w, uint32_t code, uint32_t n)
{
    const uint32_t x = w ^ (code * 0x55555555u);
    uint32_t eq = ~(x | (x >> 1)) & 0x55555555u;
    eq &= (n >= 16) ? 0xFFFFFFFFu : ((1u << (2 * n)) - 1u);
    return (uint32_t)__builtin_popcount(eq);
}
LRSC_HD uint32_t match32(uint64_t w, uint32_t code, uint32_t n)
{
    const uint64_t x = w ^ (code * 0x5555555555555555ull);
    uint64_t eq = ~(x | (x >> 1)) & 0x5555555555555555ull;
    eq &= (n >= 32) ? ~0ull : ((1ull << (2 * n)) - 1ull);
    return (uint32_t)__builtin_popcountll(eq);
}

// '$' rows in [lo, hi) from the sorted side list
LRSC_HD uint64_t dollars_in(const FmStrand& s, uint64_t lo, uint64_t hi)
{
    uint64_t a = 0, b = s.n_dollars;
    while(a < b) { const uint64_t m = (a + b) >> 1; if(s.dollars[m] < lo) a = m + 1; else b = m; }
    const uint64_t first = a;
    b = s.n_dollars;
    while(a < b) { const uint64_t m = (a + b) >> 1; if(s.dollars[m] < hi) a = m + 1; else b = m; }
    return a - first;
}

// In-register copy of one block (what a lane holds after its 64-byte load).
struct BlockRegs32 { uint32_t cnt[4]; uint32_t bits[12]; };
struct BlockRegs64 { uint64_t cnt[4]; uint64_t bits[4]; };

// count of `code` among the first `off` symbols of the block (off <= kSyms), '$' not yet removed
LRSC_HD uint32_t inblock32(const uint32_t* bits, uint32_t code, uint32_t off)
{
    uint32_t c = 0;
#pragma unroll
    for(uint32_t w = 0; w < 12; ++w) {
        const uint32_t base = w * 16;
        const uint32_t n = off > base ? (off - base) : 0;
        c += match16(bits[w], code, n);
    }
    return c;
}
LRSC_HD uint32_t inblock64(const uint64_t* bits, uint32_t code, uint32_t off)
{
    uint32_t c = 0;
#pragma unroll
    for(uint32_t w = 0; w < 4; ++w) {
        const uint32_t base = w * 32;
        const uint32_t n = off > base ? (off - base) : 0;
        c += match32(bits[w], code, n);
    }
    return c;
}

} // namespace lrsc
