// fm_device.h -- HBM layout of the FM-index and the device-side rank primitive.
//
// The reference keeps one byte per run plus a 12-byte marker every 32 symbols and a
// 48-byte marker every 8192 (SuffixTools/RLBWT.h:105-140): one Occ query touches three
// cache lines through two dependent loads.  Here one Occ query touches exactly ONE
// 64-byte, 64-byte-aligned block holding cumulative counts and the symbols as two BIT PLANES:
//
//   Block32 (N < 2^31 symbols): 4 x u32 counts of A,C,G,T before the block
//                               + 6 low-plane + 6 high-plane u32 = 192 symbols (bit i of plane word w =
//                               low/high code bit of symbol 32*w + i), grouped into 16-byte pieces (below)
//   Block64 (any N):            4 x u64 counts + lo[4] + hi[4] u32 = 128 symbols
//
// With planes, "symbols equal to code c among the first n" is popcount((lo ^ L) & (hi ^ H) & mask)
// per 32 symbols: 4 VALU ops per 32 symbols instead of ~8 per 16 for packed 2-bit codes.
//
// Symbol codes are A=0 C=1 G=2 T=3 (rank - 1).  '$' rows (one per read, ~1e-4 of the BWT) are
// stored as code 0 and listed in a sorted side array; a block that contains a '$' has the top bit
// of its A counter set, and only an A query into such a block pays the side-array lookup.
// Counts never include '$'.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LRSC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define LRSC_HD inline
#endif

namespace lrsc {

// Block32 is organised as four 16-byte PIECES so that 4 lanes can fetch one block with one coalesced
// 64-byte access and each lane owns a self-contained slice:
//   piece 0:      cnt[4]
//   piece j=1..3: { lo[2j-2], lo[2j-1], hi[2j-2], hi[2j-1] }  = both bit planes of symbols [64(j-1), 64j)
struct alignas(64) Block32 {
    static constexpr uint32_t kSyms = 192;
    static constexpr uint32_t kWords = 6;
    uint32_t cnt[4];
    uint32_t w[12];
    // index into w[] of the low / high plane word holding symbols [32*wi, 32*wi + 32)
    static constexpr uint32_t lo_index(uint32_t wi) { return (wi >> 1) * 4 + (wi & 1); }
    static constexpr uint32_t hi_index(uint32_t wi) { return (wi >> 1) * 4 + 2 + (wi & 1); }
};
struct alignas(64) Block64 {
    static constexpr uint32_t kSyms = 128;
    static constexpr uint32_t kWords = 4;
    uint64_t cnt[4];
    uint32_t lo[4];
    uint32_t hi[4];
};
static_assert(sizeof(Block32) == 64, "Block32 must be one 64-byte line");
static_assert(sizeof(Block64) == 64, "Block64 must be one 64-byte line");

// '$' directory granularity: one entry per 8 rank blocks (a '$' look-up is one directory entry + the one or two list
// entries of that group instead of a binary search over the whole list: 2 x log2(#reads) dependent loads)
constexpr uint32_t kDollarDirShift = 3;
constexpr uint32_t kFlag32 = 0x80000000u;
constexpr uint64_t kFlag64 = 0x8000000000000000ull;

// One strand of the index as the kernels see it.
struct FmStrand {
    const void* blocks;          // Block32[] or Block64[]
    const uint64_t* dollars;     // sorted BWT positions holding '$'
    const uint32_t* dollar_dir;  // dollar_dir[g] = number of '$' rows before block (g << kDollarDirShift): where a block's '$' rows start
    uint64_t n_dollars;
    uint64_t dollar_group_syms;  // symbols per directory group (block symbols << kDollarDirShift)
    uint64_t n_symbols;
    uint64_t n_blocks;
    uint64_t pred[5];            // C[$ACGT]
};
// k-mer interval tables (device-side BWTIntervalCache, SuffixTools/BWTIntervalCache.h:24-29 -- the
// reference has the class but pbcorrect leaves it unused): entry[code(w)] = {fwd.lo, fwd.hi, rvc.lo, rvc.hi}
// of findBiInterval(w) INCLUDING findInterval's early exit, i.e. exactly the state after the first k
// steps of a search.  Narrow (32-bit) indexes only.  Up to five sizes, ascending.
struct KmerTable {
    const void* entries;         // uint4[4^k]
    uint32_t k;                  // 0 = absent
    uint32_t pad;
};
struct FmIndexDev {
    FmStrand strand[2];          // [LRSC_BWT], [LRSC_RBWT]
    uint32_t wide;               // 0 -> Block32, 1 -> Block64
    uint32_t pad;
    KmerTable ktab[5];
};

// mask of the low n bits of a 32-symbol word, n clamped to [0, 32]
LRSC_HD uint32_t low_mask(int32_t n)
{
    return n >= 32 ? 0xFFFFFFFFu : (n <= 0 ? 0u : ((1u << n) - 1u));
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

} // namespace lrsc
