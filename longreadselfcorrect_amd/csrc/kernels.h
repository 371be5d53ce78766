// kernels.h -- launch wrappers of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/lrsc.h"
#include "fm_device.h"

namespace lrsc {

// device-side counters filled by the kernels.  One ctx owns kCtrShards of them, each on its own 64-byte
// line: every wavefront adds its totals to shard (blockIdx.x % kCtrShards).  (A single shared line was
// measured to throttle the grid kernel: ~49 M same-line atomics per launch at ~88 atomics/us.)
struct alignas(64) DevCounters {
    unsigned long long rank_queries;
    unsigned long long block_loads;
    unsigned long long table_loads;     // k-mer interval table look-ups (16-byte entries, one 64-byte line each)
    unsigned long long pad[5];
};
constexpr uint32_t kCtrShards = 1024;

constexpr uint32_t kMaxPool = 8;
constexpr uint32_t kChunkShift = 10;   // coarse position -> read table granularity (1024 bases)

struct GridArgs {
    const uint8_t* codes;         // one byte per base, 0..3 (A,C,G,T)
    const uint64_t* read_off;     // n_reads + 1
    const uint32_t* chunk_read;   // read containing position (chunk << kChunkShift)
    uint64_t total_bases;
    uint32_t n_reads;
    uint32_t n_k;
    uint8_t ks[kMaxPool];
    // full outputs (any may be null)
    lrsc_biinterval* out_iv;      // [pos * n_k + slot]
    uint8_t* out_size;            // [pos * n_k + slot]
    uint8_t* out_count;           // [(pos * n_k + slot) * 4]
    // compact outputs (any may be null): structure-of-arrays, slot-major
    int32_t* freq;                // [freq_index[slot] * total_bases + pos]  KmerFeature::getFreq() (-1 == fake)
    int8_t freq_index[kMaxPool];  // compact row of `freq` for a slot, -1 = not stored
    uint8_t* valid_mask;          // [pos] bit freq_index[slot] = both strands' intervals valid (KmerFeature::isValid)
    uint8_t* base_counted;        // [pos] chars counted by the base-slot search (<= ks[0])
    lrsc_biinterval* slot_iv;     // [slot * total_bases + pos] (only if non-null)
};

// LongReadProbe on the device (seeds.hip)
constexpr uint32_t kSeedInts = 8;   // seedStartPos, seedLen, maxFixedMerFreq, isRepeat, startBestK, endBestK, startKmerFreq, endKmerFreq
struct SeedArgs {
    const uint8_t* codes;
    const uint64_t* read_off;
    const uint32_t* chunk_read;
    uint64_t total_bases;
    uint32_t n_reads;
    // grid features (see GridArgs)
    const int32_t* freq;
    const uint8_t* valid_mask;
    const uint8_t* base_counted;
    int8_t row_of_k[64];          // freq row of a k-mer size, -1 if that size is not in the pool
    uint32_t base_k;
    // ProbeParameters (PacBio/LongReadProbe.h:7-40)
    int32_t start_kmer_len, scan_kmer_len, kmer_len_up_bound, pb_coverage, mode, manual, radius;
    int32_t offset[3];
    float hh_ratio;
    const float* thresholds;      // KmerThreshold table, [3][52]
    // scratch / outputs
    unsigned long long* flags;    // [pos] (repeat) | (garbage << 32), later its inclusive scan
    uint32_t* zeros;              // [pos] zero-frequency non-low-complexity scan k-mers, later its inclusive scan
    uint8_t* attribute;           // [pos] 1 unique / 2 repeat (LongReadProbe::getSeqAttribute)
    float* ratio;                 // [pos] the repeat ratio behind `attribute` (extend/<read>.log of --debugseed), or nullptr
    unsigned long long* start_bits;   // bit pos: the static k-mer at pos passes the scan's first-iteration tests (a seed can start here)
    int32_t* seeds;               // kSeedInts per seed, read r's slab starts at seed_slab(r)
    uint32_t* seed_count;         // [read]
    // --debugseed (both null otherwise): the seeds removeHitchhikingSeeds drops, same slab layout as `seeds`
    int32_t* outcasts;
    uint32_t* outcast_count;      // [read]
};
// first seed record of read r: reads can hold at most len/15 + 1 seeds (static k-mers are >= 15 long... any k >= 1: len + 1)
__host__ __device__ inline uint64_t seed_slab(uint64_t read_start, uint32_t r, uint32_t min_k) { return read_start / min_k + r; }

hipError_t launch_seed_modes(const SeedArgs& a, hipStream_t stream);
hipError_t launch_seed_attribute(const SeedArgs& a, hipStream_t stream);
hipError_t launch_seed_scan(const FmIndexDev& fm, const SeedArgs& a, uint32_t min_k, DevCounters* ctr, hipStream_t stream);
// inclusive scans of SeedArgs::flags / zeros in place (hipCUB); tmp is grown as needed
hipError_t scan_seed_flags(unsigned long long* flags, uint32_t* zeros, uint64_t n, void** tmp, size_t* tmp_cap, hipStream_t stream);

// fills a k-mer interval table (4^k uint4 entries) by running the k-step findBiInterval of every k-mer
hipError_t launch_ktab_build(const FmIndexDev& fm, uint32_t k, void* entries, uint32_t prev_k, const void* prev, hipStream_t stream);
hipError_t launch_rank(const FmIndexDev& fm, const lrsc_rank_query* q, uint64_t n, uint64_t* out,
                       DevCounters* ctr, hipStream_t stream);
// LF-walk (LongReadOverlap::retrieveStr, PacBio/LongReadOverlap.cpp:696-749): from BWT row `row` of `strand`,
// b = BWT[idx]; stop at '$' or after max_steps; idx = C[b] + Occ(b, idx - 1).  out codes (0..3), out_len.
struct LfJob { uint64_t row; uint64_t out_off; uint32_t max_steps; uint32_t strand; };
hipError_t launch_lf_walk(const FmIndexDev& fm, const LfJob* jobs, uint64_t n, uint8_t* out, uint32_t* out_len,
                          DevCounters* ctr, hipStream_t stream);
hipError_t launch_bwt_chars(const FmIndexDev& fm, int strand, const uint64_t* idx, uint64_t n, char* out,
                            hipStream_t stream);
hipError_t launch_find_kmers(const FmIndexDev& fm, const uint8_t* kmer_codes, uint32_t k, uint64_t n,
                             lrsc_biinterval* out, DevCounters* ctr, hipStream_t stream);
hipError_t launch_kmer_grid(const FmIndexDev& fm, const GridArgs& a, DevCounters* ctr, hipStream_t stream);
// ASCII -> 2-bit code per byte; *bad set to 1 if a byte is not one of ACGT
hipError_t launch_encode(const char* ascii, uint8_t* codes, uint64_t n, int* bad, hipStream_t stream);
hipError_t launch_chunk_table(const uint64_t* read_off, uint32_t n_reads, uint64_t total_bases,
                              uint32_t* chunk_read, hipStream_t stream);

} // namespace lrsc
