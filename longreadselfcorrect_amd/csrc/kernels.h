// kernels.h -- launch wrappers of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/lrsc.h"
#include "fm_device.h"

namespace lrsc {

// device-side counters filled by the kernels (one per ctx)
struct DevCounters {
    unsigned long long rank_queries;
    unsigned long long block_loads;
};

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
    int32_t* freq;                // [slot * total_bases + pos]  KmerFeature::getFreq() (-1 == fake)
    uint8_t* base_counted;        // [pos] chars counted by the base-slot search (<= ks[0])
    lrsc_biinterval* slot_iv;     // [slot * total_bases + pos] (only if non-null)
};

hipError_t launch_rank(const FmIndexDev& fm, const lrsc_rank_query* q, uint64_t n, uint64_t* out,
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
