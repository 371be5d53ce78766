// seeds.hip -- LongReadProbe on the device: getSeqAttribute + searchSeedsWithHybridKmers +
// SeedFeature::estimateBestKmerSize + removeHitchhikingSeeds
// (reference: PacBio/LongReadProbe.cpp:34-227, PacBio/SeedFeature.cpp:22-78, PacBio/KmerFeature.h).
//
// Input is the compact output of the k-mer grid kernel (kernels.hip): per position the frequency of the
// static / scan k-mers, a both-strands-valid bit and the number of bases the base search counted.
//   1. seed_modes_kernel      position-parallel: classify the scan 19-mer (repeat / garbage / zero)
//   2. hipCUB inclusive scans  so that every sliding-window count is two loads
//   3. seed_attribute_kernel  position-parallel: LongReadProbe::getSeqAttribute's per-position mode
//   4. seed_scan_kernel       ONE LANE PER READ: the greedy scan is a sequential state machine whose
//                              outer loop variable is rewritten by the inner loop (LongReadProbe.cpp:82-92),
//                              so the parallelism is across the ~1e5 reads of a batch, not inside a read.
// Floating point follows the reference's mixed float/double evaluation exactly;
// the file is compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

#include "kernels.h"
#include "rank_device.h"

namespace lrsc {

// KmerFeature::isLowComplexity (KmerFeature.h:116-126) on a composition count
__device__ __forceinline__ bool low_complexity(int c0, int c1, int c2, int c3, int size)
{
    // sort 4 ints ascending (only the two largest matter)
    int a = c0 < c1 ? c0 : c1, b = c0 < c1 ? c1 : c0;
    int c = c2 < c3 ? c2 : c3, d = c2 < c3 ? c3 : c2;
    const int top = b > d ? b : d;                               // largest
    const int second = b > d ? (a > d ? a : d) : (c > b ? c : b);  // second largest
    const float m = 0.7f, dd = 0.9f;
    const bool isMonmer = (float)top / size >= m;
    const bool isDimer = (float)(second + top) / size >= dd;
    return isMonmer || isDimer;
}

// composition of the bases a KmerFeature counted: w[0..counted) from the base search + w[base_k..size) from expand()
__device__ __forceinline__ void kmer_counts(const uint8_t* __restrict__ w, uint32_t counted, uint32_t base_k, uint32_t size, int cnt[4])
{
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    for(uint32_t t = 0; t < size; ++t) {
        const bool in = (t < base_k) ? (t < counted) : true;
        const uint32_t c = w[t];
        cnt[0] += in && c == 0; cnt[1] += in && c == 1; cnt[2] += in && c == 2; cnt[3] += in && c == 3;
    }
}

// ---- 1. per-position classification of the scan k-mer (LongReadProbe.cpp:151-169) -----------------------
__global__ __launch_bounds__(256) void seed_modes_kernel(SeedArgs a)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if(gid >= a.total_bases) return;
    const int row = a.row_of_k[a.scan_kmer_len];
    const int f = a.freq[(uint64_t)row * a.total_bases + gid];          // -1 == fake
    int freq_eff = f;
    if(f >= 0) {
        int cnt[4];
        kmer_counts(a.codes + gid, a.base_counted[gid], a.base_k, (uint32_t)a.scan_kmer_len, cnt);
        if(low_complexity(cnt[0], cnt[1], cnt[2], cnt[3], a.scan_kmer_len)) freq_eff = -1;
    }
    const float repeatValue = a.thresholds[2 * 52 + a.scan_kmer_len];
    const unsigned long long rep = (freq_eff >= 0 && (float)freq_eff >= repeatValue) ? 1ull : 0ull;
    const unsigned long long garbage = freq_eff < 0 ? 1ull : 0ull;
    a.flags[gid] = rep | (garbage << 32);
    a.zeros[gid] = freq_eff == 0 ? 1u : 0u;
}

// ---- 3. LongReadProbe::getSeqAttribute (LongReadProbe.cpp:120-182) ---------------------------------------
// box[2] is a plain +-150 window count.  box[-1] is not: entering k-mers are garbage when freq < 0, leaving
// ones when freq <= 0 (:152-156 vs :163-168), so every zero-frequency k-mer that has left the window
// stays subtracted for the rest of the read.
__global__ __launch_bounds__(256) void seed_attribute_kernel(SeedArgs a)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool start_ok = false;
    if(gid < a.total_bases) {
        uint32_t r = a.chunk_read[gid >> kChunkShift];
        while(a.read_off[r + 1] <= gid) ++r;
        const int64_t s = (int64_t)a.read_off[r], e = (int64_t)a.read_off[r + 1];
        uint8_t attr = 1;
        if(a.manual && !a.ratio) {
            attr = (uint8_t)a.mode;
        } else if(e - s >= a.start_kmer_len) {
            const int range = 300;
            int64_t left = (int64_t)gid - (range >> 1), right = (int64_t)gid + (range >> 1);
            left = left > s ? left : s;
            right = right < e - 1 ? right : e - 1;
            const unsigned long long wr = a.flags[right];
            const unsigned long long wl = left > 0 ? a.flags[left - 1] : 0ull;
            const unsigned long long d = wr - wl;                      // both halves are monotone: no borrow across bit 32
            const int box2 = (int)(uint32_t)(d & 0xFFFFFFFFull);
            int boxneg = (int)(uint32_t)(d >> 32);
            const uint32_t zl = left > 0 ? a.zeros[left - 1] : 0u;
            const uint32_t zs = s > 0 ? a.zeros[s - 1] : 0u;
            boxneg -= (int)(zl - zs);                                   // zero-frequency k-mers that already left
            const int size = (int)(right - left + 1) - boxneg;
            const float ratio = (float)((double)((float)box2 / (float)size) + 0.0005);
            if((double)ratio >= 0.02) attr = 2;
            if(a.ratio) a.ratio[gid] = ratio;
            if(a.manual) attr = (uint8_t)a.mode;          // the reference computes the attribute, then overwrites it (:39-40)
        }
        a.attribute[gid] = attr;
        // Can the greedy scan start a seed here?  Its first inner iteration (currPos == initPos, dynamicKmer == staticKmer,
        // LongReadProbe.cpp:61-80) only looks at this position: not fake, frequency >= the mode's threshold, both strands
        // valid, size within the bound (freqDiff is 1 there).  Everywhere else the scan just moves on, so it can jump.
        if(e - s >= a.start_kmer_len) {
            const int staticSize = a.start_kmer_len + a.offset[attr];
            const int row = a.row_of_k[staticSize & 63];
            const int sf = a.freq[(uint64_t)row * a.total_bases + gid];
            start_ok = sf >= 0 && !((float)sf < a.thresholds[attr * 52 + staticSize]) && ((a.valid_mask[gid] >> row) & 1u) &&
                       !(staticSize > a.kmer_len_up_bound);
        }
    }
    const unsigned long long m = __ballot(start_ok);
    if((threadIdx.x & 63) == 0) a.start_bits[gid >> 6] = m;
}

// first position >= g (global) whose start bit is set, or gend
__device__ __forceinline__ uint64_t next_start(const unsigned long long* __restrict__ bits, uint64_t g, uint64_t gend)
{
    while(g < gend) {
        const unsigned long long w = bits[g >> 6] >> (g & 63);
        if(w) { g += (uint64_t)__builtin_ctzll(w); return g < gend ? g : gend; }
        g = (g | 63) + 1;
    }
    return gend;
}

// ---- 4. the greedy scan ---------------------------------------------------------------------------------
// find `k` characters starting at codes[start], moving by `step`, optionally complemented, with
// findInterval's early exit (BWTAlgorithms.cpp:14-31).  Returns the clamped frequency.
template <bool WIDE>
__device__ __forceinline__ int64_t find_run(const FmIndexDev& fm, bool on_bwt, const StrandC<typename Lay<WIDE>::pos_t>& s,
                                            const uint8_t* __restrict__ codes, int64_t start, int step, int k, bool comp,
                                            const uint32_t* __restrict__ mtab, uint32_t& n_rank, uint32_t& n_blk)
{
    using P = typename Lay<WIDE>::pos_t;
    // The characters are consumed in the order c_0, c_1, ...  On the rBWT that is the table's fwd walk of w = c; on the BWT
    // it is the rvc walk of w = complement(c): the first T of them come from the k-mer table (frozen at the first empty
    // interval, like the loop below).
    IvT<P> iv;
    int j0 = 0;
    {
        WalkState<P> ts = walk_init<P>();
        const uint32_t tk = table_start<WIDE>(fm, [&](uint32_t t) {
            uint32_t c = codes[start + (int64_t)t * step];
            if(comp) c = 3u - c;
            return on_bwt ? 3u - c : c; }, (uint32_t)k, ts);
        if(tk != 0) {
            iv = on_bwt ? ts.rvc : ts.fwd;
            n_rank += 1;
            if(iv.lo > iv.hi) return 0;
            j0 = (int)tk;
        }
    }
    if(j0 == 0) {
        uint32_t c = codes[start];
        if(comp) c = 3u - c;
        iv = init_interval<P>(s, c);
        n_rank += 1;
        j0 = 1;
    }
    for(int j = j0; j < k; ++j) {
        uint32_t c = codes[start + (int64_t)j * step];
        if(comp) c = 3u - c;
        iv = update_interval<WIDE>(s, c, iv, mtab, n_blk);
        n_rank += 2;
        if(iv.lo > iv.hi) break;
    }
    return iv_freq(iv);
}

constexpr uint32_t kReadsPerWaveDefault = 32;       // LRSC_SEED_RPW overrides (power of two <= 64)

struct DynKmer {          // the part of KmerFeature the scan needs
    int size;
    int freq;             // frequency (valid when !fake)
    bool fake, valid, have_iv;
};

template <bool WIDE>
__global__ __launch_bounds__(64) void seed_scan_kernel(FmIndexDev fm, SeedArgs a, uint32_t min_k, uint32_t kReadsPerWave, DevCounters* ctr)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    // kReadsPerWave of the 64 lanes own a read: the scan is a divergent sequential state machine, and
    // the FM work of different lanes of one wavefront serialises; fewer reads per wave, more waves.
    const uint32_t r = blockIdx.x * kReadsPerWave + threadIdx.x / (64 / kReadsPerWave);
    const bool owner = (threadIdx.x % (64 / kReadsPerWave)) == 0;
    uint32_t n_rank = 0, n_blk = 0;
    if(owner && r < a.n_reads) {
        const StrandC<P> sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
        const StrandC<P> sR = strand_consts<P>(fm.strand[LRSC_BWT]);
        const uint64_t s = a.read_off[r], e = a.read_off[r + 1];
        const int64_t len = (int64_t)(e - s);
        const uint8_t* codes = a.codes + s;
        const uint8_t* attr = a.attribute + s;
        int32_t* out = a.seeds + seed_slab(s, r, min_k) * kSeedInts;
        uint32_t n_seeds = 0;
        int staticSize = a.start_kmer_len;
        const float hh = a.hh_ratio;
        const float inv_hh = 1 / hh;

        if(len >= staticSize) {
            for(int64_t initPos = 0; initPos < len; initPos++) {
                initPos = (int64_t)(next_start(a.start_bits, s + (uint64_t)initPos, e) - s);     // positions in between only do initPos++
                if(initPos >= len) break;
                const int dynamicMode = attr[initPos];
                staticSize += a.offset[dynamicMode];
                const int row = a.row_of_k[staticSize & 63];
                const int32_t* frow = a.freq + (uint64_t)row * a.total_bases + s;
                // dynamicKmer = KmerFeature::Log()[staticSize][initPos]
                const int f0 = frow[initPos];
                DynKmer dyn;
                dyn.fake = f0 < 0;
                dyn.freq = f0;
                dyn.size = dyn.fake ? (int)(len - initPos < staticSize ? len - initPos : staticSize) : staticSize;
                dyn.valid = (a.valid_mask[s + initPos] >> row) & 1u;
                dyn.have_iv = false;
                IvT<P> dfwd, drvc;
                dfwd.lo = dfwd.hi = drvc.lo = drvc.hi = 0;
                bool isSeed = false, isRepeat = false;
                int maxFixedMerFreq = dyn.fake ? -1 : dyn.freq;
                const int64_t seedPos = initPos;
                for(int64_t currPos = initPos; currPos < len; currPos++) {
                    const int staticMode = attr[currPos];
                    const int sf = frow[currPos];
                    if(sf < 0) break;                                   // staticKmer.isFake()
                    if(isSeed) {
                        if(!dyn.have_iv) {
                            // the static k-mer at seedPos passed isValid(): its chained interval equals a plain search
                            WalkState<P> st = walk_init<P>();
                            const uint32_t t0 = table_start<WIDE>(fm, [&](uint32_t t) { return (uint32_t)codes[seedPos + t]; }, (uint32_t)dyn.size, st);
                            for(int t = (int)t0; t < dyn.size; ++t) st = walk_step<WIDE>(sF, sR, codes[seedPos + t], 1u << 30, st, mtab);
                            n_rank += st.n_rank; n_blk += st.n_blk;
                            dfwd = st.fwd; drvc = st.rvc;
                            dyn.have_iv = true;
                        }
                        const uint32_t b = codes[currPos + staticSize - 1];
                        dfwd = update_interval<WIDE>(sF, b, dfwd, mtab, n_blk);       // KmerFeature::expand
                        drvc = update_interval<WIDE>(sR, 3u - b, drvc, mtab, n_blk);
                        n_rank += 4;
                        dyn.size++;
                        dyn.freq = (int)(iv_freq(dfwd) + iv_freq(drvc));
                        dyn.valid = dfwd.lo <= dfwd.hi && drvc.lo <= drvc.hi;
                    }
                    const float dynamicThreshold = a.thresholds[dynamicMode * 52 + dyn.size];
                    const float staticThreshold = a.thresholds[staticMode * 52 + staticSize];
                    const float repeatThreshold = (5 - ((staticMode >> 1) << 2)) * staticThreshold;
                    const int dynFreq = dyn.fake ? -1 : dyn.freq;
                    if((float)sf < staticThreshold || (float)dynFreq < dynamicThreshold || !dyn.valid ||
                       dyn.size > a.kmer_len_up_bound) {
                        if(isSeed) dyn.size--;                          // shrink(1)
                        break;
                    }
                    const float freqDiff = (float)sf / maxFixedMerFreq;
                    if(freqDiff < hh) {                                 // hitchhiking k-mer (HIGH --> LOW)
                        initPos++;
                        dyn.size--;                                     // shrink(1)
                        break;
                    } else if(freqDiff > inv_hh) {                      // hitchhiking k-mer (LOW --> HIGH)
                        initPos = currPos - 1;
                        isSeed = false;
                        break;
                    }
                    initPos = seedPos + dyn.size - 1;
                    isSeed = true;
                    isRepeat |= ((float)sf >= repeatThreshold);
                    maxFixedMerFreq = maxFixedMerFreq > sf ? maxFixedMerFreq : sf;
                }
                if(isSeed) {
                    int cnt[4];
                    kmer_counts(codes + seedPos, a.base_counted[s + seedPos], a.base_k, (uint32_t)dyn.size, cnt);
                    if(!low_complexity(cnt[0], cnt[1], cnt[2], cnt[3], dyn.size)) {
                        // SeedFeature ctor (SeedFeature.cpp:22-41) + estimateBestKmerSize (:43-78)
                        const int seedLen = dyn.size;
                        int bestK[2] = {staticSize, staticSize};
                        int bestF[2] = {0, 0};
                        const int sizeUpper = seedLen, sizeLower = staticSize;
                        const int freqUpper = a.pb_coverage >> 1, freqLower = a.pb_coverage >> 2;
                        for(int which = 0; which < 2; ++which) {
                            const bool pole = which == 0;                // true: start k-mer in the rbwt; false: end k-mer in the bwt
                            const StrandC<P>& sel = pole ? sF : sR;
                            int kmerSize = bestK[which];
                            auto occ = [&](int k) -> int {
                                int64_t f;
                                if(pole) {
                                    f = find_run<WIDE>(fm, false, sel, codes, seedPos, +1, k, false, mtab, n_rank, n_blk);
                                    f += find_run<WIDE>(fm, false, sel, codes, seedPos + k - 1, -1, k, true, mtab, n_rank, n_blk);
                                } else {
                                    f = find_run<WIDE>(fm, true, sel, codes, seedPos + seedLen - 1, -1, k, false, mtab, n_rank, n_blk);
                                    f += find_run<WIDE>(fm, true, sel, codes, seedPos + seedLen - k, +1, k, true, mtab, n_rank, n_blk);
                                }
                                return (int)f;
                            };
                            int kmerFreq = occ(kmerSize);
                            int bit = 0;
                            if(kmerFreq > freqUpper) bit = 1;
                            else if(kmerFreq < freqLower) bit = -1;
                            if(bit != 0) {
                                const int freqBound = bit > 0 ? freqUpper : freqLower;
                                const int corsFreqBound = bit > 0 ? freqLower : freqUpper;
                                const int sizeBound = bit > 0 ? sizeUpper : sizeLower;
                                while((bit ^ kmerFreq) > (bit ^ freqBound) && (bit ^ kmerSize) < (bit ^ sizeBound)) {
                                    kmerSize += bit;
                                    kmerFreq = occ(kmerSize);
                                }
                                if((bit ^ kmerFreq) < (bit ^ corsFreqBound)) {
                                    kmerSize -= bit;
                                    kmerFreq = occ(kmerSize);
                                }
                            }
                            bestK[which] = kmerSize;
                            bestF[which] = kmerFreq;
                        }
                        int32_t* o = out + (uint64_t)n_seeds * kSeedInts;
                        o[0] = (int32_t)seedPos; o[1] = seedLen; o[2] = maxFixedMerFreq; o[3] = isRepeat ? 1 : 0;
                        o[4] = bestK[0]; o[5] = bestK[1]; o[6] = bestF[0]; o[7] = bestF[1];
                        ++n_seeds;
                    }
                }
                staticSize -= a.offset[dynamicMode];
            }

            // removeHitchhikingSeeds (LongReadProbe.cpp:187-227); bit 1 of the isRepeat field marks an outcast
            if(n_seeds >= 2) {
                for(uint32_t q = 0; q + 1 < n_seeds; ++q) {
                    int32_t* query = out + (uint64_t)q * kSeedInts;
                    const int qEnd = query[0] + query[1] - 1;
                    for(uint32_t t = q + 1; t < n_seeds; ++t) {
                        int32_t* subject = out + (uint64_t)t * kSeedInts;
                        if((int)(subject[0] - qEnd) > a.radius) break;
                        const float freqDiff = (float)subject[2] / query[2];
                        if((query[3] & 1) && freqDiff < hh) subject[3] |= 2;       // HIGH --> LOW
                        if((subject[3] & 1) && freqDiff > inv_hh) query[3] |= 2;   // LOW  --> HIGH
                    }
                }
                uint32_t w = 0, n_out = 0;
                int32_t* oc = a.outcasts ? a.outcasts + (out - a.seeds) : nullptr;     // the read's slab of the outcast array
                for(uint32_t q = 0; q < n_seeds; ++q) {
                    int32_t* src = out + (uint64_t)q * kSeedInts;
                    if(src[3] & 2) {
                        if(oc) { for(uint32_t i = 0; i < kSeedInts; ++i) oc[(uint64_t)n_out * kSeedInts + i] = i == 3 ? (src[i] & 1) : src[i]; ++n_out; }
                        continue;
                    }
                    int32_t* dst = out + (uint64_t)w * kSeedInts;
                    if(w != q) for(uint32_t i = 0; i < kSeedInts; ++i) dst[i] = src[i];
                    ++w;
                }
                n_seeds = w;
                if(a.outcast_count) a.outcast_count[r] = n_out;
            } else if(a.outcast_count) a.outcast_count[r] = 0;
        } else if(a.outcast_count) a.outcast_count[r] = 0;
        a.seed_count[r] = n_seeds;
    }
    flush_counters(ctr, n_rank, n_blk);
}

// ---------------------------------------------------------------------------------------
static inline unsigned blocks_for256(uint64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_seed_modes(const SeedArgs& a, hipStream_t stream)
{
    if(a.total_bases == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_modes_kernel, dim3(blocks_for256(a.total_bases)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_seed_attribute(const SeedArgs& a, hipStream_t stream)
{
    if(a.total_bases == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_attribute_kernel, dim3(blocks_for256(a.total_bases)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_seed_scan(const FmIndexDev& fm, const SeedArgs& a, uint32_t min_k, DevCounters* ctr, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    uint32_t rpw = kReadsPerWaveDefault;
    if(const char* e = getenv("LRSC_SEED_RPW")) {
        const int v = atoi(e);
        if(v >= 1 && v <= 64 && (v & (v - 1)) == 0) rpw = (uint32_t)v;
    }
    const unsigned nb = (a.n_reads + rpw - 1) / rpw;
    if(fm.wide) hipLaunchKernelGGL(seed_scan_kernel<true>, dim3(nb), dim3(64), 0, stream, fm, a, min_k, rpw, ctr);
    else        hipLaunchKernelGGL(seed_scan_kernel<false>, dim3(nb), dim3(64), 0, stream, fm, a, min_k, rpw, ctr);
    return hipGetLastError();
}
hipError_t scan_seed_flags(unsigned long long* flags, uint32_t* zeros, uint64_t n, void** tmp, size_t* tmp_cap, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    size_t need1 = 0, need2 = 0;
    hipError_t e = hipcub::DeviceScan::InclusiveSum(nullptr, need1, flags, flags, (int64_t)n, stream);
    if(e != hipSuccess) return e;
    e = hipcub::DeviceScan::InclusiveSum(nullptr, need2, zeros, zeros, (int64_t)n, stream);
    if(e != hipSuccess) return e;
    const size_t need = need1 > need2 ? need1 : need2;
    if(need > *tmp_cap) {
        if(*tmp) (void)hipFree(*tmp);
        *tmp = nullptr; *tmp_cap = 0;
        e = hipMalloc(tmp, need);
        if(e != hipSuccess) return e;
        *tmp_cap = need;
    }
    e = hipcub::DeviceScan::InclusiveSum(*tmp, need1, flags, flags, (int64_t)n, stream);
    if(e != hipSuccess) return e;
    return hipcub::DeviceScan::InclusiveSum(*tmp, need2, zeros, zeros, (int64_t)n, stream);
}

} // namespace lrsc
