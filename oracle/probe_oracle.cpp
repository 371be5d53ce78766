// oracle/probe_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see probe_oracle.hpp).
#include "probe_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <sstream>

namespace lrsc_oracle {

// ---------------------------------------------------------------------------------------
// KmerThreshold (PacBio/KmerThreshold.cpp:11-79)
// ---------------------------------------------------------------------------------------
static const float formula[3][6] = {
    // x*x             x*y              y*y            x              y              (constant)
    {0.0004799107143, -0.008037815126, 0.03673552754, 0.1850695903, -1.572552521, 18.0522088},    // lowcov
    {0.0003348214286, -0.009112394958, 0.04286714686, 0.240519958, -1.8793367350, 21.29319228},   // unique
    {0.01714285714, -0.6193907563, 2.266956783, 17.28450630, -100.6983493, 1103.571729}            // repeat
};

void KmerThreshold::initialize(int s, int e, int c)
{
    start_ = std::max(s, 15);
    end_ = e;
    cov_ = c;
    for(int mode = 0; mode <= 2; mode++) {
        table_[mode].assign(end_ + 2, 0.0f);
        float cavity = std::numeric_limits<float>::max();
        for(int ksize = start_; ksize <= end_; ksize++) {
            cavity = std::fmin(cavity, calculate(mode, cov_, ksize));   // std::fminf on floats
            table_[mode][ksize] = cavity;
        }
    }
}

float KmerThreshold::calculate(int mode, int x, int y)
{
    const float* f = formula[mode];
    float v = f[0] * x * x + f[1] * x * y + f[2] * y * y + f[3] * x + f[4] * y + f[5];
    return std::fmax(v, 2.0f);
}

std::string KmerThreshold::table_text() const
{
    std::ostringstream out;
    out << "Coverage : " << cov_ << "\n" << "size\tlowcov\tunique\trepeat\n";
    for(int ksize = start_; ksize <= end_; ksize++)
        out << ksize << "\t" << table_[0][ksize] << "\t" << table_[1][ksize] << "\t" << table_[2][ksize] << "\n";
    return out.str();
}

// ---------------------------------------------------------------------------------------
// KmerFeature (PacBio/KmerFeature.h:37-126)
// ---------------------------------------------------------------------------------------
KmerFeature::KmerFeature(const IndexSet& indices_, const std::string& seq, size_t pos, int len, const KmerFeature* base)
{
    if(base == nullptr) {
        this->count[0] = this->count[1] = this->count[2] = this->count[3] = 0;
        this->indices = indices_;
        this->word = seq.substr(pos, len);
        this->size = (int)word.length();
        this->biInterval = find_bi_interval(this->indices, this->word, this->count);
    } else {
        *this = *base;
        assert(this->size < len);
        for(size_t i = (pos + this->size); (i < seq.length()) && (this->size < len); i++) {
            char b = seq[i];
            expand(b);
        }
    }
    this->fake = (len != this->size);
    this->frequency = (int)this->biInterval.freq();
}

void KmerFeature::expand(char b)
{
    if(b == 0) return;
    this->size++;
    this->word += b;
    update_bi_interval(this->biInterval, b, this->indices, this->count);
    this->frequency = (int)this->biInterval.freq();
}

static inline int dna_idx4(char b)
{
    switch(b) { case 'A': return 0; case 'C': return 1; case 'G': return 2; default: return 3; }
}

void KmerFeature::shrink(int len, bool update)
{
    assert(len < this->size);
    this->size -= len;
    for(std::string::iterator iter = (this->word.begin() + this->size); iter != this->word.end(); iter++)
        this->count[dna_idx4(*iter)]--;
    this->word.erase(this->size, len);
    if(!update) return;
    this->biInterval = find_bi_interval(this->indices, this->word);
    this->frequency = (int)this->biInterval.freq();
}

bool KmerFeature::isLowComplexity(float m, float d) const
{
    int copy[4];
    std::copy(this->count, this->count + 4, copy);
    std::sort(copy, (copy + 4));
    bool isMonmer = (float)copy[3] / this->size >= m;
    bool isDimer = (float)(copy[2] + copy[3]) / this->size >= d;
    return isMonmer || isDimer;
}

// ---------------------------------------------------------------------------------------
// SeedFeature (PacBio/SeedFeature.cpp:22-78)
// ---------------------------------------------------------------------------------------
SeedFeature::SeedFeature(std::string str, int startPos, int frequency, bool repeat, int kmerSize, int PBcoverage)
    : seedStr(str),
      seedLen((int)seedStr.length()),
      seedStartPos(startPos),
      seedEndPos(startPos + seedLen - 1),
      maxFixedMerFreq(frequency),
      isRepeat(repeat),
      isHitchhiked(false),
      startBestKmerSize(kmerSize),
      endBestKmerSize(kmerSize),
      sizeUpperBound(seedLen),
      sizeLowerBound(kmerSize),
      freqUpperBound(PBcoverage >> 1),
      freqLowerBound(PBcoverage >> 2)
{
}

void SeedFeature::append(const std::string& extendedStr, const SeedFeature& target)
{
    seedStr += extendedStr;
    seedLen += (int)extendedStr.length();
    startBestKmerSize = target.startBestKmerSize;
    endBestKmerSize = target.endBestKmerSize;
    isRepeat = target.isRepeat;
    maxFixedMerFreq = target.maxFixedMerFreq;
    seedStartPos = target.seedStartPos;
    seedEndPos = target.seedEndPos;
}

void SeedFeature::estimateBestKmerSize(const IndexSet& indices)
{
    modifyKmerSize(indices, true);
    modifyKmerSize(indices, false);
}

// pole(true/false) ? start : end ; bit(1/-1) > 0 ? increase : decrease
void SeedFeature::modifyKmerSize(const IndexSet& indices, bool pole)
{
    int& kmerSize = pole ? startBestKmerSize : endBestKmerSize;
    int& kmerFreq = pole ? startKmerFreq : endKmerFreq;
    const RLBwt* const pSelBWT = pole ? indices.rbwt : indices.bwt;
    std::string seed = pole ? reverse_str(seedStr) : seedStr;
    kmerFreq = (int)count_sequence_occurrences(seed.substr(seedLen - kmerSize), pSelBWT);
    int bit;
    if(kmerFreq > freqUpperBound)
        bit = 1;
    else if(kmerFreq < freqLowerBound)
        bit = -1;
    else
        return;
    const int freqBound = bit > 0 ? freqUpperBound : freqLowerBound;
    const int corsFreqBound = bit > 0 ? freqLowerBound : freqUpperBound;
    const int sizeBound = bit > 0 ? sizeUpperBound : sizeLowerBound;

    while((bit ^ kmerFreq) > (bit ^ freqBound) && (bit ^ kmerSize) < (bit ^ sizeBound)) {
        kmerSize += bit;
        kmerFreq = (int)count_sequence_occurrences(seed.substr(seedLen - kmerSize), pSelBWT);
    }
    if((bit ^ kmerFreq) < (bit ^ corsFreqBound)) {
        kmerSize -= bit;
        kmerFreq = (int)count_sequence_occurrences(seed.substr(seedLen - kmerSize), pSelBWT);
    }
}

// ---------------------------------------------------------------------------------------
// LongReadProbe (PacBio/LongReadProbe.cpp)
// ---------------------------------------------------------------------------------------
void allocateKmerLog(KmerLog& log, const std::set<int>& pool, size_t readLen)
{
    for(auto& iter : pool) log[iter] = std::unique_ptr<KmerFeature[]>(new KmerFeature[readLen]);
}

void getSeqAttribute(const ProbeParameters& m_params, KmerLog& log, const std::string& seq, int* const attribute,
                     ProbeDebug* dbg)
{
    const size_t seqLen = seq.length();
    std::fill_n(attribute, seqLen, 1);

    int range = 300;
    const int ksize = m_params.scanKmerLen;
    float repeatValue = m_params.thresholds->get(2, ksize);

    int front = 0, fear = -1;
    int leftmost = (int)(seqLen - 1), rightmost = 0;
    std::map<int, int> box;   // -1 -> garbage; 0 -> lowcov(disable); 1 -> unique; 2 -> repeat

    for(size_t pos = 0; pos < seqLen; pos++) {
        int left = (int)pos - (range >> 1);
        int right = (int)pos + (range >> 1);
        left = std::max(left, 0);
        right = std::min(right, (int)(seqLen - 1));
        while(fear < right) {
            fear++;
            KmerFeature* prev = nullptr;
            for(auto& iter : m_params.pool) {
                log[iter][fear] = KmerFeature(m_params.indices, seq, fear, iter, prev);
                prev = log[iter].get() + fear;
            }
            const KmerFeature& inKmer = log[ksize][fear];
            int freq = inKmer.isLowComplexity() ? -1 : inKmer.getFreq();
            int mode;
            if(freq < 0) mode = -1;
            else if(freq >= repeatValue) mode = 2;
            else mode = 1;
            box[mode]++;
        }
        while(front < left) {
            const KmerFeature& outKmer = log[ksize][front];
            front++;
            int freq = outKmer.isLowComplexity() ? -1 : outKmer.getFreq();
            int mode;
            if(freq <= 0) mode = -1;
            else if(freq >= repeatValue) mode = 2;
            else mode = 1;
            box[mode]--;
        }
        int size = (right - left + 1) - box[-1];
        float ratio = (float)box[2] / size + 0.0005;
        if(dbg) dbg->ratio.push_back(ratio);
        if(ratio >= 0.02) {
            attribute[pos] = 2;
            leftmost = std::min(leftmost, (int)pos);
            rightmost = std::max(rightmost, (int)pos);
        }
    }
    (void)leftmost; (void)rightmost;
}

SeedFeature::SeedVector removeHitchhikingSeeds(const ProbeParameters& m_params, SeedFeature::SeedVector initSeedVec,
                                               ProbeDebug* dbg)
{
    if(initSeedVec.size() < 2) return initSeedVec;

    for(SeedFeature::SeedVector::iterator iterQuery = initSeedVec.begin(); (iterQuery + 1) != initSeedVec.end(); iterQuery++) {
        SeedFeature& query = *iterQuery;
        SeedFeature::SeedVector::iterator iterSubject = iterQuery + 1;
        for(; iterSubject != initSeedVec.end(); iterSubject++) {
            SeedFeature& subject = *iterSubject;
            if((int)(subject.seedStartPos - query.seedEndPos) > m_params.radius) break;
            float freqDiff = (float)subject.maxFixedMerFreq / query.maxFixedMerFreq;
            subject.isHitchhiked |= (query.isRepeat && freqDiff < m_params.hhRatio);       // HIGH --> LOW
            query.isHitchhiked |= (subject.isRepeat && freqDiff > 1 / m_params.hhRatio);   // LOW  --> HIGH
        }
    }

    SeedFeature::SeedVector finalSeedVec, outcastSeedVec;
    finalSeedVec.reserve(initSeedVec.size());
    outcastSeedVec.reserve(initSeedVec.size() >> 1);
    for(const auto& iter : initSeedVec) {
        if(iter.isHitchhiked) outcastSeedVec.push_back(iter);
        else finalSeedVec.push_back(iter);
    }
    if(dbg) dbg->outcast = outcastSeedVec;
    return finalSeedVec;
}

void searchSeedsWithHybridKmers(const ProbeParameters& m_params, KmerLog& log, const std::string& readSeq,
                                SeedFeature::SeedVector& seedVec, ProbeDebug* dbg, std::vector<int>* attribute_out)
{
    const size_t readSeqLen = readSeq.length();
    int staticSize = m_params.startKmerLen;
    if((int)readSeqLen < staticSize) return;

    int* attribute = new int[readSeqLen];
    getSeqAttribute(m_params, log, readSeq, attribute, dbg);
    if(m_params.Manual) std::fill_n(attribute, readSeqLen, m_params.mode);
    if(attribute_out) attribute_out->assign(attribute, attribute + readSeqLen);
    const KmerThreshold& thr = *m_params.thresholds;

    // [init/curr]Pos indicate the initial/current position of the static-kmer.
    for(size_t initPos = 0; initPos < readSeqLen; initPos++) {
        int dynamicMode = attribute[initPos];
        staticSize += m_params.offset[dynamicMode];
        KmerFeature dynamicKmer = log[staticSize][initPos];
        bool isSeed = false, isRepeat = false;
        int maxFixedMerFreq = dynamicKmer.getFreq();
        size_t seedPos = initPos;
        for(size_t currPos = initPos; currPos < readSeqLen; currPos++) {
            int staticMode = attribute[currPos];
            const KmerFeature& staticKmer = log[staticSize][currPos];
            if(staticKmer.isFake()) break;
            if(isSeed) {
                char b = readSeq[(currPos + staticSize - 1)];
                dynamicKmer.expand(b);
            }
            float dynamicThreshold = thr.get(dynamicMode, dynamicKmer.getSize());
            float staticThreshold = thr.get(staticMode, staticKmer.getSize());
            float repeatThreshold = (5 - ((staticMode >> 1) << 2)) * staticThreshold;
            // General seed extension strategy.
            if(staticKmer.getFreq() < staticThreshold                      // 1.static frequency
               || dynamicKmer.getFreq() < dynamicThreshold                 // 2.dynamic frequency(1)
               || !dynamicKmer.isValid()                                   // 2.dynamic frequency(2)
               || dynamicKmer.getSize() > m_params.kmerLenUpBound          // 3.over length
            ) {
                if(isSeed) dynamicKmer.shrink(1);
                break;
            }
            // Kmer Hitchhike strategy.
            float freqDiff = (float)staticKmer.getFreq() / maxFixedMerFreq;
            if(freqDiff < m_params.hhRatio) {           // 4.hitchhiking kmer(1) (HIGH-->LOW)
                initPos++;
                dynamicKmer.shrink(1);
                break;
            } else if(freqDiff > 1 / m_params.hhRatio) {   // 4.hitchhiking kmer(2) (LOW-->HIGH)
                initPos = currPos - 1;
                isSeed = false;
                break;
            }
            initPos = seedPos + dynamicKmer.getSize() - 1;
            isSeed = true;
            isRepeat |= (staticKmer.getFreq() >= repeatThreshold);
            maxFixedMerFreq = std::max(maxFixedMerFreq, staticKmer.getFreq());
        }
        // Low Complexity strategy.
        if(isSeed && !dynamicKmer.isLowComplexity()) {
            seedVec.push_back(SeedFeature(dynamicKmer.getWord(), (int)seedPos, maxFixedMerFreq, isRepeat, staticSize,
                                          m_params.PBcoverage));
            seedVec.back().estimateBestKmerSize(m_params.indices);
        }
        staticSize -= m_params.offset[dynamicMode];
    }

    // Seed Hitchhike strategy.
    seedVec = removeHitchhikingSeeds(m_params, seedVec, dbg);
    delete[] attribute;
}

} // namespace lrsc_oracle
