// oracle/probe_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the solid-seed finder: KmerThreshold, KmerFeature, SeedFeature and
// LongReadProbe (PacBio/{KmerThreshold,SeedFeature,LongReadProbe}.{h,cpp}, PacBio/KmerFeature.h).
// Parity pin: KmerThreshold is checked against oracle/_ref (the reference's KmerThreshold.cpp
// compiles directly).  LongReadProbe/SeedFeature/KmerFeature include Util/HashMap.h ->
// generated config.h + google sparsehash and cannot be built here: "parity unpinned" by a
// reference build; pinned only through the FM layer underneath and the line-by-line restatement.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "fm_oracle.hpp"

namespace lrsc_oracle {

// PacBio/KmerThreshold.{h,cpp}
class KmerThreshold {
public:
    void initialize(int s, int e, int c);                 // KmerThreshold.cpp:43-63
    float get(int mode, int ksize) const { return table_[mode][ksize]; }   // KmerThreshold.h:18-23
    int start() const { return start_; }
    int end() const { return end_; }
    int cov() const { return cov_; }
    std::string table_text() const;                       // KmerThreshold.cpp:31-41,65-72 (threshold-table file)
private:
    static float calculate(int mode, int x, int y);      // KmerThreshold.cpp:74-79
    int start_ = 15, end_ = 50, cov_ = 0;
    std::vector<float> table_[3];
};

// PacBio/LongReadProbe.h:7-40
struct ProbeParameters {
    IndexSet indices;
    int startKmerLen = 19;
    int scanKmerLen = 19;
    int kmerLenUpBound = 50;
    int PBcoverage = 90;
    int mode = 1;
    int radius = 100;
    float hhRatio = 0.6;
    std::array<int, 3> offset{{0, 0, 0}};
    std::set<int> pool;
    bool Manual = false;
    const KmerThreshold* thresholds = nullptr;
};

// PacBio/KmerFeature.h:21-136
class KmerFeature {
public:
    KmerFeature() = default;
    KmerFeature(const IndexSet& indices, const std::string& seq, size_t pos, int len, const KmerFeature* base = nullptr);
    const std::string& getWord() const { return word; }
    int getSize() const { return size; }
    int getFreq() const { return fake ? -1 : frequency; }
    void expand(char b);
    void shrink(int len, bool update = false);
    bool isFake() const { return fake; }
    bool isValid() const { return biInterval.valid(); }
    bool isLowComplexity(float m = 0.7, float d = 0.9) const;

    int count[4] = {0, 0, 0, 0};
    IndexSet indices;
    std::string word;
    int size = 0;
    BiInterval biInterval;
    bool fake = false;
    int frequency = 0;
};
typedef std::map<int, std::unique_ptr<KmerFeature[]>> KmerLog;   // KmerFeature::Log()

// PacBio/SeedFeature.{h,cpp}
class SeedFeature {
public:
    typedef std::vector<SeedFeature> SeedVector;
    SeedFeature(std::string str, int startPos, int frequency, bool repeat, int kmerSize, int PBcoverage);
    void estimateBestKmerSize(const IndexSet& indices);                     // SeedFeature.cpp:43-47
    void append(const std::string& extendedStr, const SeedFeature& target);  // SeedFeature.h:22-33

    std::string seedStr;
    int seedLen;
    int seedStartPos;
    int seedEndPos;
    int maxFixedMerFreq;
    bool isRepeat;
    bool isHitchhiked;
    int startBestKmerSize;
    int endBestKmerSize;
    int startKmerFreq = 0;
    int endKmerFreq = 0;
private:
    int sizeUpperBound;
    int sizeLowerBound;
    int freqUpperBound;
    int freqLowerBound;
    void modifyKmerSize(const IndexSet& indices, bool pole);                // SeedFeature.cpp:50-78
};

// PacBio/LongReadProbe.cpp.  `log` plays KmerFeature::Log() (thread-local map in the reference).
struct ProbeDebug {                 // what --debugseed would dump (extend/<id>.log, seed/error/<id>.seed)
    std::vector<float> ratio;       // per position, getSeqAttribute
    SeedFeature::SeedVector outcast;
};
void allocateKmerLog(KmerLog& log, const std::set<int>& pool, size_t readLen);   // PacBioSelfCorrectionProcess.cpp:32-33
void getSeqAttribute(const ProbeParameters& p, KmerLog& log, const std::string& seq, int* attribute,
                     ProbeDebug* dbg = nullptr);                                    // LongReadProbe.cpp:120-182
SeedFeature::SeedVector removeHitchhikingSeeds(const ProbeParameters& p, SeedFeature::SeedVector initSeedVec,
                                               ProbeDebug* dbg = nullptr);          // LongReadProbe.cpp:187-227
void searchSeedsWithHybridKmers(const ProbeParameters& p, KmerLog& log, const std::string& readSeq,
                                SeedFeature::SeedVector& seedVec, ProbeDebug* dbg = nullptr,
                                std::vector<int>* attribute_out = nullptr);        // LongReadProbe.cpp:34-117

} // namespace lrsc_oracle
