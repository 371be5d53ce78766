// oracle/process_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the per-read workflow PacBioSelfCorrectionProcess::process
// (PacBio/PacBioSelfCorrectionProcess.{h,cpp}) and of the post-processor's FASTA / stats
// output (same file, :250-380).  "Parity unpinned" by a reference build (the file includes
// LongReadCorrectByOverlap.h -> HashMap.h -> config.h); line-by-line restatement.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "extend_oracle.hpp"
#include "probe_oracle.hpp"

namespace lrsc_oracle {

struct CorrectionParameters {        // PacBioSelfCorrectionProcess.h:24-53
    IndexSet indices;
    int PBcoverage = 90;
    double ErrorRate = 0.15;
    int startKmerLen = 19;
    int nextTarget = 1;
    int maxLeaves = 32;
    int idmerLen = 9;
    int minKmerLen = 13;
    std::set<int> pool;
    bool Split = false;
    bool OnlySeed = false;
    bool NoDp = false;
    FMextendParameters FM_params;
    ProbeParameters probe;           // LongReadProbe::m_params (global in the reference)
};

struct WalkRecord {                  // one seed-pair walk (what --debugseed's extend/<id>.ext would show, plus successes)
    int srcStartPos, trgStartPos;
    int code;                        // extendOverlap return: 1, -1, -2, -3
    int via;                         // 0 = FM-extend, 1 = DP fallback, 2 = raw copy / split
    int gap = 0, steps = 0, leaf_expansions = 0;   // of the FM attempts towards this target (test visibility)
};

struct CorrectionResult {            // PacBioSelfCorrectionProcess.h:58-94
    std::string readid;
    bool merge = false;
    std::vector<std::string> correctedStrs;
    int64_t totalReadsLen = 0, correctedLen = 0, totalSeedNum = 0, totalWalkNum = 0, highErrorNum = 0,
            exceedDepthNum = 0, exceedLeaveNum = 0, FMNum = 0, DPNum = 0, seedDis = 0;
    std::vector<WalkRecord> walks;   // not in the reference: test visibility
    SeedFeature::SeedVector seeds;   // not in the reference: test visibility
    WalkStats walk_stats;
    // not in the reference: how often the source k-mer of a walk equals the tail of the previous TARGET seed's own string
    // (what a walk-parallel schedule would assume before the previous walk has run).  [0] walks checked, [1] hits,
    // misses by what the previous walk did: [2] FM success, [3] DP consensus, [4] raw copy / split, [5] k differs
    uint64_t spec[6] = {0, 0, 0, 0, 0, 0};
};

class SelfCorrectionProcess {
public:
    explicit SelfCorrectionProcess(const CorrectionParameters& params) : m_params(params) {}
    CorrectionResult process(const std::string& id, const std::string& readSeq);     // .cpp:23-54
private:
    void initCorrect(std::string& readSeq, const SeedFeature::SeedVector& seedVec, SeedFeature::SeedVector& pieceVec,
                     CorrectionResult& result);                                       // :56-157
    int correctByFMExtension(const SeedFeature& source, const SeedFeature& target, const std::string& in,
                             std::string& out, CorrectionResult& result);             // :159-206
    bool correctByMSAlignment(const SeedFeature& source, const SeedFeature& target, const std::string& in,
                              std::string& out, CorrectionResult& result);            // :208-245
    const CorrectionParameters m_params;
};

// PacBioSelfCorrectionPostProcess (.cpp:250-380): accumulates the counters and renders
// correct.fa / discard.fa records and the integer part of the stdout stats block.
class SelfCorrectionPostProcess {
public:
    explicit SelfCorrectionPostProcess(bool split) : m_split(split) {}
    void process(const std::string& id, const std::string& readSeq, const CorrectionResult& r);   // :313-370
    std::string correct_fa, discard_fa;
    std::string stats_text() const;                                                                // :288-306 (no timer lines)
    int64_t totalReadsLen = 0, correctedLen = 0, totalSeedNum = 0, totalWalkNum = 0, highErrorNum = 0,
            exceedDepthNum = 0, exceedLeaveNum = 0, FMNum = 0, DPNum = 0, OutcastNum = 0, seedDis = 0;
private:
    bool m_split;
};

} // namespace lrsc_oracle
