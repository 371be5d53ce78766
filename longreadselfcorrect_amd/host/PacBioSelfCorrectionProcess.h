// PacBioSelfCorrectionProcess.h -- host-side mirror of the reference's processor / post-processor pair
// (PacBio/PacBioSelfCorrectionProcess.h:24-127) over the C ABI (include/lrsc.h).
#pragma once
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/lrsc.h"
#include "SequenceWorkItem.h"

namespace stride {

// PacBioSelfCorrectionParameters (reference .h:24-53): the index set becomes an lrsc_index + devices
struct PacBioSelfCorrectionParameters {
    lrsc_index* index = nullptr;
    std::vector<int> devices{0};
    std::string directory;
    int threads = 1;                // -t: host threads that pack a batch's bases and cut the corrected strings out of the device's answer
    lrsc_params p{};                // PBcoverage, ErrorRate, startKmerLen, nextTarget, maxLeaves, idmerLen, minKmerLen, Split, NoDp ...
    bool DebugExtend = false, DebugSeed = false, OnlySeed = false;
};

// PacBioSelfCorrectionResult (reference .h:58-94)
struct PacBioSelfCorrectionResult {
    std::string readid;
    bool merge = false;
    std::vector<std::string> correctedStrs;
    int64_t totalReadsLen = 0, correctedLen = 0, totalSeedNum = 0, totalWalkNum = 0, highErrorNum = 0, exceedDepthNum = 0,
            exceedLeaveNum = 0, FMNum = 0, DPNum = 0, seedDis = 0;
    double Timer_Seed = 0, Timer_FM = 0, Timer_DP = 0;
    std::vector<lrsc_seed> seeds;   // --onlyseed: the read's seeds (the reference parks them in SeedFeature::Log(), .cpp:58-62)
};

// Batched processor: one instance per device (the framework runs one worker thread per instance and deals whole
// batches -- contiguous input-order chunks, SURVEY.md section 8e -- to whichever device is free).
class PacBioSelfCorrectionProcess {
public:
    explicit PacBioSelfCorrectionProcess(const PacBioSelfCorrectionParameters& params, size_t worker = 0);
    ~PacBioSelfCorrectionProcess();
    static size_t workers(const PacBioSelfCorrectionParameters& params, int /*thread*/) { return params.devices.size(); }
    std::vector<PacBioSelfCorrectionResult> process_batch(const std::vector<SequenceWorkItem>& items);
    PacBioSelfCorrectionResult process(const SequenceWorkItem& item);     // classic concept (one-item batch)
private:
    // --debugseed / --onlyseed: the per-read files of LongReadProbe.cpp:109-113,123-175,220-225 and .cpp:71-75,130-140
    void runWithDiagnostics(const std::vector<SequenceWorkItem>& items, std::vector<PacBioSelfCorrectionResult>& results,
                            uint64_t& nPieces, uint64_t& used);
    const PacBioSelfCorrectionParameters m_params;
    lrsc_ctx* m_ctx = nullptr;
    // staging buffers, kept between batches
    std::string m_bases, m_out;
    std::vector<uint64_t> m_off, m_pieceOff;
    std::vector<lrsc_read_result> m_res;
};

// PacBioSelfCorrectionPostProcess (reference .cpp:250-380): correct.fa / discard.fa + the stats block;
// --onlyseed: total.seed (one line per read with a wrong seed) + the TOTAL line on stdout
class PacBioSelfCorrectionPostProcess {
public:
    explicit PacBioSelfCorrectionPostProcess(const PacBioSelfCorrectionParameters& params);
    ~PacBioSelfCorrectionPostProcess();
    void process(const SequenceWorkItem& workItem, const PacBioSelfCorrectionResult& result);
private:
    const PacBioSelfCorrectionParameters m_params;
    std::vector<char> m_bufCorrect = std::vector<char>(4u << 20), m_bufDiscard = std::vector<char>(1u << 20);   // stream buffers (declared before the streams)
    std::ofstream m_correct, m_discard;
    int64_t m_totalReadsLen = 0, m_correctedLen = 0, m_totalSeedNum = 0, m_totalWalkNum = 0, m_highErrorNum = 0,
            m_exceedDepthNum = 0, m_exceedLeaveNum = 0, m_FMNum = 0, m_DPNum = 0, m_OutcastNum = 0, m_seedDis = 0;
    double m_Timer_Seed = 0, m_Timer_FM = 0, m_Timer_DP = 0;
    static void summarize(FILE* out, const size_t* status, const std::string& subject);       // reference :372-380
    FILE* m_pStatusWriter = nullptr;
    size_t m_status[3] = {0, 0, 0};            // seeds that are correct / wrong / outside every barcode block
};

} // namespace stride
