// oracle/process_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see process_oracle.hpp).
#include "process_oracle.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <sstream>

#include "dp_oracle.hpp"

namespace lrsc_oracle {

// PacBioSelfCorrectionProcess::process (PacBio/PacBioSelfCorrectionProcess.cpp:23-54)
CorrectionResult SelfCorrectionProcess::process(const std::string& id, const std::string& readSeqIn)
{
    CorrectionResult result;
    result.readid = id;
    std::string readSeq = readSeqIn;
    const size_t readSeqLen = readSeq.length();
    SeedFeature::SeedVector seedVec, pieceVec;

    // allocate space for kmers on the sequence
    KmerLog log;
    allocateKmerLog(log, m_params.pool, readSeqLen);

    // Part 1: start searching seeds
    searchSeedsWithHybridKmers(m_params.probe, log, readSeq, seedVec);
    result.totalSeedNum = seedVec.size();
    result.seeds = seedVec;

    // Part 2: start correcting sequence
    initCorrect(readSeq, seedVec, pieceVec, result);

    log.clear();

    result.merge = !pieceVec.empty();
    result.totalReadsLen = readSeq.length();
    for(const auto& iter : pieceVec) result.correctedStrs.push_back(iter.seedStr);
    return result;
}

// initCorrect (:56-157)
void SelfCorrectionProcess::initCorrect(std::string& readSeq, const SeedFeature::SeedVector& seedVec,
                                        SeedFeature::SeedVector& pieceVec, CorrectionResult& result)
{
    if(m_params.OnlySeed) return;
    if(seedVec.size() < 2) return;

    // push first seed into vector and reserve space for fast expansion
    pieceVec.push_back(seedVec[0]);
    pieceVec.back().seedStr.reserve(readSeq.length());

    const SeedFeature* lastSeed = &seedVec[0];     // instrumentation only: the raw seed the source currently ends with
    int lastVia = -1;                              // -1 first seed, 0 FM, 1 DP, 2 raw/split
    int case_number = 1;
    for(SeedFeature::SeedVector::const_iterator iterTarget = seedVec.begin() + 1; iterTarget != seedVec.end();
        iterTarget++, case_number++) {
        int isFMExtensionSuccess = 0, firstFMExtensionType = 0;
        SeedFeature& source = pieceVec.back();
        std::string mergedSeq;
        const int walkSrcStart = source.seedStartPos;
        const uint64_t steps_before = result.walk_stats.steps, exp_before = result.walk_stats.leaf_expansions;
        const int gap_now = iterTarget->seedStartPos - source.seedEndPos - 1;
        {   // instrumentation only (no effect on the result): predicted vs actual source k-mer of this walk
            const SeedFeature& target = *iterTarget;
            auto ksize = [&](int endBest, bool srcRepeat, int srcLen) {
                int k = std::min(endBest, target.startBestKmerSize) - 2;
                if(srcRepeat || target.isRepeat) { k = std::min(srcLen, target.seedLen); k = std::min(k, m_params.startKmerLen + 2); }
                return k;
            };
            const int k_true = ksize(source.endBestKmerSize, source.isRepeat, source.seedLen);
            const int k_pred = ksize(lastSeed->endBestKmerSize, lastSeed->isRepeat, lastSeed->seedLen);
            result.spec[0]++;
            if(k_true != k_pred || k_pred > lastSeed->seedLen || k_pred < 0) result.spec[5]++;
            else if(source.seedStr.substr(source.seedLen - k_true) == lastSeed->seedStr.substr(lastSeed->seedLen - k_pred)) result.spec[1]++;
            else result.spec[2 + (lastVia < 0 ? 2 : lastVia)]++;
        }

        for(int next = 0; next < m_params.nextTarget && (iterTarget + next) != seedVec.end(); next++) {
            const SeedFeature& target = *(iterTarget + next);
            isFMExtensionSuccess = correctByFMExtension(source, target, readSeq, mergedSeq, result);
            firstFMExtensionType = (next == 0 ? isFMExtensionSuccess : firstFMExtensionType);
            if(isFMExtensionSuccess > 0) {
                result.totalWalkNum++;
                result.walks.push_back({walkSrcStart, target.seedStartPos, isFMExtensionSuccess, 0, gap_now, (int)(result.walk_stats.steps - steps_before), (int)(result.walk_stats.leaf_expansions - exp_before)});
                source.append(mergedSeq, target);
                iterTarget += next;
                case_number += next;
                lastSeed = &*iterTarget; lastVia = 0;
                break;
            }
        }

        if(isFMExtensionSuccess <= 0) {
            const SeedFeature& target = *iterTarget;
            switch(firstFMExtensionType) {
                case -1: result.highErrorNum++; break;
                case -2: result.exceedDepthNum++; break;
                case -3: result.exceedLeaveNum++; break;
                default:
                    std::cerr << "Does it really happen?\n";
                    exit(EXIT_FAILURE);
            }

            result.totalWalkNum++;
            bool isMSAlignmentSuccess = correctByMSAlignment(source, target, readSeq, mergedSeq, result);
            result.walks.push_back({walkSrcStart, target.seedStartPos, firstFMExtensionType, isMSAlignmentSuccess ? 1 : 2, gap_now, (int)(result.walk_stats.steps - steps_before), (int)(result.walk_stats.leaf_expansions - exp_before)});
            if(isMSAlignmentSuccess)
                source.append(mergedSeq, target);
            else {
                if(m_params.Split)
                    pieceVec.push_back(target);
                else {
                    mergedSeq = readSeq.substr((source.seedEndPos + 1), (target.seedEndPos - source.seedEndPos));
                    source.append(mergedSeq, target);
                }
                result.correctedLen += target.seedStr.length();
            }
            lastSeed = &target; lastVia = isMSAlignmentSuccess ? 1 : 2;
        }
    }
}

// correctByFMExtension (:159-206)
int SelfCorrectionProcess::correctByFMExtension(const SeedFeature& source, const SeedFeature& target,
                                                const std::string& in, std::string& out, CorrectionResult& result)
{
    int interval = target.seedStartPos - source.seedEndPos - 1;
    int extendKmerSize = std::min(source.endBestKmerSize, target.startBestKmerSize) - 2;
    if(source.isRepeat || target.isRepeat) {
        extendKmerSize = std::min(source.seedLen, target.seedLen);
        extendKmerSize = std::min(extendKmerSize, m_params.startKmerLen + 2);
    }
    std::string src, trg, path;
    src = source.seedStr.substr(source.seedLen - extendKmerSize);
    trg = target.seedStr;
    path = in.substr(source.seedEndPos + 1, interval);
    int min_SA_threshold = 3, isFMExtensionSuccess = 0;
    min_SA_threshold = m_params.PBcoverage > 60 ? ((m_params.PBcoverage / 60) * 3) : min_SA_threshold;
    bool isFromRtoU = source.isRepeat && !target.isRepeat;
    if(isFromRtoU) {
        std::swap(src, trg);
        src = reverse_complement(src);
        trg = reverse_complement(trg);
        path = reverse_complement(path);
    }

    FMWalkResult2 fmwalkresult;
    LongReadSelfCorrectByOverlap OverlapTree(src, path, trg, interval, extendKmerSize, extendKmerSize + 2,
                                             m_params.FM_params, min_SA_threshold);
    isFMExtensionSuccess = OverlapTree.extendOverlap(fmwalkresult);
    result.walk_stats.steps += OverlapTree.stats.steps;
    result.walk_stats.leaf_expansions += OverlapTree.stats.leaf_expansions;
    result.walk_stats.refine_calls += OverlapTree.stats.refine_calls;

    if(isFMExtensionSuccess < 0) return isFMExtensionSuccess;
    if(isFromRtoU) {
        fmwalkresult.mergedSeq = reverse_complement(fmwalkresult.mergedSeq);
        fmwalkresult.mergedSeq += reverse_complement(src).substr(extendKmerSize);
    }
    out = fmwalkresult.mergedSeq;
    out.erase(0, extendKmerSize);
    result.correctedLen += out.length();
    result.seedDis += interval;
    result.FMNum++;
    return isFMExtensionSuccess;
}

// correctByMSAlignment (:208-245)
bool SelfCorrectionProcess::correctByMSAlignment(const SeedFeature& source, const SeedFeature& target,
                                                 const std::string& in, std::string& out, CorrectionResult& result)
{
    if(m_params.NoDp) return false;
    int interval = target.seedStartPos - source.seedEndPos - 1;
    int extendKmerSize = std::min(source.endBestKmerSize, target.startBestKmerSize) - 2;
    if(source.isRepeat || target.isRepeat) {
        extendKmerSize = std::min(source.seedLen, target.seedLen);
        extendKmerSize = std::min(extendKmerSize, m_params.startKmerLen + 2);
    }
    std::string src, trg, path;
    src = source.seedStr.substr(source.seedLen - extendKmerSize);
    trg = target.seedStr;
    path = in.substr(source.seedEndPos + 1, interval);
    path = src + path + trg;
    double identity = 0.65;
    size_t totalMaxFixedMerFreq = source.maxFixedMerFreq + target.maxFixedMerFreq, min_call_coverage = 15;
    identity += (totalMaxFixedMerFreq > 50 ? 0.05 : 0);
    identity += (totalMaxFixedMerFreq > 100 ? 0.05 : 0);
    min_call_coverage = totalMaxFixedMerFreq > 50 ? totalMaxFixedMerFreq * 0.4 : min_call_coverage;

    MultipleAlignment maquery = buildMultipleAlignment(path, extendKmerSize, extendKmerSize, path.length() / 10, identity,
                                                       m_params.PBcoverage, m_params.indices);

    if(maquery.getNumRows() <= 3) return false;
    out = maquery.calculateBaseConsensus(min_call_coverage, -1);
    out.erase(0, extendKmerSize);
    result.correctedLen += out.length();
    result.seedDis += interval;
    result.DPNum++;
    return true;
}

// PacBioSelfCorrectionPostProcess::process (:313-370)
void SelfCorrectionPostProcess::process(const std::string& id, const std::string& readSeq, const CorrectionResult& result)
{
    if(result.merge) {
        totalReadsLen += result.totalReadsLen;
        correctedLen += result.correctedLen;
        totalSeedNum += result.totalSeedNum;
        totalWalkNum += result.totalWalkNum;
        highErrorNum += result.highErrorNum;
        exceedDepthNum += result.exceedDepthNum;
        exceedLeaveNum += result.exceedLeaveNum;
        FMNum += result.FMNum;
        DPNum += result.DPNum;
        seedDis += result.seedDis;
        for(size_t index = 0; index < result.correctedStrs.size(); ++index) {
            std::string flag = m_split ? ("_" + std::to_string(index)) : "";
            // SeqItem::write (Util/Util.h:57-61): ">" id "\n" seq "\n"
            correct_fa += ">" + id + flag + "\n" + result.correctedStrs[index] + "\n";
        }
    } else {
        discard_fa += ">" + id + "\n" + readSeq + "\n";
    }
}

// the integer lines of the destructor's stats block (:288-306); float ratios and the three timer
// lines are excluded from parity
std::string SelfCorrectionPostProcess::stats_text() const
{
    std::ostringstream o;
    if(totalWalkNum > 0 && totalReadsLen > 0) {
        const int64_t outcast = totalWalkNum - FMNum - DPNum;
        o << "TotalReadsLen: " << totalReadsLen << "\n"
          << "CorrectedLen: " << correctedLen << "\n"
          << "TotalSeedNum: " << totalSeedNum << "\n"
          << "TotalWalkNum: " << totalWalkNum << "\n"
          << "FMNum: " << FMNum << "\n"
          << "DPNum: " << DPNum << "\n"
          << "OutcastNum: " << outcast << "\n"
          << "HighErrorNum: " << highErrorNum << "\n"
          << "ExceedDepthNum: " << exceedDepthNum << "\n"
          << "ExceedLeaveNum: " << exceedLeaveNum << "\n"
          << "DisBetweenSeeds: " << seedDis / totalWalkNum << "\n";
    }
    return o.str();
}

} // namespace lrsc_oracle
