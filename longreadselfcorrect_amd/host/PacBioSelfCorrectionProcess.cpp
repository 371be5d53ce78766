// PacBioSelfCorrectionProcess.cpp -- batched processor over the C ABI + the reference's post-processor
// (PacBio/PacBioSelfCorrectionProcess.cpp:250-380: FASTA records and the statistics block on stdout).
#include "PacBioSelfCorrectionProcess.h"

#include <cstdlib>
#include <iostream>
#include <thread>

namespace stride {

static void orDie(int st, const char* what)
{
    if(st != LRSC_OK) {
        std::cerr << what << ": " << lrsc_strerror(st) << " (" << lrsc_last_error() << ")\n";
        exit(EXIT_FAILURE);
    }
}

PacBioSelfCorrectionProcess::PacBioSelfCorrectionProcess(const PacBioSelfCorrectionParameters& params) : m_params(params)
{
    for(int d : m_params.devices) {
        lrsc_ctx* ctx = nullptr;
        orDie(lrsc_ctx_create(m_params.index, &m_params.p, d, &ctx), "lrsc_ctx_create");
        m_ctx.push_back(ctx);
    }
}

PacBioSelfCorrectionProcess::~PacBioSelfCorrectionProcess()
{
    for(lrsc_ctx* c : m_ctx) lrsc_ctx_destroy(c);
}

namespace {
struct Shard {
    size_t first = 0, count = 0;
    std::string bases, out;
    std::vector<uint64_t> off, pieceOff;
    std::vector<lrsc_read_result> res;
    int status = LRSC_OK;
    std::string error;
};
}

std::vector<PacBioSelfCorrectionResult> PacBioSelfCorrectionProcess::process_batch(const std::vector<SequenceWorkItem>& items)
{
    // contiguous input-order chunks, one per device; each device has its own ctx
    const size_t nd = m_ctx.size();
    std::vector<Shard> shards(nd);
    const size_t per = (items.size() + nd - 1) / nd;
    for(size_t d = 0; d < nd; ++d) {
        Shard& s = shards[d];
        s.first = std::min(items.size(), d * per);
        s.count = std::min(items.size(), (d + 1) * per) - s.first;
        s.off.assign(1, 0);
        for(size_t i = 0; i < s.count; ++i) {
            s.bases += items[s.first + i].read.seq;
            s.off.push_back(s.bases.size());
        }
    }
    auto run = [&](size_t d) {
        Shard& s = shards[d];
        if(s.count == 0) return;
        s.res.resize(s.count);
        s.pieceOff.resize(2 * s.count + 16);
        s.out.resize(s.bases.size() * 2 + 4096);
        uint64_t nPieces = 0, used = 0;
        int st = lrsc_correct_reads(m_ctx[d], s.bases.data(), s.off.data(), (uint32_t)s.count, s.res.data(), s.pieceOff.data(),
                                    s.pieceOff.size(), &s.out[0], s.out.size(), &nPieces, &used);
        if(st == LRSC_ERR_CAPACITY) {
            s.pieceOff.resize(nPieces + 1);
            s.out.resize(used);
            st = lrsc_correct_reads(m_ctx[d], s.bases.data(), s.off.data(), (uint32_t)s.count, s.res.data(), s.pieceOff.data(),
                                    s.pieceOff.size(), &s.out[0], s.out.size(), &nPieces, &used);
        }
        s.status = st;
        if(st != LRSC_OK) s.error = lrsc_last_error();
    };
    std::vector<std::thread> th;
    for(size_t d = 1; d < nd; ++d) th.emplace_back(run, d);
    run(0);
    for(auto& t : th) t.join();

    std::vector<PacBioSelfCorrectionResult> results(items.size());
    for(size_t d = 0; d < nd; ++d) {
        const Shard& s = shards[d];
        if(s.status != LRSC_OK) {
            std::cerr << "lrsc_correct_reads: " << lrsc_strerror(s.status) << " (" << s.error << ")\n";
            exit(EXIT_FAILURE);
        }
        for(size_t i = 0; i < s.count; ++i) {
            const lrsc_read_result& r = s.res[i];
            PacBioSelfCorrectionResult& o = results[s.first + i];
            o.readid = items[s.first + i].read.id;
            o.merge = r.merge != 0;
            o.totalReadsLen = r.total_reads_len; o.correctedLen = r.corrected_len; o.totalSeedNum = r.total_seed_num;
            o.totalWalkNum = r.total_walk_num; o.highErrorNum = r.high_error_num; o.exceedDepthNum = r.exceed_depth_num;
            o.exceedLeaveNum = r.exceed_leave_num; o.FMNum = r.fm_num; o.DPNum = r.dp_num; o.seedDis = r.seed_dis;
            for(uint64_t p = r.piece_first; p < r.piece_first + r.n_pieces; ++p)
                o.correctedStrs.push_back(s.out.substr(s.pieceOff[p], s.pieceOff[p + 1] - s.pieceOff[p]));
        }
    }
    return results;
}

PacBioSelfCorrectionResult PacBioSelfCorrectionProcess::process(const SequenceWorkItem& item)
{
    return process_batch(std::vector<SequenceWorkItem>(1, item))[0];
}

// ---- post-processor ---------------------------------------------------------------------------------------
PacBioSelfCorrectionPostProcess::PacBioSelfCorrectionPostProcess(const PacBioSelfCorrectionParameters& params) : m_params(params)
{
    m_correct.open((m_params.directory + "correct.fa").c_str());
    m_discard.open((m_params.directory + "discard.fa").c_str());
    if(!m_correct || !m_discard) {
        std::cerr << "Error: could not open " << m_params.directory << "correct.fa / discard.fa for write\n";
        exit(EXIT_FAILURE);
    }
}

PacBioSelfCorrectionPostProcess::~PacBioSelfCorrectionPostProcess()
{
    // reference :288-306 (same text, same float formatting)
    if(m_totalWalkNum > 0 && m_totalReadsLen > 0) {
        m_OutcastNum = m_totalWalkNum - m_FMNum - m_DPNum;
        std::cout << "\n"
                  << "TotalReadsLen: " << m_totalReadsLen << "\n"
                  << "CorrectedLen: " << m_correctedLen << ", ratio: " << (float)(m_correctedLen) / m_totalReadsLen << "\n"
                  << "TotalSeedNum: " << m_totalSeedNum << "\n"
                  << "TotalWalkNum: " << m_totalWalkNum << "\n"
                  << "FMNum: " << m_FMNum << ", ratio: " << (float)(m_FMNum * 100) / m_totalWalkNum << "%\n"
                  << "DPNum: " << m_DPNum << ", ratio: " << (float)(m_DPNum * 100) / m_totalWalkNum << "%\n"
                  << "OutcastNum: " << m_OutcastNum << ", ratio: " << (float)(m_OutcastNum * 100) / m_totalWalkNum << "%\n"
                  << "HighErrorNum: " << m_highErrorNum << ", ratio: " << (float)(m_highErrorNum * 100) / (m_DPNum + m_OutcastNum) << "%\n"
                  << "ExceedDepthNum: " << m_exceedDepthNum << ", ratio: " << (float)(m_exceedDepthNum * 100) / (m_DPNum + m_OutcastNum) << "%\n"
                  << "ExceedLeaveNum: " << m_exceedLeaveNum << ", ratio: " << (float)(m_exceedLeaveNum * 100) / (m_DPNum + m_OutcastNum) << "%\n"
                  << "DisBetweenSeeds: " << m_seedDis / m_totalWalkNum << "\n"
                  << "Time of searching Seeds: " << m_Timer_Seed << "\n"
                  << "Time of searching FM: " << m_Timer_FM << "\n"
                  << "Time of searching DP: " << m_Timer_DP << "\n";
    }
}

void PacBioSelfCorrectionPostProcess::process(const SequenceWorkItem& workItem, const PacBioSelfCorrectionResult& result)
{
    if(result.merge) {
        m_totalReadsLen += result.totalReadsLen;
        m_correctedLen += result.correctedLen;
        m_totalSeedNum += result.totalSeedNum;
        m_totalWalkNum += result.totalWalkNum;
        m_highErrorNum += result.highErrorNum;
        m_exceedDepthNum += result.exceedDepthNum;
        m_exceedLeaveNum += result.exceedLeaveNum;
        m_FMNum += result.FMNum;
        m_DPNum += result.DPNum;
        m_seedDis += result.seedDis;
        m_Timer_Seed += result.Timer_Seed;
        m_Timer_FM += result.Timer_FM;
        m_Timer_DP += result.Timer_DP;
        for(size_t index = 0; index < result.correctedStrs.size(); ++index) {
            const std::string flag = m_params.p.split ? ("_" + std::to_string(index)) : "";
            m_correct << ">" << workItem.read.id << flag << "\n" << result.correctedStrs[index] << "\n";   // SeqItem::write (Util/Util.h:57-61)
        }
    } else {
        m_discard << ">" << workItem.read.id << "\n" << workItem.read.seq << "\n";
    }
}

} // namespace stride
