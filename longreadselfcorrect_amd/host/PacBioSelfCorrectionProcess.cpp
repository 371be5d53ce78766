// PacBioSelfCorrectionProcess.cpp -- batched processor over the C ABI + the reference's post-processor
// (PacBio/PacBioSelfCorrectionProcess.cpp:250-380: FASTA records and the statistics block on stdout).
#include "PacBioSelfCorrectionProcess.h"
#include "BCode.h"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <thread>

namespace stride {

static void orDie(int st, const char* what)
{
    if(st != LRSC_OK) {
        std::cerr << what << ": " << lrsc_strerror(st) << " (" << lrsc_last_error() << ")\n";
        // called from a worker thread while the reader, the other workers and the post-processor may be inside HIP calls: leave
        // without running static destructors / the HIP runtime's teardown under them (a plain exit() can hang there)
        std::cerr.flush(); std::cout.flush();
        std::_Exit(EXIT_FAILURE);
    }
}

PacBioSelfCorrectionProcess::PacBioSelfCorrectionProcess(const PacBioSelfCorrectionParameters& params, size_t worker) : m_params(params)
{
    const int device = m_params.devices[worker % m_params.devices.size()];
    orDie(lrsc_ctx_create(m_params.index, &m_params.p, device, &m_ctx), "lrsc_ctx_create");
}

PacBioSelfCorrectionProcess::~PacBioSelfCorrectionProcess()
{
    lrsc_ctx_destroy(m_ctx);
}

// f(lo, hi) over [0, n) on up to `threads` host threads
template <class F>
static void parallelFor(size_t n, int threads, F f)
{
    const size_t t = std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), n / 256 + 1));
    if(t == 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    for(size_t i = 1; i < t; ++i) th.emplace_back(f, n * i / t, n * (i + 1) / t);
    f((size_t)0, n / t);
    for(std::thread& x : th) x.join();
}

std::vector<PacBioSelfCorrectionResult> PacBioSelfCorrectionProcess::process_batch(const std::vector<SequenceWorkItem>& items)
{
    const size_t n = items.size();
    std::vector<PacBioSelfCorrectionResult> results(n);
    if(n == 0) return results;
    // the batch as the C ABI takes it: one block of bases + offsets
    m_off.assign(n + 1, 0);
    for(size_t i = 0; i < n; ++i) m_off[i + 1] = m_off[i] + items[i].read.seq.size();
    m_bases.resize(m_off[n]);
    parallelFor(n, m_params.threads, [&](size_t lo, size_t hi) {
        for(size_t i = lo; i < hi; ++i) items[i].read.seq.copy(&m_bases[m_off[i]], std::string::npos);
    });
    m_res.resize(n);
    m_pieceOff.resize(std::max<size_t>(m_pieceOff.size(), 2 * n + 16));
    // the device budgets up to 3 x |read| + 390 per seed for a read's output slot (DP consensuses can outgrow their queries): size for it
    // at once -- a second call after LRSC_ERR_CAPACITY would run the whole correction again
    m_out.resize(std::max<size_t>(m_out.size(), m_bases.size() * 3 + 400 * n + 4096));
    uint64_t nPieces = 0, used = 0;
    if(m_params.DebugSeed || m_params.OnlySeed) {
        runWithDiagnostics(items, results, nPieces, used);
        if(m_params.OnlySeed) return results;
    } else {
        int st = lrsc_correct_reads(m_ctx, m_bases.data(), m_off.data(), (uint32_t)n, m_res.data(), m_pieceOff.data(), m_pieceOff.size(),
                                    &m_out[0], m_out.size(), &nPieces, &used);
        if(st == LRSC_ERR_CAPACITY) {
            m_pieceOff.resize(nPieces + 1);
            m_out.resize(used);
            st = lrsc_correct_reads(m_ctx, m_bases.data(), m_off.data(), (uint32_t)n, m_res.data(), m_pieceOff.data(), m_pieceOff.size(),
                                    &m_out[0], m_out.size(), &nPieces, &used);
        }
        orDie(st, "lrsc_correct_reads");
    }
    parallelFor(n, m_params.threads, [&](size_t lo, size_t hi) {
        for(size_t i = lo; i < hi; ++i) {
            const lrsc_read_result& r = m_res[i];
            PacBioSelfCorrectionResult& o = results[i];
            o.readid = items[i].read.id;
            o.merge = r.merge != 0;
            o.totalReadsLen = r.total_reads_len; o.correctedLen = r.corrected_len; o.totalSeedNum = r.total_seed_num;
            o.totalWalkNum = r.total_walk_num; o.highErrorNum = r.high_error_num; o.exceedDepthNum = r.exceed_depth_num;
            o.exceedLeaveNum = r.exceed_leave_num; o.FMNum = r.fm_num; o.DPNum = r.dp_num; o.seedDis = r.seed_dis;
            o.correctedStrs.reserve(r.n_pieces);
            for(uint64_t p = r.piece_first; p < r.piece_first + r.n_pieces; ++p)
                o.correctedStrs.emplace_back(m_out.data() + m_pieceOff[p], m_pieceOff[p + 1] - m_pieceOff[p]);
        }
    });
    for(size_t i = 0; i < n; ++i)
        if(m_res[i].status != LRSC_READ_OK)
            std::cerr << "Warning: read " << items[i].read.id << " exceeds an internal capacity of the device path (status " << m_res[i].status
                      << ", include/lrsc.h lrsc_read_status): written to discard.fa uncorrected\n";
    return results;
}

static void writeSeeds(const std::string& path, const std::string& seq, const lrsc_seed* s, size_t n)
{
    std::ofstream w(path.c_str());
    for(size_t i = 0; i < n; ++i)         // operator<<(SeedVector), PacBio/SeedFeature.cpp:11-20
        w << seq.substr((size_t)s[i].start, (size_t)s[i].len) << '\t' << s[i].max_fixed_mer_freq << '\t' << s[i].start << '\t'
          << (s[i].is_repeat ? "Yes" : "No") << '\n';
}

void PacBioSelfCorrectionProcess::runWithDiagnostics(const std::vector<SequenceWorkItem>& items,
                                                     std::vector<PacBioSelfCorrectionResult>& results, uint64_t& nPieces, uint64_t& used)
{
    const size_t n = items.size();
    const std::string& dir = m_params.directory;
    lrsc_batch* b = nullptr;
    orDie(lrsc_batch_create(m_ctx, m_bases.data(), m_off.data(), (uint32_t)n, &b), "lrsc_batch_create");
    orDie(lrsc_batch_set_debug(b, LRSC_DEBUG_OUTCASTS | LRSC_DEBUG_RATIO | (m_params.OnlySeed ? 0 : LRSC_DEBUG_WALKS)), "lrsc_batch_set_debug");
    orDie(lrsc_batch_find_seeds(m_ctx, b), "lrsc_batch_find_seeds");
    std::vector<uint32_t> count(n), outCount(n);
    uint64_t total = 0, outTotal = 0;
    orDie(lrsc_batch_seeds(m_ctx, b, count.data(), nullptr, ~0ull, &total, nullptr), "lrsc_batch_seeds");
    std::vector<lrsc_seed> seeds(total + 1);
    orDie(lrsc_batch_seeds(m_ctx, b, count.data(), seeds.data(), seeds.size(), &total, nullptr), "lrsc_batch_seeds");
    orDie(lrsc_batch_outcast_seeds(m_ctx, b, outCount.data(), nullptr, ~0ull, &outTotal), "lrsc_batch_outcast_seeds");
    std::vector<lrsc_seed> outcasts(outTotal + 1);
    orDie(lrsc_batch_outcast_seeds(m_ctx, b, outCount.data(), outcasts.data(), outcasts.size(), &outTotal), "lrsc_batch_outcast_seeds");
    std::vector<float> ratio(m_bases.size() + 1);
    orDie(lrsc_batch_repeat_ratio(m_ctx, b, ratio.data()), "lrsc_batch_repeat_ratio");
    std::vector<uint64_t> first(n + 1, 0), outFirst(n + 1, 0);
    for(size_t i = 0; i < n; ++i) { first[i + 1] = first[i] + count[i]; outFirst[i + 1] = outFirst[i] + outCount[i]; }

    const int startKmerLen = m_params.p.start_kmer_len;
    parallelFor(n, m_params.threads, [&](size_t lo, size_t hi) {
        for(size_t i = lo; i < hi; ++i) {
            const std::string& id = items[i].read.id;
            const std::string& seq = items[i].read.seq;
            if((int)seq.size() < startKmerLen) continue;                      // searchSeedsWithHybridKmers returns before any file (:37)
            {   // extend/<id>.log: position, repeat ratio (getSeqAttribute, :123-124,170-171)
                std::ofstream w((dir + "extend/" + id + ".log").c_str());
                const float* r = ratio.data() + m_off[i];
                for(size_t pos = 0; pos < seq.size(); ++pos) w << pos << '\t' << r[pos] << '\n';
            }
            if(count[i] + outCount[i] >= 2)                                   // removeHitchhikingSeeds returns early below two seeds (:189)
                writeSeeds(dir + "seed/error/" + id + ".seed", seq, outcasts.data() + outFirst[i], outCount[i]);
            writeSeeds(dir + "seed/" + id + ".seed", seq, seeds.data() + first[i], count[i]);
        }
    });
    if(m_params.OnlySeed) {
        for(size_t i = 0; i < n; ++i) {
            results[i].readid = items[i].read.id;
            results[i].totalSeedNum = count[i];
            results[i].totalReadsLen = (int64_t)items[i].read.seq.size();
            results[i].seeds.assign(seeds.begin() + first[i], seeds.begin() + first[i + 1]);
        }
        lrsc_batch_destroy(b);
        return;
    }
    int st = lrsc_batch_correct(m_ctx, b, m_res.data(), m_pieceOff.data(), m_pieceOff.size(), &m_out[0], m_out.size(), &nPieces, &used);
    if(st == LRSC_ERR_CAPACITY) {
        m_pieceOff.resize(nPieces + 1);
        m_out.resize(used);
        st = lrsc_batch_correct(m_ctx, b, m_res.data(), m_pieceOff.data(), m_pieceOff.size(), &m_out[0], m_out.size(), &nPieces, &used);
    }
    orDie(st, "lrsc_batch_correct");
    std::vector<uint8_t> log(total + 1);
    orDie(lrsc_batch_walk_log(m_ctx, b, log.data(), log.size()), "lrsc_batch_walk_log");
    lrsc_batch_destroy(b);
    parallelFor(n, m_params.threads, [&](size_t lo, size_t hi) {
        for(size_t i = lo; i < hi; ++i) {
            if(count[i] < 2 || m_res[i].status != LRSC_READ_OK) continue;     // initCorrect opens the writers after its size check (:63-75)
            const std::string& id = items[i].read.id;
            std::ofstream ext((dir + "extend/" + id + ".ext").c_str()), dp((dir + "extend/" + id + ".dp").c_str());
            const lrsc_seed* s = seeds.data() + first[i];
            const uint8_t* l = log.data() + first[i];
            for(uint32_t t = 1; t < count[i]; ++t) {
                if(l[t] == 0) continue;
                ext << s[t - 1].start << "\t" << s[t].start << "\t" << (int)(l[t] & 15) << "\n";      // :130-131
                if(l[t] & 16) dp << s[t - 1].start << "\t" << s[t].start << "\n";                     // :139-140
            }
        }
    });
}

PacBioSelfCorrectionResult PacBioSelfCorrectionProcess::process(const SequenceWorkItem& item)
{
    return process_batch(std::vector<SequenceWorkItem>(1, item))[0];
}

// ---- post-processor ---------------------------------------------------------------------------------------
PacBioSelfCorrectionPostProcess::PacBioSelfCorrectionPostProcess(const PacBioSelfCorrectionParameters& params) : m_params(params)
{
    if(m_params.OnlySeed) {
        m_pStatusWriter = fopen((m_params.directory + "total.seed").c_str(), "w");
        if(!m_pStatusWriter) {
            std::cerr << "Error: could not open " << m_params.directory << "total.seed for write\n";
            exit(EXIT_FAILURE);
        }
        return;
    }
    m_correct.rdbuf()->pubsetbuf(m_bufCorrect.data(), (std::streamsize)m_bufCorrect.size());
    m_discard.rdbuf()->pubsetbuf(m_bufDiscard.data(), (std::streamsize)m_bufDiscard.size());
    m_correct.open((m_params.directory + "correct.fa").c_str());
    m_discard.open((m_params.directory + "discard.fa").c_str());
    if(!m_correct || !m_discard) {
        std::cerr << "Error: could not open " << m_params.directory << "correct.fa / discard.fa for write\n";
        exit(EXIT_FAILURE);
    }
}

PacBioSelfCorrectionPostProcess::~PacBioSelfCorrectionPostProcess()
{
    if(m_params.OnlySeed) {
        summarize(stdout, m_status, "TOTAL");
        fclose(m_pStatusWriter);
        return;
    }
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

void PacBioSelfCorrectionPostProcess::summarize(FILE* out, const size_t* status, const std::string& subject)
{
    const size_t sum = status[0] + status[1] + status[2];
    const double crt = (double)(100 * status[0]) / sum, err = (double)(100 * status[1]) / sum, non = (double)(100 * status[2]) / sum;
    if(status[1] > 0) fprintf(out, "%s [%ld] %.2lf%% %.2lf%% %.2lf%%\n", subject.c_str(), (long)sum, crt, err, non);
}

void PacBioSelfCorrectionPostProcess::process(const SequenceWorkItem& workItem, const PacBioSelfCorrectionResult& result)
{
    if(m_params.OnlySeed) {
        // every seed: 0 = lies in a barcode block and is a correct k-mer, 1 = lies in one and is not, 2 = in no block (reference :317-335)
        size_t status[3] = {0, 0, 0};
        const std::string& seq = workItem.read.seq;
        const BCode::BCodeVector& blocks = BCode::Log()[workItem.read.id];
        for(const lrsc_seed& s : result.seeds) {
            int m = 2;
            for(const BCode& blk : blocks)
                if(s.start >= blk.getStart() && s.start + s.len - 1 <= blk.getEnd()) {
                    m = BCode::validate(s.start, s.len, blk, seq) ? 0 : 1;
                    break;
                }
            status[m]++;
        }
        summarize(m_pStatusWriter, status, result.readid);
        for(int j = 0; j < 3; ++j) m_status[j] += status[j];
        return;
    }
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
