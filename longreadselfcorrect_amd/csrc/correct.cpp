// correct.cpp -- the per-read workflow of PacBioSelfCorrectionProcess::process / initCorrect /
// correctByFMExtension (reference: PacBio/PacBioSelfCorrectionProcess.cpp:23-206) on top of the device
// stages: seeds from lrsc_batch_find_seeds, seed-to-seed walks from lrsc_extend_walks.
//
// The walks of one read form a sequential chain (the next source k-mer is the tail of what was just
// appended, :163-170), so the batch advances in ROUNDS: round j runs the j-th pending walk of every
// read that still has one -- ~1e5 independent walks per launch for a 100k-read batch -- and the host
// stitches the results between rounds.  Pure host C++ over the C ABI; no device code here.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lrsc.h"

namespace {

struct Piece {               // SeedFeature as the correction loop uses it (PacBio/SeedFeature.h:22-45)
    std::string seedStr;
    int seedLen, seedStartPos, seedEndPos, maxFixedMerFreq, startBestKmerSize, endBestKmerSize;
    bool isRepeat;
    void append(const std::string& ext, const Piece& target)     // SeedFeature::append (SeedFeature.h:22-33)
    {
        seedStr += ext;
        seedLen += (int)ext.length();
        startBestKmerSize = target.startBestKmerSize;
        endBestKmerSize = target.endBestKmerSize;
        isRepeat = target.isRepeat;
        maxFixedMerFreq = target.maxFixedMerFreq;
        seedStartPos = target.seedStartPos;
        seedEndPos = target.seedEndPos;
    }
};

Piece make_piece(const lrsc_seed& s, const char* read)
{
    Piece p;
    p.seedStr.assign(read + s.start, (size_t)s.len);
    p.seedLen = s.len;
    p.seedStartPos = s.start;
    p.seedEndPos = s.start + s.len - 1;
    p.maxFixedMerFreq = s.max_fixed_mer_freq;
    p.isRepeat = s.is_repeat != 0;
    p.startBestKmerSize = s.start_best_kmer_size;
    p.endBestKmerSize = s.end_best_kmer_size;
    return p;
}

std::string revcomp(const std::string& s)
{
    std::string o(s.size(), 'A');
    for(size_t i = 0; i < s.size(); ++i) {
        const char c = s[s.size() - 1 - i];
        o[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    return o;
}

struct ReadState {
    std::vector<Piece> seeds;        // seedVec
    std::vector<Piece> pieces;       // pieceVec
    size_t it = 1;                   // iterTarget (index into seeds)
    int next = 0;                    // the inner `next` loop of initCorrect (:85)
    int firstType = 0;               // firstFMExtensionType
    bool active = false;
    // pending walk
    bool rtou = false;
    int extendKmerSize = 0, interval = 0;
    std::string src;                 // as passed to the walk (after the R-to-U swap)
};

} // namespace

extern "C" int lrsc_correct_reads(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                                  lrsc_read_result* res, uint64_t* piece_off, uint64_t piece_cap, char* out,
                                  uint64_t out_cap, uint64_t* n_pieces_out, uint64_t* out_used)
{
    if(!ctx || !reads || !read_off || !res || !n_pieces_out || !out_used) return LRSC_ERR_ARG;
    lrsc_params params_v;
    if(lrsc_ctx_get_params(ctx, &params_v) != LRSC_OK) return LRSC_ERR_ARG;
    const lrsc_params* params = &params_v;
    *n_pieces_out = 0; *out_used = 0;
    if(n_reads == 0) return LRSC_OK;
    // Default: everything on the device (lrsc_batch_correct).  LRSC_CORRECT_MODE=rounds keeps the host-stitched
    // rounds below, which the tests run as an independent cross-check of the persistent kernel.
    const char* mode = std::getenv("LRSC_CORRECT_MODE");
    if(!(mode && std::strcmp(mode, "rounds") == 0) || !params->no_dp) {     // the rounds cross-check covers --nodp only
        lrsc_batch* db = nullptr;
        int dst = lrsc_batch_create(ctx, reads, read_off, n_reads, &db);
        if(dst != LRSC_OK) return dst;
        dst = lrsc_batch_correct(ctx, db, res, piece_off, piece_cap, out, out_cap, n_pieces_out, out_used);
        lrsc_batch_destroy(db);
        return dst;
    }

    // ---- Part 1: seeds (LongReadProbe::searchSeedsWithHybridKmers) on the device ----------------------------------
    lrsc_batch* batch = nullptr;
    int st = lrsc_batch_create(ctx, reads, read_off, n_reads, &batch);
    if(st != LRSC_OK) return st;
    st = lrsc_batch_find_seeds(ctx, batch);
    std::vector<uint32_t> seed_count(n_reads);
    std::vector<lrsc_seed> seeds;
    if(st == LRSC_OK) {
        uint64_t n = 0;
        st = lrsc_batch_seeds(ctx, batch, seed_count.data(), nullptr, 0, &n, nullptr);
        if(st == LRSC_OK || st == LRSC_ERR_CAPACITY) {
            seeds.resize(n);
            st = lrsc_batch_seeds(ctx, batch, seed_count.data(), seeds.data(), n, &n, nullptr);
        }
    }
    lrsc_batch_destroy(batch);
    if(st != LRSC_OK) return st;

    // ---- Part 2: initCorrect (:56-157) -------------------------------------------------------------------------------
    std::vector<ReadState> rs(n_reads);
    size_t k = 0;
    for(uint32_t r = 0; r < n_reads; ++r) {
        std::memset(&res[r], 0, sizeof(res[r]));
        res[r].total_reads_len = (int64_t)(read_off[r + 1] - read_off[r]);
        res[r].total_seed_num = seed_count[r];
        ReadState& s = rs[r];
        const char* read = reads + read_off[r];
        for(uint32_t i = 0; i < seed_count[r]; ++i) s.seeds.push_back(make_piece(seeds[k + i], read));
        k += seed_count[r];
        if(s.seeds.size() >= 2) {
            s.pieces.push_back(s.seeds[0]);
            s.active = true;
        }
    }
    const int min_SA = params->pb_coverage > 60 ? ((params->pb_coverage / 60) * 3) : 3;     // :174-175

    std::vector<uint32_t> act;
    std::vector<lrsc_walk_desc> descs;
    std::vector<lrsc_walk_result> wres;
    std::string seq, arena;
    while(true) {
        act.clear(); descs.clear(); seq.clear();
        for(uint32_t r = 0; r < n_reads; ++r) {
            ReadState& s = rs[r];
            if(!s.active) continue;
            if(s.it >= s.seeds.size()) { s.active = false; continue; }
            // correctByFMExtension (:159-190) for target = *(iterTarget + next)
            const Piece& source = s.pieces.back();
            const Piece& target = s.seeds[s.it + (size_t)s.next];
            const int interval = target.seedStartPos - source.seedEndPos - 1;
            int extendKmerSize = std::min(source.endBestKmerSize, target.startBestKmerSize) - 2;
            if(source.isRepeat || target.isRepeat) {
                extendKmerSize = std::min(source.seedLen, target.seedLen);
                extendKmerSize = std::min(extendKmerSize, params->start_kmer_len + 2);
            }
            if(extendKmerSize <= 0 || extendKmerSize > source.seedLen || interval < 0) return LRSC_ERR_ARG;
            const char* read = reads + read_off[r];
            std::string src = source.seedStr.substr((size_t)(source.seedLen - extendKmerSize));
            std::string trg = target.seedStr;
            std::string path(read + source.seedEndPos + 1, (size_t)interval);
            s.rtou = source.isRepeat && !target.isRepeat;
            if(s.rtou) {
                std::swap(src, trg);
                src = revcomp(src);
                trg = revcomp(trg);
                path = revcomp(path);
            }
            s.extendKmerSize = extendKmerSize;
            s.interval = interval;
            s.src = src;
            lrsc_walk_desc d;
            std::memset(&d, 0, sizeof(d));
            d.seq_off = seq.size();
            d.src_len = (uint32_t)src.size(); d.path_len = (uint32_t)path.size(); d.trg_len = (uint32_t)trg.size();
            d.dis = interval; d.init_kmer = (uint32_t)extendKmerSize; d.max_overlap = (uint32_t)extendKmerSize + 2;
            d.min_sa_threshold = (uint32_t)min_SA;
            seq += src; seq += path; seq += trg;
            descs.push_back(d);
            act.push_back(r);
        }
        if(act.empty()) break;

        wres.resize(descs.size());
        uint64_t used = 0;
        if(arena.size() < seq.size() * 2 + 4096) arena.resize(seq.size() * 2 + 4096);
        st = lrsc_extend_walks(ctx, seq.data(), seq.size(), descs.data(), (uint32_t)descs.size(), wres.data(), &arena[0],
                               arena.size(), &used);
        if(st == LRSC_ERR_CAPACITY && used > arena.size()) {
            arena.resize(used);
            st = lrsc_extend_walks(ctx, seq.data(), seq.size(), descs.data(), (uint32_t)descs.size(), wres.data(), &arena[0],
                                   arena.size(), &used);
        }
        if(st != LRSC_OK) return st;

        for(size_t a = 0; a < act.size(); ++a) {
            const uint32_t r = act[a];
            ReadState& s = rs[r];
            lrsc_read_result& R = res[r];
            Piece& source = s.pieces.back();
            const int code = wres[a].code;
            if(s.next == 0) s.firstType = code;
            if(code > 0) {
                // :194-205
                std::string merged(arena.data() + wres[a].out_off, wres[a].out_len);
                if(s.rtou) {
                    merged = revcomp(merged);
                    merged += revcomp(s.src).substr((size_t)s.extendKmerSize);
                }
                merged.erase(0, (size_t)s.extendKmerSize);
                R.corrected_len += (int64_t)merged.length();
                R.seed_dis += s.interval;
                R.fm_num++;
                R.total_walk_num++;
                const Piece target = s.seeds[s.it + (size_t)s.next];
                source.append(merged, target);
                s.it += (size_t)s.next + 1;          // iterTarget += next; then the for-loop's iterTarget++
                s.next = 0;
                continue;
            }
            // failed: try the next target of the inner loop, if any (:85)
            if(s.next + 1 < params->next_target && s.it + (size_t)s.next + 1 < s.seeds.size()) {
                s.next++;
                continue;
            }
            // :110-149
            switch(s.firstType) {
                case -1: R.high_error_num++; break;
                case -2: R.exceed_depth_num++; break;
                case -3: R.exceed_leave_num++; break;
                default: return LRSC_ERR_UNSUPPORTED;       // "Does it really happen?" (:125-127)
            }
            R.total_walk_num++;
            const Piece target = s.seeds[s.it];
            // correctByMSAlignment returns false under --nodp (:211)
            if(params->split)
                s.pieces.push_back(target);
            else {
                const char* read = reads + read_off[r];
                const std::string raw(read + source.seedEndPos + 1, (size_t)(target.seedEndPos - source.seedEndPos));
                source.append(raw, target);
            }
            R.corrected_len += (int64_t)target.seedStr.length();
            s.it += 1;
            s.next = 0;
        }
    }

    // ---- results (process, :49-53) --------------------------------------------------------------------------------
    uint64_t n_pieces = 0, used = 0;
    for(uint32_t r = 0; r < n_reads; ++r) {
        res[r].merge = rs[r].pieces.empty() ? 0 : 1;
        res[r].piece_first = n_pieces;
        res[r].n_pieces = (uint32_t)rs[r].pieces.size();
        for(const Piece& p : rs[r].pieces) {
            if(piece_off && n_pieces < piece_cap) piece_off[n_pieces] = used;
            if(out && used + p.seedStr.size() <= out_cap) std::memcpy(out + used, p.seedStr.data(), p.seedStr.size());
            used += p.seedStr.size();
            ++n_pieces;
        }
    }
    if(piece_off && n_pieces <= piece_cap) piece_off[n_pieces < piece_cap ? n_pieces : piece_cap] = used;
    *n_pieces_out = n_pieces;
    *out_used = used;
    if(!out || !piece_off || used > out_cap || n_pieces + 1 > piece_cap) return LRSC_ERR_CAPACITY;
    return LRSC_OK;
}
