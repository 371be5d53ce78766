// tests/host_tools/saipb_driver.cpp -- TEST INFRASTRUCTURE: runs the host SAIPBSelfCorrectTree (longreadselfcorrect_amd/host/
// SAIPBSelfCTree.cpp) the way the reference's commented-out call site does (PacBioHybridCorrectionProcess.cpp:1083-1122).
//   saipb_driver oracle BWT RBWT     FM access = the CPU oracle's RLBWT (liblrsc_oracle.so), for `-m "not gpu"` tests
//   saipb_driver device BWT RBWT     FM access = the C ABI (liblrsc_hip.so: lrsc_find_kmers, lrsc_rank, lrsc_lf_walk), `-m gpu`
//   saipb_driver align               stdin: "s1 s2" per line -> "matches score columns" (host/GlobalAlign.h)
// stdin: "source between target dis" per line  ->  "code merged" per line ("-" for an empty string)
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/lrsc.h"
#include "../../longreadselfcorrect_amd/host/GlobalAlign.h"
#include "../../longreadselfcorrect_amd/host/SAIPBSelfCTree.h"

using namespace stride;

#ifdef SAIPB_WITH_ORACLE
extern "C" {
void* orc_bwt_load(const char* path);
uint64_t orc_bwt_pc(void* h, char b);
void orc_bwt_occ_batch(void* h, const char* b, const int64_t* idx, uint64_t n, uint64_t* out);
void orc_bwt_char_batch(void* h, const uint64_t* idx, uint64_t n, char* out);
void orc_find_intervals(void* h, const char* kmers, uint32_t k, uint64_t n, int64_t* out);
}
class OracleFMAccess : public FMAccess {
public:
    OracleFMAccess(void* bwt, void* rbwt) { h[LRSC_BWT] = bwt; h[LRSC_RBWT] = rbwt; }
    void findBiIntervals(const std::vector<std::string>& kmers, std::vector<lrsc_biinterval>& out) override
    {
        out.resize(kmers.size());
        for(size_t i = 0; i < kmers.size(); ++i) {
            std::string rev(kmers[i].rbegin(), kmers[i].rend()), rc = rev;
            for(char& c : rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
            int64_t v[2];
            orc_find_intervals(h[LRSC_RBWT], rev.data(), (uint32_t)rev.size(), 1, v);
            out[i].fwd.lower = v[0]; out[i].fwd.upper = v[1];
            orc_find_intervals(h[LRSC_BWT], rc.data(), (uint32_t)rc.size(), 1, v);
            out[i].rvc.lower = v[0]; out[i].rvc.upper = v[1];
        }
    }
    void occ(const std::vector<lrsc_rank_query>& q, std::vector<uint64_t>& out) override
    {
        out.resize(q.size());
        for(size_t i = 0; i < q.size(); ++i) {
            const char b = (char)q[i].base;
            orc_bwt_occ_batch(h[q[i].strand], &b, &q[i].idx, 1, &out[i]);
        }
    }
    uint64_t pc(int strand, char base) const override { return orc_bwt_pc(h[strand], base); }
    void lfWalks(int strand, const std::vector<uint64_t>& rows, uint32_t max_steps, std::vector<std::string>& out) override
    {
        out.assign(rows.size(), std::string());
        for(size_t i = 0; i < rows.size(); ++i) {
            uint64_t row = rows[i];
            for(uint32_t s = 0; s < max_steps; ++s) {
                char c;
                orc_bwt_char_batch(h[strand], &row, 1, &c);
                if(c == '$') break;
                out[i].push_back(c);
                const int64_t before = (int64_t)row - 1;
                uint64_t o;
                orc_bwt_occ_batch(h[strand], &c, &before, 1, &o);
                row = orc_bwt_pc(h[strand], c) + o;
            }
        }
    }
private:
    void* h[2];
};
#endif

static int runPairs(FMAccess& fm)
{
    std::string source, between, target;
    int dis;
    while(std::cin >> source >> between >> target >> dis) {
        if(between == "-") between.clear();
        const double maxRatio = 1.1, minRatio = 0.9;
        const int minOffSet = 30;
        const size_t extendKmerSize = 15, srcKmerSize = 17;
        SAIPBSelfCorrectTree tree(fm, between, 2);
        std::string srcStr = source.substr(source.length() - srcKmerSize);
        const size_t srcMaxLength = (size_t)(maxRatio * (dis + minOffSet) + srcStr.length() + extendKmerSize);
        tree.addHashBySingleSeed(source.substr(source.length() - srcKmerSize * 2, srcKmerSize), srcKmerSize, extendKmerSize, srcMaxLength, true);
        tree.addHashBySingleSeed(source.substr(source.length() - srcKmerSize * 3, srcKmerSize), srcKmerSize, extendKmerSize, srcMaxLength, true);
        tree.addHashBySingleSeed(source.substr(source.length() - (size_t)(srcKmerSize * 1.5), srcKmerSize), srcKmerSize, extendKmerSize, srcMaxLength, true);
        std::string rvcTarget(target.rbegin(), target.rend());
        for(char& c : rvcTarget) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        const int targetMaxLength = (int)(maxRatio * (dis + minOffSet) + rvcTarget.length() + srcKmerSize);
        size_t expectedLength = (size_t)dis + rvcTarget.length();
        tree.addHashBySingleSeed(rvcTarget, srcKmerSize, extendKmerSize, (size_t)targetMaxLength, true, (int)expectedLength);
        int srcMinLength = (int)(minRatio * (dis - minOffSet) + srcStr.length() + extendKmerSize);
        if(srcMinLength < 0) srcMinLength = 0;
        expectedLength = srcStr.length() + (size_t)dis + target.length();
        std::string pbseq;
        const int rc = tree.mergeTwoSeedsUsingHash(srcStr, target, pbseq, extendKmerSize, 32, (size_t)srcMinLength, srcMaxLength, expectedLength);
        const std::string merged = pbseq.empty() ? std::string() : source + pbseq.substr(srcKmerSize);
        std::cout << rc << ' ' << (merged.empty() ? "-" : merged) << '\n';
    }
    return 0;
}

int main(int argc, char** argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    if(mode == "align") {
        std::string a, b;
        while(std::cin >> a >> b) {
            const GlobalAlignment g = globalAlignPacBio(a, b);
            std::cout << g.matches << ' ' << g.score << ' ' << g.columns << '\n';
        }
        return 0;
    }
#ifdef SAIPB_WITH_ORACLE
    if(mode == "oracle" && argc > 3) {
        void* bwt = orc_bwt_load(argv[2]);
        void* rbwt = orc_bwt_load(argv[3]);
        if(!bwt || !rbwt) { std::cerr << "cannot load the index\n"; return 1; }
        OracleFMAccess fm(bwt, rbwt);
        return runPairs(fm);
    }
#else
    if(mode == "device" && argc > 3) {
        lrsc_index* idx = nullptr;
        lrsc_ctx* ctx = nullptr;
        lrsc_params p;
        if(lrsc_index_open(argv[2], argv[3], &idx) != LRSC_OK || lrsc_index_upload(idx, 0) != LRSC_OK || lrsc_params_default(5, 90, &p) != LRSC_OK ||
           lrsc_ctx_create(idx, &p, 0, &ctx) != LRSC_OK) {
            std::cerr << "device set-up failed: " << lrsc_last_error() << "\n";
            return 1;
        }
        LrscFMAccess fm(ctx, idx);
        const int rc = runPairs(fm);
        lrsc_ctx_destroy(ctx);
        lrsc_index_close(idx);
        return rc;
    }
#endif
    std::cerr << "usage: saipb_driver oracle|device BWT RBWT | align\n";
    return 2;
}
