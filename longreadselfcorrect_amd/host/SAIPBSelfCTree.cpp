// SAIPBSelfCTree.cpp -- see SAIPBSelfCTree.h.  Reference behaviour: PacBio/SAIPBSelfCTree.cpp (line numbers in the comments).
#include "SAIPBSelfCTree.h"

#include <cstdlib>
#include <cstring>
#include <iostream>

#include "GlobalAlign.h"

namespace stride {

namespace {

inline char complementOf(char b) { return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : b; }
inline std::string revcomp(const std::string& s)
{
    std::string o(s.size(), 'N');
    for(size_t i = 0; i < s.size(); ++i) o[s.size() - 1 - i] = complementOf(s[i]);
    return o;
}
inline bool validIv(const lrsc_interval& v) { return v.lower <= v.upper; }
inline int64_t sizeOf(const lrsc_interval& v) { return v.upper - v.lower + 1; }
inline int codeOf(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

void orDie(int st, const char* what)
{
    if(st != LRSC_OK) {
        std::cerr << what << ": " << lrsc_strerror(st) << " (" << lrsc_last_error() << ")\n";
        // called from a worker thread while the reader, the other workers and the post-processor may be inside HIP calls: leave
        // without running static destructors / the HIP runtime's teardown under them (a plain exit() can hang there)
        std::cerr.flush(); std::cout.flush();
        std::_Exit(EXIT_FAILURE);
    }
}

} // namespace

// ---- FMAccess over the C ABI -----------------------------------------------------------------------------------
LrscFMAccess::LrscFMAccess(lrsc_ctx* ctx, const lrsc_index* index) : m_ctx(ctx)
{
    orDie(lrsc_index_info_get(index, &m_info), "lrsc_index_info_get");
}
void LrscFMAccess::findBiIntervals(const std::vector<std::string>& kmers, std::vector<lrsc_biinterval>& out)
{
    out.resize(kmers.size());
    if(kmers.empty()) return;
    const size_t k = kmers[0].size();
    std::string flat;
    flat.reserve(k * kmers.size());
    for(const std::string& s : kmers) flat += s;
    orDie(lrsc_find_kmers(m_ctx, flat.data(), (uint32_t)k, kmers.size(), out.data()), "lrsc_find_kmers");
}
void LrscFMAccess::occ(const std::vector<lrsc_rank_query>& q, std::vector<uint64_t>& out)
{
    out.resize(q.size());
    if(!q.empty()) orDie(lrsc_rank(m_ctx, q.data(), q.size(), out.data()), "lrsc_rank");
}
uint64_t LrscFMAccess::pc(int strand, char base) const
{
    static const char kOrder[] = "$ACGT";
    for(int r = 0; r < 5; ++r) if(kOrder[r] == base) return m_info.pred_count[strand][r];
    return 0;
}
void LrscFMAccess::lfWalks(int strand, const std::vector<uint64_t>& rows, uint32_t max_steps, std::vector<std::string>& out)
{
    out.assign(rows.size(), std::string());
    if(rows.empty() || max_steps == 0) return;
    std::vector<uint8_t> strands(rows.size(), (uint8_t)strand);
    std::vector<uint32_t> steps(rows.size(), max_steps), lens(rows.size(), 0);
    std::vector<uint64_t> offs(rows.size());
    for(size_t i = 0; i < rows.size(); ++i) offs[i] = (uint64_t)i * max_steps;
    std::string buf((size_t)rows.size() * max_steps + 1, '\0');
    orDie(lrsc_lf_walk(m_ctx, rows.data(), strands.data(), steps.data(), offs.data(), rows.size(), &buf[0], buf.size(), lens.data()), "lrsc_lf_walk");
    for(size_t i = 0; i < rows.size(); ++i) out[i].assign(buf.data() + offs[i], lens[i]);
}

// ---- the tree ------------------------------------------------------------------------------------------------
SAIPBSelfCorrectTree::SAIPBSelfCorrectTree(FMAccess& fm, const std::string& rawSeq, size_t srcmaxLength, size_t min_SA_threshold, int maxLeavesAllowed)
    : m_fm(fm), m_rawSeq(rawSeq), m_maxLength(srcmaxLength), m_minSAThreshold(min_SA_threshold), m_maxLeavesAllowed((size_t)maxLeavesAllowed)
{
}

bool SAIPBSelfCorrectTree::pack(const std::string& s, size_t from, size_t len, uint64_t& key)
{
    if(len > 31 || from + len > s.size()) return false;
    uint64_t k = 1;                                   // leading 1: k-mers of different lengths never share a key
    for(size_t i = 0; i < len; ++i) {
        const int c = codeOf(s[from + i]);
        if(c < 0) return false;
        k = (k << 2) | (uint64_t)c;
    }
    key = k;
    return true;
}

void SAIPBSelfCorrectTree::insertKmer(uint64_t key, long long pos, size_t maxLength)
{
    auto it = m_hash.find(key);
    if(it == m_hash.end()) {
        Feature f;
        f.freq.assign(maxLength / 35 + 1, 0);          // KmerFeatures(pos, maxIntervalSize = maxLength, intervalSize = 35)
        it = m_hash.emplace(key, std::move(f)).first;
    }
    std::vector<long long>& fr = it->second.freq;
    long long index = pos / 35;
    if(index < 0) index = 0; else if(index > (long long)fr.size() - 1) index = (long long)fr.size() - 1;
    fr[(size_t)index]++;
}

long long SAIPBSelfCorrectTree::sumOfFreq(const Feature& f, long long pos) const
{
    const long long index = pos / 35;
    if(index < 0 || index >= (long long)f.freq.size()) return 0;        // the reference reads out of range here
    long long s = f.freq[(size_t)index];
    if(index > 0) s += f.freq[(size_t)index - 1];
    if(index < (long long)f.freq.size() - 1) s += f.freq[(size_t)index + 1];
    return s;
}

// The k-mers along the reads that contain the seed, one LF-walk per row of the seed's interval (at most 30 rows; :713,:727-757).
// strand RBWT: the walk spells the read onwards from the seed; strand BWT: it spells the other strand's read backwards.
void SAIPBSelfCorrectTree::collectAlong(const std::string& seedStr, const lrsc_interval& iv, int strand, size_t smallKmerSize, size_t maxLength,
                                        int expectedLength)
{
    if(!validIv(iv)) return;
    const size_t seedLen = seedStr.length();
    std::vector<uint64_t> rows;
    for(int64_t r = iv.lower; r <= iv.upper && r - iv.lower < 30; ++r) rows.push_back((uint64_t)r);
    std::vector<std::string> walked;
    m_fm.lfWalks(strand, rows, maxLength > seedLen ? (uint32_t)(maxLength - seedLen) : 0u, walked);
    const std::string tail = seedStr.substr(seedLen - smallKmerSize);
    auto positionOf = [&](size_t currentLength) -> long long {       // insertKmerToHash's position (:891-914), size_t arithmetic as written
        return expectedLength < 0 ? (long long)(currentLength - seedLen) : (long long)((size_t)expectedLength - currentLength + smallKmerSize);
    };
    for(const std::string& w : walked) {
        uint64_t key = 0;
        if(strand == LRSC_RBWT) {
            // window = last smallKmerSize characters of seed + walked prefix
            std::string text = tail + w;
            for(size_t t = 0; t + smallKmerSize <= text.size(); ++t)
                if(pack(text, t, smallKmerSize, key)) insertKmer(key, positionOf(seedLen + t), maxLength);
        } else {
            // the k-mer grows to the left: each walked character is prepended to revcomp(tail)
            std::string text = revcomp(tail);
            if(pack(text, 0, smallKmerSize, key)) insertKmer(key, positionOf(seedLen), maxLength);
            for(size_t t = 0; t < w.size(); ++t) {
                text = w[t] + text.substr(0, smallKmerSize - 1);
                if(pack(text, 0, smallKmerSize, key)) insertKmer(key, positionOf(seedLen + t + 1), maxLength);
            }
        }
    }
}

size_t SAIPBSelfCorrectTree::addHashBySingleSeed(const std::string& seedStr, size_t largeKmerSize, size_t smallKmerSize, size_t maxLength,
                                                 bool skipRepeat, int expectedLength)
{
    m_hashKmerSize = smallKmerSize;
    std::vector<lrsc_biinterval> bi;
    m_fm.findBiIntervals(std::vector<std::string>(1, seedStr.substr(seedStr.length() - largeKmerSize)), bi);
    size_t kmerFreq = 0;
    kmerFreq += validIv(bi[0].fwd) ? (size_t)sizeOf(bi[0].fwd) : 0;
    kmerFreq += validIv(bi[0].fwd) ? (size_t)sizeOf(bi[0].rvc) : 0;        // sic (:720): guarded by the forward interval
    if(skipRepeat && kmerFreq > 128) return kmerFreq;
    collectAlong(seedStr, bi[0].fwd, LRSC_RBWT, smallKmerSize, maxLength, expectedLength);
    collectAlong(seedStr, bi[0].rvc, LRSC_BWT, smallKmerSize, maxLength, expectedLength);
    return kmerFreq;
}

size_t SAIPBSelfCorrectTree::hashkmerfreqs(const std::string& fwdkmer, size_t kmerposition) const
{
    uint64_t k1 = 0, k2 = 0;
    size_t f = 0;
    if(fwdkmer.size() == m_hashKmerSize && pack(fwdkmer, 0, fwdkmer.size(), k1)) {
        const auto it = m_hash.find(k1);
        if(it != m_hash.end()) f += (size_t)sumOfFreq(it->second, (long long)kmerposition);
    }
    const std::string rc = revcomp(fwdkmer);
    if(rc.size() == m_hashKmerSize && pack(rc, 0, rc.size(), k2)) {
        const auto it = m_hash.find(k2);
        if(it != m_hash.end()) f += (size_t)sumOfFreq(it->second, (long long)kmerposition);
    }
    return f;
}

void SAIPBSelfCorrectTree::refine(size_t kmerSize)                       // refineSAInterval (:1178-1187): one launch for all leaves
{
    std::vector<std::string> kmers;
    for(const Leaf& l : m_leaves) kmers.push_back(l.seq.size() >= kmerSize ? l.seq.substr(l.seq.size() - kmerSize) : l.seq);
    std::vector<lrsc_biinterval> bi;
    // k-mers of one launch must be equally long: leaves shorter than kmerSize (never with the reference's parameters) go alone
    bool same = true;
    for(const std::string& k : kmers) same = same && k.size() == kmers[0].size();
    if(same) {
        m_fm.findBiIntervals(kmers, bi);
        for(size_t i = 0; i < m_leaves.size(); ++i) m_leaves[i].iv = bi[i];
    } else
        for(size_t i = 0; i < m_leaves.size(); ++i) {
            m_fm.findBiIntervals(std::vector<std::string>(1, kmers[i]), bi);
            m_leaves[i].iv = bi[0];
        }
}

// getFMIndexRightExtensions (:1213-1253) of every leaf: one batch of Occ queries (2 per valid strand and base)
void SAIPBSelfCorrectTree::extensionsOfLeaves(std::vector<std::vector<Ext>>& out, size_t cutoff)
{
    static const char kBases[4] = {'A', 'C', 'G', 'T'};
    std::vector<lrsc_rank_query> q;
    for(const Leaf& l : m_leaves)
        for(int b = 0; b < 4; ++b) {
            if(validIv(l.iv.fwd)) {
                lrsc_rank_query a{}; a.base = (uint8_t)kBases[b]; a.strand = LRSC_RBWT; a.idx = l.iv.fwd.lower - 1; q.push_back(a);
                a.idx = l.iv.fwd.upper; q.push_back(a);
            }
            if(validIv(l.iv.rvc)) {
                lrsc_rank_query a{}; a.base = (uint8_t)complementOf(kBases[b]); a.strand = LRSC_BWT; a.idx = l.iv.rvc.lower - 1; q.push_back(a);
                a.idx = l.iv.rvc.upper; q.push_back(a);
            }
        }
    std::vector<uint64_t> occ;
    m_fm.occ(q, occ);
    out.assign(m_leaves.size(), std::vector<Ext>());
    size_t at = 0;
    for(size_t li = 0; li < m_leaves.size(); ++li) {
        const Leaf& l = m_leaves[li];
        for(int b = 0; b < 4; ++b) {
            lrsc_biinterval p = l.iv;
            if(validIv(l.iv.fwd)) {
                const uint64_t pcb = m_fm.pc(LRSC_RBWT, kBases[b]);
                p.fwd.lower = (int64_t)(pcb + occ[at]); p.fwd.upper = (int64_t)(pcb + occ[at + 1]) - 1; at += 2;
            }
            if(validIv(l.iv.rvc)) {
                const uint64_t pcb = m_fm.pc(LRSC_BWT, complementOf(kBases[b]));
                p.rvc.lower = (int64_t)(pcb + occ[at]); p.rvc.upper = (int64_t)(pcb + occ[at + 1]) - 1; at += 2;
            }
            size_t bcount = 0;
            if(validIv(p.fwd)) bcount += (size_t)sizeOf(p.fwd);
            if(validIv(p.rvc)) bcount += (size_t)sizeOf(p.rvc);
            if(bcount >= cutoff) out[li].push_back(Ext{kBases[b], p});
        }
    }
}

bool SAIPBSelfCorrectTree::isExtensionValid(const std::string& fwdkmer, double currAvgFreq, size_t& kmerFreq, size_t bcount)     // :1131-1175
{
    uint64_t k1 = 0, k2 = 0;
    Feature* f1 = nullptr;
    if(fwdkmer.size() == m_hashKmerSize && pack(fwdkmer, 0, fwdkmer.size(), k1)) {
        auto it = m_hash.find(k1);
        if(it != m_hash.end()) f1 = &it->second;
    }
    if(f1 && m_leaves.size() > 8 && currAvgFreq < f1->maxAvgFreq) return false;          // bubble removal once the frontier is wide
    if(f1 && currAvgFreq > f1->maxAvgFreq) f1->maxAvgFreq = currAvgFreq;
    const Feature* f2 = nullptr;
    const std::string rc = revcomp(fwdkmer);
    if(rc.size() == m_hashKmerSize && pack(rc, 0, rc.size(), k2)) {
        auto it = m_hash.find(k2);
        if(it != m_hash.end()) f2 = &it->second;
    }
    const long long here = m_currentLength - m_seedLength;
    kmerFreq = f1 ? (size_t)sumOfFreq(*f1, here) : 0;
    kmerFreq += f2 ? (size_t)sumOfFreq(*f2, here) : 0;
    return kmerFreq >= m_minSAThreshold || (bcount >= 7 && kmerFreq >= 1);
}

void SAIPBSelfCorrectTree::attemptToExtend(const std::vector<std::vector<Ext>>& exts, std::vector<Leaf>& next, size_t hashKmerSize)   // :977-1111
{
    for(size_t li = 0; li < m_leaves.size(); ++li) {
        const Leaf& leaf = m_leaves[li];
        const std::string stem = leaf.seq.size() >= hashKmerSize - 1 ? leaf.seq.substr(leaf.seq.size() - (hashKmerSize - 1)) : leaf.seq;
        const double currAvgFreq = (double)leaf.kmerCount / (m_currentLength + 1000000);
        // one extension: the leaf itself grows; several: one child per accepted extension (a child inherits the count)
        for(const Ext& e : exts[li]) {
            size_t kmerfreqs = 0;
            const size_t bcount = (size_t)(sizeOf(e.iv.fwd) + sizeOf(e.iv.rvc));          // raw sizes (:1022,:1061)
            if(!isExtensionValid(stem + e.base, currAvgFreq, kmerfreqs, bcount)) continue;
            Leaf n;
            n.seq = leaf.seq + e.base;
            n.iv = e.iv;
            n.kmerCount = leaf.kmerCount + kmerfreqs;
            next.push_back(std::move(n));
        }
    }
}

int SAIPBSelfCorrectTree::mergeTwoSeedsUsingHash(const std::string& src, const std::string& dest, std::string& mergedseq, size_t hashKmerSize,
                                                 size_t maxLeaves, size_t minLength, size_t maxLength, size_t expectedLength)
{
    // initializeSearchTree / initializeTerminalIntervals (:52-88): one launch for the two k-mers
    std::vector<lrsc_biinterval> bi;
    m_fm.findBiIntervals({src.substr(src.length() - hashKmerSize), dest.substr(0, hashKmerSize)}, bi);
    m_leaves.assign(1, Leaf{src, 0, bi[0]});
    m_terminal = bi[1];
    m_seedLength = m_currentLength = (int)src.length();

    struct Result { std::string thread; size_t coverage; };
    std::vector<Result> results;
    while(!m_leaves.empty() && m_leaves.size() <= maxLeaves && (size_t)m_currentLength <= maxLength) {
        refine(hashKmerSize - 1);
        std::vector<std::vector<Ext>> exts;
        extensionsOfLeaves(exts, 2);
        std::vector<Leaf> next;
        attemptToExtend(exts, next, hashKmerSize);
        if(next.empty()) {                                    // once more with the local-frequency threshold lowered by one (:108-113)
            m_minSAThreshold--;
            attemptToExtend(exts, next, hashKmerSize);
            m_minSAThreshold++;
        }
        if(!next.empty()) m_currentLength++;
        m_leaves.swap(next);
        if((size_t)m_currentLength >= minLength)              // isTerminated (:1258-1294): terminated leaves stay in the frontier
            for(const Leaf& l : m_leaves) {
                const bool f = validIv(l.iv.fwd) && l.iv.fwd.lower >= m_terminal.fwd.lower && l.iv.fwd.upper <= m_terminal.fwd.upper;
                const bool r = validIv(l.iv.rvc) && l.iv.rvc.lower >= m_terminal.rvc.lower && l.iv.rvc.upper <= m_terminal.rvc.upper;
                if(f || r) results.push_back(Result{l.seq, l.kmerCount});
            }
    }

    if(!results.empty()) {
        double maxKmerCoverage = 0, maxMatchPercent = -100;
        int minLengthDiff = 100000;
        for(const Result& res : results) {
            const std::string tmpseq = dest.length() > hashKmerSize ? res.thread + dest.substr(hashKmerSize) : res.thread;
            const int currLengthDiff = std::abs((int)tmpseq.length() - (int)expectedLength);
            const double avgCov = (double)res.coverage / (tmpseq.length() + 1000000);
            const bool isLengthDiffBetter = currLengthDiff < minLengthDiff && std::abs(currLengthDiff - minLengthDiff) > 3;
            const bool isKmerCoverageBetter = std::abs(currLengthDiff - minLengthDiff) <= 3 && maxKmerCoverage < avgCov;
            if(results.size() > 1) {
                const double matchPercent = (double)globalAlignPacBio(m_rawSeq, tmpseq).matches / m_rawSeq.length();
                if(maxMatchPercent < matchPercent) { maxMatchPercent = matchPercent; mergedseq = tmpseq; }
            } else if(isLengthDiffBetter || isKmerCoverageBetter) {
                minLengthDiff = currLengthDiff;
                maxKmerCoverage = avgCov;
                mergedseq = tmpseq;
            }
        }
        return 1;
    }
    const int half = (int)(expectedLength - m_seedLength) / 2 + m_seedLength;
    if(m_leaves.empty() && m_currentLength >= half) return -1;
    if((size_t)m_currentLength > maxLength) return -2;
    if(m_leaves.size() > maxLeaves) return -3;
    if(m_leaves.empty() && m_currentLength < half) return -4;
    return -5;
}

} // namespace stride
