// KmerCheckProcess.cpp -- see KmerCheckProcess.h (reference behaviour: PacBio/KmerCheckProcess.cpp:11-64).
#include "KmerCheckProcess.h"

#include <cstdlib>
#include <iostream>

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

KmerCheckProcess::KmerCheckProcess(const KmerCheckParameters& params, size_t worker) : m_params(params)
{
    lrsc_params p;
    orDie(lrsc_params_default(10, m_params.coverage, &p), "lrsc_params_default");
    orDie(lrsc_ctx_create(m_params.index, &p, m_params.devices[worker % m_params.devices.size()], &m_ctx), "lrsc_ctx_create");
}

KmerCheckProcess::~KmerCheckProcess() { lrsc_ctx_destroy(m_ctx); }

std::vector<KmerCheckResult> KmerCheckProcess::process_batch(const std::vector<SequenceWorkItem>& items)
{
    std::vector<KmerCheckResult> results(items.size());
    for(size_t i = 0; i < items.size(); ++i) results[i].readid = items[i].read.id;
    struct Probe { uint32_t item; const BCode* block; int pos; };
    std::vector<Probe> probes;
    std::string kmers;
    std::vector<lrsc_biinterval> iv;
    for(int k = m_params.lower; k <= m_params.upper; k += m_params.step) {
        probes.clear();
        kmers.clear();
        for(size_t i = 0; i < items.size(); ++i) {
            const std::string& seq = items[i].read.seq;
            // find(), not operator[]: several workers read the log concurrently and must not insert into it
            const auto entry = BCode::Log().find(items[i].read.id);
            if(entry == BCode::Log().end()) continue;
            for(const BCode& block : entry->second)
                for(int pos = block.getStart(); pos <= block.getEnd() - k; ++pos) {          // scan(), reference :25-27
                    if(pos < 0 || (size_t)pos + (size_t)k > seq.size()) {
                        std::cerr << "kmercheck: block " << block.getStart() << "-" << block.getEnd() << " of " << items[i].read.id
                                  << " reaches beyond the read\n";                           // assert(!curr.isFake()) there
                        exit(EXIT_FAILURE);
                    }
                    probes.push_back(Probe{(uint32_t)i, &block, pos});
                    kmers.append(seq, (size_t)pos, (size_t)k);
                }
        }
        if(probes.empty()) continue;
        iv.resize(probes.size());
        orDie(lrsc_find_kmers(m_ctx, kmers.data(), (uint32_t)k, probes.size(), iv.data()), "lrsc_find_kmers");
        for(size_t j = 0; j < probes.size(); ++j) {
            const lrsc_biinterval& b = iv[j];
            const int64_t f = (b.fwd.lower <= b.fwd.upper ? b.fwd.upper - b.fwd.lower + 1 : 0) +
                              (b.rvc.lower <= b.rvc.upper ? b.rvc.upper - b.rvc.lower + 1 : 0);      // BiBWTInterval::getFreq
            const Probe& q = probes[j];
            if(f == 0) {
                std::cerr << "kmercheck: a k-mer of " << items[q.item].read.id << " is absent from the index (is it the index of these reads?)\n";
                exit(EXIT_FAILURE);                                                          // assert(curr.getFreq() != 0) there
            }
            if(f == 1) continue;
            const bool find = BCode::validate(q.pos, k, *q.block, items[q.item].read.seq);
            (find ? results[q.item].crtKdMap : results[q.item].errKdMap)[k].add((int)f);
        }
    }
    return results;
}

KmerCheckPostProcess::KmerCheckPostProcess(const KmerCheckParameters& params) : m_params(params)
{
    m_total.open((m_params.directory + "total.box").c_str(), std::ios_base::app);
    m_value.open((m_params.directory + "value.box").c_str(), std::ios_base::app);
    if(!m_total || !m_value) {
        std::cerr << "Error: could not open " << m_params.directory << "total.box / value.box for write\n";
        exit(EXIT_FAILURE);
    }
}

KmerCheckPostProcess::~KmerCheckPostProcess()
{
    for(int k = m_params.lower; k <= m_params.upper; k += m_params.step)
        compare(m_total, m_value, m_params.coverage, k, m_crtKdMap[k], m_errKdMap[k]);
}

void KmerCheckPostProcess::process(const SequenceWorkItem&, const KmerCheckResult& result)
{
    for(int k = m_params.lower; k <= m_params.upper; k += m_params.step) {
        kdMap::const_iterator crt = result.crtKdMap.find(k), err = result.errKdMap.find(k);
        if(crt != result.crtKdMap.end()) m_crtKdMap[k] += crt->second;
        if(err != result.errKdMap.end()) m_errKdMap[k] += err->second;
    }
}

} // namespace stride
