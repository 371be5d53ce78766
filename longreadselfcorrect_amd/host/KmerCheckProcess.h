// KmerCheckProcess.h -- `stride kmercheck`: frequency distributions of correct and wrong k-mers inside the barcode blocks of
// the reads (the reference's PacBio/KmerCheckProcess.{h,cpp}), batched over the C ABI: all k-mers of one size of one batch
// are one lrsc_find_kmers call.
#pragma once
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../../include/lrsc.h"
#include "BCode.h"
#include "KmerDistribution.h"
#include "SequenceWorkItem.h"

namespace stride {

typedef std::map<int, KmerDistribution> kdMap;

struct KmerCheckParameters {      // reference .h:21-29 (indices -> lrsc_index + devices)
    lrsc_index* index = nullptr;
    std::vector<int> devices{0};
    std::string directory;
    int coverage = 90, lower = 15, upper = 35, step = 1;
};

struct KmerCheckResult {
    kdMap crtKdMap, errKdMap;
    std::string readid;
};

class KmerCheckProcess {
public:
    explicit KmerCheckProcess(const KmerCheckParameters& params, size_t worker = 0);
    ~KmerCheckProcess();
    static size_t workers(const KmerCheckParameters& params, int) { return params.devices.size(); }
    std::vector<KmerCheckResult> process_batch(const std::vector<SequenceWorkItem>& items);
private:
    const KmerCheckParameters m_params;
    lrsc_ctx* m_ctx = nullptr;
};

class KmerCheckPostProcess {
public:
    explicit KmerCheckPostProcess(const KmerCheckParameters& params);
    ~KmerCheckPostProcess();          // compare() per k-mer size -> total.box, value.box (appended)
    void process(const SequenceWorkItem& workItem, const KmerCheckResult& result);
private:
    const KmerCheckParameters m_params;
    kdMap m_crtKdMap, m_errKdMap;
    std::ofstream m_total, m_value;
};

} // namespace stride
