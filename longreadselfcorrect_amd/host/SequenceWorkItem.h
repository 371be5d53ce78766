// SequenceWorkItem.h -- work items of the dispatch framework, same fields as the reference's
// Concurrency/SequenceWorkItem.h:15-21 and Util/Util.h:51-129 (SeqRecord), plus a FASTA/FASTQ
// reader with Util/SeqReader.cpp:26-135's rules.
#pragma once
#include <cstddef>
#include <fstream>
#include <string>

namespace stride {

struct SeqRecord {
    std::string id;
    std::string seq;     // upper-cased, validated ACGT
    std::string qual;
};

struct SequenceWorkItem {
    size_t idx = 0;
    SeqRecord read;
};

// FASTA (multi-line allowed) / FASTQ; id = header up to the first space or tab; the sequence is
// upper-cased and a non-ACGT base is a fatal error (message + exit, as the reference does).
class SeqReader {
public:
    explicit SeqReader(const std::string& filename);
    bool get(SeqRecord& sr);
private:
    std::ifstream m_in;
};

template <class INPUT>
class WorkItemGenerator {       // Concurrency/SequenceWorkItem.h:31-93
public:
    explicit WorkItemGenerator(SeqReader* pReader) : m_pReader(pReader), m_numConsumedTotal(0) {}
    bool generate(SequenceWorkItem& out)
    {
        SeqRecord read;
        if(!m_pReader->get(read)) return false;
        out.idx = m_numConsumedTotal;
        out.read = read;
        m_numConsumedTotal += 1;
        return true;
    }
    size_t getNumConsumed() const { return m_numConsumedTotal; }
private:
    SeqReader* m_pReader;
    size_t m_numConsumedTotal;
};

} // namespace stride
