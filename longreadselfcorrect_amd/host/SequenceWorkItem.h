// SequenceWorkItem.h -- work items of the dispatch framework, same fields as the reference's
// Concurrency/SequenceWorkItem.h:15-21 and Util/Util.h:51-129 (SeqRecord), plus the FASTA/FASTQ reader.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

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

// FASTA (multi-line allowed) / FASTQ, plain or gzip (.gz, Util/Util.cpp:276-309).  The whole input is one memory block
// (mmap for plain files, inflated once for .gz) and records are cut out of it with memchr: no stream, no per-line string.
// Record rules = the reference's (Util/SeqReader.cpp:26-135), including its corner cases:
//   * lines before the first '>' / '@' header are skipped; id = header up to the first space or tab;
//   * FASTA: the sequence is every following non-empty line up to a line that starts with '>' or '@'; a last line
//     without a newline is not part of it; a record without sequence ends the input;
//   * FASTQ: sequence, separator and quality line; a record whose quality line hits the end of the input without a
//     newline is dropped; a warning if sequence or quality is empty;
//   * the sequence is upper-cased, and a base other than ACGT is a fatal error (message + exit).
class SeqReader {
public:
    explicit SeqReader(const std::string& filename);
    ~SeqReader();
    SeqReader(const SeqReader&) = delete;
    SeqReader& operator=(const SeqReader&) = delete;
    bool get(SeqRecord& sr);
private:
    struct Span { const char* p; size_t n; };
    bool nextLine(Span& line, bool& hit_eof);      // hit_eof: the line ended at the end of the input, not at a newline
    bool nextRecord(Span& header, Span& qual, bool& fastq);
    const char* m_data = nullptr;
    size_t m_size = 0, m_pos = 0;
    bool m_good = true;                            // the stream's good(): cleared once a read ran into the end of the input
    bool m_mapped = false;
    std::vector<char> m_owned;                     // inflated .gz contents
    std::vector<Span> m_lines;                     // sequence lines of the record being cut
};

template <class INPUT>
class WorkItemGenerator {       // Concurrency/SequenceWorkItem.h:31-93
public:
    explicit WorkItemGenerator(SeqReader* pReader) : m_pReader(pReader), m_numConsumedTotal(0) {}
    bool generate(SequenceWorkItem& out)
    {
        if(!m_pReader->get(out.read)) return false;
        out.idx = m_numConsumedTotal;
        m_numConsumedTotal += 1;
        return true;
    }
    size_t getNumConsumed() const { return m_numConsumedTotal; }
private:
    SeqReader* m_pReader;
    size_t m_numConsumedTotal;
};

} // namespace stride
