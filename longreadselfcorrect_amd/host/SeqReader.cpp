// SeqReader.cpp -- FASTA/FASTQ reader with the reference's record rules (Util/SeqReader.cpp:26-135).
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <iostream>

#include "SequenceWorkItem.h"

namespace stride {

SeqReader::SeqReader(const std::string& filename) : m_in(filename.c_str())
{
    if(!m_in) {
        // createReader / assertFileOpen (Util/Util.cpp:276-339)
        std::cerr << "Error: could not open " << filename << " for read\n";
        exit(EXIT_FAILURE);
    }
}

bool SeqReader::get(SeqRecord& sr)
{
    enum { RT_UNKNOWN, RT_FASTA, RT_FASTQ } rt = RT_UNKNOWN;
    std::string header;
    while(m_in.good()) {
        std::getline(m_in, header);
        if(header.empty()) continue;
        if(header[0] == '>') { rt = RT_FASTA; break; }
        if(header[0] == '@') { rt = RT_FASTQ; break; }
    }
    if(rt == RT_UNKNOWN) return false;

    bool validRecord = false;
    std::string seq, qual;
    if(rt == RT_FASTA) {
        std::string temp;
        while(m_in.good() && m_in.peek() != '>' && m_in.peek() != '@') {
            std::getline(m_in, temp);
            if(m_in.good() && temp.size() > 0) seq.append(temp);
        }
        validRecord = seq.size() > 0;
    } else {
        std::string temp;
        std::getline(m_in, seq);
        std::getline(m_in, temp);
        std::getline(m_in, qual);
        if(seq.empty() || qual.empty()) std::cerr << "Warning, read " << header << " has no sequence or quality values\n";
        validRecord = !m_in.eof();
    }
    if(validRecord) {
        const size_t endPos = std::min(header.find_first_of(' '), header.find_first_of('\t'));
        sr.id = endPos != std::string::npos ? header.substr(1, endPos - 1) : header.substr(1);
        std::transform(seq.begin(), seq.end(), seq.begin(), ::toupper);
        if(seq.find_first_not_of("ACGT") != std::string::npos) {
            std::cerr << "Error: read " << sr.id << " contains non-ACGT characters.\n";
            std::cerr << "Please run sga preprocess on the data first.\n";
            exit(EXIT_FAILURE);
        }
        sr.seq = seq;
        sr.qual = qual;
    }
    return validRecord;
}

} // namespace stride
