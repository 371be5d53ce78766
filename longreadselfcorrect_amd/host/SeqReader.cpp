// SeqReader.cpp -- block FASTA/FASTQ parser (record rules: SequenceWorkItem.h; reference behaviour: Util/SeqReader.cpp:26-135).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "SequenceWorkItem.h"

namespace stride {

static void openFailure(const std::string& filename)
{
    std::cerr << "Error: could not open " << filename << " for read\n";       // assertFileOpen (Util/Util.cpp:312-339)
    exit(EXIT_FAILURE);
}

SeqReader::SeqReader(const std::string& filename)
{
    const bool gz = filename.size() > 3 && filename.compare(filename.size() - 3, 3, ".gz") == 0;
    if(gz) {
        gzFile f = gzopen(filename.c_str(), "rb");
        if(!f) openFailure(filename);
        gzbuffer(f, 1 << 20);
        size_t used = 0;
        m_owned.resize(1 << 24);
        while(true) {
            if(m_owned.size() - used < (1u << 22)) m_owned.resize(m_owned.size() * 2);
            const int got = gzread(f, m_owned.data() + used, (unsigned)std::min<size_t>(m_owned.size() - used, 1u << 30));
            if(got < 0) { std::cerr << "Error: could not inflate " << filename << "\n"; exit(EXIT_FAILURE); }
            if(got == 0) break;
            used += (size_t)got;
        }
        gzclose(f);
        m_data = m_owned.data();
        m_size = used;
        return;
    }
    const int fd = open(filename.c_str(), O_RDONLY);
    if(fd < 0) openFailure(filename);
    struct stat st;
    if(fstat(fd, &st) != 0) { close(fd); openFailure(filename); }
    m_size = (size_t)st.st_size;
    if(m_size > 0) {
        void* p = mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if(p == MAP_FAILED) { close(fd); openFailure(filename); }
        madvise(p, m_size, MADV_SEQUENTIAL);
        m_data = static_cast<const char*>(p);
        m_mapped = true;
    }
    close(fd);
}

SeqReader::~SeqReader()
{
    if(m_mapped) munmap(const_cast<char*>(m_data), m_size);
}

// One line without its '\n'.  false = nothing left.
bool SeqReader::nextLine(Span& line, bool& hit_eof)
{
    hit_eof = false;
    if(m_pos >= m_size) { hit_eof = true; line = Span{m_data + m_size, 0}; return false; }
    const char* b = m_data + m_pos;
    const char* nl = static_cast<const char*>(memchr(b, '\n', m_size - m_pos));
    if(nl) { line = Span{b, (size_t)(nl - b)}; m_pos = (size_t)(nl - m_data) + 1; }
    else { line = Span{b, m_size - m_pos}; m_pos = m_size; hit_eof = true; }
    return true;
}

// Cuts the next record: header line, sequence lines (m_lines), quality line.  false = no further valid record.
bool SeqReader::nextRecord(Span& header, Span& qual, bool& fastq)
{
    m_lines.clear();
    qual = Span{nullptr, 0};
    bool found = false, eof = false;
    while(m_good) {
        Span l;
        const bool any = nextLine(l, eof);
        if(eof) m_good = false;                 // a read ran into the end of the input (also when it still delivered characters)
        if(!any || l.n == 0) continue;
        if(l.p[0] == '>') { fastq = false; header = l; found = true; break; }
        if(l.p[0] == '@') { fastq = true; header = l; found = true; break; }
    }
    if(!found) return false;
    if(!fastq) {
        // every following line up to one that starts a new record; a line that ended at the end of the input does not count
        while(m_good) {
            if(m_pos >= m_size) { m_good = false; break; }            // looking at the end of the input
            const char c = m_data[m_pos];
            if(c == '>' || c == '@') break;
            Span l;
            nextLine(l, eof);
            if(eof) { m_good = false; break; }
            if(l.n > 0) m_lines.push_back(l);
        }
        return !m_lines.empty();
    }
    Span s, sep;
    nextLine(s, eof);   if(eof) m_good = false;
    nextLine(sep, eof); if(eof) m_good = false;
    nextLine(qual, eof);
    const bool at_eof = eof;
    if(eof) m_good = false;
    if(s.n == 0 || qual.n == 0)
        std::cerr << "Warning, read " << std::string(header.p, header.n) << " has no sequence or quality values\n";
    m_lines.push_back(s);
    return !at_eof;
}

static inline size_t idEnd(const char* h, size_t n)
{
    for(size_t i = 0; i < n; ++i) if(h[i] == ' ' || h[i] == '\t') return i;
    return n;
}

bool SeqReader::get(SeqRecord& sr)
{
    Span header, qual;
    bool fastq = false;
    if(!nextRecord(header, qual, fastq)) return false;
    sr.id.assign(header.p + 1, idEnd(header.p + 1, header.n - 1));
    size_t total = 0;
    for(const Span& l : m_lines) total += l.n;
    sr.seq.resize(total);
    char* d = total ? &sr.seq[0] : nullptr;
    unsigned bad = 0;
    for(const Span& l : m_lines)
        for(size_t i = 0; i < l.n; ++i) {
            const char c = l.p[i];
            const char u = (c >= 'a' && c <= 'z') ? (char)(c - 32) : c;          // the reference upper-cases with toupper
            *d++ = u;
            bad |= (unsigned)!(u == 'A' || u == 'C' || u == 'G' || u == 'T');
        }
    if(bad) {
        std::cerr << "Error: read " << sr.id << " contains non-ACGT characters.\n";
        std::cerr << "Please run sga preprocess on the data first.\n";
        exit(EXIT_FAILURE);
    }
    sr.qual.assign(qual.p ? qual.p : "", qual.n);
    return true;
}

} // namespace stride
