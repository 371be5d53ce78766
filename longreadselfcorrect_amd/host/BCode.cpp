// BCode.cpp -- barcode blocks and the k-mer correctness check (behaviour of the reference's PacBio/BCode.cpp:27-153).
//
// The reference phrases the check with Python-style string slices (`in[pos::step]`, BCode.cpp:50-58).  Here the same
// walks are index loops over the code string; every verdict, and every std::out_of_range the reference can raise, is kept.
// Where the reference would read a std::string out of range (undefined there) the k-mer is reported as not correct.
#include "BCode.h"

#include <zlib.h>

#include <cstdlib>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace stride {

std::map<std::string, BCode::BCodeVector>& BCode::Log()
{
    static std::map<std::string, BCodeVector> log;
    return log;
}

void BCode::load(const std::string& barcode)
{
    if(!Log().empty()) {
        std::cerr << "2nd loading is forbidden.\n";
        exit(EXIT_FAILURE);
    }
    std::cerr << "Loading BARCODE: " << barcode << '\n';
    gzFile f = gzopen(barcode.c_str(), "rb");              // transparent for plain files (createReader, Util/Util.cpp:276-309)
    if(!f) {
        std::cerr << "Error: could not open " << barcode << " for read\n";
        exit(EXIT_FAILURE);
    }
    std::string text;
    char buf[1 << 16];
    for(int got; (got = gzread(f, buf, sizeof(buf))) > 0;) text.append(buf, (size_t)got);
    gzclose(f);
    std::istringstream in(text);
    std::string qname, tname, code, rvc, sup;
    int qstart, qend, tstart, tend;
    // nine white-space separated fields per block.  (After a final newline the reference's loop runs once more on the
    // failed stream and files one block of indeterminate content under the empty read name; that entry is not made here.)
    while(in >> qname) {
        if(!(in >> qstart >> qend >> tname >> tstart >> tend >> code >> rvc >> sup)) {
            std::cerr << "Error: malformed barcode block for " << qname << " in " << barcode << "\n";
            exit(EXIT_FAILURE);
        }
        Log()[qname].push_back(BCode(qstart, qend, code, rvc == "True"));
    }
}

namespace {

inline int hexDigit(char c)                       // char_int.at(c)
{
    if(c >= '0' && c <= '9') return c - '0';
    if(c >= 'a' && c <= 'f') return c - 'a' + 10;
    throw std::out_of_range("BCode: not a hex digit");
}

inline int baseBit(char c)                        // base_hex.at(c)
{
    switch(c) {
        case 'a': case 'A': return 1;
        case 't': case 'T': return 2;
        case 'c': case 'C': return 4;
        case 'g': case 'G': return 8;
        default: throw std::out_of_range("BCode: not a base");
    }
}

inline int pyIndex(int pos, int len)              // getPys: a negative position counts from the end
{
    if(pos < 0) pos += len;
    if(pos < 0) throw std::logic_error("BCode: position before the start");      // the reference asserts
    return pos;
}

// A strided view: element t is data[t * step] for t < n.
struct Strided {
    const char* data;
    int n, step;
    char at(int t) const { return data[(long)t * step]; }
    bool has(int t) const { return t >= 0 && t < n; }
};

// for x in view[pos::dir]: f(x) until f returns false
template <class F>
void walk(const Strided& v, int pos, int dir, F f)
{
    for(int t = pyIndex(pos, v.n); t >= 0 && t < v.n; t += dir)
        if(!f(v.at(t))) return;
}

} // namespace

bool BCode::validate(int pos, int ksize, const BCode& block, const std::string& seq)
{
    const std::string& code = block.getCode();
    const int base = block.getStart();
    const long first = (long)(pos - base) * 2;
    if(first < 0 || (size_t)first > code.size() || (size_t)pos > seq.size()) throw std::out_of_range("BCode: k-mer outside the block");
    // the k-mer's slice of the code: two digits per base, without the deletion digit of the last base
    const int infoLen = (int)std::min<size_t>((size_t)(2 * ksize - 1), code.size() - (size_t)first);
    const char* info = code.data() + first;
    const Strided insDigits{info, (infoLen + 1) / 2, 2};          // info[0::2]
    const Strided delDigits{info + 1, infoLen / 2, 2};            // info[1::2]
    const Strided infoAll{info, infoLen, 1};
    const Strided blockIns{code.data(), ((int)code.size() + 1) / 2, 2};     // code[0::2]
    const bool rvc = block.getRvc();
    const int sign = rvc ? -1 : 1;
    const int bit = rvc ? 0 : 1;
    const int pole = rvc ? pos : pos + ksize;      // the k-mer end the alignment grows from: its last base + 1, or its first base
    auto kmerAt = [&](int p) -> char {             // kmer[getPys(p, ksize)]
        const int q = pyIndex(p, ksize);
        return (size_t)(pos + q) < seq.size() && q < ksize ? seq[pos + q] : '\0';
    };

    // ---- insertions: they may only form one run at the pole, and the inserted bases must repeat what follows
    int upper = 0;
    for(int t = 0; t < insDigits.n; ++t) upper += hexDigit(insDigits.at(t));
    if(upper > 0) {
        int igap = 0, n = 0;
        walk(infoAll, -bit, -sign * 2, [&](char c) {
            const int v = hexDigit(c);
            if(!((igap == 0 && (v == 0 || v == 1)) || (igap > 0 && v == 1))) return false;
            n += 1;
            igap += v;
            return true;
        });
        if(upper != igap) return false;
        if(igap > 0) {
            int ioffset = 0;       // inserted bases that continue beyond the k-mer
            walk(blockIns, pole - base + bit - 1, sign, [&](char c) {
                if(hexDigit(c) != 1) return false;
                ioffset += 1;
                return true;
            });
            if((n - igap) > 0 && ioffset > 0) return false;
            for(int i = 0; i < n; ++i) {
                const int shift = sign * (1 - bit + ioffset + i) - sign * (n - igap);
                const int inBlock = pole - base + shift, inSeq = pole + shift;
                if(!blockIns.has(inBlock) || inSeq < 0 || (size_t)inSeq >= seq.size()) return false;
                if(!(blockIns.at(inBlock) == '0' && kmerAt(-sign * (n + bit - 1 - i)) == seq[inSeq])) return false;
            }
        }
    }

    // ---- deletions: at most one, next to the pole, and made of the base(s) around it (a homopolymer contraction)
    int lower = 0;
    for(int t = 0; t < delDigits.n; ++t) lower += hexDigit(delDigits.at(t));
    if(lower > 0) {
        int dgap = 0, m = 0, hex = 0;
        walk(infoAll, -sign * (1 + bit), -sign * 2, [&](char c) {
            const int v = hexDigit(c);
            if(dgap != 0) return false;
            hex |= baseBit(kmerAt(-sign * (bit + m)));
            m += 1;
            dgap += v;
            return true;
        });
        if(lower != dgap) return false;
        if(dgap > 0) {
            const int bits = (dgap & 1) + ((dgap >> 1) & 1) + ((dgap >> 2) & 1) + ((dgap >> 3) & 1);
            if(!(dgap == hex || (m == 1 && (dgap & hex) > 0 && bits == 2))) return false;
        }
    }
    return true;
}

} // namespace stride
