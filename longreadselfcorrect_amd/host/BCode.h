// BCode.h -- alignment barcodes of raw reads, the truth source of `--onlyseed -b FILE` and `stride kmercheck`
// (interface of the reference's PacBio/BCode.h:10-41; the checks of BCode.cpp:81-153 restated with index arithmetic).
//
// A barcode file has one block per line:  qname qstart qend tname tstart tend code rvc sup  (BCode.cpp:39-45).  `code`
// holds two hex digits per read base of the block: digit 2*i describes an insertion at base qstart+i, digit 2*i+1 a
// deletion after it (a bit set of the deleted reference bases, A=1 T=2 C=4 G=8).  A k-mer of the read is "correct" when
// it carries no error the index could not have reproduced: see BCode::validate.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace stride {

class BCode {
public:
    BCode(int s, int e, const std::string& c, bool r) : start(s), end(e), code(c), rvc(r) {}

    typedef std::vector<BCode> BCodeVector;
    static std::map<std::string, BCodeVector>& Log();
    // Reads the file once (a second call is fatal, BCode.cpp:29-33).  Plain or .gz.
    static void load(const std::string& barcodefile);
    // Is seq[pos, pos+ksize) free of uncorrectable errors according to `block`?  Throws std::out_of_range where the
    // reference would (a digit outside 0-9a-f, a k-mer that starts beyond the block's code).
    static bool validate(int pos, int ksize, const BCode& block, const std::string& seq);

    int getStart() const { return start; }
    int getEnd() const { return end; }
    const std::string& getCode() const { return code; }
    bool getRvc() const { return rvc; }

private:
    int start, end;
    std::string code;
    bool rvc;
};

} // namespace stride
