// tools.cpp -- `stride kmerfreq` and `stride kmercheck` on the MI355X back end: the reference's two k-mer diagnostics
// (StriDe/kmerfreq.cpp:60-156, StriDe/kmercheck.cpp:70-225) with their option surface, prompts, output text and exit codes.
#include <getopt.h>

#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/lrsc.h"
#include "BCode.h"
#include "KmerCheckProcess.h"
#include "SequenceProcessFramework.h"

#define PACKAGE_NAME "StriDe"
#define PACKAGE_VERSION "0.0.1"
#define BWT_EXT ".bwt"
#define RBWT_EXT ".rbwt"

namespace stride {

static void lrscOrDie(int st, const char* what)
{
    if(st != LRSC_OK) {
        std::cerr << what << ": " << lrsc_strerror(st) << " (" << lrsc_last_error() << ")\n";
        exit(EXIT_FAILURE);
    }
}

static lrsc_index* openIndex(const std::string& prefix, const std::vector<int>& devices)
{
    std::cerr << "Loading BWT: " << prefix + BWT_EXT << "\n" << "Loading RBWT: " << prefix + RBWT_EXT << "\n";
    lrsc_index* idx = nullptr;
    lrscOrDie(lrsc_index_open((prefix + BWT_EXT).c_str(), (prefix + RBWT_EXT).c_str(), &idx), "lrsc_index_open");
    for(int d : devices) lrscOrDie(lrsc_index_upload(idx, d), "lrsc_index_upload");
    return idx;
}

static std::vector<int> parseDevices(const std::string& s)
{
    std::vector<int> out;
    std::istringstream in(s);
    std::string tok;
    while(std::getline(in, tok, ',')) out.push_back(atoi(tok.c_str()));
    return out;
}

static int64_t freqOf(const lrsc_biinterval& b)           // BiBWTInterval::getFreq (SuffixTools/BWTInterval.h:29,72)
{
    return (b.fwd.lower <= b.fwd.upper ? b.fwd.upper - b.fwd.lower + 1 : 0) + (b.rvc.lower <= b.rvc.upper ? b.rvc.upper - b.rvc.lower + 1 : 0);
}

// ---- kmerfreq ------------------------------------------------------------------------------------------------
static const char* KMERFREQ_USAGE_MESSAGE =
    "Usage: " PACKAGE_NAME " kmerfreq [OPTION]\n"
    "Get sequences kmer frequency\n"
    "  -p, --prefix=PREFIX       Use PREFIX for the names of the index files\n"
    "  -c, --PBcoverage=N        Coverage of PacBio reads (default: 90)\n"
    "  -v, --verbose             Display verbose output\n"
    "      --device=N            HIP device to use (default: 0)\n"
    "      --help                Display this help and exit\n"
    "      --version             Display version\n";

int kmerfreqMain(int argc, char** argv)
{
    std::string prefix;
    int PBcoverage = 90, device = 0;
    enum { OPT_HELP = 1, OPT_VERSION, OPT_DEVICE };
    static const struct option longopts[] = {{"prefix", required_argument, nullptr, 'p'}, {"PBcoverage", required_argument, nullptr, 'c'},
                                             {"verbose", no_argument, nullptr, 'v'},      {"help", no_argument, nullptr, OPT_HELP},
                                             {"version", no_argument, nullptr, OPT_VERSION}, {"device", required_argument, nullptr, OPT_DEVICE},
                                             {nullptr, 0, nullptr, 0}};
    bool die = false;
    optind = 1;
    for(int c; (c = getopt_long(argc, argv, "p:c:v", longopts, nullptr)) != -1;) {
        std::istringstream arg(optarg != nullptr ? optarg : "");
        switch(c) {
            case 'p': arg >> prefix; break;
            case 'c': arg >> PBcoverage; break;
            case 'v': break;
            case OPT_DEVICE: arg >> device; break;
            case OPT_HELP: std::cout << KMERFREQ_USAGE_MESSAGE; exit(EXIT_SUCCESS);
            case OPT_VERSION: std::cout << "kmerfreq Version " PACKAGE_VERSION " (MI355X back end)\n\n"; exit(EXIT_SUCCESS);
            default: die = true; break;
        }
    }
    if(prefix.empty()) { std::cerr << "kmerfreq: no prefix\n"; die = true; }
    if(PBcoverage <= 0) { std::cerr << "kmerfreq: invalid number of coverage: " << PBcoverage << ", must be greater than zero\n"; die = true; }
    if(die) { std::cout << "\n" << KMERFREQ_USAGE_MESSAGE; exit(EXIT_FAILURE); }

    lrsc_index* idx = openIndex(prefix, std::vector<int>(1, device));
    lrsc_params p;
    lrscOrDie(lrsc_params_default(10, PBcoverage, &p), "lrsc_params_default");
    lrsc_ctx* ctx = nullptr;
    lrscOrDie(lrsc_ctx_create(idx, &p, device, &ctx), "lrsc_ctx_create");
    const int end = 100;                                            // KmerThreshold::initialize(-1, 100, coverage, "") (kmerfreq.cpp:75)
    std::vector<float> thr(3 * (end + 2));
    lrscOrDie(lrsc_kmer_thresholds_range(PBcoverage, end, thr.data()), "lrsc_kmer_thresholds_range");
    auto threshold = [&](int mode, int k) -> float {
        if(mode < 0 || mode > 2 || k < 0 || k > end + 1) {           // the reference asserts (KmerThreshold.h:20-21)
            std::cerr << "kmerfreq: mode must be 0..2 and the k-mer size 0.." << end + 1 << "\n";
            exit(EXIT_FAILURE);
        }
        return thr[(size_t)mode * (end + 2) + k];
    };

    std::string query;
    int staticSize, mode;
    std::cerr << "Please enter query sequence, kmer size and mode:\n";
    std::vector<lrsc_biinterval> fixed, growing;
    std::string kmers;
    while(std::cin >> query >> staticSize >> mode) {
        // row `pos`: the fixed-size k-mer at pos, and the prefix of the query of size staticSize + pos (a k-mer grown one base per
        // row, KmerFeature::expand), each with its frequency and the threshold of (mode, size)
        const int queryLen = (int)query.length();
        const int rows = staticSize > 0 ? queryLen - staticSize + 1 : 0;
        if(rows > 0) {
            kmers.clear();
            for(int pos = 0; pos < rows; ++pos) kmers.append(query, (size_t)pos, (size_t)staticSize);
            fixed.resize((size_t)rows);
            lrscOrDie(lrsc_find_kmers(ctx, kmers.data(), (uint32_t)staticSize, (uint64_t)rows, fixed.data()), "lrsc_find_kmers");
            growing.resize((size_t)rows);
            for(int pos = 0; pos < rows; ++pos)
                lrscOrDie(lrsc_find_kmers(ctx, query.data(), (uint32_t)(staticSize + pos), 1, &growing[(size_t)pos]), "lrsc_find_kmers");
        }
        for(int pos = 0; pos < rows; ++pos) {
            const int dynamicSize = staticSize + pos;
            std::cout << pos << '\t' << query.substr((size_t)pos, (size_t)staticSize) << '\t' << freqOf(fixed[(size_t)pos]) << " <-> "
                      << threshold(mode, staticSize) << '\t' << query.substr(0, (size_t)dynamicSize) << '\t' << freqOf(growing[(size_t)pos])
                      << " <-> " << threshold(mode, dynamicSize) << '\n';
        }
        std::cout << "-\n";
    }
    std::cerr << "Exit successfully!\n";
    lrsc_ctx_destroy(ctx);
    lrsc_index_close(idx);
    return 0;
}

// ---- kmercheck -----------------------------------------------------------------------------------------------
static const char* KMERCHECK_USAGE_MESSAGE =
    "Usage: " PACKAGE_NAME " kmercheck [OPTION] ... READSFILE\n"
    "Get sequences kmer frequency\n"
    "  -t, --threads=NUM         Use NUM threads for the computation (default: 1)\n"
    "  -c, --coverage=NUM        Coverage of PacBio reads (default: 90)\n"
    "  -p, --prefix=PREFIX       Use PREFIX for the names of the index files\n"
    "  -o, --directory=PATH      Put results in the directory\n"
    "  -b, --barcode=FILE        Use the barcode to check kmer \n"
    "  -l, --lower=NUM           Kmer size lower bound (default: 15)\n"
    "  -u, --upper=NUM           Kmer size upper bound (default: 35)\n"
    "  -s, --step=NUM            Kmer size step (default: 1)\n"
    "  -v, --verbose             Display verbose output\n"
    "      --devices=LIST        HIP devices to use (default: 0)\n"
    "      --batch=N             Reads per device batch (default: 256)\n"
    "      --help                Display this help and exit\n"
    "      --version             Display version\n";

int kmercheckMain(int argc, char** argv)
{
    int thread = 1, coverage = 90, lower = 15, upper = 35, step = 1;
    std::string prefix, directory, barcode, readsFile;
    std::vector<int> devices(1, 0);
    size_t batch = 256;
    enum { OPT_HELP = 1, OPT_VERSION, OPT_DEVICES, OPT_BATCH };
    static const struct option longopts[] = {
        {"threads", required_argument, nullptr, 't'},   {"coverage", required_argument, nullptr, 'c'}, {"prefix", required_argument, nullptr, 'p'},
        {"directory", required_argument, nullptr, 'o'}, {"barcode", required_argument, nullptr, 'b'},  {"lower", required_argument, nullptr, 'l'},
        {"upper", required_argument, nullptr, 'u'},     {"step", required_argument, nullptr, 's'},     {"verbose", no_argument, nullptr, 'v'},
        {"help", no_argument, nullptr, OPT_HELP},       {"version", no_argument, nullptr, OPT_VERSION}, {"devices", required_argument, nullptr, OPT_DEVICES},
        {"batch", required_argument, nullptr, OPT_BATCH}, {nullptr, 0, nullptr, 0}};
    optind = 1;
    bool die = false;
    for(int c; (c = getopt_long(argc, argv, "t:c:p:o:b:l:u:s:v", longopts, nullptr)) != -1;) {
        std::istringstream arg(optarg != nullptr ? optarg : "");
        switch(c) {
            case 't': arg >> thread; break;
            case 'c': arg >> coverage; break;
            case 'p': arg >> prefix; break;
            case 'o': arg >> directory; break;
            case 'b': arg >> barcode; break;
            case 'l': arg >> lower; break;
            case 'u': arg >> upper; break;
            case 's': arg >> step; break;
            case 'v': break;
            case OPT_DEVICES: devices = parseDevices(arg.str()); break;
            case OPT_BATCH: arg >> batch; break;
            case OPT_HELP: std::cerr << KMERCHECK_USAGE_MESSAGE; exit(EXIT_SUCCESS);
            case OPT_VERSION: std::cerr << "kmercheck Version " PACKAGE_VERSION " (MI355X back end)\n\n"; exit(EXIT_SUCCESS);
            default: die = true; break;
        }
    }
    if(argc - optind < 1) { std::cerr << "kmercheck: missing arguments\n"; die = true; }
    else if(argc - optind > 1) { std::cerr << "kmercheck: too many arguments\n"; die = true; }
    if(thread <= 0) { std::cerr << "kmercheck: invalid number of threads: " << thread << "\n"; die = true; }
    if(coverage <= 0) { std::cerr << "kmercheck: invalid coverage: " << coverage << "\n"; die = true; }
    if(prefix.empty()) { std::cerr << "kmercheck: no prefix\n"; die = true; }
    if(directory.empty()) { std::cerr << "kmercheck: no directory\n"; die = true; }
    else {
        directory += "/";
        if(system(("mkdir -p " + directory).c_str()) != 0) { std::cerr << "kmercheck: something wrong in directory: " << directory << "\n"; die = true; }
    }
    if(barcode.empty()) { std::cerr << "kmercheck: no barcode\n"; die = true; }
    if(!(lower >= 9 && upper >= lower)) { std::cerr << "kmercheck" << "invalid range of kmer size:" << lower << " - " << upper << '\n'; die = true; }
    if(step <= 0) { std::cerr << "kmercheck" << "invalid step size: " << step << '\n'; die = true; }
    if(devices.empty() || batch == 0) { std::cerr << "kmercheck: invalid --devices / --batch\n"; die = true; }
    if(die) { std::cerr << "\n" << KMERCHECK_USAGE_MESSAGE; exit(EXIT_FAILURE); }
    readsFile = argv[optind++];

    KmerCheckParameters kcParams;
    kcParams.index = openIndex(prefix, devices);
    BCode::load(barcode);
    kcParams.devices = devices;
    kcParams.directory = directory;
    kcParams.coverage = coverage;
    kcParams.lower = lower; kcParams.upper = upper; kcParams.step = step;
    std::cerr << "Using kmer size : " << lower << " - " << upper << " (" << step << ")\n";
    SequenceProcessFramework::processSequences<SequenceWorkItem, KmerCheckResult, KmerCheckProcess, KmerCheckPostProcess, KmerCheckParameters>(
        thread, readsFile, kcParams, batch);
    lrsc_index_close(kcParams.index);
    return 0;
}

} // namespace stride
