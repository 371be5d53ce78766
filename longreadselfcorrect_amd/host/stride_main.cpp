// stride_main.cpp -- `stride pbcorrect` (a.k.a. PacBioSelfCorrection) and `stride index` on the
// MI355X back end.  Option surface, defaults, validation messages and exit codes follow the reference's
// StriDe/PacBioSelfCorrection.cpp:32-140,262-434 and StriDe/StriDe.cpp:62-126; extra flags: --devices, --batch.
#include <getopt.h>

#include <array>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/lrsc.h"
#include "BCode.h"
#include "PacBioSelfCorrectionProcess.h"
#include "SequenceProcessFramework.h"

#define PACKAGE_NAME "StriDe"
#define PACKAGE_VERSION "0.0.1"
#define PACKAGE_BUGREPORT "ythuang@cs.ccu.edu.tw"
#define SUBPROGRAM "PacBioSelfCorrection"
#define BWT_EXT ".bwt"
#define RBWT_EXT ".rbwt"

using namespace stride;
namespace stride {
int kmerfreqMain(int argc, char** argv);       // tools.cpp
int kmercheckMain(int argc, char** argv);
}

static const char* CORRECT_VERSION_MESSAGE = SUBPROGRAM " Version " PACKAGE_VERSION " (MI355X back end)\n";

static const char* CORRECT_USAGE_MESSAGE =
    "Usage: " PACKAGE_NAME " " SUBPROGRAM " [OPTION] ... READSFILE\n"
    "Correct PacBio reads via FM-index walk\n"
    "\n"
    "      -t, --thread=NUM                 Use NUM threads for the computation (default: 1)\n"
    "      -p, --prefix=PREFIX              Use PREFIX for the names of the index files\n"
    "      -o, --output=DIR                 Output results in the directory\n"
    "      -b, --barcode=FILE               Barcode of raw reads\n"
    "\nPacBio correction parameters:\n"
    "      -c, --PBcoverage=N               Coverage of PacBio reads (default: 90)\n"
    "      -e, --error-rate=N               The error rate of PacBio reads.(default:0.15)\n"
    "      -k, --kmer-size=N                The start kmer length (default: 19 (PacBioS).)\n"
    "      -n, --next-target                The number of next FMWalk target seed(default: 1)\n"
    "      -l, --max-leaves=N               Number of maximum leaves in the search tree. (default: 32)\n"
    "      -i, --idmer-length=N             The length of the kmer to identify similar reads.(default: 9)\n"
    "      -s, --min-kmer-size=N            The minimum length of the kmer to use. (default: 13.)\n"
    "      -g, --genome=(5/10/100)[m]       Genome size of the species (default: 10m)\n"
    "      -m, --mode=(0/1/2)               Mode in seed-searching (default: 1)\n"
    "      -v, --verbose                    Display verbose output\n"
    "      --help                           Display this help and exit\n"
    "      --version                        Display version and exit\n"
    "      --debugseed                      Output seeds file for each reads (default: false)\n"
    "      --debugextend                    Show extension information (default: false)\n"
    "      --onlyseed                       Only search seeds file for each reads (default: false)\n"
    "      --nodp                           Don't use dp (default: false)\n"
    "      --split                          Split the uncorrected reads (default: false)\n"
    "      --devices=LIST                   HIP devices to use, e.g. 0,1,2,3 (default: 0)\n"
    "      --workers-per-device=N           batches in flight per device (default: 2; two are ~13 % faster than one)\n"
    "      --batch=N                        Reads per device batch (default: 100000)\n"
    "\nReport bugs to " PACKAGE_BUGREPORT "\n\n";

namespace opt {
static int thread = 1;
static std::string prefix, directory, barcode, readsFile;
static size_t PBcoverage = 90;
static double ErrorRate = 0.15;
static int startKmerLen = 19, nextTarget = 1, maxLeaves = 32, idmerLen = 9, minKmerLen = 13, genome = 10, mode = 1, verbose = 0;
static bool Split = false, DebugExtend = false, DebugSeed = false, OnlySeed = false, NoDp = false, Manual = false, Adjust = false;
static std::array<int, 3> offset = {{0, 0, 0}};
static std::vector<int> devices(1, 0);
static int workersPerDevice = 2;
static size_t batch = 100000;
}

static const char* shortopts = "t:p:o:b:c:e:k:u:r:n:l:i:s:g:m:v";
enum { OPT_HELP = 1, OPT_VERSION, OPT_SPLIT, OPT_FIRST, OPT_DEBUGEXTEND, OPT_DEBUGSEED, OPT_ONLYSEED, OPT_NODP, OPT_DEVICES, OPT_BATCH, OPT_WORKERS };
static const struct option longopts[] = {
    {"thread", required_argument, nullptr, 't'},       {"prefix", required_argument, nullptr, 'p'},
    {"output", required_argument, nullptr, 'o'},       {"barcode", required_argument, nullptr, 'b'},
    {"PBcoverage", required_argument, nullptr, 'c'},   {"error-rate", required_argument, nullptr, 'e'},
    {"kmer-size", required_argument, nullptr, 'k'},    {"unique-offset", required_argument, nullptr, 'u'},
    {"repeat-offset", required_argument, nullptr, 'r'}, {"next-target", required_argument, nullptr, 'n'},
    {"max-leaves", required_argument, nullptr, 'l'},   {"idmer-length", required_argument, nullptr, 'i'},
    {"min-kmer-size", required_argument, nullptr, 's'}, {"genome", required_argument, nullptr, 'g'},
    {"mode", required_argument, nullptr, 'm'},         {"verbose", no_argument, nullptr, 'v'},
    {"help", no_argument, nullptr, OPT_HELP},          {"version", no_argument, nullptr, OPT_VERSION},
    {"split", no_argument, nullptr, OPT_SPLIT},        {"debugextend", no_argument, nullptr, OPT_DEBUGEXTEND},
    {"debugseed", no_argument, nullptr, OPT_DEBUGSEED}, {"onlyseed", no_argument, nullptr, OPT_ONLYSEED},
    {"nodp", no_argument, nullptr, OPT_NODP},          {"devices", required_argument, nullptr, OPT_DEVICES},
    {"batch", required_argument, nullptr, OPT_BATCH},  {"workers-per-device", required_argument, nullptr, OPT_WORKERS},
    {nullptr, 0, nullptr, 0}};

static void lrscOrDie(int st, const char* what)
{
    if(st != LRSC_OK) {
        std::cerr << what << ": " << lrsc_strerror(st) << " (" << lrsc_last_error() << ")\n";
        exit(EXIT_FAILURE);
    }
}

static void parsePacBioSelfCorrectionOptions(int argc, char** argv)
{
    optind = 1;
    bool die = false;
    for(int c; (c = getopt_long(argc, argv, shortopts, longopts, nullptr)) != -1;) {
        std::istringstream arg(optarg != nullptr ? optarg : "");
        switch(c) {
            case 't': arg >> opt::thread; break;
            case 'p': arg >> opt::prefix; break;
            case 'o': arg >> opt::directory; break;
            case 'b': arg >> opt::barcode; break;
            case 'c': arg >> opt::PBcoverage; break;
            case 'e': arg >> opt::ErrorRate; break;
            case 'k': arg >> opt::startKmerLen; opt::Adjust = true; break;
            case 'u': arg >> opt::offset[1]; opt::Adjust = true; break;
            case 'r': arg >> opt::offset[2]; opt::Adjust = true; break;
            case 'n': arg >> opt::nextTarget; break;
            case 'l': arg >> opt::maxLeaves; break;
            case 'i': arg >> opt::idmerLen; break;
            case 's': arg >> opt::minKmerLen; break;
            case 'g': arg >> opt::genome; break;
            case 'm': arg >> opt::mode; opt::Manual = true; break;
            case 'v': opt::verbose++; break;
            case OPT_HELP: std::cerr << CORRECT_USAGE_MESSAGE; exit(EXIT_SUCCESS);
            case OPT_VERSION: std::cerr << CORRECT_VERSION_MESSAGE; exit(EXIT_SUCCESS);
            case OPT_SPLIT: opt::Split = true; break;
            case OPT_DEBUGEXTEND: opt::DebugExtend = true; break;
            case OPT_DEBUGSEED: opt::DebugSeed = true; break;
            case OPT_NODP: opt::NoDp = true; break;
            case OPT_ONLYSEED: opt::DebugSeed = true; opt::OnlySeed = true; break;
            case OPT_DEVICES: {
                opt::devices.clear();
                std::string tok;
                while(std::getline(arg, tok, ',')) opt::devices.push_back(atoi(tok.c_str()));
                break;
            }
            case OPT_BATCH: arg >> opt::batch; break;
            case OPT_WORKERS: arg >> opt::workersPerDevice; break;
            default: die = true; break;
        }
    }
    if(argc - optind < 1) { std::cerr << SUBPROGRAM ": missing arguments\n"; die = true; }
    else if(argc - optind > 1) { std::cerr << SUBPROGRAM ": too many arguments\n"; die = true; }
    if(opt::thread <= 0) { std::cerr << SUBPROGRAM ": invalid number of threads: " << opt::thread << "\n"; die = true; }
    if(opt::prefix.empty()) { std::cerr << SUBPROGRAM << ": no prefix\n"; die = true; }
    if(opt::directory.empty()) { std::cerr << SUBPROGRAM << ": no directory\n"; die = true; }
    else {
        opt::directory += "/";
        // reference :346-363: with --debugseed the per-read files go to extend/ and seed/ (+ seed/error/)
        std::vector<std::string> subdir(1, std::string(""));
        if(opt::DebugSeed) { subdir.clear(); subdir.push_back("extend/"); subdir.push_back("seed/error/"); }
        for(const std::string& sub : subdir)
            if(system(("mkdir -p " + opt::directory + sub).c_str()) != 0) {
                std::cerr << SUBPROGRAM << ": something wrong making directory: " << opt::directory << "\n";
                die = true;
            }
    }
    if(opt::PBcoverage <= 0) { std::cerr << SUBPROGRAM ": invalid number of coverage: " << opt::PBcoverage << ", must be greater than zero\n"; die = true; }
    if(opt::ErrorRate < 0 || opt::ErrorRate > 1) { std::cerr << SUBPROGRAM ":invalid error rate: " << opt::ErrorRate << ", must be 0 ~ 1\n"; die = true; }
    if(opt::startKmerLen <= 0) { std::cerr << SUBPROGRAM ": invalid start kmer length: " << opt::startKmerLen << ", must be greater than zero\n"; die = true; }
    if(opt::nextTarget <= 0) { std::cerr << SUBPROGRAM ": invalid number of next target: " << opt::nextTarget << ", must be greater than zero\n"; die = true; }
    if(opt::maxLeaves <= 0) { std::cerr << SUBPROGRAM ":invalid number of max leaves:" << opt::maxLeaves << ", must be greater than zero\n"; die = true; }
    if(opt::idmerLen <= 0) { std::cerr << SUBPROGRAM ":invalid kmer length to identify similar reads" << opt::idmerLen << ", must be greater than zero\n"; die = true; }
    if(opt::minKmerLen <= 0) { std::cerr << SUBPROGRAM ":invalid min kmer length:" << opt::minKmerLen << ", must be greater than zero\n"; die = true; }
    if(opt::genome != 5 && opt::genome != 10 && opt::genome != 100) { std::cerr << SUBPROGRAM ": invalid genome size: " << opt::genome << ", must be (5/10/100)[m]\n"; die = true; }
    if(opt::mode < 0 || opt::mode > 2) { std::cerr << SUBPROGRAM ": invalid mode: " << opt::mode << ", must be (0/1/2)\n"; die = true; }
    if(opt::OnlySeed && opt::barcode.empty()) { std::cerr << SUBPROGRAM ": no barcode\n"; die = true; }
    if(opt::devices.empty() || opt::batch == 0 || opt::workersPerDevice < 1 || opt::workersPerDevice > 8) { std::cerr << SUBPROGRAM ": invalid --devices / --batch / --workers-per-device\n"; die = true; }
    if(die) { std::cerr << "\n" << CORRECT_USAGE_MESSAGE; exit(EXIT_FAILURE); }
    opt::readsFile = argv[optind++];
}

static int PacBioSelfCorrectionMain(int argc, char** argv)
{
    parsePacBioSelfCorrectionOptions(argc, argv);
    // --debugextend is accepted and inert: in the reference its only consumer (the debugExtInfo FASTA dump) is commented out
    // (PacBioSelfCorrectionProcess.cpp:87-98).
    PacBioSelfCorrectionParameters ecParams;
    std::cerr << "Loading BWT: " << opt::prefix + BWT_EXT << "\n" << "Loading RBWT: " << opt::prefix + RBWT_EXT << "\n";
    lrsc_index* idx = nullptr;
    lrscOrDie(lrsc_index_open((opt::prefix + BWT_EXT).c_str(), (opt::prefix + RBWT_EXT).c_str(), &idx), "lrsc_index_open");
    for(int d : opt::devices) lrscOrDie(lrsc_index_upload(idx, d), "lrsc_index_upload");

    lrsc_params p;
    lrscOrDie(lrsc_params_default(opt::genome, (int)opt::PBcoverage, &p), "lrsc_params_default");
    if(opt::Adjust) {                       // -k/-u/-r given: no automatic derivation (reference :195-200)
        p.start_kmer_len = opt::startKmerLen;
        p.offset[0] = opt::offset[0]; p.offset[1] = opt::offset[1]; p.offset[2] = opt::offset[2];
    }
    p.error_rate = opt::ErrorRate; p.next_target = opt::nextTarget; p.max_leaves = opt::maxLeaves;
    p.idmer_len = opt::idmerLen; p.min_kmer_len = opt::minKmerLen; p.mode = opt::mode; p.manual = opt::Manual ? 1 : 0;
    p.split = opt::Split ? 1 : 0; p.no_dp = opt::NoDp ? 1 : 0;

    // one framework worker (host thread + lrsc_ctx) per entry: every device is listed workersPerDevice times, device-major order
    std::vector<int> workers;
    for(int w = 0; w < opt::workersPerDevice; ++w) for(int d : opt::devices) workers.push_back(d);
    ecParams.index = idx; ecParams.devices = workers; ecParams.directory = opt::directory; ecParams.p = p;
    ecParams.threads = opt::thread;
    ecParams.DebugExtend = opt::DebugExtend; ecParams.DebugSeed = opt::DebugSeed; ecParams.OnlySeed = opt::OnlySeed;
    if(opt::OnlySeed) BCode::load(opt::barcode);                   // reference :191

    {   // <out>/threshold-table (KmerThreshold.cpp:31-41,65-72)
        float thr[3 * 52];
        lrscOrDie(lrsc_kmer_thresholds(p.pb_coverage, thr), "lrsc_kmer_thresholds");
        std::ofstream t((opt::directory + "threshold-table").c_str());
        t << "Coverage : " << p.pb_coverage << "\n" << "size\tlowcov\tunique\trepeat\n";
        for(int k = 15; k <= 50; ++k) t << k << "\t" << thr[k] << "\t" << thr[52 + k] << "\t" << thr[104 + k] << "\n";
    }

    std::cerr << "\nCorrecting PacBio reads for " << opt::readsFile << " using--\n"
              << "number of threads:\t" << opt::thread << "\n"
              << "PB reads coverage:\t" << opt::PBcoverage << "\n"
              << "num of next Targets:\t" << opt::nextTarget << "\n"
              << "large kmer size:\t" << p.start_kmer_len << "\n"
              << "small kmer size:\t" << opt::minKmerLen << "\n"
              << "max leaves:\t" << opt::maxLeaves << "\n"
              << "max depth:\t1.2~0.8* (length between two seeds +- 20)" << "\n"
              << "devices:\t" << opt::devices.size() << " x " << opt::workersPerDevice << " workers\n";

    SequenceProcessFramework::processSequences<SequenceWorkItem, PacBioSelfCorrectionResult, PacBioSelfCorrectionProcess,
                                               PacBioSelfCorrectionPostProcess, PacBioSelfCorrectionParameters>(
        opt::thread, opt::readsFile, ecParams, opt::batch);
    lrsc_index_close(idx);
    return 0;
}

// `stride index -p PREFIX READS`: multi-string BWT of the reads and of the reversed reads on the GPU
static int indexMain(int argc, char** argv)
{
    std::string prefix, reads;
    int device = 0;
    for(int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if((a == "-p" || a == "--prefix") && i + 1 < argc) prefix = argv[++i];
        else if(a.rfind("--prefix=", 0) == 0) prefix = a.substr(9);
        else if((a == "-t" || a == "-a" || a == "-d") && i + 1 < argc) ++i;     // accepted for compatibility, unused
        else if(a.rfind("--device=", 0) == 0) device = atoi(a.c_str() + 9);
        else if(a == "--help") { std::cerr << "Usage: " PACKAGE_NAME " index [-p PREFIX] [--device=N] READSFILE\n"; return 0; }
        else if(!a.empty() && a[0] != '-') reads = a;
    }
    if(reads.empty()) { std::cerr << "index: missing arguments\n"; return EXIT_FAILURE; }
    if(prefix.empty()) { prefix = reads.substr(reads.find_last_of('/') + 1); prefix = prefix.substr(0, prefix.find_last_of('.')); }
    SeqReader reader(reads);
    SeqRecord r;
    std::string bases;
    std::vector<uint64_t> off(1, 0);
    while(reader.get(r)) { bases += r.seq; off.push_back(bases.size()); }
    const uint32_t n = (uint32_t)(off.size() - 1);
    std::cout << "Building index for " << reads << " on the GPU\n";
    for(int rev = 0; rev < 2; ++rev) {
        uint8_t* units = nullptr; uint64_t nu = 0;
        lrscOrDie(lrsc_build_bwt(bases.data(), off.data(), n, rev, device, &units, &nu), "lrsc_build_bwt");
        lrscOrDie(lrsc_write_bwt_file((prefix + (rev ? RBWT_EXT : BWT_EXT)).c_str(), units, nu, n, bases.size() + n), "lrsc_write_bwt_file");
        lrsc_buffer_free(units);
        // .sai / .rsai: lexicographic rank -> read index (SampledSuffixArray::buildLexicoIndex + writeLexicoIndex,
        // SuffixTools/SampledSuffixArray.cpp:158-190,248-258; text format of SAWriter.cpp:32-54).  The reference LF-walks each
        // read back to its '$' row; that row's rank among the '$' rows is the rank of the read among all reads compared as
        // strings ('$' < A < C < G < T, so a proper prefix sorts first) with equal reads in input order (sentinel order
        // MR_SO_IO) -- computed directly here.  `pbcorrect` only needs the file to exist.
        std::vector<uint32_t> order(n);
        for(uint32_t i = 0; i < n; ++i) order[i] = i;
        const char* B = bases.data();
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const uint64_t lx = off[x + 1] - off[x], ly = off[y + 1] - off[y];
            const uint64_t m = lx < ly ? lx : ly;
            for(uint64_t t = 0; t < m; ++t) {
                const char cx = rev ? B[off[x + 1] - 1 - t] : B[off[x] + t], cy = rev ? B[off[y + 1] - 1 - t] : B[off[y] + t];
                if(cx != cy) return cx < cy;
            }
            if(lx != ly) return lx < ly;
            return x < y;
        });
        std::ofstream sai((prefix + (rev ? ".rsai" : ".sai")).c_str());
        sai << 51914 << "\n" << n << "\n" << n << "\n";
        for(uint32_t i = 0; i < n; ++i) sai << order[i] << " 0\n";
        if(!sai) { std::cerr << "index: cannot write " << prefix << (rev ? ".rsai" : ".sai") << "\n"; return EXIT_FAILURE; }
    }
    return 0;
}

int main(int argc, char** argv)
{
    // several workers on one device (--devices 0,0) only overlap if their streams get hardware queues of their own (default: 4)
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    if(argc <= 1) { std::cerr << "Usage: " PACKAGE_NAME " <command> [options]\nCommands: index, pbcorrect, kmerfreq, kmercheck\n"; return EXIT_FAILURE; }
    const std::string command(argv[1]);
    if(command == "help" || command == "--help") { std::cout << "Usage: " PACKAGE_NAME " <command> [options]\nCommands: index, pbcorrect, kmerfreq, kmercheck\n"; return 0; }
    if(command == "pbcorrect" || command == SUBPROGRAM) return PacBioSelfCorrectionMain(argc - 1, argv + 1);
    if(command == "index") return indexMain(argc - 1, argv + 1);
    if(command == "kmerfreq") return kmerfreqMain(argc - 1, argv + 1);
    if(command == "kmercheck") return kmercheckMain(argc - 1, argv + 1);
    std::cerr << "Unrecognized command: " << command << "\n";
    return EXIT_FAILURE;
}
