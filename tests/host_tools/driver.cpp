// tests/host_tools/driver.cpp -- TEST INFRASTRUCTURE: command-line access to the host-side helpers of `stride --onlyseed` /
// `stride kmercheck` (longreadselfcorrect_amd/host/BCode.cpp, KmerDistribution.h) so that tests/test_host_tools.py can
// compare them with golden vectors made from the reference's own object code (tests/golden/make_host_tools.py).
//   driver validate          stdin: "pos ksize start end rvc code seq" per line -> 1 / 0 / -1 (std::out_of_range) / -2 (assert)
//   driver load FILE         -> "qname start end rvc code" per block, map order
//   driver compare COV K     stdin: line 1 = frequencies of correct k-mers, line 2 = of wrong ones -> total.box + value.box lines
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>

#include "../../longreadselfcorrect_amd/host/BCode.h"
#include "../../longreadselfcorrect_amd/host/KmerDistribution.h"

using namespace stride;

int main(int argc, char** argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    if(cmd == "validate") {
        int pos, ksize, start, end, rvc;
        std::string code, seq;
        while(std::cin >> pos >> ksize >> start >> end >> rvc >> code >> seq) {
            int verdict;
            try {
                verdict = BCode::validate(pos, ksize, BCode(start, end, code, rvc != 0), seq) ? 1 : 0;
            } catch(const std::out_of_range&) {
                verdict = -1;
            } catch(const std::logic_error&) {
                verdict = -2;
            }
            std::cout << verdict << '\n';
        }
        return 0;
    }
    if(cmd == "load" && argc > 2) {
        BCode::load(argv[2]);
        for(const auto& kv : BCode::Log())
            for(const BCode& b : kv.second)
                std::cout << kv.first << ' ' << b.getStart() << ' ' << b.getEnd() << ' ' << (b.getRvc() ? 1 : 0) << ' ' << b.getCode() << '\n';
        return 0;
    }
    if(cmd == "compare" && argc > 3) {
        KmerDistribution c, e;
        std::string line;
        for(KmerDistribution* d : {&c, &e}) {
            std::getline(std::cin, line);
            std::istringstream in(line);
            for(int f; in >> f;) d->add(f);
        }
        compare(std::cout, std::cout, atoi(argv[2]), atoi(argv[3]), c, e);
        return 0;
    }
    std::cerr << "usage: driver validate | load FILE | compare COV K\n";
    return 2;
}
