// sort_order.cpp -- test hook: runs the product's emulation of libstdc++'s std::sort (csrc/introsort_emul.h, the tie order
// of the interval-tree build in the FM-extend engine) on the host, so that tests can compare it with the real std::sort and
// with the reference's object code.
#include <cstdint>
#include <vector>

#include "../csrc/introsort_emul.h"
#include "lrsc_testkit.h"

using namespace lrsc;

extern "C" int lrsc_debug_sort_order(const uint64_t* keys, uint32_t n, uint32_t* perm_out)
{
    if((!keys || !perm_out) && n) return -1;
    std::vector<SortItem> v(n);
    for(uint32_t i = 0; i < n; ++i) { v[i].key = keys[i]; v[i].val = i; v[i].pad = 0; }
    introsort(v.data(), (int64_t)n);
    for(uint32_t i = 0; i < n; ++i) perm_out[i] = v[i].val;
    return 0;
}
