// introsort_emul.h -- an exact re-implementation of libstdc++'s std::sort (introsort: median-of-3
// quicksort with a 2*floor(log2 n) depth limit, heapsort fallback, final insertion sort with the
// 16-element threshold), for host and device.
//
// Why: the reference sorts the per-walk interval lists with std::sort (PacBio/IntervalTree.cpp:18,
// comparator a.start > b.start).  Entries of the same k-mer have equal keys, std::sort is unstable, and
// the order it leaves them in decides which supporting seed isSupportedByNewSeed picks
// (LongReadCorrectByOverlap.cpp:587-630).  Bit-identical output therefore needs the same permutation,
// i.e. the same algorithm, step for step.  Algorithm structure follows GCC's bits/stl_algo.h and
// bits/stl_heap.h (behaviour restated, not copied); tests/test_host_logic.py checks it against the
// real std::sort through the reference's own IntervalTree object code.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LRSC_SORT_HD __host__ __device__ inline
#else
#define LRSC_SORT_HD inline
#endif

namespace lrsc {

struct SortItem {
    uint64_t key;     // interval start
    uint32_t val;     // query offset
    uint32_t pad;
};

// comp(a, b) == std::greater on TreeInterval == a.start > b.start  (descending by key)
LRSC_SORT_HD bool sort_comp(const SortItem& a, const SortItem& b) { return a.key > b.key; }

LRSC_SORT_HD void sort_swap(SortItem* a, int64_t i, int64_t j) { const SortItem t = a[i]; a[i] = a[j]; a[j] = t; }

LRSC_SORT_HD void unguarded_linear_insert(SortItem* a, int64_t last)
{
    const SortItem val = a[last];
    int64_t next = last - 1;
    while(sort_comp(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}

LRSC_SORT_HD void insertion_sort(SortItem* a, int64_t first, int64_t last)
{
    if(first == last) return;
    for(int64_t i = first + 1; i != last; ++i) {
        if(sort_comp(a[i], a[first])) {
            const SortItem val = a[i];
            for(int64_t j = i; j > first; --j) a[j] = a[j - 1];     // move_backward(first, i, i + 1)
            a[first] = val;
        } else
            unguarded_linear_insert(a, i);
    }
}

LRSC_SORT_HD void push_heap_(SortItem* a, int64_t first, int64_t holeIndex, int64_t topIndex, SortItem value)
{
    int64_t parent = (holeIndex - 1) / 2;
    while(holeIndex > topIndex && sort_comp(a[first + parent], value)) {
        a[first + holeIndex] = a[first + parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    a[first + holeIndex] = value;
}

LRSC_SORT_HD void adjust_heap(SortItem* a, int64_t first, int64_t holeIndex, int64_t len, SortItem value)
{
    const int64_t topIndex = holeIndex;
    int64_t secondChild = holeIndex;
    while(secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if(sort_comp(a[first + secondChild], a[first + (secondChild - 1)])) secondChild--;
        a[first + holeIndex] = a[first + secondChild];
        holeIndex = secondChild;
    }
    if((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        a[first + holeIndex] = a[first + (secondChild - 1)];
        holeIndex = secondChild - 1;
    }
    push_heap_(a, first, holeIndex, topIndex, value);
}

// std::__partial_sort(first, last, last): make_heap + sort_heap
LRSC_SORT_HD void heap_sort(SortItem* a, int64_t first, int64_t last)
{
    const int64_t len = last - first;
    if(len >= 2) {
        int64_t parent = (len - 2) / 2;
        while(true) {
            const SortItem value = a[first + parent];
            adjust_heap(a, first, parent, len, value);
            if(parent == 0) break;
            parent--;
        }
    }
    int64_t l = last;
    while(l - first > 1) {
        --l;
        const SortItem value = a[l];      // __pop_heap(first, l, l)
        a[l] = a[first];
        adjust_heap(a, first, 0, l - first, value);
    }
}

LRSC_SORT_HD int64_t unguarded_partition_pivot(SortItem* a, int64_t first, int64_t last)
{
    const int64_t mid = first + (last - first) / 2;
    // __move_median_to_first(first, first + 1, mid, last - 1)
    const int64_t A = first + 1, B = mid, C = last - 1;
    if(sort_comp(a[A], a[B])) {
        if(sort_comp(a[B], a[C])) sort_swap(a, first, B);
        else if(sort_comp(a[A], a[C])) sort_swap(a, first, C);
        else sort_swap(a, first, A);
    } else if(sort_comp(a[A], a[C])) sort_swap(a, first, A);
    else if(sort_comp(a[B], a[C])) sort_swap(a, first, C);
    else sort_swap(a, first, B);
    // __unguarded_partition(first + 1, last, pivot = first)
    int64_t lo = first + 1, hi = last;
    while(true) {
        while(sort_comp(a[lo], a[first])) ++lo;
        --hi;
        while(sort_comp(a[first], a[hi])) --hi;
        if(!(lo < hi)) return lo;
        sort_swap(a, lo, hi);
        ++lo;
    }
}

// std::sort(a, a + n, greater-by-key)
#ifdef __HIPCC__
static __host__ __device__ __attribute__((noinline, unused)) void introsort(SortItem* a, int64_t n)
#else
inline void introsort(SortItem* a, int64_t n)
#endif
{
    if(n <= 0) return;
    // __lg(n) * 2
    int depth = 0;
    for(int64_t t = n; t > 1; t >>= 1) ++depth;
    depth *= 2;
    // __introsort_loop with an explicit stack for the recursive [cut, last) calls
    struct Frame { int64_t first, last; int depth; };
    Frame stack[96];
    int sp = 0;
    stack[sp++] = Frame{0, n, depth};
    while(sp > 0) {
        Frame f = stack[--sp];
        int64_t first = f.first, last = f.last;
        int d = f.depth;
        while(last - first > 16) {
            if(d == 0) {
                heap_sort(a, first, last);
                break;
            }
            --d;
            const int64_t cut = unguarded_partition_pivot(a, first, last);
            // recursion: __introsort_loop(cut, last, d) runs to completion BEFORE the loop continues on
            // [first, cut).  The two ranges are disjoint, so deferring [first, cut) and processing
            // [cut, last) first gives the same result: push the left part, continue with the right.
            stack[sp++] = Frame{first, cut, d};
            first = cut;
        }
    }
    // __final_insertion_sort
    if(n > 16) {
        insertion_sort(a, 0, 16);
        for(int64_t i = 16; i != n; ++i) unguarded_linear_insert(a, i);
    } else
        insertion_sort(a, 0, n);
}

} // namespace lrsc
