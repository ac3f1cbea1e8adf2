// libstdcxx_sort.hpp — libstdc++'s std::sort (bits/stl_algo.h: __introsort_loop with median-of-three
// __unguarded_partition_pivot down to 16 elements, heap sort (__partial_sort) when the depth limit 2 floor(log2 n) runs out, then
// __final_insertion_sort) restated step by step on an INDEX array, for device code that has to reproduce the order in which the
// reference's std::sort leaves EQUAL keys (the standard leaves it open; the reference's results are whatever this implementation
// does: src/pose_proposal.cpp:224 sorts limb candidates by confidence and pops them from the back).  The sequence of
// comparisons and moves is a pure function of the comparator's answers.  `comp(a, b)` compares the ELEMENTS with indices a and b
// (the reference's comparator on the elements themselves).  Meant for one lane; v lives in LDS or registers.
#pragma once
#include <hip/hip_runtime.h>

namespace hp {

template <class Comp>
__device__ inline void libstdcxx_adjust_heap(int* v, int first, int hole, int len, int value, Comp comp)
{
    // bits/stl_heap.h __adjust_heap(first, holeIndex, len, value, comp) followed by __push_heap
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(v[first + child], v[first + child - 1]))
            --child;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && comp(v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}

// returns false only if the explicit stack overflowed (cannot happen: its depth is bounded by the depth limit)
template <class Comp>
__device__ inline bool libstdcxx_sort(int* v, int n, Comp comp, bool* used_heap = nullptr)
{
    if (n <= 1)
        return true;
    bool ok = true;
    int lg = 0;
    while ((2 << lg) <= n)
        ++lg;
    auto swp = [&](int i, int j) {
        const int t = v[i];
        v[i] = v[j];
        v[j] = t;
    };
    // __introsort_loop(first, last, depth): `while (last - first > 16) { ...; __introsort_loop(cut, last, depth); last = cut; }`: the
    // recursion takes the RIGHT part first; an explicit stack of (first, last, depth) continuations reproduces the same sequence
    int stk_f[64], stk_l[64], stk_d[64], sp = 0;
    stk_f[0] = 0, stk_l[0] = n, stk_d[0] = 2 * lg, sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_f[sp], last = stk_l[sp], depth = stk_d[sp];
        while (last - first > 16) {
            if (depth == 0) {
                // std::__partial_sort(first, last, last): __heap_select (= __make_heap, its scan over [middle, last) is empty) + __sort_heap
                const int len = last - first;
                for (int parent = (len - 2) / 2;; --parent) {
                    libstdcxx_adjust_heap(v, first, parent, len, v[first + parent], comp);
                    if (parent == 0)
                        break;
                }
                for (int l = last; l - first > 1;) {
                    --l;
                    const int value = v[l]; // __pop_heap(first, l, l)
                    v[l] = v[first];
                    libstdcxx_adjust_heap(v, first, 0, l - first, value, comp);
                }
                if (used_heap)
                    *used_heap = true;
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            { // __move_median_to_first(result = first, a = first + 1, b = mid, c = last - 1)
                const int a = first + 1, b = mid, c = last - 1;
                if (comp(v[a], v[b])) {
                    if (comp(v[b], v[c]))
                        swp(first, b);
                    else if (comp(v[a], v[c]))
                        swp(first, c);
                    else
                        swp(first, a);
                } else if (comp(v[a], v[c]))
                    swp(first, a);
                else if (comp(v[b], v[c]))
                    swp(first, c);
                else
                    swp(first, b);
            }
            int lo = first + 1, hi = last; // __unguarded_partition(first + 1, last, pivot = first)
            for (;;) {
                while (comp(v[lo], v[first]))
                    ++lo;
                --hi;
                while (comp(v[first], v[hi]))
                    --hi;
                if (!(lo < hi))
                    break;
                swp(lo, hi);
                ++lo;
            }
            const int cut = lo;
            if (sp < 63) {
                stk_f[sp] = first, stk_l[sp] = cut, stk_d[sp] = depth, ++sp; // continued after the right part is done
            } else
                ok = false;
            first = cut;
        }
    }
    // __final_insertion_sort: guarded insertion sort of the first 16, unguarded linear inserts for the rest
    const int head = n > 16 ? 16 : n;
    for (int i = 1; i < head; ++i) {
        const int val = v[i];
        if (comp(val, v[0])) {
            for (int k = i; k > 0; --k)
                v[k] = v[k - 1];
            v[0] = val;
        } else {
            int k = i;
            while (comp(val, v[k - 1])) {
                v[k] = v[k - 1];
                --k;
            }
            v[k] = val;
        }
    }
    for (int i = head; i < n; ++i) {
        const int val = v[i];
        int k = i;
        while (k > 0 && comp(val, v[k - 1])) { // (k > 0 never decides after a completed introsort loop; kept as a guard)
            v[k] = v[k - 1];
            --k;
        }
        v[k] = val;
    }
    return ok;
}

} // namespace hp
