// dbtk_sort.h — GCC libstdc++ std::sort, restated for index arrays in LDS.
//
// Why this exists: the reference orders a pair's unique k-mers by their number
// of mapped loci with an UNSTABLE std::sort (getSortedIndex / fillstats,
// src/aQueryFasta_thread.cpp:247-250, 320-327) and then votes with early
// termination, so the tie permutation of GCC's introsort decides which locus a
// chimeric pair is assigned to (SURVEY.md finding 3a).  This is the algorithm of
// bits/stl_algo.h:1800-1957 + bits/stl_heap.h (GCC 11): introsort loop with
// threshold 16, median-of-3 moved to `first`, unguarded Hoare partition, depth
// limit 2*floor(log2 n) falling back to heapsort, then insertion sort of the
// first 16 and unguarded insertion of the rest.
//
// idx: uint16_t indices (values 0..n-1), key: the sort key per ORIGINAL index.
// Comparator: key[a] < key[b].  The recursion of __introsort_loop is replaced
// by an explicit stack: the sub-ranges are disjoint, so the order in which they
// are processed does not change the result.
#ifndef DBTK_SORT_H_
#define DBTK_SORT_H_

#include "dbtk_tables.h"

namespace dbtk {

struct SortLt {  // indices compared through a key array: getSortedIndex's comparator, literally
    typedef uint16_t elem_t;
    const uint32_t* key;
    DBTK_HD bool operator()(uint16_t a, uint16_t b) const { return key[a] < key[b]; }
};
// The same order on packed words (key << 9 | index): one LDS read per element instead of two.
// The comparator looks at the key bits only, so ties behave exactly as in the index form.
struct PackedLt {
    typedef uint32_t elem_t;
    DBTK_HD bool operator()(uint32_t a, uint32_t b) const { return (a >> 9) < (b >> 9); }
};

template <class Lt> DBTK_HD void s_unguarded_linear_insert(typename Lt::elem_t* a, int last, const Lt& lt) {
    const typename Lt::elem_t val = a[last];
    int next = last - 1;
    while (lt(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}

template <class Lt> DBTK_HD void s_insertion_sort(typename Lt::elem_t* a, int first, int last, const Lt& lt) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (lt(a[i], a[first])) {
            const typename Lt::elem_t val = a[i];
            for (int j = i; j > first; --j) a[j] = a[j - 1];  // move_backward
            a[first] = val;
        } else {
            s_unguarded_linear_insert(a, i, lt);
        }
    }
}

template <class Lt> DBTK_HD void s_push_heap(typename Lt::elem_t* a, int first, int hole, int top, typename Lt::elem_t value, const Lt& lt) {
    int parent = (hole - 1) / 2;
    while (hole > top && lt(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <class Lt> DBTK_HD void s_adjust_heap(typename Lt::elem_t* a, int first, int hole, int len, typename Lt::elem_t value, const Lt& lt) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(a[first + child], a[first + child - 1])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    s_push_heap(a, first, hole, top, value, lt);
}

template <class Lt> DBTK_HD void s_heapsort(typename Lt::elem_t* a, int first, int last, const Lt& lt) {  // __partial_sort(first,last,last)
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const typename Lt::elem_t v = a[first + parent];
            s_adjust_heap(a, first, parent, len, v, lt);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        const typename Lt::elem_t v = a[last];
        a[last] = a[first];
        s_adjust_heap(a, first, 0, last - first, v, lt);
    }
}

template <class E> DBTK_HD void s_swap(E* a, int i, int j) {
    const E t = a[i];
    a[i] = a[j];
    a[j] = t;
}

template <class Lt> DBTK_HD int s_partition_pivot(typename Lt::elem_t* a, int first, int last, const Lt& lt) {
    const int mid = first + (last - first) / 2;
    {  // __move_median_to_first(first, first+1, mid, last-1)
        const int x = first + 1, y = mid, z = last - 1;
        if (lt(a[x], a[y])) {
            if (lt(a[y], a[z])) s_swap(a, first, y);
            else if (lt(a[x], a[z])) s_swap(a, first, z);
            else s_swap(a, first, x);
        } else if (lt(a[x], a[z])) s_swap(a, first, x);
        else if (lt(a[y], a[z])) s_swap(a, first, z);
        else s_swap(a, first, y);
    }
    int lo = first + 1, hi = last;  // __unguarded_partition(first+1, last, pivot = first)
    for (;;) {
        while (lt(a[lo], a[first])) ++lo;
        --hi;
        while (lt(a[first], a[hi])) --hi;
        if (!(lo < hi)) return lo;
        s_swap(a, lo, hi);
        ++lo;
    }
}

// std::sort(a, a + n, lt).  stack: caller-provided int[3 * 40] scratch (first, last, depth triples).
template <class Lt> DBTK_HD void gcc_sort(typename Lt::elem_t* a, int n, const Lt& lt, int* stack) {
    if (n == 0) return;
    int sp = 0;
    stack[0] = 0;
    stack[1] = n;
    stack[2] = 2 * (31 - __builtin_clz((unsigned)n));  // std::__lg(n) * 2
    sp = 1;
    while (sp > 0) {
        --sp;
        int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {
            if (depth == 0) {
                s_heapsort(a, first, last, lt);
                break;
            }
            --depth;
            const int cut = s_partition_pivot(a, first, last, lt);
            stack[3 * sp] = cut;  // the recursive call __introsort_loop(cut, last, depth)
            stack[3 * sp + 1] = last;
            stack[3 * sp + 2] = depth;
            ++sp;
            last = cut;
        }
    }
    if (n > 16) {  // __final_insertion_sort
        s_insertion_sort(a, 0, 16, lt);
        for (int i = 16; i != n; ++i) s_unguarded_linear_insert(a, i, lt);
    } else {
        s_insertion_sort(a, 0, n, lt);
    }
}

// ---- the same std::sort carried out by a whole wavefront on packed words in LDS.
// Introsort's partitions are data-parallel once stated without the two walking pointers:
// with A = positions (ascending) whose element is not less than the pivot and B = positions
// (descending) whose element is not greater, __unguarded_partition swaps A[j] <-> B[j] for
// j < J = #{j : A[j] < B[j]} (a prefix, since A ascends and B descends; the pairs are disjoint)
// and returns min(A[J], B[J-1]): after the J-th swap the left pointer stops at the next
// original stopper or at the element it has just swapped into B[J-1], whichever comes first.
// The recursion tree is walked with the same explicit stack as gcc_sort, so every sub-range is
// partitioned exactly as libstdc++ does it; depth-limit exhaustion falls back to one lane running
// the heapsort.  __final_insertion_sort is a stable sort of whatever the partitions left
// (insertion moves an element only past strictly greater ones), done here by ranking.
// a: n packed words, sorted result in `out`; apos/bpos: n uint16 each; stack: int[3 * 40].
template <class X>
DBTK_HD void wave_gcc_sort_packed(X& x, uint32_t* a, int n, uint16_t* apos, uint16_t* bpos, uint32_t* out, int* stack) {
    const int lane = x.lane();
    const uint64_t below = (1ull << lane) - 1;
    if (n <= 0) return;
    int sp = 0, first = 0, last = n, depth = 2 * (31 - __builtin_clz((unsigned)n));
    for (;;) {
        while (last - first > 16) {
            if (depth == 0) {
                x.sync();
                if (lane == 0) s_heapsort(a, first, last, PackedLt{});
                x.sync();
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            const uint32_t vf = x.uni(a[first]), vx = x.uni(a[first + 1]), vy = x.uni(a[mid]), vz = x.uni(a[last - 1]);
            const uint32_t kx = vx >> 9, ky = vy >> 9, kz = vz >> 9;
            int msel;  // __move_median_to_first(first, first+1, mid, last-1): 0 = first+1, 1 = mid, 2 = last-1
            if (kx < ky) msel = (ky < kz) ? 1 : ((kx < kz) ? 2 : 0);
            else msel = (kx < kz) ? 0 : ((ky < kz) ? 2 : 1);
            const int mpos = msel == 0 ? first + 1 : (msel == 1 ? mid : last - 1);
            const uint32_t vm = msel == 0 ? vx : (msel == 1 ? vy : vz);
            x.sync();
            if (lane == 0) { a[first] = vm; a[mpos] = vf; }
            x.sync();
            const uint32_t pv = vm >> 9;
            int totA = 0, totB = 0;
            for (int c0 = first + 1; c0 < last; c0 += 64) {
                const int p = c0 + lane;
                const bool in = p < last;
                const uint32_t key = in ? a[p] >> 9 : 0u;
                const bool ga = in && !(key < pv), gb = in && !(pv < key);
                const uint64_t ma = x.ballot(ga), mb = x.ballot(gb);
                if (ga) apos[totA + __builtin_popcountll(ma & below)] = (uint16_t)p;
                if (gb) bpos[totB + __builtin_popcountll(mb & below)] = (uint16_t)p;  // ascending; B[j] = bpos[totB - 1 - j]
                totA += __builtin_popcountll(ma);
                totB += __builtin_popcountll(mb);
            }
            x.sync();
            const int nmin = totA < totB ? totA : totB;
            int J = 0;
            for (int j0 = 0; j0 < nmin; j0 += 64) {
                const int j = j0 + lane;
                int pa = 0, pb = 0;
                bool pr = false;
                if (j < nmin) { pa = apos[j]; pb = bpos[totB - 1 - j]; pr = pa < pb; }
                const uint64_t mk = x.ballot(pr);
                if (pr) { const uint32_t va = a[pa], vb = a[pb]; a[pa] = vb; a[pb] = va; }
                const int c = __builtin_popcountll(mk);
                J += c;
                if (c < 64) break;
            }
            const int ca = J < totA ? (int)x.uni(apos[J]) : 0x7FFFFFFF, cb = J > 0 ? (int)x.uni(bpos[totB - J]) : 0x7FFFFFFF;
            const int cut = ca < cb ? ca : cb;
            x.sync();
            stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth;  // the recursive call __introsort_loop(cut, last, depth)
            ++sp;
            last = cut;
        }
        if (sp == 0) break;
        --sp;
        first = (int)x.uni((uint32_t)stack[3 * sp]); last = (int)x.uni((uint32_t)stack[3 * sp + 1]); depth = (int)x.uni((uint32_t)stack[3 * sp + 2]);
    }
    x.sync();
    // stable sort by key of the partitioned array: one ranking pass per distinct key, ascending
    int placed = 0;
    bool started = false;
    uint32_t prev = 0;
    while (placed < n) {
        uint32_t mymin = 0xFFFFFFFFu;
        for (int p = lane; p < n; p += 64) {
            const uint32_t k = a[p] >> 9;
            if ((!started || k > prev) && k < mymin) mymin = k;
        }
        const uint32_t kmin = x.wave_min(mymin);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int p = c0 + lane;
            const uint32_t v = p < n ? a[p] : 0u;
            const bool in = p < n && (v >> 9) == kmin;
            const uint64_t m = x.ballot(in);
            if (in) out[placed + __builtin_popcountll(m & below)] = v;
            placed += __builtin_popcountll(m);
        }
        prev = kmin;
        started = true;
    }
    x.sync();
}

// getSortedIndex (src/aQueryFasta_thread.cpp:247-250): iota, then std::sort by key[index].
DBTK_HD void gcc_sort_index(uint16_t* a, int n, const uint32_t* key, int* stack) {
    for (int i = 0; i < n; ++i) a[i] = (uint16_t)i;  // std::iota
    gcc_sort(a, n, SortLt{key}, stack);
}

}  // namespace dbtk
#endif
