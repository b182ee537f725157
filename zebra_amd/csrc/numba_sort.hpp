// Exact emulation of numba's float64 argsort for the T-PPR top-k prune.
//
// The reference selects `np.argsort(values)[-k:]` inside @jitclass code
// (utils/util.py:258,555,658,762,851), i.e. numba's quicksort
// (numba/misc/quicksort.py, numba 0.54.1; lt(a,b) = isnan(b) or a < b).  That
// sort is not stable, and both the membership of the top-k under ties and the
// ORDER of the selected entries (which is the dictionary order the next merge
// sees) follow from its partition dynamics.  Two paths, same result:
//
//   * fast: wave-parallel rank counting.  If no group of equal values reaches
//     into the top-k, the last k entries of ANY correct ascending sort are the
//     same sequence, so rank - (n-k) is the output position.
//   * exact: one lane replays the quicksort on the LDS copy.
#pragma once

#include "common.hpp"

namespace zt {

__device__ __forceinline__ bool lt_f(double a, double b) { return (b != b) || a < b; }

// Sequential replay (run by ONE lane).  a[0..n) values, r[0..n) permutation out.
__device__ inline void numba_argsort_seq(const double *a, int n, int *r, int *stk /* >= 2*48 ints */)
{
    for (int i = 0; i < n; ++i) r[i] = i;
    if (n < 2) return;
    int sp = 0;
    stk[0] = 0;
    stk[1] = n - 1;
    sp = 1;
    while (sp > 0) {
        --sp;
        int low = stk[2 * sp], high = stk[2 * sp + 1];
        while (high - low >= 15) {
            int mid = (low + high) >> 1, t;
            if (lt_f(a[r[mid]], a[r[low]])) { t = r[low]; r[low] = r[mid]; r[mid] = t; }
            if (lt_f(a[r[high]], a[r[mid]])) { t = r[high]; r[high] = r[mid]; r[mid] = t; }
            if (lt_f(a[r[mid]], a[r[low]])) { t = r[low]; r[low] = r[mid]; r[mid] = t; }
            const double pivot = a[r[mid]];
            t = r[high]; r[high] = r[mid]; r[mid] = t;
            int i = low, j = high - 1;
            for (;;) {
                while (i < high && lt_f(a[r[i]], pivot)) ++i;
                while (j >= low && lt_f(pivot, a[r[j]])) --j;
                if (i >= j) break;
                t = r[i]; r[i] = r[j]; r[j] = t;
                ++i; --j;
            }
            t = r[i]; r[i] = r[high]; r[high] = t;
            if (high - i > i - low) {
                if (high > i) { stk[2 * sp] = i + 1; stk[2 * sp + 1] = high; ++sp; }
                high = i - 1;
            } else {
                if (i > low) { stk[2 * sp] = low; stk[2 * sp + 1] = i - 1; ++sp; }
                low = i + 1;
            }
        }
        for (int p = low + 1; p <= high; ++p) {
            const int kk = r[p];
            const double v = a[kk];
            int q = p;
            while (q > low && lt_f(v, a[r[q - 1]])) { r[q] = r[q - 1]; --q; }
            r[q] = kk;
        }
    }
}

// Wave-cooperative top-k selection.  On return sel[0..k) holds the indices of
// np.argsort(a)[-k:] in that order.  Requires n > k.  `a`, `sel`, `perm`, `stk`
// are LDS arrays owned by this wave (perm >= n ints, sel >= k ints).
// Returns true when the exact (sequential) path was needed.
__device__ inline bool topk_select_wave(const double *a, int n, int k, int *sel, int *perm, int *stk)
{
    const int lane = lane_id();
    const int drop = n - k;
    bool slow = false;
    // rank counting: every candidate c (strided over lanes) counts smaller /
    // equal values; all lanes read a[q] at the same address (LDS broadcast).
    for (int c = lane; c < ((n + WAVE - 1) / WAVE) * WAVE; c += WAVE) {
        int lt = 0, eq = 0;
        const bool live = c < n;
        const double v = live ? a[c] : 0.0;
        bool nan = live && (v != v);
        for (int q = 0; q < n; ++q) {
            const double x = a[q];
            lt += (x < v) ? 1 : 0;
            eq += (x == v) ? 1 : 0;
        }
        eq -= 1;  // itself
        // a tie group [lt, lt+eq] that reaches rank >= drop makes the order
        // depend on the quicksort; NaNs always do.
        const bool bad = live && (nan || (eq > 0 && lt + eq >= drop));
        if (__ballot(bad) != 0ull) slow = true;
        if (!slow && live && lt >= drop) sel[lt - drop] = c;
    }
    wave_sync();
    if (!slow) return false;
    if (lane == 0) {
        numba_argsort_seq(a, n, perm, stk);
        for (int q = 0; q < k; ++q) sel[q] = perm[drop + q];
    }
    wave_sync();
    return true;
}

}  // namespace zt
