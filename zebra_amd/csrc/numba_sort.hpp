// Exact emulation of numba's float64 argsort for the T-PPR top-k prune.
//
// The reference selects `np.argsort(values)[-k:]` inside @jitclass code
// (utils/util.py:258,555,658,762,851), i.e. numba's quicksort
// (numba/misc/quicksort.py, numba 0.54.1; lt(a,b) = isnan(b) or a < b).  That
// sort is not stable, and both the membership of the top-k under ties and the
// ORDER of the selected entries (which is the dictionary order the next merge
// sees) follow from its partition dynamics.  Three paths, same result:
//
//   * fast: wave-parallel rank counting.  If no group of equal values reaches
//     into the top-k, the last k entries of ANY correct ascending sort are the
//     same sequence, so rank - (n-k) is the output position.
//   * exact, wave-parallel (n <= 128, no NaN): the quicksort is replayed
//     partition by partition, each Hoare partition in O(1) wave steps:
//     the i-scan stops exactly at the positions holding a value >= pivot, the
//     j-scan at those holding a value <= pivot (the scans only ever look at
//     positions no swap has touched yet), so the m-th i-stop is swapped with
//     the m-th j-stop while it lies to its left; ranks within the two stop
//     lists come from ballots.  Segments shorter than 16 are finished by a
//     stable rank (numba's insertion sort with strict < is a stable sort).
//   * exact, sequential: one lane replays the algorithm literally (NaNs,
//     n > 128: the pruning strategy's long candidate lists).
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

// LDS scratch of the wave-parallel sort (one per wave).
struct SortLds {
    int r[128];        // permutation being sorted
    int r2[128];       // final permutation
    int ilist[130];    // i-stops of the current partition (ascending) + sentinel
    int jlist[130];    // j-stops (descending) + sentinel
    double v[128];     // values in position order (for the final stable rank)
    short seg_lo[128]; // finished segment of every position
    short seg_hi[128];
    int stk[96];
};

// One Hoare partition of r[low..high] (high - low >= 15), wave-parallel.
// Returns the pivot's final position (uniform across the wave).
__device__ inline int partition_wave(const double *a, SortLds &S, int low, int high)
{
    const int lane = lane_id();
    const int mid = (low + high) >> 1;
    int rl = S.r[low], rm = S.r[mid], rh = S.r[high];
    double al = a[rl], am = a[rm], ah = a[rh];
    if (lt_f(am, al)) { int t = rl; rl = rm; rm = t; double x = al; al = am; am = x; }
    if (lt_f(ah, am)) { int t = rh; rh = rm; rm = t; double x = ah; ah = am; am = x; }
    if (lt_f(am, al)) { int t = rl; rl = rm; rm = t; double x = al; al = am; am = x; }
    const double pivot = am;
    wave_sync();                                   // all lanes have read r[low], r[mid], r[high]
    if (lane == 0) { S.r[low] = rl; S.r[mid] = rh; S.r[high] = rm; }   // pivot stashed at `high`
    wave_sync();
    // stop masks over positions low .. high-1 (two slots per lane)
    const int p0 = low + lane, p1 = low + lane + WAVE;
    const bool in0 = p0 <= high - 1, in1 = p1 <= high - 1;
    const double v0 = in0 ? a[S.r[p0]] : 0.0, v1 = in1 ? a[S.r[p1]] : 0.0;
    const bool ge0 = in0 && !lt_f(v0, pivot), ge1 = in1 && !lt_f(v1, pivot);
    const bool le0 = in0 && !lt_f(pivot, v0), le1 = in1 && !lt_f(pivot, v1);
    const u64 GE0 = __ballot(ge0), GE1 = __ballot(ge1), LE0 = __ballot(le0), LE1 = __ballot(le1);
    const u64 below = lanemask_lt();
    const u64 above = ~below & ~(1ull << lane);
    const int nI = __popcll(GE0) + __popcll(GE1), nJ = __popcll(LE0) + __popcll(LE1);
    if (ge0) S.ilist[__popcll(GE0 & below)] = p0;
    if (ge1) S.ilist[__popcll(GE0) + __popcll(GE1 & below)] = p1;
    if (le1) S.jlist[__popcll(LE1 & above)] = p1;
    if (le0) S.jlist[__popcll(LE1) + __popcll(LE0 & above)] = p0;
    if (lane == 0) { S.ilist[nI] = high; S.jlist[nJ] = low - 1; }
    wave_sync();
    // the m-th i-stop and the m-th j-stop are swapped while the former lies to the left
    const int npair = (nI < nJ ? nI : nJ);          // sentinels never swap
    int M = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int m = lane + s * WAVE;
        bool sw = false;
        int pi = 0, pj = 0;
        if (m < npair) { pi = S.ilist[m]; pj = S.jlist[m]; sw = pi < pj; }
        if (sw) { const int x = S.r[pi], y = S.r[pj]; S.r[pi] = y; S.r[pj] = x; }
        M += __popcll(__ballot(sw));
    }
    // where the i-scan finally stops: the next original stop, or the slot the
    // last swap filled with a >= pivot value, whichever comes first
    int ifin = S.ilist[M];
    if (M > 0) { const int jl = S.jlist[M - 1]; ifin = jl < ifin ? jl : ifin; }
    ifin = __builtin_amdgcn_readfirstlane(ifin);   // same in every lane; tell the compiler
    wave_sync();
    if (lane == 0) { const int t = S.r[ifin]; S.r[ifin] = S.r[high]; S.r[high] = t; }
    wave_sync();
    return ifin;
}

// Full argsort, wave-parallel.  Requires 2 <= n <= 128 and no NaN in a[0..n).
// Result in S.r2[0..n).
__device__ inline void numba_argsort_wave(const double *a, int n, SortLds &S)
{
    const int lane = lane_id();
    for (int p = lane; p < n; p += WAVE) S.r[p] = p;
    if (lane == 0) { S.stk[0] = 0; S.stk[1] = n - 1; }
    wave_sync();
    int sp = 1;
    while (sp > 0) {
        --sp;
        int low = __builtin_amdgcn_readfirstlane(S.stk[2 * sp]);
        int high = __builtin_amdgcn_readfirstlane(S.stk[2 * sp + 1]);
        wave_sync();
        while (high - low >= 15) {
            const int i = partition_wave(a, S, low, high);
            if (lane == 0) { S.seg_lo[i] = (short)i; S.seg_hi[i] = (short)i; }   // the pivot is in its final place
            if (high - i > i - low) {
                if (high > i) { if (lane == 0) { S.stk[2 * sp] = i + 1; S.stk[2 * sp + 1] = high; } ++sp; }
                high = i - 1;
            } else {
                if (i > low) { if (lane == 0) { S.stk[2 * sp] = low; S.stk[2 * sp + 1] = i - 1; } ++sp; }
                low = i + 1;
            }
        }
        // [low, high] is finished by insertion sort: remember the segment
        if (low + lane <= high) { S.seg_lo[low + lane] = (short)low; S.seg_hi[low + lane] = (short)high; }
        wave_sync();
    }
    // stable rank inside every segment (= insertion sort with strict <)
    for (int p = lane; p < n; p += WAVE) S.v[p] = a[S.r[p]];
    wave_sync();
    for (int p = lane; p < n; p += WAVE) {
        const int lo = S.seg_lo[p], hi = S.seg_hi[p];
        const double v = S.v[p];
        int rank = 0;
        for (int q = lo; q <= hi; ++q) {
            const double x = S.v[q];
            rank += (x < v || (x == v && q < p)) ? 1 : 0;
        }
        S.r2[lo + rank] = S.r[p];
    }
    wave_sync();
}

// Wave-cooperative top-k selection.  On return sel[0..k) holds the indices of
// np.argsort(a)[-k:] in that order.  Requires n > k.  `a` and `sel` are LDS
// arrays owned by this wave.  Returns 0 = fast path, 1 = exact wave-parallel,
// 2 = exact sequential.
__device__ inline int topk_select_wave(const double *a, int n, int k, int *sel, SortLds &S, int *seq_perm,
                                       int *seq_stk)
{
    const int lane = lane_id();
    const int drop = n - k;
    bool slow = false, has_nan = false;
    // rank counting: every candidate c (strided over lanes) counts smaller /
    // equal values; all lanes read a[q] at the same address (LDS broadcast).
    for (int c = lane; c < ((n + WAVE - 1) / WAVE) * WAVE; c += WAVE) {
        int lt = 0, eq = 0;
        const bool live = c < n;
        const double v = live ? a[c] : 0.0;
        const bool nan = live && (v != v);
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
        if (__ballot(nan) != 0ull) has_nan = true;
        if (!slow && live && lt >= drop) sel[lt - drop] = c;
    }
    wave_sync();
    if (!slow) return 0;
    if (n <= 128 && !has_nan) {
        numba_argsort_wave(a, n, S);
        if (lane < k) sel[lane] = S.r2[drop + lane];
        wave_sync();
        return 1;
    }
    if (lane == 0) {       // seq_perm: caller scratch of >= n ints, seq_stk: >= 96 ints
        numba_argsort_seq(a, n, seq_perm, seq_stk);
        for (int q = 0; q < k; ++q) sel[q] = seq_perm[drop + q];
    }
    wave_sync();
    return 2;
}

}  // namespace zt
