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
        for (int q0 = lo; q0 <= hi; q0 += 8) {       // segments hold at most 15 entries
            double x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = S.v[(q0 + t) <= hi ? (q0 + t) : hi];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = q0 + t;
                rank += (q <= hi && (x[t] < v || (x[t] == v && q < p))) ? 1 : 0;
            }
        }
        S.r2[lo + rank] = S.r[p];
    }
    wave_sync();
}

// ---------------------------------------------------------------------------
// Register-resident, level-synchronous replay for n <= 64 (no NaN): lane p IS
// position p and holds (candidate index r, value v).  Every segment that still
// needs a partition (size >= 16) is partitioned in the same pass: segments are
// disjoint lane ranges, ballots are masked per segment, the three
// median-of-3 reads, the pair swaps and the final pivot swap are cross-lane
// permutes.  The order in which numba pops segments off its stack does not
// change the result (partitions of disjoint ranges commute), so levels can be
// processed side by side.  Result: S.r2[0..n) = argsort.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double shfl_f64(double x, int src)
{
    const long long b = __double_as_longlong(x);
    const int lo = __shfl((int)(unsigned)(b & 0xffffffffll), src), hi = __shfl((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double readlane_f64(double x, int src /* wave-uniform */)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ inline void numba_argsort_reg(const double *a, int n, SortLds &S)
{
    const int p = lane_id();
    const bool live = p < n;
    int r = p;
    double v = live ? a[p] : 0.0;
    int lo = 0, hi = live ? n - 1 : -1;               // my segment (inclusive)
    const u64 below = lanemask_lt();
    const u64 above = ~below & ~(1ull << p);
    for (;;) {
        const bool act = live && (hi - lo >= 15);
        const u64 actm = __ballot(act);
        if (actm == 0ull) break;
        // ---- median of three of MY segment (every lane of a segment computes the same) ----
        const int mid = (lo + hi) >> 1;
        int rl, rm, rh;
        double al, am, ah;
        const int first = __ffsll((long long)actm) - 1;
        const int lo0 = __builtin_amdgcn_readlane(lo, first), hi0 = __builtin_amdgcn_readlane(hi, first);
        if (__ballot(act && lo != lo0) == 0ull) {      // one segment in this level: scalar lane reads, no LDS
            const int mid0 = (lo0 + hi0) >> 1;
            rl = __builtin_amdgcn_readlane(r, lo0); rm = __builtin_amdgcn_readlane(r, mid0);
            rh = __builtin_amdgcn_readlane(r, hi0);
            al = readlane_f64(v, lo0); am = readlane_f64(v, mid0); ah = readlane_f64(v, hi0);
        } else {
            const int sl = act ? lo : p, sm = act ? mid : p, sh = act ? hi : p;
            rl = __shfl(r, sl); rm = __shfl(r, sm); rh = __shfl(r, sh);
            al = shfl_f64(v, sl); am = shfl_f64(v, sm); ah = shfl_f64(v, sh);
        }
        if (am < al) { int t = rl; rl = rm; rm = t; double x = al; al = am; am = x; }
        if (ah < am) { int t = rh; rh = rm; rm = t; double x = ah; ah = am; am = x; }
        if (am < al) { int t = rl; rl = rm; rm = t; double x = al; al = am; am = x; }
        const double pivot = am;
        if (act) {                                     // pivot stashed at `hi`
            if (p == lo) { r = rl; v = al; }
            else if (p == mid) { r = rh; v = ah; }
            else if (p == hi) { r = rm; v = am; }
        }
        // ---- stops over lo .. hi-1, ranks inside the segment ----
        const bool in = act && p <= hi - 1;
        const bool ge = in && !(v < pivot), le = in && !(pivot < v);   // no NaN here: lt(a,b) == a < b
        const u64 GE = __ballot(ge), LE = __ballot(le);
        // bits lo .. hi-1 of my segment (act => hi-1 >= lo, hi-1 <= 62)
        const u64 seg = act ? ((((1ull << (hi - 1)) << 1) - 1ull) & ~((1ull << lo) - 1ull)) : 0ull;
        const u64 GEs = GE & seg, LEs = LE & seg;
        const int nI = __popcll(GEs), nJ = __popcll(LEs);
        const int mi = __popcll(GEs & below), mj = __popcll(LEs & above);
        if (ge) S.ilist[lo + mi] = p;                  // slots lo.. are private to the segment
        if (le) S.jlist[lo + mj] = p;
        wave_sync();
        // partner of an i-stop of rank m: the j-stop of rank m (and vice versa); swap while I < J
        int src1 = p;
        bool swi = false, swj = false;
        if (ge && mi < nJ) { const int q = S.jlist[lo + mi]; if (p < q) { src1 = q; swi = true; } }
        if (le && mj < nI) { const int q = S.ilist[lo + mj]; if (q < p) { src1 = q; swj = true; } }
        // where the i-scan finally stops: the first i-stop that did not swap (or `hi`), or the
        // slot the last swap filled with a >= pivot value (the lowest swapped j-stop), whichever is first
        const u64 stay = __ballot(ge && !swi) & seg, swapped_j = __ballot(swj) & seg;
        int ifin = stay ? __ffsll((long long)stay) - 1 : hi;
        if (swapped_j) { const int jl = __ffsll((long long)swapped_j) - 1; ifin = jl < ifin ? jl : ifin; }
        // pair swaps, then pivot (stashed at hi) <-> ifin, applied as ONE permutation
        const int src2 = act ? (p == ifin ? hi : (p == hi ? ifin : p)) : p;
        const int csrc = __shfl(src1, src2);
        r = __shfl(r, csrc);
        v = shfl_f64(v, csrc);
        if (act) {
            if (p < ifin) hi = ifin - 1;
            else if (p > ifin) lo = ifin + 1;
            else { lo = p; hi = p; }
        }
        wave_sync();                                   // the lists are rewritten by the next level
    }
    // ---- stable rank inside every finished segment (insertion sort with strict <) ----
    if (live) S.v[p] = v;
    wave_sync();
    if (live) {
        int rank = 0;
        for (int q0 = lo; q0 <= hi; q0 += 8) {
            double x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = S.v[(q0 + t) <= hi ? (q0 + t) : hi];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = q0 + t;
                rank += (q <= hi && (x[t] < v || (x[t] == v && q < p))) ? 1 : 0;
            }
        }
        S.r2[lo + rank] = r;
    }
    wave_sync();
}

// Rank counting in registers (n <= 64, no NaN): lane c holds candidate c's value and counts the
// smaller ones from n scalar-lane broadcasts.  With all kept values distinct the argsort is simply
// the rank order.  Returns false -- sel is then undefined -- when a tie group reaches the kept
// ranks [n-k, n): either the group straddles the cut (fewer than k lanes see rank >= n-k) or two
// kept lanes claim the same slot.  Ties among dropped candidates do not matter.  *lt_out = number
// of strictly smaller candidates (equal values <=> equal counts), which is all the quicksort
// replay below needs to know about the values.
__device__ inline bool topk_rank_reg(double v, int n, int k, int *sel, int *lt_out)
{
    const int lane = lane_id();
    const int drop = n - k;
    // lanes past n hold +inf: never smaller than anything, so whole groups of 8 lanes can be
    // broadcast without a bounds test; constant lane numbers and four counters keep the VALU busy
    const double vc = lane < n ? v : __longlong_as_double(0x7ff0000000000000ll);
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
    for (int q0 = 0; q0 < WAVE; q0 += 8) {
        if (q0 >= n) break;
        l0 += (readlane_f64(vc, q0 + 0) < vc) ? 1 : 0;
        l1 += (readlane_f64(vc, q0 + 1) < vc) ? 1 : 0;
        l2 += (readlane_f64(vc, q0 + 2) < vc) ? 1 : 0;
        l3 += (readlane_f64(vc, q0 + 3) < vc) ? 1 : 0;
        l0 += (readlane_f64(vc, q0 + 4) < vc) ? 1 : 0;
        l1 += (readlane_f64(vc, q0 + 5) < vc) ? 1 : 0;
        l2 += (readlane_f64(vc, q0 + 6) < vc) ? 1 : 0;
        l3 += (readlane_f64(vc, q0 + 7) < vc) ? 1 : 0;
    }
    const int lt = (l0 + l1) + (l2 + l3);
    *lt_out = lt;
    const bool keep = lane < n && lt >= drop;
    if (__popcll(__ballot(keep)) != k) return false;
    if (keep) sel[lt - drop] = lane;
    wave_sync();
    const bool clash = keep && sel[lt - drop] != lane;
    const bool ok = __ballot(clash) == 0ull;
    wave_sync();
    return ok;
}

// ---- cross-lane helpers of the register-resident merge ---------------------------------------------
// push: lane i's value goes to lane dest_i (ds_permute_b32; lanes nobody writes read 0, of several
// writers the highest lane wins).  Every lane must be active.
__device__ __forceinline__ int push_i32(int v, int dest) { return __builtin_amdgcn_ds_permute(dest << 2, v); }
__device__ __forceinline__ u64 push_u64(u64 v, int dest)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_permute(dest << 2, (int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_permute(dest << 2, (int)(unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double push_f64(double v, int dest)
{
    return __longlong_as_double((long long)push_u64((u64)__double_as_longlong(v), dest));
}

// OR of x over the wave (uniform result): inclusive prefix OR inside each row of 16 by DPP shifts
// (OR is idempotent, overlaps do not matter), then the row totals are chained with the two row
// broadcasts; lane 63 holds the total.
__device__ __forceinline__ unsigned wave_or(unsigned x)
{
    int v = (int)x;
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);    // row_bcast:15 into rows 1 and 3
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);    // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int mbcnt64(u64 m)      // bits of m below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// ---------------------------------------------------------------------------
// Exact top-k when ties decide (n <= 64, no NaN): numba's quicksort replayed on the value RANKS
// (`lt` from topk_rank_reg), one segment at a time with everything about the segment in scalar
// registers.  Lane p is position p and holds e = (rank << 8) | candidate.  Per partition: three
// scalar lane reads + scalar compares for the median of three, two ballots for the scan stops, one
// LDS round trip to pair the m-th i-stop with the m-th j-stop, one cross-lane permute for all the
// swaps, scalar lane writes for the pivot.  Segments that lie wholly below the cut n-k are neither
// partitioned nor ordered (partitions of disjoint ranges are independent).  Finished segments
// (< 16 entries) get the stable rank = insertion sort with strict <.  Writes sel[0..k).
// ---------------------------------------------------------------------------
// sel != nullptr: writes sel[0..k).  sel == nullptr: returns in *f_out the final position of the candidate
// *c_out this lane ends up holding (or -1 if that position is below the cut / not resolved).
__device__ inline void topk_ties_reg(int lt, int n, int k, int *sel, SortLds &S, int *f_out = nullptr, int *c_out = nullptr,
                                     int cand_id = -1)
{
    // Everything about the segment being partitioned (lo, hi, the pivot, the stack of pending segments) is
    // wave-uniform; readfirstlane pins those values to scalar registers so that the bookkeeping runs on the
    // scalar unit and only the per-position work (keys, stop ranks, the permutation) on the vector unit.
#define ZT_U(x) __builtin_amdgcn_readfirstlane(x)
    const int p = lane_id();
    const int drop = ZT_U(n - k);
    int e = (lt << 8) | (cand_id >= 0 ? cand_id : p);   // (rank, candidate); the candidate defaults to the position
    int mylo = p, myhi = p - 1;                     // finished segment holding position p (empty: none)
    unsigned pend_lo = 0u, pend_hi = 0u;            // stack of segments still to partition, 16 bits each
    int depth = 0;
    int lo = 0, hi = ZT_U(n - 1);
    bool work = true;
    if (hi - lo < 15) {
        if (p <= hi) { mylo = lo; myhi = hi; }
        work = false;
    }
    while (work) {
        lo = ZT_U(lo); hi = ZT_U(hi);
        // ---- median of three (uniform values, scalar unit) ----
        const int mid = (lo + hi) >> 1;
        int el = __builtin_amdgcn_readlane(e, lo), em = __builtin_amdgcn_readlane(e, mid),
            eh = __builtin_amdgcn_readlane(e, hi);
        if ((em >> 8) < (el >> 8)) { const int t = el; el = em; em = t; }
        if ((eh >> 8) < (em >> 8)) { const int t = eh; eh = em; em = t; }
        if ((em >> 8) < (el >> 8)) { const int t = el; el = em; em = t; }
        const int pk = em >> 8;
        e = (p == lo ? el : e);
        e = (p == mid ? eh : e);
        e = (p == hi ? em : e);                     // pivot stashed at `hi`
        // ---- stops of the two scans over lo .. hi-1 ----
        const int key = e >> 8;
        const u64 seg = (((1ull << hi) - 1ull) >> lo) << lo;          // bits lo .. hi-1 (hi <= 62)
        const u64 GE = __ballot(key >= pk) & seg, LE = __ballot(key <= pk) & seg;
        const bool ge = (GE >> p) & 1ull, le = (LE >> p) & 1ull;
        // Hoare's scans swap the m-th i-stop (ascending) with the m-th j-stop (descending) while the former lies
        // to the left.  Number of swaps: with a(x) = i-stops at positions <= x and b(x) = j-stops at positions
        // > x, pair m swaps iff some x has a(x) > m and b(x) > m, so S = max_x min(a(x), b(x)); a rises and b
        // falls with x, so the maximum sits at the first position where a >= b, or just before it.
        const int nJ = __popcll(LE);
        const int mi = mbcnt64(GE), lb = mbcnt64(LE);                  // stops strictly below this position
        const int a_p = mi + (ge ? 1 : 0), b_p = nJ - lb - (le ? 1 : 0);
        const u64 C = __ballot(((seg >> p) & 1ull) && a_p >= b_p);     // never empty: b(hi-1) = 0
        const int px = ZT_U(__ffsll((long long)C) - 1);
        const u64 below_px = (1ull << px) - 1ull;
        const int a_prev = __popcll(GE & below_px), b_at = __popcll(LE & ~below_px & ~(1ull << px));
        const int Sw = ZT_U(a_prev > b_at ? a_prev : b_at);            // swaps of this partition
        // The swapped i-stops are the Sw lowest, the swapped j-stops the Sw highest, and every swapped i-stop
        // lies left of every swapped j-stop: the swaps REVERSE the sequence of elements at these 2*Sw
        // positions.  Element number t of that sequence goes to compact lane 2*Sw-1-t (push), then position
        // number t fetches compact lane t (pull): two register permutes, no lists in LDS.
        const int mj = nJ - 1 - lb;                                    // rank among the j-stops, from the top
        const bool swi = ge && mi < Sw, swj = le && mj < Sw;
        const int t = swi ? mi : 2 * Sw - 1 - mj;
        const int staged = push_i32(e, (swi || swj) ? 2 * Sw - 1 - t : 63);
        const int got = __shfl(staged, (swi || swj) ? t : p);
        // where the i-scan ends: the first i-stop that did not swap, or the lowest swapped j-stop
        // (it received a >= pivot value), or `hi`
        const u64 stay = GE & ~__ballot(swi), sj = __ballot(swj);
        int ifin = stay ? __ffsll((long long)stay) - 1 : hi;
        if (sj) { const int jl = __ffsll((long long)sj) - 1; ifin = jl < ifin ? jl : ifin; }
        ifin = ZT_U(ifin);
        e = (swi || swj) ? got : e;                 // all pair swaps at once
        const int x = __builtin_amdgcn_readlane(e, ifin);
        e = (p == ifin ? em : e);                   // pivot <-> ifin
        e = (p == hi ? x : e);
        if (p == ifin) { mylo = p; myhi = p; }
        // ---- children: only those reaching the kept ranks matter ----
        const int lhi = ifin - 1, rlo = ifin + 1;
        const bool lneed = lhi >= lo && lhi >= drop, rneed = hi >= rlo && hi >= drop;
        const bool lpart = lneed && lhi - lo >= 15, rpart = rneed && hi - rlo >= 15;
        if (lneed && !lpart && p >= lo && p <= lhi) { mylo = lo; myhi = lhi; }
        if (rneed && !rpart && p >= rlo && p <= hi) { mylo = rlo; myhi = hi; }
        if (lpart) {
            if (rpart) {                            // push the right child
                pend_hi = (pend_hi << 16) | (pend_lo >> 16);
                pend_lo = (pend_lo << 16) | (unsigned)(rlo | (hi << 8));
                ++depth;
            }
            hi = lhi;
        } else if (rpart) {
            lo = rlo;
        } else if (depth > 0) {
            lo = (int)(pend_lo & 0xffu); hi = (int)((pend_lo >> 8) & 0xffu);
            pend_lo = (pend_lo >> 16) | (pend_hi << 16);
            pend_hi >>= 16;
            --depth;
        } else {
            work = false;
        }
    }
#undef ZT_U
    // ---- stable rank inside every finished segment that reaches the cut ----
    if (p < n) S.r[p] = (e & ~0xff) | p;            // (rank, current position)
    wave_sync();
    if (myhi >= mylo) {
        const int cp = (e & ~0xff) | p;
        int rank = 0;
        for (int q0 = mylo; q0 <= myhi; q0 += 8) {  // segments hold at most 15 entries
            int x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = S.r[(q0 + t) <= myhi ? (q0 + t) : myhi];
#pragma unroll
            for (int t = 0; t < 8; ++t) rank += ((q0 + t) <= myhi && x[t] < cp) ? 1 : 0;
        }
        const int f = mylo + rank;
        if (sel != nullptr) { if (f >= drop) sel[f - drop] = e & 0xff; }
        else *f_out = f >= drop ? f : -1;
    } else if (sel == nullptr) {
        *f_out = -1;
    }
    if (sel == nullptr) *c_out = e & 0xff;
    wave_sync();
}

// max of x >= 0 over the wave (uniform result); same scheme as wave_or (0 is the identity)
__device__ __forceinline__ int wave_max0(int x)
{
    int v = x;
#define ZT_MAXDPP(ctrl, rm) { const int o = __builtin_amdgcn_update_dpp(0, v, ctrl, rm, 0xf, true); v = o > v ? o : v; }
    ZT_MAXDPP(0x111, 0xf) ZT_MAXDPP(0x112, 0xf) ZT_MAXDPP(0x114, 0xf) ZT_MAXDPP(0x118, 0xf)
    ZT_MAXDPP(0x142, 0xa) ZT_MAXDPP(0x143, 0xc)
#undef ZT_MAXDPP
    return __builtin_amdgcn_readlane(v, 63);
}

// Top-k of n <= 63 candidates held one per lane, not necessarily in adjacent lanes, in two steps so that a
// caller can act on the kept SET before the ORDER is known (hub chains, tppr_stream.hip).
//
// rank_pass: `live` (uniform) marks the lanes that hold a candidate, v the value (no NaN), n = popcount(live)
// > k, k <= 31.  *lt = number of strictly smaller candidates, *keep = this lane's candidate is certainly
// kept.  Returns 1 if no tie reaches the kept ranks (slot = lt - drop is the answer), 2 if ties decide the
// ORDER but the kept set is known (exactly k lanes have *keep), 3 if a tie group straddles the cut (the set
// itself follows from the quicksort's dynamics).  No LDS.
__device__ inline int rank_pass(double v, u64 live, int n, int k, int *lt_out, bool *keep_out, unsigned *claimed_out = nullptr)
{
    const int lane = lane_id();
    const int drop = n - k;
    const bool mine = (live >> lane) & 1ull;
    const double vc = mine ? v : __longlong_as_double(0x7ff0000000000000ll);    // dead lanes: +inf, never smaller
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
    for (int q0 = 0; q0 < WAVE; q0 += 8) {
        if (((live >> q0) & 0xffull) == 0ull) continue;                         // uniform: a scalar branch
        l0 += (readlane_f64(vc, q0 + 0) < vc) ? 1 : 0;
        l1 += (readlane_f64(vc, q0 + 1) < vc) ? 1 : 0;
        l2 += (readlane_f64(vc, q0 + 2) < vc) ? 1 : 0;
        l3 += (readlane_f64(vc, q0 + 3) < vc) ? 1 : 0;
        l0 += (readlane_f64(vc, q0 + 4) < vc) ? 1 : 0;
        l1 += (readlane_f64(vc, q0 + 5) < vc) ? 1 : 0;
        l2 += (readlane_f64(vc, q0 + 6) < vc) ? 1 : 0;
        l3 += (readlane_f64(vc, q0 + 7) < vc) ? 1 : 0;
    }
    const int lt = (l0 + l1) + (l2 + l3);
    const bool keep = mine && lt >= drop;
    *lt_out = lt;
    *keep_out = keep;
    // tie-free among the kept <=> exactly k lanes are kept and their ranks cover [drop, n) (equal values
    // have equal counts, so a tie leaves a rank unclaimed)
    const unsigned claimed = wave_or(keep ? 1u << (lt - drop) : 0u);
    if (claimed_out) *claimed_out = claimed;       // bit r: some kept candidate has rank drop + r (a run of g equal values claims one bit)
    const bool full = __popcll(__ballot(keep)) == k;
    if (full && claimed == (1u << k) - 1u) return 1;
    return full ? 2 : 3;
}

// ties_order: the quicksort replay.  `pos` = this lane's place in the candidate LIST (the reference's
// dictionary order, 0..n-1: numba's argsort depends on it).  Returns the slot [0, k) this lane's candidate
// takes in np.argsort(values)[-k:], or -1 if it is dropped.
__device__ inline int ties_order(int lt, u64 live, int pos, int n, int k, SortLds &S)
{
    const int lane = lane_id();
    const int drop = n - k;
    const bool mine = (live >> lane) & 1ull;
    // bring (rank, lane) into list order (lane p = position p) and replay
    const int e_at_pos = push_i32(mine ? ((lt << 8) | lane) : 0, mine ? pos : 63);
    int f, c;
    topk_ties_reg(e_at_pos >> 8, n, k, nullptr, S, &f, &c, e_at_pos & 0xff);
    // lane p holds the candidate of lane c at final position f: tell lane c its slot
    const bool kept = lane < n && f >= drop;
    const int got = push_i32(kept ? f - drop + 1 : 0, kept ? c : 63);
    return (mine && lane != 63) ? got - 1 : -1;
}

// both steps at once.  Returns the path taken (0 ranks, 4 quicksort replay).
__device__ inline int topk_reg(double v, u64 live, int pos, int n, int k, SortLds &S, int *out_slot)
{
    int lt;
    bool keep;
    if (rank_pass(v, live, n, k, &lt, &keep) == 1) {
        *out_slot = keep ? lt - (n - k) : -1;
        return 0;
    }
    *out_slot = ties_order(lt, live, pos, n, k, S);
    return 4;
}

// ---------------------------------------------------------------------------
// The same exact top-k for 64 < n <= 128 candidates (no NaN), two list positions per lane (p and p + 64): the
// pruning strategy's candidate lists (width + width^2 = 110 states at width 10, depth 2).  Ranks (`lt`: strictly smaller
// candidates, equal values <=> equal ranks) come from ONE pass of broadcast reads; if no tie reaches the kept ranks
// the ranks are the answer; otherwise numba's quicksort is replayed on the ranks as in topk_ties_reg -- segment
// bookkeeping in scalar registers, Hoare's pair swaps as the reversal of the swapped stops -- with the elements
// (rank << 8 | candidate) of the two positions in registers between partitions and one round through LDS (write,
// read) for the permutation of a partition.  Segments wholly below the cut are neither partitioned nor ordered.
// Writes sel[0..k).  Returns 0 (ranks decided) or 5 (replay).  a, sel and S are LDS of this wave.
// ---------------------------------------------------------------------------
__device__ inline int topk_select_wave128(const double *a, int n, int k, int *sel, SortLds &S)
{
#define ZT_U(x) __builtin_amdgcn_readfirstlane(x)
    const int lane = lane_id();
    const int drop = ZT_U(n - k);
    const int p0 = lane, p1 = lane + WAVE;
    const bool has0 = p0 < n, has1 = p1 < n;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const double v0 = has0 ? a[p0] : inf, v1 = has1 ? a[p1] : inf;
    // ---- ranks: one broadcast read per candidate, two counters per lane ----
    int lt0 = 0, lt1 = 0;
    for (int q0 = 0; q0 < n; q0 += 8) {
        double x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = a[(q0 + t) < n ? (q0 + t) : 0];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool in = (q0 + t) < n;
            lt0 += (in && x[t] < v0) ? 1 : 0;
            lt1 += (in && x[t] < v1) ? 1 : 0;
        }
    }
    const bool keep0 = has0 && lt0 >= drop, keep1 = has1 && lt1 >= drop;
    // tie-free among the kept <=> exactly k are kept and no two of them have the same rank
    if (lane < k) sel[lane] = -1;
    wave_sync();
    if (keep0) sel[lt0 - drop] = p0;
    if (keep1) sel[lt1 - drop] = p1;
    wave_sync();
    const bool clash = (keep0 && sel[lt0 - drop] != p0) || (keep1 && sel[lt1 - drop] != p1);
    const bool plain = __popcll(__ballot(keep0)) + __popcll(__ballot(keep1)) == k && __ballot(clash) == 0ull;
    wave_sync();
    if (plain) return 0;
    // ---- replay on the ranks ----
    int *E = S.r, *stage = S.r2;
    int e0 = (lt0 << 8) | p0, e1 = (lt1 << 8) | p1;            // (rank, candidate) at positions p0, p1
    if (has0) { S.seg_lo[p0] = (short)1; S.seg_hi[p0] = (short)0; }      // finished segment of the position: none yet
    if (has1) { S.seg_lo[p1] = (short)1; S.seg_hi[p1] = (short)0; }
    unsigned pend_lo = 0u, pend_hi = 0u;                        // stack of pending segments, 16 bits each: lo | hi << 8
    int depth = 0;
    int lo = 0, hi = ZT_U(n - 1);
    bool work = true;
    auto finish = [&](int slo, int shi) {                       // [slo, shi]: fewer than 16 entries, reaches the cut
        if (p0 >= slo && p0 <= shi) { S.seg_lo[p0] = (short)slo; S.seg_hi[p0] = (short)shi; }
        if (p1 >= slo && p1 <= shi) { S.seg_lo[p1] = (short)slo; S.seg_hi[p1] = (short)shi; }
    };
    if (hi - lo < 15) { finish(lo, hi); work = false; }
    const u64 below = lanemask_lt();
    while (work) {
        lo = ZT_U(lo); hi = ZT_U(hi);
        const int mid = (lo + hi) >> 1;
        auto at = [&](int pos) { return pos < WAVE ? __builtin_amdgcn_readlane(e0, pos) : __builtin_amdgcn_readlane(e1, pos - WAVE); };
        int el = at(lo), em = at(mid), eh = at(hi);
        if ((em >> 8) < (el >> 8)) { const int t = el; el = em; em = t; }
        if ((eh >> 8) < (em >> 8)) { const int t = eh; eh = em; em = t; }
        if ((em >> 8) < (el >> 8)) { const int t = el; el = em; em = t; }
        const int pk = em >> 8;
        e0 = p0 == lo ? el : (p0 == mid ? eh : (p0 == hi ? em : e0));   // pivot stashed at `hi`
        e1 = p1 == lo ? el : (p1 == mid ? eh : (p1 == hi ? em : e1));
        // ---- stops of the two scans over lo .. hi-1 ----
        const bool in0 = p0 >= lo && p0 < hi, in1 = p1 >= lo && p1 < hi;
        const u64 GE0 = __ballot(in0 && (e0 >> 8) >= pk), GE1 = __ballot(in1 && (e1 >> 8) >= pk);
        const u64 LE0 = __ballot(in0 && (e0 >> 8) <= pk), LE1 = __ballot(in1 && (e1 >> 8) <= pk);
        const bool ge0 = (GE0 >> lane) & 1ull, ge1 = (GE1 >> lane) & 1ull, le0 = (LE0 >> lane) & 1ull, le1 = (LE1 >> lane) & 1ull;
        const int nJ = __popcll(LE0) + __popcll(LE1);
        const int mi0 = __popcll(GE0 & below), mi1 = __popcll(GE0) + __popcll(GE1 & below);        // i-stops strictly below
        const int lb0 = __popcll(LE0 & below), lb1 = __popcll(LE0) + __popcll(LE1 & below);        // j-stops strictly below
        const int a_0 = mi0 + (ge0 ? 1 : 0), b_0 = nJ - lb0 - (le0 ? 1 : 0);
        const int a_1 = mi1 + (ge1 ? 1 : 0), b_1 = nJ - lb1 - (le1 ? 1 : 0);
        const u64 C0 = __ballot(in0 && a_0 >= b_0), C1 = __ballot(in1 && a_1 >= b_1);            // never both empty: b(hi-1) = 0
        const int px = ZT_U(C0 ? __ffsll((long long)C0) - 1 : WAVE + __ffsll((long long)C1) - 1);
        // i-stops below px, j-stops above px
        const u64 bl0 = px < WAVE ? ((1ull << px) - 1ull) : ~0ull, bl1 = px < WAVE ? 0ull : ((1ull << (px - WAVE)) - 1ull);
        const u64 at0 = px < WAVE ? (1ull << px) : 0ull, at1 = px < WAVE ? 0ull : (1ull << (px - WAVE));
        const int a_prev = __popcll(GE0 & bl0) + __popcll(GE1 & bl1);
        const int b_at = __popcll(LE0 & ~bl0 & ~at0) + __popcll(LE1 & ~bl1 & ~at1);
        const int Sw = ZT_U(a_prev > b_at ? a_prev : b_at);                                        // swaps of this partition
        // the swaps reverse the sequence of elements at the Sw lowest i-stops and the Sw highest j-stops
        const int mj0 = nJ - 1 - lb0, mj1 = nJ - 1 - lb1;                                          // rank among the j-stops, from the top
        const bool swi0 = ge0 && mi0 < Sw, swj0 = le0 && mj0 < Sw, swi1 = ge1 && mi1 < Sw, swj1 = le1 && mj1 < Sw;
        const int t0 = swi0 ? mi0 : 2 * Sw - 1 - mj0, t1 = swi1 ? mi1 : 2 * Sw - 1 - mj1;
        if (swi0 || swj0) stage[2 * Sw - 1 - t0] = e0;
        if (swi1 || swj1) stage[2 * Sw - 1 - t1] = e1;
        wave_sync();
        if (swi0 || swj0) e0 = stage[t0];
        if (swi1 || swj1) e1 = stage[t1];
        // where the i-scan ends: the first i-stop that did not swap, or the lowest swapped j-stop, or `hi`
        const u64 st0 = GE0 & ~__ballot(swi0), st1 = GE1 & ~__ballot(swi1), sj0 = __ballot(swj0), sj1 = __ballot(swj1);
        int ifin = st0 ? __ffsll((long long)st0) - 1 : (st1 ? WAVE + __ffsll((long long)st1) - 1 : hi);
        const int jl = sj0 ? __ffsll((long long)sj0) - 1 : (sj1 ? WAVE + __ffsll((long long)sj1) - 1 : 1 << 30);
        ifin = ZT_U(jl < ifin ? jl : ifin);
        wave_sync();                                            // the stage is rewritten by the next partition
        const int x = at(ifin);
        e0 = p0 == ifin ? em : e0; e1 = p1 == ifin ? em : e1;   // pivot <-> ifin
        e0 = p0 == hi ? x : e0; e1 = p1 == hi ? x : e1;
        if (ifin >= drop) finish(ifin, ifin);
        // ---- children: only those reaching the kept ranks matter ----
        const int lhi = ifin - 1, rlo = ifin + 1;
        const bool lneed = lhi >= lo && lhi >= drop, rneed = hi >= rlo && hi >= drop;
        const bool lpart = lneed && lhi - lo >= 15, rpart = rneed && hi - rlo >= 15;
        if (lneed && !lpart) finish(lo, lhi);
        if (rneed && !rpart) finish(rlo, hi);
        if (lpart) {
            if (rpart) {                                        // push the right child
                pend_hi = (pend_hi << 16) | (pend_lo >> 16);
                pend_lo = (pend_lo << 16) | (unsigned)(rlo | (hi << 8));
                ++depth;
            }
            hi = lhi;
        } else if (rpart) {
            lo = rlo;
        } else if (depth > 0) {
            lo = (int)(pend_lo & 0xffu); hi = (int)((pend_lo >> 8) & 0xffu);
            pend_lo = (pend_lo >> 16) | (pend_hi << 16);
            pend_hi >>= 16;
            --depth;
        } else {
            work = false;
        }
    }
    // ---- stable rank inside every finished segment that reaches the cut (insertion sort with strict <) ----
    if (has0) E[p0] = (e0 & ~0xff) | p0;                        // (rank, current position): p < 128 fits the low byte
    if (has1) E[p1] = (e1 & ~0xff) | p1;
    wave_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = h ? p1 : p0, e = h ? e1 : e0;
        if (p >= n) continue;
        const int slo = S.seg_lo[p], shi = S.seg_hi[p];
        if (shi < slo) continue;
        const int cp = (e & ~0xff) | p;
        int rank = 0;
        for (int q0 = slo; q0 <= shi; q0 += 8) {                // segments hold at most 15 entries
            int x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = E[(q0 + t) <= shi ? (q0 + t) : shi];
#pragma unroll
            for (int t = 0; t < 8; ++t) rank += ((q0 + t) <= shi && x[t] < cp) ? 1 : 0;
        }
        const int f = slo + rank;
        if (f >= drop) sel[f - drop] = e & 0xff;
    }
    wave_sync();
#undef ZT_U
    return 5;
}

// Wave-cooperative top-k selection.  On return sel[0..k) holds the indices of
// np.argsort(a)[-k:] in that order.  Requires n > k.  `a` and `sel` are LDS
// arrays owned by this wave.  Returns the path taken: 0 = rank fast path (no
// tie reaches the kept ranks), 4 = quicksort replay on ranks in registers
// (n <= 64), 1 = exact wave-parallel in LDS, 2 = exact sequential.
// (numba_argsort_reg, path 3, is the same replay on the values themselves; it
// is kept as an independently tested cross-check.)
__device__ inline int topk_select_wave(const double *a, int n, int k, int *sel, SortLds &S, int *seq_perm,
                                       int *seq_stk)
{
    const int lane = lane_id();
    const int drop = n - k;
    bool slow = false, has_nan = false;
    if (n <= WAVE) {
        // one lane per candidate: ranks by counting; the quicksort replay only when ties decide
        const double v = lane < n ? a[lane] : 0.0;
        if (__ballot(v != v) == 0ull) {
            int lt;
            if (topk_rank_reg(v, n, k, sel, &lt)) return 0;
            topk_ties_reg(lt, n, k, sel, S);
            return 4;
        }
    }
    if (n > WAVE && n <= 2 * WAVE) {
        // two candidates per lane (the pruning strategy's lists)
        const double w0 = lane < n ? a[lane] : 0.0, w1 = lane + WAVE < n ? a[lane + WAVE] : 0.0;
        if (__ballot(w0 != w0 || w1 != w1) == 0ull) return topk_select_wave128(a, n, k, sel, S);
    }
    // rank counting: every candidate c (strided over lanes) counts smaller /
    // equal values; all lanes read a[q] at the same address (LDS broadcast).
    for (int c = lane; c < ((n + WAVE - 1) / WAVE) * WAVE; c += WAVE) {
        int lt = 0, eq = 0;
        const bool live = c < n;
        const double v = live ? a[c] : 0.0;
        const bool nan = live && (v != v);
        for (int q0 = 0; q0 < n; q0 += 8) {          // 8 broadcast reads in flight per step
            double x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = a[(q0 + t) < n ? (q0 + t) : (n - 1)];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = (q0 + t) < n;
                lt += (in && x[t] < v) ? 1 : 0;
                eq += (in && x[t] == v) ? 1 : 0;
            }
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

// ---------------------------------------------------------------------------
// The wave-parallel replay for MORE than 128 candidates (round 6: kept sets wider than a wavefront, tppr_wide.hpp -- up to
// 2 x 255 + 1 candidates).  partition_wave / numba_argsort_wave above hold two positions per lane; here SLOTS of them (the
// same partition: stop lists from ballots, the m-th i-stop swapped with the m-th j-stop while it lies to its left), over an
// LDS block of its own size.  Kept apart from the 128-candidate form, which sits on k_stream's and k_pruned_topk's hot
// paths with its LDS footprint.
// ---------------------------------------------------------------------------
template <int SLOTS>
struct SortLdsN {
    static constexpr int CAP = WAVE * SLOTS;
    int r[CAP], r2[CAP];
    int ilist[CAP + 2], jlist[CAP + 2];
    double v[CAP];
    short seg_lo[CAP], seg_hi[CAP];
    int stk[96];
};

template <int SLOTS>
__device__ inline int partition_wave_n(const double *a, SortLdsN<SLOTS> &S, int low, int high)
{
    const int lane = lane_id();
    const int mid = (low + high) >> 1;
    int rl = S.r[low], rm = S.r[mid], rh = S.r[high];
    double al = a[rl], am = a[rm], ah = a[rh];
    if (lt_f(am, al)) { int t = rl; rl = rm; rm = t; double x = al; al = am; am = x; }
    if (lt_f(ah, am)) { int t = rh; rh = rm; rm = t; double x = ah; ah = am; am = x; }
    if (lt_f(am, al)) { int t = rl; rl = rm; rm = t; double x = al; al = am; am = x; }
    const double pivot = am;
    wave_sync();
    if (lane == 0) { S.r[low] = rl; S.r[mid] = rh; S.r[high] = rm; }   // pivot stashed at `high`
    wave_sync();
    const u64 below = lanemask_lt();
    const u64 above = ~below & ~(1ull << lane);
    u64 GE[SLOTS], LE[SLOTS];
    bool ge[SLOTS], le[SLOTS];
    const int nslot = (high - low + WAVE - 1) / WAVE;     // slots this segment reaches into (wave-uniform): the others are skipped
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        GE[q] = 0ull; LE[q] = 0ull; ge[q] = false; le[q] = false;
        if (q >= nslot) continue;
        const int p = low + lane + q * WAVE;
        const bool in = p <= high - 1;
        const double v = in ? a[S.r[p]] : 0.0;
        ge[q] = in && !lt_f(v, pivot);
        le[q] = in && !lt_f(pivot, v);
        GE[q] = __ballot(ge[q]);
        LE[q] = __ballot(le[q]);
    }
    int nI = 0, nJ = 0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {                     // i-stops ascending: slot by slot, lanes upwards
        if (q >= nslot) continue;
        if (ge[q]) S.ilist[nI + __popcll(GE[q] & below)] = low + lane + q * WAVE;
        nI += __popcll(GE[q]);
    }
#pragma unroll
    for (int q = SLOTS - 1; q >= 0; --q) {                // j-stops descending: the last slot first, lanes downwards
        if (q >= nslot) continue;
        if (le[q]) S.jlist[nJ + __popcll(LE[q] & above)] = low + lane + q * WAVE;
        nJ += __popcll(LE[q]);
    }
    if (lane == 0) { S.ilist[nI] = high; S.jlist[nJ] = low - 1; }
    wave_sync();
    const int npair = (nI < nJ ? nI : nJ);                // sentinels never swap
    int M = 0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        if (q * WAVE >= npair) continue;                  // (wave-uniform)
        const int m = lane + q * WAVE;
        bool sw = false;
        int pi = 0, pj = 0;
        if (m < npair) { pi = S.ilist[m]; pj = S.jlist[m]; sw = pi < pj; }
        if (sw) { const int x = S.r[pi], y = S.r[pj]; S.r[pi] = y; S.r[pj] = x; }
        M += __popcll(__ballot(sw));
    }
    int ifin = S.ilist[M];
    if (M > 0) { const int jl = S.jlist[M - 1]; ifin = jl < ifin ? jl : ifin; }
    ifin = __builtin_amdgcn_readfirstlane(ifin);
    wave_sync();
    if (lane == 0) { const int t = S.r[ifin]; S.r[ifin] = S.r[high]; S.r[high] = t; }
    wave_sync();
    return ifin;
}

// Full argsort for 2 <= n <= 64 SLOTS, no NaN.  Result in S.r2[0..n).  (numba_argsort_wave's body over the wider block.)
template <int SLOTS>
__device__ inline void numba_argsort_wave_n(const double *a, int n, SortLdsN<SLOTS> &S)
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
            const int i = partition_wave_n<SLOTS>(a, S, low, high);
            if (lane == 0) { S.seg_lo[i] = (short)i; S.seg_hi[i] = (short)i; }   // the pivot is in its final place
            if (high - i > i - low) {
                if (high > i) { if (lane == 0) { S.stk[2 * sp] = i + 1; S.stk[2 * sp + 1] = high; } ++sp; }
                high = i - 1;
            } else {
                if (i > low) { if (lane == 0) { S.stk[2 * sp] = low; S.stk[2 * sp + 1] = i - 1; } ++sp; }
                low = i + 1;
            }
        }
        for (int p = low + lane; p <= high; p += WAVE) { S.seg_lo[p] = (short)low; S.seg_hi[p] = (short)high; }   // (<= 15 entries)
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

// np.argsort(values)[-k:] in numba's order for ANY k (kept sets wider than a wavefront: tppr_wide.hpp, the pruning query with
// k > ZT_MAX_K) and any n: ranks by counting where no tie reaches the kept ranks, else the wave-parallel replay while the
// candidates fit the sort block (128 for SortLds, 64 SLOTS for SortLdsN), one lane's literal replay beyond (and for NaN).
// a, sel (>= k), S, seq_perm (>= n), seq_stk (>= 96): LDS of this wave.
__device__ __forceinline__ int sort_cap(const SortLds &) { return 128; }
template <int SLOTS> __device__ __forceinline__ int sort_cap(const SortLdsN<SLOTS> &) { return WAVE * SLOTS; }
__device__ __forceinline__ void sort_wave(const double *a, int n, SortLds &S) { numba_argsort_wave(a, n, S); }
template <int SLOTS> __device__ __forceinline__ void sort_wave(const double *a, int n, SortLdsN<SLOTS> &S) { numba_argsort_wave_n<SLOTS>(a, n, S); }

template <class SL>
__device__ inline void topk_select_any(const double *a, int n, int k, int *sel, SL &S, int *seq_perm, int *seq_stk)
{
    const int lane = lane_id();
    const int drop = n - k;
    bool slow = false, has_nan = false;
    for (int c = lane; c < ((n + WAVE - 1) / WAVE) * WAVE; c += WAVE) {
        int lt = 0, eq = 0;
        const bool live = c < n;
        const double v = live ? a[c] : 0.0;
        const bool nan = live && (v != v);
        for (int q0 = 0; q0 < n; q0 += 8) {          // 8 broadcast reads in flight per step
            double x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = a[(q0 + t) < n ? (q0 + t) : (n - 1)];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = (q0 + t) < n;
                lt += (in && x[t] < v) ? 1 : 0;
                eq += (in && x[t] == v) ? 1 : 0;
            }
        }
        eq -= 1;                                     // itself
        const bool bad = live && (nan || (eq > 0 && lt + eq >= drop));
        if (__ballot(bad) != 0ull) slow = true;
        if (__ballot(nan) != 0ull) has_nan = true;
        if (!slow && live && !nan && lt >= drop) sel[lt - drop] = c;
    }
    wave_sync();
    if (!slow) return;
    if (n <= sort_cap(S) && !has_nan) {
        sort_wave(a, n, S);
        for (int q = lane; q < k; q += WAVE) sel[q] = S.r2[drop + q];
        wave_sync();
        return;
    }
    if (lane == 0) {
        numba_argsort_seq(a, n, seq_perm, seq_stk);
        for (int q = 0; q < k; ++q) sel[q] = seq_perm[drop + q];
    }
    wave_sync();
}

}  // namespace zt
