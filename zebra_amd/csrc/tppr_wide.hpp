// Streaming T-PPR for dictionaries WIDER than a wavefront: ZT_MAX_K < k <= ZT_MAX_K_WIDE (the reference's --topk is
// unbounded, train.py:46).  k_stream (tppr_stream.hip) is built on "lane j owns entry j of a row" -- registers, the hub
// chains' mailboxes, the six-stage network of a hop all follow from it -- and stops at k = 63.  This file is the path
// beyond: CORRECT FIRST, not tuned.  One wavefront per T-PPR model applies the launch's edges one after the other, exactly
// as the reference's loop does (utils/util.py:473-576): no prepass, no tags to poll, no chains -- a wave that works alone
// needs none of them.  The state is the SAME row of granules (tppr_state.hpp: header + six entry-major arrays of k), so
// reset, copy, export / import, checkpoints and snapshots are the code every other k uses; tags are written as 0.
//   per edge (u, v, fake), per model:  the rows of u, v and the negative sample into LDS (all requested together: one memory
//   round trip up to k = 128, two beyond), emit the three rows as they are (:504-506),
//   build t_s1_PPR for (u <- v) and (v <- u) from the OLD rows (:509-547: scale, merge by key in dictionary order, the new
//   key last), keep the top k in numba's argsort order when there are more (:555-564; numba_sort.hpp), store, move the norms.
// Cost with full rows and both models of C5's (alpha, beta) side by side (tools/exp/wide_k_cost.py, profiles/r6/experiments/
// wide_k_cost.log): 80 / 154 / 591 us per edge at k = 64 / 100 / 255 -- the numba-order selection over 2k + 1 candidates is
// most of it (a rank pass of n reads per candidate, then, where ties reach the kept set, the wave-parallel quicksort replay
// over up to 512 positions: numba_sort.hpp, SortLdsN) -- against 5 us at k = 63 on the same stream.  A wide k is a quality knob
// of the reference, not the configuration its throughput is quoted on (BASELINE.json: k = 20 / 40).
#pragma once

#include "tppr_hop.hpp"

namespace {

constexpr int WIDE_K = ZT_MAX_K_WIDE;            // 255
constexpr int WIDE_ROW = 256;                    // entries of one old row in LDS
constexpr int WIDE_CAND = 2 * WIDE_ROW;          // candidates of one update: k + k + 1 <= 511

struct WideLds {
    u64 rk[3][WIDE_ROW];                         // the old rows of u (side 0), v (side 1) and the negative sample (side 2)
    double rt[3][WIDE_ROW], rw[3][WIDE_ROW];
    u64 ck[WIDE_CAND];                           // t_s1_PPR in dictionary order
    double ct[WIDE_CAND], cw[WIDE_CAND];
    int sel[WIDE_ROW];                           // the kept candidates in output order
    int perm[WIDE_CAND];                         // scratch of the sequential argsort replay
    int stk[96];
    SortLdsN<WIDE_CAND / WAVE> sort;             // the wave-parallel argsort replay over all 512 candidate positions
    int len[3];
    double norm[3];
};

// ids of a launch: the conditions of the prepass's k_count (tppr_prepass.hip: d_count) -- every node id in [0, N), every
// edge id in [0, 2^31) -- or the launch is rejected as a whole (ZT_ERR_RANGE: state untouched, empty output rows)
__global__ void k_wide_check(const int *__restrict__ nodes, const long long *__restrict__ eidx, long long role_stride, int B,
                             int n_roles, long long N, int *ctl, int *latch)
{
    for (long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x; a < (long long)B * n_roles;
         a += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(a / B), i = (int)(a % B);
        const int x = nodes[(long long)r * role_stride + i];
        bool ok = x >= 0 && x < N;
        if (r == 0) { const long long e = eidx[i]; ok = ok && e >= 0 && e <= 0x7fffffffll; }
        if (!ok) { atomicExch(&ctl[2], ZT_ERR_RANGE); latch_failure(latch, ZT_ERR_RANGE); }
    }
}
__global__ void k_wide_begin(int *ctl) { if (threadIdx.x == 0) { ctl[2] = 0; ctl[13] = 0; } }

__device__ __forceinline__ u64 wide_lo(u64 g) { return (u64)(unsigned)g; }

// The rows of an edge's nodes into LDS in as few memory round trips as the wave's load queue allows: the headers and the
// entries of ALL sides are requested together, 128 entries per side and trip, whatever the rows' lengths (entries beyond a
// row's length are never looked at).  n_sides = 2: u and v; 3: the negative sample as well.
__device__ inline void wide_load(u64 *const *base, int n_sides, int k, int lane, WideLds &L)
{
    u64 h[3] = {0ull, 0ull, 0ull};
#pragma unroll
    for (int sd = 0; sd < 3; ++sd)
        if (sd < n_sides && lane < 3) h[sd] = ld_agent(base[sd] + lane);
    for (int j0 = lane; j0 < k; j0 += 2 * WAVE) {
        u64 g[3][2][6];
#pragma unroll
        for (int sd = 0; sd < 3; ++sd)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = j0 + t * WAVE;
#pragma unroll
                for (int c = 0; c < 6; ++c) g[sd][t][c] = (sd < n_sides && j < k) ? ld_agent(base[sd] + HDR + j + c * k) : 0ull;
            }
#pragma unroll
        for (int sd = 0; sd < 3; ++sd)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = j0 + t * WAVE;
                if (sd < n_sides && j < k) {
                    L.rk[sd][j] = (wide_lo(g[sd][t][1]) << 32) | wide_lo(g[sd][t][0]);
                    L.rt[sd][j] = __longlong_as_double((long long)((wide_lo(g[sd][t][3]) << 32) | wide_lo(g[sd][t][2])));
                    L.rw[sd][j] = __longlong_as_double((long long)((wide_lo(g[sd][t][5]) << 32) | wide_lo(g[sd][t][4])));
                }
            }
    }
#pragma unroll
    for (int sd = 0; sd < 3; ++sd) {
        const unsigned h0 = (unsigned)__shfl((unsigned)h[sd], 0), h1 = (unsigned)__shfl((unsigned)h[sd], 1), h2 = (unsigned)__shfl((unsigned)h[sd], 2);
        if (sd < n_sides && lane == 0) { L.len[sd] = (int)h0; L.norm[sd] = __longlong_as_double((long long)(((u64)h2 << 32) | h1)); }
    }
}

// extract_streaming_tppr (utils/util.py:447-469) for one row, from its copy in LDS
__device__ inline void wide_emit(const WideLds &L, int side, int k, int lane, double tnow, int *on, int *oe, float *od, float *ow)
{
    const int len = L.len[side];
    for (int j = lane; j < k; j += WAVE) {
        const bool a = j < len;
        const u64 key = a ? L.rk[side][j] : 0ull;
        on[j] = (int)(unsigned)(key & 0xffffffffull);
        oe[j] = (int)(unsigned)(key >> 32);
        ow[j] = a ? (float)L.rw[side][j] : 0.f;
        const float tsf = a ? (float)L.rt[side][j] : 0.f;      // tmp_timestamps is float32
        od[j] = len == 0 ? 0.f : (float)(tnow - (double)tsf);  // (an empty dictionary leaves the zero row: :450)
    }
}

// One wavefront per emitted model; edges in order.
__global__ __launch_bounds__(WAVE) void k_stream_wide(zt_tppr h, StreamArgs A)
{
    __shared__ WideLds L;
    const int lane = threadIdx.x, k = h.k, mo = blockIdx.x, m = A.m_lo + mo;
    const long long rg = h.rg;
    const long long ob = (long long)mo * A.out_rows * k;
    if (ld_agent(h.ctl + 2) == ZT_ERR_RANGE) {          // rejected by k_wide_check: the output rows read as empty dictionaries
        if (A.emit)
            for (long long q = lane; q < (long long)A.n_roles * A.B * k; q += WAVE) {
                const long long c = q % k, row = q / k, i = row % A.B, role = row / A.B;
                const long long o = ob + (role * A.role_stride + i) * k + c;
                A.out_nodes[o] = 0; A.out_eidx[o] = 0; A.out_dt[o] = 0.f; A.out_w[o] = 0.f;
            }
        return;
    }
    const double alpha = h.alpha[m], beta = h.beta[m];
    u64 *rows = h.rows + (long long)m * h.N * rg;
    for (int i = 0; i < A.B; ++i) {
        const long long u = A.nodes[i], v = A.nodes[A.role_stride + i];
        const double tnow = A.tsv[i];
        const u64 eid = (u64)A.eidx[i];
        u64 *bu = rows + u * rg, *bv = rows + v * rg;
        // ---- the rows of u, v (and the negative sample) as they are BEFORE this edge: one or two memory round trips ----
        const bool fake = A.emit && A.n_roles == 3;
        u64 *bases[3] = {bu, bv, fake ? rows + (long long)A.nodes[2 * A.role_stride + i] * rg : bu};
        wide_load(bases, fake ? 3 : 2, k, lane, L);
        wave_sync();
        // ---- the three output rows (:504-506) ----
        if (A.emit) {
            const long long ou = ob + (long long)i * k, ov = ob + (A.role_stride + i) * k, og = ob + (2 * A.role_stride + i) * k;
            wide_emit(L, 0, k, lane, tnow, A.out_nodes + ou, A.out_eidx + ou, A.out_dt + ou, A.out_w + ou);
            wide_emit(L, 1, k, lane, tnow, A.out_nodes + ov, A.out_eidx + ov, A.out_dt + ov, A.out_w + ov);
            if (fake) wide_emit(L, 2, k, lane, tnow, A.out_nodes + og, A.out_eidx + og, A.out_dt + og, A.out_w + og);
        }
        // ---- both updates read the OLD rows (PPR_list is replaced only after both, :566-574): side 0 = u, side 1 = v ----
        const int n_pairs = u != v ? 2 : 1;
        for (int p = 0; p < n_pairs; ++p) {
            const int s1 = p, s2 = u != v ? 1 - p : 0;
            const long long x2 = p == 0 ? v : u;                       // s2's node id: the new key's node
            const int n1 = L.len[s1], n2 = L.len[s2];
            const double norm1 = L.norm[s1], norm2 = L.norm[s2];
            int n = 0;
            double scale_s2;
            if (norm1 == 0.0) {                                        // :514-519
                scale_s2 = 1.0 - alpha;
            } else {                                                   // :520-526
                const double new_norm = norm1 * beta + beta;
                const double scale_s1 = norm1 / new_norm * beta;
                scale_s2 = beta / new_norm * (1.0 - alpha);
                for (int j = lane; j < n1; j += WAVE) { L.ck[j] = L.rk[s1][j]; L.ct[j] = L.rt[s1][j]; L.cw[j] = L.rw[s1][j] * scale_s1; }
                n = n1;
            }
            wave_sync();
            const int n_own = n;                                       // entries that came from s1: the only ones an s2 key can meet
            if (norm2 != 0.0) {                                        // :532-538, s2's entries in dictionary order
                for (int c0 = 0; c0 < n2; c0 += WAVE) {
                    const int j2 = c0 + lane;
                    const bool live = j2 < n2;
                    const u64 key2 = live ? L.rk[s2][j2] : 0ull;
                    int mi = -1;
                    for (int q = 0; q < n_own; ++q) if (L.ck[q] == key2) mi = q;
                    const double add = live ? L.rw[s2][j2] * scale_s2 : 0.0;
                    const bool fresh = live && mi < 0;
                    if (live && mi >= 0) L.cw[mi] = L.cw[mi] + add;    // (s2's keys are distinct: nobody else touches entry mi)
                    const u64 fm = __ballot(fresh);
                    if (fresh) {
                        const int pos = n + __popcll(fm & ((1ull << lane) - 1ull));
                        L.ck[pos] = key2; L.ct[pos] = L.rt[s2][j2]; L.cw[pos] = add;
                    }
                    n += __popcll(fm);
                }
                wave_sync();
            }
            {   // the new key (edge_idx, s2, timestamp), last (:529, :540-541); an existing key keeps its place
                const u64 nk = (eid << 32) | (u64)(unsigned)x2;
                const double nw = alpha != 0.0 ? scale_s2 * alpha : scale_s2;
                int at = -1;
                for (int c0 = 0; c0 < n; c0 += WAVE) {
                    const u64 hit = __ballot(c0 + lane < n && L.ck[c0 + lane] == nk);
                    if (hit != 0ull) at = c0 + __ffsll((long long)hit) - 1;
                }
                if (lane == 0) {
                    if (at >= 0) L.cw[at] = nw;
                    else { L.ck[n] = nk; L.ct[n] = tnow; L.cw[n] = nw; }
                }
                if (at < 0) n += 1;
                wave_sync();
            }
            // ---- keep the top k (:549-564) and store the row; the norm moves by norm * beta + beta (:567-574) ----
            const int n_new = n <= k ? n : k;
            if (n > k) topk_select_any(L.cw, n, k, L.sel, L.sort, L.perm, L.stk);
            u64 *dst = p == 0 ? bu : bv;
            for (int j = lane; j < n_new; j += WAVE) {
                const int c = n > k ? L.sel[j] : j;
                const u64 kk = L.ck[c];
                const u64 tt = (u64)__double_as_longlong(L.ct[c]), ww = (u64)__double_as_longlong(L.cw[c]);
                u64 *e = dst + HDR + j;
                st_agent(e, wide_lo(kk));
                st_agent(e + k, kk >> 32);
                st_agent(e + 2 * k, wide_lo(tt));
                st_agent(e + 3 * k, tt >> 32);
                st_agent(e + 4 * k, wide_lo(ww));
                st_agent(e + 5 * k, ww >> 32);
            }
            if (lane < 3) {
                const u64 nn = (u64)__double_as_longlong(norm1 * beta + beta);
                st_agent(dst + lane, lane == 0 ? (u64)(unsigned)n_new : (lane == 1 ? wide_lo(nn) : nn >> 32));
            }
            wave_sync();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the next edge may read the rows just written
    }
}

}  // namespace
