// fc1 + ReLU + T-PPR-weighted k-reduction for WIDE edge features (F = 172: Wikipedia / Reddit, BASELINE configs
// C1-C3) over the projected memory table -- the persistent counterpart of k_fc1_agg_reg (aggregate.hip, F <= 4).
//
// Replaces, like the other aggregate kernels, the neighbour half of
// GraphDiffusionEmbedding.compute_embedding_tppr_ensemble (reference modules/embedding_module.py:250-276, transform
// :320-328, TimeEncode model/time_encoding.py:23-28): for every gathered neighbour row
//   h = relu(W_m memory[nbr] + W_e efeat[eidx] + W_t cos(dt w) + b1),  reduced over k with weights w / sum(w).
//
// Shape of the problem on this chip.  The contraction runs over F + T = 272 columns: 68 k-steps x 7 N-tiles of
// v_mfma_f32_16x16x4_f32 per 16 gathered rows = 476 B operands per lane -- they do not fit one wave's registers (the
// F <= 4 kernel keeps its 182 in AGPRs).  So the [272 x 112] fc1 block lives in LDS for the kernel's life (119 KB of
// the CU's 160 KB, stored in exactly the order the MFMAs consume it: one conflict-free ds_read_b128 feeds four MFMAs),
// and everything else follows k_fc1_agg_reg: one workgroup of four waves per CU, persistent, ONE barrier (after the
// weight fill); every wave loops by itself over M-tiles of 16 gathered rows -- no LDS tile, no staging:
//   * the edge-feature part of the A operand comes straight from memory into the lanes that feed it to the MFMA:
//     lane (row r, k-slot g) loads float4 at columns 16 j + 4 g (a 64-byte segment per row and instruction); the
//     k-steps are permuted accordingly (step 4 j + c contracts columns 16 j + 4 g + c), which the packed weights
//     mirror.  A tile's features are requested ONE TILE AHEAD, each register right after the MFMAs that consumed its
//     previous value; the per-row scalars (ids, dt, w) two tiles ahead;
//   * the time encoding is evaluated in the operand lanes (time_cosf_rev, 6 instructions + v_cos_f32);
//   * the accumulators start from the bias, the projected rows P[nbr] (requested at the top of the tile) are added in
//     the epilogue, then ReLU and the normalised weight.
// The unit of work is an M-tile, not a group of query rows: C2 has 1500 M-tiles for 896 waves, so 80-row tiles (four
// query rows, as in the F <= 4 kernel) would leave two thirds of the chip idle.  A query row's k = 20 rows then
// straddle M-tiles, so the kernel writes the partial sums of every GROUP OF FOUR rows (one 16-lane group of the MFMA's
// output layout: no cross-lane traffic at all), G[m][n][k/4][D], and k_embed_out adds a query row's k / 4 groups first
// to last -- the association k_fc1_agg_reg uses, independent of where a row falls in a tile or a shard.
#include "common.hpp"

using namespace zt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WIDE_THREADS = 256, WIDE_WAVES = 4;
constexpr int WD = 100, WT = 100, WDP = 112, WNB = 7;       // hidden width, time width, padded width, N-tiles
constexpr int WF = 172;                                     // edge-feature width
constexpr int EF_V4 = 10;                                   // float4 loads per lane: columns 0..159, permuted k order
constexpr int EF_TAIL = 3;                                  // columns 160..171 in natural order: 3 k-steps
constexpr int EF_STEPS = 4 * EF_V4 + EF_TAIL;               // 43
constexpr int COS_STEPS = WT / 4;                           // 25
constexpr int NSTEP = EF_STEPS + COS_STEPS;                 // 68
constexpr int NGRP = NSTEP / 4;                             // 17 groups of four k-steps: one ds_read_b128 per N-tile each
constexpr int WLDS_FLOATS = NGRP * WNB * 64 * 4;            // 30464 floats = 121 856 bytes
constexpr int WLDS_PAD = (WLDS_FLOATS / 4 + WIDE_THREADS - 1) / WIDE_THREADS * WIDE_THREADS * 4;   // whole rounds of the fill: 30720 floats
// behind the image, in the order a lane reads them with 16-byte loads: the time frequencies of k-slot g4 ([4][28], entry s =
// time_w[4 s + g4]) and the biases of column r16 ([16][8], entry b = b1[16 b + r16])
constexpr int WFREQ_OFF = WLDS_PAD, WBIAS_OFF = WFREQ_OFF + 4 * 28, WPACK_FLOATS = WBIAS_OFF + 16 * 8;
static_assert(NSTEP % 4 == 0 && 16 * EF_V4 + 4 * EF_TAIL == WF, "k-step layout");

// column of [ef | time] that k-step t contracts in the lanes of k-slot g4
__host__ __device__ inline int wide_krow(int t, int g4)
{
    if (t < 4 * EF_V4) return 16 * (t >> 2) + 4 * g4 + (t & 3);
    if (t < EF_STEPS) return 16 * EF_V4 + 4 * (t - 4 * EF_V4) + g4;
    return WF + 4 * (t - EF_STEPS) + g4;
}

// fc1_w [D][D + F + T] (torch layout) -> the LDS image [NGRP][WNB][64 lanes][4 steps]: the B operand of N-tile b,
// k-step 4 u + c for lane (r16 = column, g4 = k-slot)
__global__ void k_pack_wide(const float *__restrict__ fc1_w, const float *__restrict__ time_w, const float *__restrict__ b1,
                            float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= WPACK_FLOATS) return;
    if (i >= WBIAS_OFF) { const int r = (i - WBIAS_OFF) >> 3, b = (i - WBIAS_OFF) & 7; out[i] = (b < WNB && 16 * b + r < WD) ? b1[16 * b + r] : 0.f; return; }
    if (i >= WFREQ_OFF) { const int g = (i - WFREQ_OFF) / 28, sidx = (i - WFREQ_OFF) % 28; out[i] = sidx < COS_STEPS ? time_w[4 * sidx + g] : 0.f; return; }
    if (i >= WLDS_FLOATS) { out[i] = 0.f; return; }
    const int c = i & 3, lane = (i >> 2) & 63, rest = i >> 8, b = rest % WNB, u = rest / WNB;
    const int n = 16 * b + (lane & 15), kr = wide_krow(4 * u + c, lane >> 4);
    out[i] = n < WD ? fc1_w[(size_t)n * (WD + WF + WT) + WD + kr] : 0.f;
}

#ifdef ZT_WIDE_STAMP
// diagnostic build only (tools/exp/wide_bench.py): per wave: shader cycles spent at the top of the tiles, in the MFMA
// groups, in the epilogues; tiles; wall-clock start / end (100 MHz); cycles before the first tile
__device__ unsigned long long g_wide[1024 * 8];
__device__ unsigned long long g_wide2[1024 * 8];
#define PSTAMP(i) st_p[i] = __builtin_amdgcn_s_memtime() - st_c0
#define WSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define WSTAMP(var) do { } while (0)
#define PSTAMP(i) do { } while (0)
#endif

struct Scal {            // per-row scalars of one M-tile: lanes 0..15 hold row l; wq: this lane's share of the weights of
    int nb, ei;          // the query row its wave HALF sums (lanes 0..31: the first query row the tile touches, 32..63: the
    float dt, w, wq[2];  // second; entry l32 and, for k > 32, entry l32 + 32)
};

template <int KK>
__global__ __launch_bounds__(WIDE_THREADS, 1) void k_fc1_agg_wide(
    const float *__restrict__ P, const float *__restrict__ efeat, const float *__restrict__ time_w, long long num_nodes,
    long long num_edges, long long N, int M, const int *__restrict__ nbr, const int *__restrict__ eix,
    const float *__restrict__ dt, const float *__restrict__ w, const float *__restrict__ Wl, const float *__restrict__ b1,
    float *__restrict__ G, float *__restrict__ S, int *status, const int *gate_word, int gate_target, int *gate_latch)
{
    static_assert(KK % 4 == 0 && KK >= 16 && KK <= 64, "groups of 4 rows must not straddle query rows; a tile touches <= 2 query rows");
    constexpr int HG = KK / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef ZT_WIDE_STAMP
    const unsigned long long st_wall0 = __builtin_amdgcn_s_memrealtime(), st_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_top = 0, st_mfma = 0, st_epi = 0, st_tiles = 0, st_pro = 0, st_p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    float *wl = reinterpret_cast<float *>(smem);                           // [NGRP][WNB][64][4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g4 = lane >> 4, half = lane >> 5, l32 = lane & 31;

    // 32-bit tile arithmetic (the launcher refuses N k M >= 2^31): a 64-bit division is ~100 instructions, and a tile
    // needs half a dozen
    const int rpm = (int)N * KK;                                            // gathered rows per model
    const int tpm = (rpm + 15) / 16;                                        // M-tiles per model
    const int n_tiles = tpm * M;
    const int stride = (int)gridDim.x * WIDE_WAVES;
    int tile = (int)blockIdx.x * WIDE_WAVES + wave;

    // Loads only, no control flow: whoever tests a value here (or loads under a branch, whose join needs the value)
    // waits for the loads it has just issued.  Indices are clamped to the model's rows; `settle` zeroes what was not there.
    auto fetch = [&](int t, Scal &sc) {
        const int tc = t < n_tiles ? t : 0;
        const int m = tc / tpm, rb = (tc - m * tpm) * 16;
        const size_t mb = (size_t)m * rpm;
        const int r = rb + r16;
        const size_t i = mb + (size_t)(r < rpm ? r : rpm - 1);
        sc.nb = nbr[i]; sc.ei = eix[i]; sc.dt = dt[i]; sc.w = w[i];
        const int qrow = (rb / KK + half) * KK;                             // first row of this half's query row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            sc.wq[h] = 0.f;
            if (32 * h < KK) {                                              // (compile time)
                const int e = qrow + l32 + 32 * h;
                sc.wq[h] = w[mb + (size_t)(e < rpm ? e : rpm - 1)];
            }
        }
    };
    // ... and once they have arrived: rows and entries beyond the model's (or the launch's) end read as zero, ids are
    // range-checked
    auto settle = [&](int t, Scal &sc) {
        const bool live = t < n_tiles;
        const int tc = live ? t : 0;
        const int m = tc / tpm, rb = (tc - m * tpm) * 16;
        if (!live || rb + r16 >= rpm) { sc.nb = 0; sc.ei = 0; sc.dt = 0.f; sc.w = 0.f; }
        const int qrow = (rb / KK + half) * KK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (!live || l32 + 32 * h >= KK || qrow + l32 + 32 * h >= rpm) sc.wq[h] = 0.f;
        if (sc.nb < 0 || sc.nb >= num_nodes || sc.ei < 0 || sc.ei >= num_edges) { atomicExch(status, ZT_ERR_RANGE); sc.nb = 0; sc.ei = 0; }
    };
    // Everything a tile needs from its scalars, in registers, without LDS: sum(w) of the two query rows by a fixed
    // reduction tree inside each wave half (rotations inside the rows of 16 lanes, lane 0's association; then row 0 +
    // row 1) -- a query row's sum is the same function of its k weights wherever the row falls --, w / sum(w) (0 where
    // the sum is 0: modules/embedding_module.py:267-270), and the MFMA-lane views of dt, w / sum(w) and the neighbour ids.
    float dta = 0.f;
    f32x4 wn = {0.f, 0.f, 0.f, 0.f};
    int nb4[4] = {0, 0, 0, 0};
    float *s_ptr = nullptr;                                                 // the S flag this lane owes (stored with the held results)
    float s_val = 0.f;
    auto prep = [&](const Scal &sc, int t) {
        const int m = t / tpm, rb = (t - m * tpm) * 16, qa = rb / KK;
        float sm = sc.wq[0] + sc.wq[1];
        sm += dpp_f<0x128>(sm);                                             // row_ror:8, 4, 2, 1
        sm += dpp_f<0x124>(sm);
        sm += dpp_f<0x122>(sm);
        sm += dpp_f<0x121>(sm);
        auto at = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), l)); };
        const float sum_a = at(0) + at(16), sum_b = at(32) + at(48);
        // the tile in which a query row STARTS writes its flag (lane 0: the first query row, lane 32: the second)
        s_ptr = nullptr;
        if (l32 == 0 && t < n_tiles) {
            const int q = qa + half, qrow = q * KK;
            if (q < N && qrow >= rb && qrow < rb + 16) { s_ptr = S + (size_t)m * N + q; s_val = ((half ? sum_b : sum_a) == 0.f) ? 0.f : 1.f; }
        }
        const float qs = ((rb + r16) / KK - qa) ? sum_b : sum_a;            // lanes 0..15: row l's query row
        const float wnv = (qs == 0.f) ? 0.f : sc.w / qs;
        dta = __shfl(sc.dt, r16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { wn[j] = __shfl(wnv, 4 * g4 + j); nb4[j] = __shfl(sc.nb, 4 * g4 + j); }
    };

    // ---- once per launch: the weights into LDS, frequencies, biases.  ONE memory round trip for all of it: every load
    // is issued before the first value is looked at -- 30 x 16 bytes of weights per thread in registers, frequencies and
    // biases packed for 16-byte loads (a wave has 63 loads in flight at most: 25 + 7 single ones on top of the weights
    // and the scalars made it two round trips), the weights first (their addresses need no arithmetic) ----
    constexpr int NFILL = WLDS_PAD / 4 / WIDE_THREADS;                      // 30 (the image is padded to whole rounds)
    f32x4 fill[NFILL];
#pragma unroll
    for (int q = 0; q < NFILL; ++q) fill[q] = reinterpret_cast<const f32x4 *>(Wl)[threadIdx.x + q * WIDE_THREADS];
    float freq[28], bias[8];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(Wl + WFREQ_OFF + 28 * g4 + 4 * q);
        freq[4 * q] = v[0]; freq[4 * q + 1] = v[1]; freq[4 * q + 2] = v[2]; freq[4 * q + 3] = v[3];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(Wl + WBIAS_OFF + 8 * r16 + 4 * q);
        bias[4 * q] = v[0]; bias[4 * q + 1] = v[1]; bias[4 * q + 2] = v[2]; bias[4 * q + 3] = v[3];
    }
    // (the T-PPR rows of this batch may still be on their way -- a launch group released batch by batch, common.hpp:
    //  member_gate --: the wait sits behind the requests for the weights, in front of the first read of a row)
    member_gate_enter(gate_word, gate_target, status, gate_latch);
    Scal cur, nxt;
    fetch(tile, cur);
    fetch(tile + stride, nxt);
    PSTAMP(0);
    PSTAMP(1);
    // this lane's share of a tile's edge features: row r16, columns 16 j + 4 g4 .. + 3 and 160 + 4 s + g4
    f32x4 efv[EF_V4];
    float eft[EF_TAIL];
    const float *ef_row;
    auto ef_base = [&](const Scal &sc) { ef_row = efeat + (size_t)__shfl(sc.ei, r16) * WF + 4 * g4; };
    settle(tile, cur);
    PSTAMP(2);
    ef_base(cur);
#pragma unroll
    for (int j = 0; j < EF_V4; ++j) efv[j] = *reinterpret_cast<const f32x4 *>(ef_row + 16 * j);
#pragma unroll
    for (int s = 0; s < EF_TAIL; ++s) eft[s] = ef_row[16 * EF_V4 + 4 * s - 3 * g4];
#pragma unroll
    for (int q = 0; q < NFILL; ++q) reinterpret_cast<f32x4 *>(wl)[threadIdx.x + q * WIDE_THREADS] = fill[q];
    PSTAMP(3);
    __syncthreads();                                                        // the only workgroup barrier
    PSTAMP(4);
    prep(cur, tile);
    PSTAMP(5);
    settle(tile + stride, nxt);
    ef_base(nxt);
    fetch(tile + 2 * stride, cur);          // roles from here on: one set = the NEXT tile's scalars, the other = in flight
    // results of a tile are stored during the NEXT one (after its first wait for loads): on this chip a wait for a load
    // with a store in flight is a wait for everything (loads and stores complete out of order with each other), and at
    // the top of a tile that was a microsecond of store latency
    float hv[WNB];
    float *hp = nullptr;
#pragma unroll
    for (int b = 0; b < WNB; ++b) hv[b] = 0.f;

#ifdef ZT_WIDE_STAMP
    st_pro = __builtin_amdgcn_s_memtime() - st_c0;
#endif
    // One tile.  On entry A = the scalars of the NEXT tile (arrived, validated; ef_row points at its features), B = those
    // of the tile after it (in flight).  The two sets swap roles from tile to tile -- the loop below calls the body twice
    // -- because a register copy of a set whose loads are in flight is a wait for them.
    auto body = [&](Scal &A, Scal &B) __attribute__((always_inline)) {
        WSTAMP(t_a);
        const int m = tile / tpm, tt = tile - m * tpm, rb = tt * 16;
        const bool has_next = tile + stride < n_tiles;
        float Pv[WNB][4];
        {
            const float *p0 = P + (size_t)nb4[0] * WDP + r16, *p1 = P + (size_t)nb4[1] * WDP + r16,
                        *p2 = P + (size_t)nb4[2] * WDP + r16, *p3 = P + (size_t)nb4[3] * WDP + r16;
#pragma unroll
            for (int b = 0; b < WNB; ++b) { Pv[b][0] = p0[16 * b]; Pv[b][1] = p1[16 * b]; Pv[b][2] = p2[16 * b]; Pv[b][3] = p3[16 * b]; }
        }
        f32x4 acc[WNB];
#pragma unroll
        for (int b = 0; b < WNB; ++b) acc[b] = f32x4{bias[b], bias[b], bias[b], bias[b]};
        f32x4 Bf[2][WNB];
#pragma unroll
        for (int b = 0; b < WNB; ++b) Bf[0][b] = *reinterpret_cast<const f32x4 *>(wl + ((0 * WNB + b) * 64 + lane) * 4);
        float cs[4] = {0.f, 0.f, 0.f, 0.f};                                 // cosines of the group after this one
        WSTAMP(t_b);
#pragma unroll
        for (int u = 0; u < NGRP; ++u) {
            if (u + 1 < NGRP) {
#pragma unroll
                for (int b = 0; b < WNB; ++b)
                    Bf[(u + 1) & 1][b] = *reinterpret_cast<const f32x4 *>(wl + (((u + 1) * WNB + b) * 64 + lane) * 4);
            }
            float a[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * u + c;
                if (t < 4 * EF_V4) a[c] = efv[u][c];
                else if (t < EF_STEPS) a[c] = eft[t - 4 * EF_V4];
                else a[c] = cs[c];
            }
            // the cosines of group u + 1 (vector instructions between this group's MFMAs)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * (u + 1) + c;
                if (t >= EF_STEPS && t < NSTEP) cs[c] = time_cosf_rev(dta * freq[t - EF_STEPS]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int b = 0; b < WNB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], Bf[u & 1][b][c], acc[b], 0, 0, 0);
            if (u == 0) {
                // the PREVIOUS tile's results and this tile's S flags: the edge features were waited for, nothing else
                // is waited for until the epilogue
                if (hp != nullptr) {
#pragma unroll
                    for (int b = 0; b < WNB; ++b)
                        if (16 * b + r16 < WD) hp[16 * b + r16] = hv[b];
                }
                if (s_ptr != nullptr) *s_ptr = s_val;
            }
            // the NEXT tile's edge features into the registers this group has just consumed
            if (has_next) {
                if (u < EF_V4) efv[u] = *reinterpret_cast<const f32x4 *>(ef_row + 16 * u);
                if (u == EF_V4) {
#pragma unroll
                    for (int s = 0; s < EF_TAIL; ++s) eft[s] = ef_row[16 * EF_V4 + 4 * s - 3 * g4];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        WSTAMP(t_c);
        // ---- epilogue: + projected row, ReLU, x w / sum(w); the lane's 4 rows are ONE group of a query row ----
        const int grow = rb + 4 * g4;                                      // first row of this lane's group
        hp = grow < rpm ? G + ((size_t)m * N * HG + (size_t)(grow / 4)) * WD : nullptr;
#pragma unroll
        for (int b = 0; b < WNB; ++b) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[b][j] + Pv[b][j];
                v = v > 0.f ? v : 0.f;
                part += v * wn[j];
            }
            hv[b] = part;
        }
        // ---- the next tile: its scalars arrived a tile ago ----
        if (has_next) prep(A, tile + stride);
        settle(tile + 2 * stride, B);
        ef_base(B);
        fetch(tile + 3 * stride, A);
        tile += stride;
#ifdef ZT_WIDE_STAMP
        { const unsigned long long t_d = __builtin_amdgcn_s_memtime(); st_top += t_b - t_a; st_mfma += t_c - t_b; st_epi += t_d - t_c; ++st_tiles; }
#endif
    };
#pragma nounroll
    while (tile < n_tiles) {
        body(nxt, cur);
        if (tile >= n_tiles) break;
        body(cur, nxt);
    }
    if (hp != nullptr) {
#pragma unroll
        for (int b = 0; b < WNB; ++b)
            if (16 * b + r16 < WD) hp[16 * b + r16] = hv[b];
    }
#ifdef ZT_WIDE_STAMP
    if (lane == 0 && blockIdx.x * WIDE_WAVES + wave < 1024) {
        unsigned long long *o = g_wide + 8 * (blockIdx.x * WIDE_WAVES + wave);
        o[0] = st_top; o[1] = st_mfma; o[2] = st_epi; o[3] = st_tiles; o[4] = st_wall0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = st_pro; o[7] = __builtin_amdgcn_s_memtime() - st_c0;
        for (int q = 0; q < 8; ++q) g_wide2[8 * (blockIdx.x * WIDE_WAVES + wave) + q] = st_p[q];
    }
#endif
}

}  // namespace

#ifdef ZT_WIDE_STAMP
extern "C" int zt_debug_wide(unsigned long long *host)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wide), sizeof(unsigned long long) * 8 * 1024));
    ZT_HIP(hipMemcpyFromSymbol(host + 8 * 1024, HIP_SYMBOL(g_wide2), sizeof(unsigned long long) * 8 * 1024));
    return ZT_OK;
}
#endif

namespace zt {

bool fc1_agg_wide_supported(int D, int F, int T, int k)
{
    return D == WD && T == WT && F == WF && (k == 20 || k == 40);
}

size_t fc1_agg_wide_weight_bytes() { return (size_t)WPACK_FLOATS * 4; }

void fc1_agg_wide_pack(const float *fc1_w_dev, const float *time_w_dev, const float *fc1_b_dev, float *packed_dev, hipStream_t s)
{
    k_pack_wide<<<(WPACK_FLOATS + 255) / 256, 256, 0, s>>>(fc1_w_dev, time_w_dev, fc1_b_dev, packed_dev);
}

// H_groups: [M][N][k/4][D] partial sums (k_embed_out adds them); S: [M][N].  cus: compute units of the stream.
int fc1_agg_wide_launch(const float *P, const float *efeat, const float *time_w, long long num_nodes, long long num_edges,
                        long long N, int M, int k, const int *nbr, const int *eix, const float *dt, const float *w,
                        const float *packed, const float *b1, float *G, float *S, int *status, int cus, hipStream_t s,
                        const member_gate *gate)
{
    const int *gw = gate ? gate->word : nullptr;
    const int gt = gate ? gate->target : 0;
    int *gl = gate ? gate->latch : nullptr;
    const size_t lds = (size_t)WLDS_PAD * 4;
    static size_t attr[2] = {0, 0};
    const int ki = k == 20 ? 0 : 1;
    if (lds > attr[ki]) {
        const void *fn = k == 20 ? reinterpret_cast<const void *>(k_fc1_agg_wide<20>) : reinterpret_cast<const void *>(k_fc1_agg_wide<40>);
        ZT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[ki] = lds;
    }
    const long long tiles = ((N * k + 15) / 16) * M;
    if (N * k * M >= (1ll << 31)) { set_error("k_fc1_agg_wide: N k M = %lld does not fit the kernel's 32-bit tile arithmetic", N * k * M); return ZT_ERR_UNSUPPORTED; }
    long long wgs = (tiles + WIDE_WAVES - 1) / WIDE_WAVES;
    if (wgs > cus) wgs = cus;                                              // persistent: one workgroup (4 waves) per CU
    if (wgs < 1) wgs = 1;
    if (k == 20)
        k_fc1_agg_wide<20><<<(unsigned)wgs, WIDE_THREADS, lds, s>>>(P, efeat, time_w, num_nodes, num_edges, N, M, nbr, eix, dt, w,
                                                                    packed, b1, G, S, status, gw, gt, gl);
    else
        k_fc1_agg_wide<40><<<(unsigned)wgs, WIDE_THREADS, lds, s>>>(P, efeat, time_w, num_nodes, num_edges, N, M, nbr, eix, dt, w,
                                                                    packed, b1, G, S, status, gw, gt, gl);
    return ZT_OK;
}

}  // namespace zt
