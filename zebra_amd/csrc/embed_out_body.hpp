// The output layers of the embedding (k_embed_out: fc2 on the reduced rows, transform_source on memory[nodes]) as a device
// function: aggregate.hip launches it as a kernel of its own, memory_update.hip beside the GRU update in ONE launch
// (k_out_gru: the two are independent, each is bound by one workgroup's latency, and a launch boundary between them costs
// more than either's arithmetic at C2-C4's batch sizes).
#pragma once

#include "common.hpp"

namespace {

using zt::fastdiv;
using zt::fastdiv_magic;
using zt::wave_sync;

#ifndef ZT_F32X4_DEFINED
#define ZT_F32X4_DEFINED
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

constexpr int AGG_THREADS = 256;
constexpr int AGG_WAVES = 4;
constexpr int NTW = 2;             // N-tiles per wave -> D <= 128

// ---------------------------------------------------------------------------
// out[n] = [ fc2s(relu(fc1s(memory[nodes[n]]))) | fc2(H_0[n]) + b2*S_0[n] | ... ]
// Three small D x D layers on f32 MFMA.  One workgroup per 32 rows; the input
// rows sit in LDS, weights ([Dp][Dp], zero padded) stream from L2 as b128
// fragments; wave w owns output N-tiles {w, w+4}.
// ---------------------------------------------------------------------------
constexpr int OUT_ROWS = 32;
constexpr int OUT_MT = OUT_ROWS / 16;
constexpr int SG_CH = 8;             // k-steps of weight fragments in flight (small_gemm)
constexpr int EO_GU = 8;             // staged elements in flight per thread (k_embed_out)

// acc[a][b] = X[a-th 16 rows] * W[b-th owned N-tile]^T   (X in LDS [32][ldx], W padded [Dp][Dp])
__device__ __forceinline__ void small_gemm(const float *X, int ldx, const float *__restrict__ Wp, int Dp, int NT,
                                           int wave, int lane, f32x4 (&acc)[OUT_MT][NTW])
{
    const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int a = 0; a < OUT_MT; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The weight fragments of SG_CH k-steps are fetched together, before the first MFMA that needs one: a
    // workgroup has one tile of 32 rows, so nothing else hides the L2 round trip of a load issued per k-step.
    const int KC = Dp / 16;
    for (int kc0 = 0; kc0 < KC; kc0 += SG_CH) {
        f32x4 bv[SG_CH][NTW];
#pragma unroll
        for (int c = 0; c < SG_CH; ++c)
#pragma unroll
            for (int b = 0; b < NTW; ++b) {
                const int nt = wave + b * AGG_WAVES;
                bv[c][b] = (kc0 + c < KC && nt < NT)
                               ? *reinterpret_cast<const f32x4 *>(Wp + (size_t)(nt * 16 + r16) * Dp + 16 * (kc0 + c) + 4 * g4)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int c = 0; c < SG_CH; ++c) {
            if (kc0 + c >= KC) break;
            f32x4 av[OUT_MT];
#pragma unroll
            for (int a = 0; a < OUT_MT; ++a)
                av[a] = *reinterpret_cast<const f32x4 *>(X + (size_t)(a * 16 + r16) * ldx + 16 * (kc0 + c) + 4 * g4);
#pragma unroll
            for (int b = 0; b < NTW; ++b) {
                if (wave + b * AGG_WAVES >= NT) continue;
#pragma unroll
                for (int a = 0; a < OUT_MT; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], bv[c][b][j], acc[a][b], 0, 0, 0);
            }
        }
    }
}

// what the output layers need (by value: the fused kernel of memory_update.hip takes it beside the GRU's arguments)
struct EmbedOutArgs {
    const float *memory;
    long long num_nodes;
    const int *nodes;
    long long N;
    int D, M;
    const float *H, *S, *fc2_p, *fc2_b, *fc1s_p, *fc1s_b, *fc2s_p, *fc2s_b;
    float *out;
    int *status;
};

// bx = 32-row tile, path = 0 (source path) or 1 + model; the first AGG_THREADS threads of the workgroup.  src_read != nullptr:
// a source-path workgroup adds 1 there once its memory rows are in LDS (k_out_gru: the GRU half waits for all of them before
// it writes the table).
template <int HG>                    // partial-sum groups per query row in H (k_fc1_agg_wide: k / 4; else 1)
__device__ __forceinline__ void embed_out_body(const EmbedOutArgs &E, char *smem, int bx, int path, int *src_read)
{
    const float *__restrict__ memory = E.memory;
    const long long num_nodes = E.num_nodes, N = E.N;
    const int *__restrict__ nodes = E.nodes;
    const int D = E.D, M = E.M;
    const float *__restrict__ H = E.H, *__restrict__ S = E.S, *__restrict__ fc2_p = E.fc2_p, *__restrict__ fc2_b = E.fc2_b;
    const float *__restrict__ fc1s_p = E.fc1s_p, *__restrict__ fc1s_b = E.fc1s_b, *__restrict__ fc2s_p = E.fc2s_p, *__restrict__ fc2s_b = E.fc2s_b;
    float *__restrict__ out = E.out;
    int *status = E.status;
    const int Dp = (D + 15) / 16 * 16, ldx = Dp + 4, NT = Dp / 16;
    float *X = reinterpret_cast<float *>(smem);          // [32][ldx] layer input
    float *Y = X + OUT_ROWS * ldx;                       // [32][ldx] hidden rows of the source path
    int *rid = reinterpret_cast<int *>(Y + OUT_ROWS * ldx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const long long r0 = (long long)bx * OUT_ROWS;
    const int nr = (int)((N - r0) < OUT_ROWS ? (N - r0) : OUT_ROWS);
    const int OW = D * (M + 1);
    const unsigned mL = fastdiv_magic((unsigned)ldx);
    f32x4 acc[OUT_MT][NTW];

    // blockIdx.y = 0: the source path (two dependent layers); 1 + m: fc2 of model m.  The paths are independent,
    // and the kernel's time is the latency of ONE workgroup (there are fewer workgroups than the chip holds).
    if (path == 0) {
        if (tid < OUT_ROWS) {
            int nd = 0;
            if (tid < nr) {
                nd = nodes[r0 + tid];
                if (nd < 0 || nd >= num_nodes) { atomicExch(status, ZT_ERR_RANGE); nd = 0; }
            }
            rid[tid] = nd;
        }
        __syncthreads();
        // ---- source path: memory[nodes] -> fc1s -> relu -> fc2s ----
        // (EO_GU loads in flight per thread before the first LDS store: a load per iteration was a memory round trip
        //  per iteration, 15 of them in a row)
        for (int f0 = tid; f0 < OUT_ROWS * ldx; f0 += AGG_THREADS * EO_GU) {
            float v[EO_GU];
#pragma unroll
            for (int u = 0; u < EO_GU; ++u) {
                const int f = f0 + u * AGG_THREADS, g = fastdiv(f, mL), c = f - g * ldx;
                v[u] = (f < OUT_ROWS * ldx && g < nr && c < D) ? memory[(size_t)rid[g] * D + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < EO_GU; ++u)
                if (f0 + u * AGG_THREADS < OUT_ROWS * ldx) X[f0 + u * AGG_THREADS] = v[u];
        }
        __syncthreads();
        if (src_read != nullptr && tid == 0) atomicAdd(src_read, 1);      // (the rows are in LDS: every load has returned)
        small_gemm(X, ldx, fc1s_p, Dp, NT, wave, lane, acc);
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            const int col = (wave + b * AGG_WAVES) * 16 + r16;
            if (col >= Dp) continue;
            const float bias = col < D ? fc1s_b[col] : 0.f;
#pragma unroll
            for (int a = 0; a < OUT_MT; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[a][b][j] + bias;
                    Y[(size_t)(a * 16 + g4 * 4 + j) * ldx + col] = (col < D && v > 0.f) ? v : 0.f;
                }
        }
        __syncthreads();
        small_gemm(Y, ldx, fc2s_p, Dp, NT, wave, lane, acc);
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            const int col = (wave + b * AGG_WAVES) * 16 + r16;
            if (col >= D) continue;
            const float bias = fc2s_b[col];
#pragma unroll
            for (int a = 0; a < OUT_MT; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int g = a * 16 + g4 * 4 + j;
                    if (g < nr) out[(size_t)(r0 + g) * OW + col] = acc[a][b][j] + bias;
                }
        }
        return;
    }
    // ---- neighbour path: fc2 on the reduced rows of model m ----
    {
        const int m = path - 1;
        for (int f0 = tid; f0 < OUT_ROWS * ldx; f0 += AGG_THREADS * EO_GU) {
            float v[EO_GU];
#pragma unroll
            for (int u = 0; u < EO_GU; ++u) {
                const int f = f0 + u * AGG_THREADS, g = fastdiv(f, mL), c = f - g * ldx;
                v[u] = 0.f;
                if (f < OUT_ROWS * ldx && g < nr && c < D) {
                    // hg > 1 (k_fc1_agg_wide): H holds the partial sums of every group of four neighbour rows; a query row's
                    // groups are added first to last, whichever tile or shard computed them
                    const float *hp = H + (((size_t)m * N + r0 + g) * HG) * D + c;
                    v[u] = hp[0];
#pragma unroll
                    for (int q = 1; q < HG; ++q) v[u] += hp[(size_t)q * D];
                }
            }
#pragma unroll
            for (int u = 0; u < EO_GU; ++u)
                if (f0 + u * AGG_THREADS < OUT_ROWS * ldx) X[f0 + u * AGG_THREADS] = v[u];
        }
        __syncthreads();
        small_gemm(X, ldx, fc2_p, Dp, NT, wave, lane, acc);
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            const int col = (wave + b * AGG_WAVES) * 16 + r16;
            if (col >= D) continue;
            const float bias = fc2_b[col];
#pragma unroll
            for (int a = 0; a < OUT_MT; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int g = a * 16 + g4 * 4 + j;
                    if (g < nr)
                        out[(size_t)(r0 + g) * OW + (size_t)D * (m + 1) + col] =
                            acc[a][b][j] + bias * S[(size_t)m * N + r0 + g];
                }
        }
    }
}

template <int HG>
__global__ __launch_bounds__(AGG_THREADS) void k_embed_out(EmbedOutArgs E)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_out_body<HG>(E, smem, blockIdx.x, blockIdx.y, nullptr);
}

// ---------------------------------------------------------------------------
// k_embed_out2: the same three layers, organised for LATENCY (round 4).  k_embed_out's time was never its arithmetic:
// at C2's batch (600 rows) it ran 27 us for 40 MFLOP -- a chain of dependent memory round trips per workgroup (ids ->
// row staging through LDS in a loop the compiler cannot unroll, one round trip per iteration -> weight fragments per
// k-chunk -> second layer -> ...) on 57 workgroups.  Here one WAVE owns (16 rows, path, one N-tile of the output):
//   * its weight fragments -- fc2's [16 x Dp] slice, for the source path also ALL of fc1s -- are requested first,
//     before anything is looked at, and stay in registers while the wave strides over row tiles;
//   * the A operand comes straight from memory into the MFMA lanes (lane (row r, k-slot g) loads float4 at columns
//     16 c + 4 g -- the k order the padded weights already have), no LDS staging; partial-sum groups (k_fc1_agg_wide)
//     are added first to last as they arrive;
//   * the source path's hidden layer is computed by each of the NT waves of a row tile (0.4 MFLOP, redundant on purpose:
//     sharing it would cost a barrier and a round trip through LDS or memory) and turned from the MFMA's output layout
//     into its input layout through 7 KB of the wave's own LDS.
// One memory round trip for the neighbour paths, two for the source path (ids, then rows).  D % 4 == 0.
// ---------------------------------------------------------------------------
// ONE WAVE's work: (bx of gx) = the tiles it strides over, path, b = its N-tile; Y = 16 x (16 NT + 4) floats of LDS of its own.
// src_read != nullptr: a source-path wave adds 1 there once its last memory rows have arrived (k_out_gru2).
template <int NT, int HG>
__device__ __forceinline__ void embed_out2_body(const EmbedOutArgs &E, float *Y, int lane, int bx, int gx, int path, int b, int *src_read)
{
    const float *__restrict__ memory = E.memory;
    const long long num_nodes = E.num_nodes, N = E.N;
    const int *__restrict__ nodes = E.nodes;
    const int D = E.D, M = E.M;
    const float *__restrict__ H = E.H, *__restrict__ S = E.S, *__restrict__ fc2_p = E.fc2_p, *__restrict__ fc2_b = E.fc2_b;
    const float *__restrict__ fc1s_p = E.fc1s_p, *__restrict__ fc1s_b = E.fc1s_b, *__restrict__ fc2s_p = E.fc2s_p, *__restrict__ fc2s_b = E.fc2s_b;
    float *__restrict__ out = E.out;
    int *status = E.status;
    constexpr int Dp = NT * 16, KC = NT, ldy = Dp + 4;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int col = 16 * b + r16, OW = D * (M + 1);
    const long long tiles = (N + 15) / 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // columns this lane's float4 of chunk c covers: all four inside the row, or none (D % 4 == 0)
    bool cin[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) cin[c] = 16 * c + 4 * g4 < D;

    if (path == 0) {
        f32x4 w1[NT][KC], w2[KC];
#pragma unroll
        for (int bb = 0; bb < NT; ++bb)
#pragma unroll
            for (int c = 0; c < KC; ++c) w1[bb][c] = *reinterpret_cast<const f32x4 *>(fc1s_p + (size_t)(16 * bb + r16) * Dp + 16 * c + 4 * g4);
#pragma unroll
        for (int c = 0; c < KC; ++c) w2[c] = *reinterpret_cast<const f32x4 *>(fc2s_p + (size_t)col * Dp + 16 * c + 4 * g4);
        float b1v[NT];
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) b1v[bb] = 16 * bb + r16 < D ? fc1s_b[16 * bb + r16] : 0.f;
        const float b2v = col < D ? fc2s_b[col] : 0.f;
        long long t = bx;
        int nd = (t < tiles && t * 16 + r16 < N) ? nodes[t * 16 + r16] : 0;
        for (; t < tiles; t += gx) {
            const long long r0 = t * 16;
            if (nd < 0 || nd >= num_nodes) { atomicExch(status, ZT_ERR_RANGE); nd = 0; }
            const bool rin = r0 + r16 < N;
            f32x4 a[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c)
                a[c] = (rin && cin[c]) ? *reinterpret_cast<const f32x4 *>(memory + (size_t)nd * D + 16 * c + 4 * g4) : zero4;
            const long long tn = t + gx;
            nd = (tn < tiles && tn * 16 + r16 < N) ? nodes[tn * 16 + r16] : 0;      // the next tile's ids, a tile ahead
            f32x4 acc1[NT];
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) acc1[bb] = zero4;
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int bb = 0; bb < NT; ++bb)
                        acc1[bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], w1[bb][c][j], acc1[bb], 0, 0, 0);
#pragma unroll
            for (int bb = 0; bb < NT; ++bb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc1[bb][j] + b1v[bb];
                    Y[(4 * g4 + j) * ldy + 16 * bb + r16] = (16 * bb + r16 < D && v > 0.f) ? v : 0.f;
                }
            wave_sync();
            f32x4 acc = zero4;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const f32x4 y = *reinterpret_cast<const f32x4 *>(Y + r16 * ldy + 16 * c + 4 * g4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[j], w2[c][j], acc, 0, 0, 0);
            }
            if (col < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0 + 4 * g4 + j < N) out[(size_t)(r0 + 4 * g4 + j) * OW + col] = acc[j] + b2v;
            }
            wave_sync();                                                   // Y is free for the next tile
        }
        // (every row this wave reads has been consumed by an MFMA: the loads have returned)
        if (src_read != nullptr && lane == 0) atomicAdd(src_read, 1);
        return;
    }
    // ---- neighbour path: fc2 on the reduced rows of model m ----
    const int m = path - 1;
    f32x4 w2[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) w2[c] = *reinterpret_cast<const f32x4 *>(fc2_p + (size_t)col * Dp + 16 * c + 4 * g4);
    const float b2v = col < D ? fc2_b[col] : 0.f;
    for (long long t = bx; t < tiles; t += gx) {
        const long long r0 = t * 16;
        const bool rin = r0 + r16 < N;
        const float *hp = H + (((size_t)m * N + (rin ? r0 + r16 : 0)) * HG) * D + 4 * g4;
        f32x4 g[HG][KC];
#pragma unroll
        for (int q = 0; q < HG; ++q)
#pragma unroll
            for (int c = 0; c < KC; ++c) g[q][c] = (rin && cin[c]) ? *reinterpret_cast<const f32x4 *>(hp + (size_t)q * D + 16 * c) : zero4;
        float sv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sv[j] = r0 + 4 * g4 + j < N ? S[(size_t)m * N + r0 + 4 * g4 + j] : 0.f;
        f32x4 acc = zero4;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            f32x4 a = g[0][c];
#pragma unroll
            for (int q = 1; q < HG; ++q) a += g[q][c];                     // a query row's groups, first to last
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w2[c][j], acc, 0, 0, 0);
        }
        if (col < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r0 + 4 * g4 + j < N) out[(size_t)(r0 + 4 * g4 + j) * OW + (size_t)D * (m + 1) + col] = acc[j] + b2v * sv[j];
        }
    }
}

template <int NT, int HG>
__global__ __launch_bounds__(64) void k_embed_out2(EmbedOutArgs E)
{
    __shared__ __attribute__((aligned(16))) float Y[16 * (NT * 16 + 4)];
    embed_out2_body<NT, HG>(E, Y, threadIdx.x, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, nullptr);
}

}  // namespace
