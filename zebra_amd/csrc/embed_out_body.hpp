// The output layers of the embedding (k_embed_out: fc2 on the reduced rows, transform_source on memory[nodes]) as a device
// function: aggregate.hip launches it as a kernel of its own, memory_update.hip beside the GRU update in ONE launch
// (k_out_gru: the two are independent, each is bound by one workgroup's latency, and a launch boundary between them costs
// more than either's arithmetic at C2-C4's batch sizes).
#pragma once

#include "common.hpp"

namespace {

using zt::fastdiv;
using zt::fastdiv_magic;

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

}  // namespace
