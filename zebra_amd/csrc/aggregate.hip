// Top-k neighbour gather + TimeEncode + transform + T-PPR-weighted sum (P2).
//
// Replaces the device half of GraphDiffusionEmbedding.compute_embedding_tppr_ensemble
// (reference modules/embedding_module.py:243-276, transform/transform_source
// :320-328, TimeEncode model/time_encoding.py:23-28), eval forward.
//
// The reference materialises cat[memory[nbr] | edge_feat[eidx] | cos(dt*w)] as
// an [N,k,2D+F] tensor, runs two Linear layers over all N*k rows and then
// reduces over k.  Here:
//   k_fc1_agg   one workgroup per (tile of RQ query rows, model): gathers the
//               RQ*k neighbour rows straight into an LDS tile (the cat is never
//               written to HBM), evaluates the time encoding in-register,
//               runs fc1 on exact-f32 MFMA (v_mfma_f32_16x16x4_f32; weights
//               streamed from L2 as pre-padded fragments), applies bias+ReLU
//               and the normalised T-PPR weight, and reduces over k in LDS.
//               fc2 is linear, so sum_k w_k*fc2(h_k) = fc2(sum_k w_k*h_k) +
//               b2*sum_k w_k: only the [N,D] reduced rows go through fc2.
//   k_embed_out fc2 on the reduced rows, transform_source on memory[nodes]
//               (three D x D layers, also f32 MFMA) and the concat into out[N, D*(M+1)].
// Arithmetic is float32 throughout (parity tolerance for embeddings: 1e-4).
#include "common.hpp"
#include "embed_out_body.hpp"     // f32x4, AGG_THREADS / AGG_WAVES / NTW, small_gemm, the output layers' body (shared with memory_update.hip)

using namespace zt;

namespace {

constexpr int MAX_MT = 5;          // M-tiles (16 gathered rows each) per workgroup
constexpr int MAX_MT_BIG = 16;     // ... of the instantiation for 80 < k <= 255 (k_fc1_agg<TAB, MAX_MT_BIG>)
constexpr int LDS_BUDGET = 150 * 1024;

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

#ifdef ZT_AGG_STAMP
// diagnostic build only: shader-clock time per phase of k_fc1_agg, summed over workgroups
__device__ unsigned long long g_agg[8];
#define AGG_STAMP(i) do { if (threadIdx.x == 0) { const long long t__ = clock64(); atomicAdd(&g_agg[i], (unsigned long long)(t__ - t_prev__)); t_prev__ = t__; } } while (0)
#else
#define AGG_STAMP(i) do { } while (0)
#endif

// Zero-padded copy of the block W[0..rows)[c0..c0+cols) of a matrix with leading dimension ld -> Wp[rows_p][cols_p].
__global__ void k_pad_matrix(const float *__restrict__ W, int ld, int c0, int rows, int cols, float *__restrict__ Wp,
                             int rows_p, int cols_p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_p * cols_p) return;
    const int r = i / cols_p, c = i % cols_p;
    Wp[i] = (r < rows && c < cols) ? W[(size_t)r * ld + c0 + c] : 0.f;
}

// ---------------------------------------------------------------------------
// fc1 + ReLU + weighted k-reduction.
//   A tile in LDS: [mt*16][lda] floats, row g = q*k + j holds
//   [memory[nbr] (D) | efeat[eidx] (F) | cos(dt*w_t) (T) | 0-pad].
//   W1p: [Dp][K1p] zero padded (Dp = NT*16, K1p = round_up(K1,16)).
//   H:   [M][N][D] reduced hidden rows;  S: [M][N] = 1 if sum(w) != 0 else 0.
// ---------------------------------------------------------------------------
// TAB = true: the memory columns are not gathered and not multiplied.  fc1 is linear, so
//   fc1([mem | ef | cos]) = W_m mem + W_e ef + W_t cos + b1, and W_m mem[v] is a per-NODE quantity: it is kept in
// a projected table P[v] = W_m memory[v] ([num_nodes][Dp], maintained by k_project_rows whenever a memory row
// changes: at most 2B rows per batch instead of 3B*k*M gathered rows).  The accumulators START from the
// gathered P rows, the tile holds only [ef | cos] and the contraction runs over F + T columns: for F = 1
// half the matrix work and none of the memory-row staging.  `memory` then points at P and D is the row stride Dp.
// MMT: M-tiles a workgroup can hold.  MAX_MT (5: 80 gathered rows) for every k <= 80; MAX_MT_BIG (16) is instantiated for
// 80 < k <= 255 -- a query row of that many neighbours must still fit ONE workgroup's tile (dictionaries wider than a
// wavefront: tppr_wide.hpp; correct first, not tuned).
template <bool TAB, int MMT = MAX_MT>
__global__ __launch_bounds__(AGG_THREADS) void k_fc1_agg(
    const float *__restrict__ memory, const float *__restrict__ efeat, const float *__restrict__ time_w,
    long long num_nodes, long long num_edges, int D, int F, int T, long long N, int k, int rq, int mt_count, int lda,
    const int *__restrict__ nbr, const int *__restrict__ eix, const float *__restrict__ dt,
    const float *__restrict__ w, const float *__restrict__ W1p, int K1p, const float *__restrict__ b1,
    float *__restrict__ H, float *__restrict__ S, int *status, int Dout, const int *__restrict__ row_map,
    const float *__restrict__ overlay, unsigned drop_lo = 0u, unsigned drop_hi = 0u, unsigned drop_thr = 0u,
    float drop_inv = 1.f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *A = reinterpret_cast<float *>(smem);                       // [mt*16][lda]
    float *wn = A + (size_t)mt_count * 16 * lda;                      // [mt*16] normalised weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = blockIdx.y;
    const long long q0 = (long long)blockIdx.x * rq;
    const int nq = (int)((N - q0) < rq ? (N - q0) : rq);
    const int rows = nq * k, rows_p = mt_count * 16;
    const int CM = TAB ? 0 : D;                                       // memory columns held in the tile
    const int K1 = CM + F + T;
    const size_t mb = ((size_t)m * N + q0) * k;                       // first entry of this tile in [M][N][k]
    // ---- this wave's N-tiles {wave, wave+4}; weight fragments stream from L2 one chunk ahead of their MFMAs
    // ---- (keeping all of them in registers was measured: no gain, and it halves the occupancy)
    const int NT = (Dout + 15) / 16;
    const int r16 = lane & 15, g4 = lane >> 4;
    const float *bp[NTW];
    bool live[NTW];
#pragma unroll
    for (int b = 0; b < NTW; ++b) {
        const int nt = wave + b * AGG_WAVES;
        live[b] = nt < NT;
        bp[b] = W1p + (size_t)((live[b] ? nt : 0) * 16 + r16) * K1p + 4 * g4;
    }
#ifdef ZT_AGG_STAMP
    long long t_prev__ = clock64();
#endif

    // per-row gather indices, staged once so that the row loads below are independent
    int *g_nb = reinterpret_cast<int *>(wn + rows_p);                 // [mt*16]
    int *g_ei = g_nb + rows_p;
    float *g_dt = reinterpret_cast<float *>(g_ei + rows_p);
    float *tw = g_dt + rows_p;                                        // [T] time-encoding frequencies

    // ---- stage the per-row scalars (coalesced), then normalise the weights in LDS ----
    for (int g = tid; g < rows_p; g += AGG_THREADS) {
        int nb = 0, ei = 0;
        float d = 0.f, wv = 0.f;
        if (g < rows) {
            nb = nbr[mb + g]; ei = eix[mb + g]; d = dt[mb + g]; wv = w[mb + g];
            if (nb < 0 || nb >= num_nodes || ei < 0 || ei >= num_edges) {
                atomicExch(status, ZT_ERR_RANGE);
                nb = 0; ei = 0;
            }
            // training: rows of the lazily updated memory live in a compact overlay (aggregate_bwd.hip)
            if (!TAB && row_map != nullptr) { const int ov = row_map[nb]; if (ov >= 0) nb = -ov - 1; }
        }
        g_nb[g] = nb; g_ei[g] = ei; g_dt[g] = d; wn[g] = wv;
    }
    for (int c = tid; c < T; c += AGG_THREADS) tw[c] = time_w[c];
    __syncthreads();
    // w / sum(w), 0 where the sum is 0 (:267-270); sum in entry order like torch.sum(dim=1)
    float my_sum = 0.f;
    if (tid < rows) {
        const int q = tid / k;
        for (int j = 0; j < k; ++j) my_sum += wn[q * k + j];
    }
    __syncthreads();
    if (tid < rows) {
        wn[tid] = (my_sum == 0.f) ? 0.f : wn[tid] / my_sum;
        if (tid % k == 0) S[(size_t)m * N + q0 + tid / k] = (my_sum == 0.f) ? 0.f : 1.f;
    }

    auto mem_row = [&](int s) { return s >= 0 ? memory + (size_t)s * D : overlay + (size_t)(-s - 1) * D; };
    AGG_STAMP(0);
    // accumulators: zero, or (TAB) the projected rows P[nbr[row]][col] -- issued before the staging below so that the
    // memory round trip is hidden behind it; the MFMAs then accumulate on top of them
    f32x4 acc[MMT][NTW];
#pragma unroll
    for (int a = 0; a < MMT; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TAB && a < mt_count && live[b]) {
                const int col = (wave + b * AGG_WAVES) * 16 + r16;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][b][j] = memory[(size_t)g_nb[a * 16 + g4 * 4 + j] * D + col];
            }
        }
    // ---- gather: flat (row, column) loops, GU loads in flight per thread before any LDS store
    // ---- (a row-at-a-time loop serialises on HBM latency: 20 rows x ~3 us per wave)
    constexpr int GU = 8;
    const unsigned mD = fastdiv_magic((unsigned)D), mF = fastdiv_magic((unsigned)(F > 0 ? F : 1)),
                   mT = fastdiv_magic((unsigned)(T > 0 ? T : 1));
    // rows are whole float4s, 16-byte aligned in HBM and in the tile
    const bool vecD = TAB || ((D & 3) == 0 && ((size_t)memory & 15) == 0);
    const bool vecF = vecD && F > 0 && (F & 3) == 0 && ((size_t)efeat & 15) == 0;
    // cos(dt * w_c) for every (row, frequency); reads only LDS.  A thread keeps ONE frequency in a register and
    // walks down a block of rows, four dt values per (broadcast) LDS read: no dependent LDS read and no index
    // arithmetic per element, and the four cosines of a step are independent instructions.
    auto time_encode = [&]() {
        if (T > AGG_THREADS || T <= 0) {                                // (not a shape the reference uses)
            for (int f = tid; f < rows_p * T; f += AGG_THREADS) {
                const int g = fastdiv(f, mT), c = f - g * T;
                A[g * lda + CM + F + c] = g < rows ? time_cosf(g_dt[g] * tw[c]) : 0.f;
            }
            return;
        }
        const int TC = T <= 64 ? 64 : (T <= 128 ? 128 : 256);          // threads per row block
        const int RG = AGG_THREADS / TC, per = rows_p / RG;             // rows_p is a multiple of 16
        const int c = tid & (TC - 1), g0 = (tid / TC) * per;
        if (c >= T) return;
        const float wc = tw[c];
        float *dst = A + (size_t)g0 * lda + CM + F + c;
        for (int i = 0; i < per; i += 4) {
            const f32x4 d4 = *reinterpret_cast<const f32x4 *>(g_dt + g0 + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                dst[(size_t)(i + u) * lda] = (g0 + i + u) < rows ? time_cosf(d4[u] * wc) : 0.f;     // cos(t*w + 0)
        }
    };
    if (vecD) {
        // One memory round trip for the whole gather, with the time encoding computed underneath it: the
        // first GU 16-byte loads of every thread (all of them for D = 100, k = 20) and the first batch
        // of edge-feature loads are issued, then the cosines are evaluated, then the loaded values are
        // stored to the tile.  (Memory latency under the gather load is ~3-4 us per dependent access.)
        const int D4 = D >> 2, F4 = F >> 2;
        const unsigned mD4 = fastdiv_magic((unsigned)D4), mF4 = fastdiv_magic((unsigned)(F4 > 0 ? F4 : 1));
        f32x4 vm[GU], vf[GU];
        float sf[GU];
        if (!TAB) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = tid + u * AGG_THREADS;
                const int g = fastdiv(f, mD4), c = f - g * D4;
                vm[u] = (f < rows_p * D4 && g < rows) ? *reinterpret_cast<const f32x4 *>(mem_row(g_nb[g]) + 4 * c)
                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (vecF) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = tid + u * AGG_THREADS;
                const int g = fastdiv(f, mF4), c = f - g * F4;
                vf[u] = (f < rows_p * F4 && g < rows)
                            ? *reinterpret_cast<const f32x4 *>(efeat + (size_t)g_ei[g] * F + 4 * c)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = tid + u * AGG_THREADS;
                const int g = fastdiv(f, mF), c = f - g * F;
                sf[u] = (f < rows_p * F && g < rows) ? efeat[(size_t)g_ei[g] * F + c] : 0.f;
            }
        }
        AGG_STAMP(1);
        time_encode();
        AGG_STAMP(2);
        if (!TAB) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = tid + u * AGG_THREADS;
                const int g = fastdiv(f, mD4), c = f - g * D4;
                if (f < rows_p * D4) *reinterpret_cast<f32x4 *>(A + (size_t)g * lda + 4 * c) = vm[u];
            }
        }
        if (vecF) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = tid + u * AGG_THREADS;
                const int g = fastdiv(f, mF4), c = f - g * F4;
                if (f < rows_p * F4) *reinterpret_cast<f32x4 *>(A + (size_t)g * lda + CM + 4 * c) = vf[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = tid + u * AGG_THREADS;
                const int g = fastdiv(f, mF), c = f - g * F;
                if (f < rows_p * F) A[(size_t)g * lda + CM + c] = sf[u];
            }
        }
        // whatever does not fit the first batch
        for (int f0 = tid + AGG_THREADS * GU; !TAB && f0 < rows_p * D4; f0 += AGG_THREADS * GU) {
            f32x4 v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = f0 + u * AGG_THREADS;
                const int g = fastdiv(f, mD4), c = f - g * D4;
                v[u] = (f < rows_p * D4 && g < rows)
                           ? *reinterpret_cast<const f32x4 *>(mem_row(g_nb[g]) + 4 * c)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = f0 + u * AGG_THREADS;
                const int g = fastdiv(f, mD4), c = f - g * D4;
                if (f < rows_p * D4) *reinterpret_cast<f32x4 *>(A + (size_t)g * lda + 4 * c) = v[u];
            }
        }
        if (vecF) {
            for (int f0 = tid + AGG_THREADS * GU; f0 < rows_p * F4; f0 += AGG_THREADS * GU) {
                f32x4 v[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF4), c = f - g * F4;
                    v[u] = (f < rows_p * F4 && g < rows)
                               ? *reinterpret_cast<const f32x4 *>(efeat + (size_t)g_ei[g] * F + 4 * c)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF4), c = f - g * F4;
                    if (f < rows_p * F4) *reinterpret_cast<f32x4 *>(A + (size_t)g * lda + CM + 4 * c) = v[u];
                }
            }
        } else {
            for (int f0 = tid + AGG_THREADS * GU; f0 < rows_p * F; f0 += AGG_THREADS * GU) {
                float v[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF), c = f - g * F;
                    v[u] = (f < rows_p * F && g < rows) ? efeat[(size_t)g_ei[g] * F + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF), c = f - g * F;
                    if (f < rows_p * F) A[(size_t)g * lda + CM + c] = v[u];
                }
            }
        }
    } else {
        for (int f0 = tid; f0 < rows_p * D; f0 += AGG_THREADS * GU) {
            float v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = f0 + u * AGG_THREADS;
                const int g = fastdiv(f, mD), c = f - g * D;
                v[u] = (f < rows_p * D && g < rows) ? mem_row(g_nb[g])[c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = f0 + u * AGG_THREADS;
                const int g = fastdiv(f, mD), c = f - g * D;
                if (f < rows_p * D) A[(size_t)g * lda + c] = v[u];
            }
        }
        AGG_STAMP(1);
        for (int f0 = tid; f0 < rows_p * F; f0 += AGG_THREADS * GU) {
            float v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = f0 + u * AGG_THREADS;
                const int g = fastdiv(f, mF), c = f - g * F;
                v[u] = (f < rows_p * F && g < rows) ? efeat[(size_t)g_ei[g] * F + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int f = f0 + u * AGG_THREADS;
                const int g = fastdiv(f, mF), c = f - g * F;
                if (f < rows_p * F) A[(size_t)g * lda + CM + c] = v[u];
            }
        }
        AGG_STAMP(2);
        time_encode();
    }
    const int padw = K1p - K1;
    const unsigned mP = fastdiv_magic((unsigned)(padw > 0 ? padw : 1));
    for (int f = tid; f < rows_p * padw; f += AGG_THREADS) {
        const int g = fastdiv(f, mP), c = f - g * padw;
        A[(size_t)g * lda + K1 + c] = 0.f;
    }
    AGG_STAMP(3);
    __syncthreads();
    AGG_STAMP(4);

    // ---- fc1 on f32 MFMA: wave handles N-tiles {wave, wave+4}, all M-tiles ----
    {
        const int nchunk = K1p / 16;
        f32x4 bcur[NTW], bnext[NTW];
#pragma unroll
        for (int b = 0; b < NTW; ++b) bcur[b] = *reinterpret_cast<const f32x4 *>(bp[b]);
        for (int kc = 0; kc < nchunk; ++kc) {
            if (kc + 1 < nchunk) {
#pragma unroll
                for (int b = 0; b < NTW; ++b) bnext[b] = *reinterpret_cast<const f32x4 *>(bp[b] + 16 * (kc + 1));
            }
            f32x4 av[MMT];
#pragma unroll
            for (int a = 0; a < MMT; ++a)
                av[a] = a < mt_count
                            ? *reinterpret_cast<const f32x4 *>(A + (size_t)(a * 16 + r16) * lda + 16 * kc + 4 * g4)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < MMT; ++a)
#pragma unroll
                    for (int b = 0; b < NTW; ++b)
                        if (a < mt_count && live[b])
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], bcur[b][j], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NTW; ++b) bcur[b] = bnext[b];
        }
    }
    AGG_STAMP(5);
    __syncthreads();   // every wave is done reading the A tile: reuse it for the hidden rows

    // ---- bias + ReLU + weight, staged as Hs[g][col] in the A region ----
    float *Hs = A;                               // [rows_p][ldh]
    const int ldh = NT * 16 + 1;
#pragma unroll
    for (int b = 0; b < NTW; ++b) {
        if (!live[b]) continue;
        const int col = (wave + b * AGG_WAVES) * 16 + r16;
        const float bias = col < Dout ? b1[col] : 0.f;
#pragma unroll
        for (int a = 0; a < MMT; ++a) {
            if (a >= mt_count) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = a * 16 + g4 * 4 + j;
                float v = acc[a][b][j] + bias;
                v = v > 0.f ? v : 0.f;
                if (drop_thr != 0u) v *= drop_scale(drop_lo, drop_hi, drop_thr, drop_inv, (unsigned long long)(mb + g) * Dout + col);
                Hs[(size_t)g * ldh + col] = v * wn[g];
            }
        }
    }
    __syncthreads();
    AGG_STAMP(6);
    // ---- reduce over the k neighbours of each query row ----
    const unsigned mDo = fastdiv_magic((unsigned)Dout);
    for (int idx = tid; idx < nq * Dout; idx += AGG_THREADS) {
        const int q = fastdiv(idx, mDo), c = idx - q * Dout;
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += Hs[(size_t)(q * k + j) * ldh + c];
        H[((size_t)m * N + q0 + q) * Dout + c] = s;
    }
    AGG_STAMP(7);
}

// ---------------------------------------------------------------------------
// The same kernel (table path) for the reference's default widths -- memory_dim = time_dim = 100
// (train.py:33-36) -- and k in {10, 20, 40}: a tile is exactly 80 gathered rows (5 M-tiles), the hidden layer 7
// N-tiles.  k_fc1_agg<true> spends more SIMD time on vector instructions (index arithmetic, bounds, generic
// loops: ~2400 per wave and tile) than on its 245 MFMAs; with the widths fixed at compile time the vector
// work is ~3x smaller:
//   * every thread owns one time-encoding COLUMN (its frequency stays in a register) and walks down 40 rows,
//     four dt values per broadcast LDS read; the 28 idle column lanes write the zero padding of the tile;
//   * the projected rows are fetched with one 64-bit address per row and immediate offsets per column;
//   * bias / ReLU / weight and the k-reduction use immediate LDS offsets (no divisions anywhere).
// Waves 0-2 own N-tiles {w, w+4}, wave 3 owns tile 3 only (7 tiles): two instantiations of the MFMA loop.
// (Splitting the 35 (M, N) tile pairs 9 / 9 / 9 / 8 instead of 10 / 10 / 10 / 5 was measured: no difference -- the
// kernel is bound by the latency of a workgroup's serial phases at 4 workgroups per CU, not by MFMA issue.)
// ---------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void d100_mfma(const float *A, int lda, const float *__restrict__ bp, int K1p, int nchunk, int nrem,
                                          int r16, int g4, f32x4 (&acc)[5][2])
{
    // nchunk whole 16-column chunks (one b128 fragment per lane: k = 16 kc + 4 g4 + j), then nrem single MFMA steps
    // over the last columns (k = 16 nchunk + 4 j + g4, scalar fragments): the contraction is padded to a multiple
    // of 4 columns, not 16 (F + T = 101 -> 26 steps instead of 28).
    f32x4 bcur[NB], bnext[NB];
    float bre[NB][3];
#pragma unroll
    for (int b = 0; b < NB; ++b) bcur[b] = *reinterpret_cast<const f32x4 *>(bp + (size_t)b * 64 * K1p);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            bre[b][j] = j < nrem ? bp[(size_t)b * 64 * K1p - 4 * g4 + 16 * nchunk + 4 * j + g4] : 0.f;
    const float *ap = A + (size_t)r16 * lda + 4 * g4;
    for (int kc = 0; kc < nchunk; ++kc) {
        if (kc + 1 < nchunk) {
#pragma unroll
            for (int b = 0; b < NB; ++b) bnext[b] = *reinterpret_cast<const f32x4 *>(bp + (size_t)b * 64 * K1p + 16 * (kc + 1));
        }
        f32x4 av[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) av[a] = *reinterpret_cast<const f32x4 *>(ap + (size_t)a * 16 * lda + 16 * kc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], bcur[b][j], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) bcur[b] = bnext[b];
    }
    const float *ar = A + (size_t)r16 * lda + 16 * nchunk + g4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j >= nrem) break;
        float av[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) av[a] = ar[(size_t)a * 16 * lda + 4 * j];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bre[b][j], acc[a][b], 0, 0, 0);
    }
}

template <int KK>
__global__ __launch_bounds__(AGG_THREADS, 4) void k_fc1_agg_d100(
    const float *__restrict__ P, const float *__restrict__ efeat, const float *__restrict__ time_w, long long num_nodes,
    long long num_edges, int F, long long N, int lda, const int *__restrict__ nbr, const int *__restrict__ eix,
    const float *__restrict__ dt, const float *__restrict__ w, const float *__restrict__ W1p, int K1p,
    const float *__restrict__ b1, float *__restrict__ H, float *__restrict__ S, int *status)
{
    constexpr int D = 100, T = 100, DP = 112, ROWS = 80, RQ = ROWS / KK, LDH = DP + 1;
    static_assert(ROWS % KK == 0, "a tile holds whole query rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *A = reinterpret_cast<float *>(smem);                       // [80][lda]: [ef (F) | cos (T) | 0-pad]
    float *wn = A + (size_t)ROWS * lda;                               // raw T-PPR weights
    float *wnn = wn + ROWS;                                           // w / sum(w)
    int *g_nb = reinterpret_cast<int *>(wnn + ROWS);
    int *g_ei = g_nb + ROWS;
    float *g_dt = reinterpret_cast<float *>(g_ei + ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int m = blockIdx.y;
    const long long q0 = (long long)blockIdx.x * RQ;
    const int nq = (int)((N - q0) < RQ ? (N - q0) : RQ);
    const int rows = nq * KK;
    const size_t mb = ((size_t)m * N + q0) * KK;
#ifdef ZT_AGG_STAMP
    long long t_prev__ = clock64();
#endif

    // ---- per-row scalars; a thread's own edge feature (F = 1) and time frequency ride on the same round trip ----
    float ef1 = 0.f;
    if (tid < ROWS) {
        int nb = 0, ei = 0;
        float d = 0.f, wv = 0.f;
        if (tid < rows) {
            nb = nbr[mb + tid]; ei = eix[mb + tid]; d = dt[mb + tid]; wv = w[mb + tid];
            if (nb < 0 || nb >= num_nodes || ei < 0 || ei >= num_edges) {
                atomicExch(status, ZT_ERR_RANGE);
                nb = 0; ei = 0;
            }
            if (F == 1) ef1 = efeat[ei];
        }
        g_nb[tid] = nb; g_ei[tid] = ei; g_dt[tid] = d; wn[tid] = wv;
    }
    const int cc = tid & 127;                                         // this thread's column of the time block
    const float wc = cc < T ? time_w[cc] : 0.f;
    const int col0 = wave * 16 + r16;                                 // this lane's output columns: col0, col0 + 64
    const float bias0 = col0 < D ? b1[col0] : 0.f, bias1 = (wave < 3 && col0 + 64 < D) ? b1[col0 + 64] : 0.f;
    __syncthreads();
    AGG_STAMP(0);

    // ---- accumulators start from the projected rows P[nbr[row]][col]: one address per row ----
    f32x4 acc[5][2];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int nb4[4] = {g_nb[a * 16 + g4 * 4 + 0], g_nb[a * 16 + g4 * 4 + 1], g_nb[a * 16 + g4 * 4 + 2], g_nb[a * 16 + g4 * 4 + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *pr = P + (size_t)nb4[j] * DP + col0;
            acc[a][0][j] = pr[0];
            acc[a][1][j] = pr[wave < 3 ? 64 : 0];                     // wave 3 has no second N-tile: any valid address
        }
    }
    // ---- w / sum(w), 0 where the sum is 0 (:267-270); sum in entry order like torch.sum(dim=1) ----
    if (tid < ROWS) {
        const int q = tid / KK;
        float my_sum = 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j) my_sum += wn[q * KK + j];
        wnn[tid] = (my_sum == 0.f) ? 0.f : wn[tid] / my_sum;
        if (tid < rows && tid - q * KK == 0) S[(size_t)m * N + q0 + q] = (my_sum == 0.f) ? 0.f : 1.f;
    }
    AGG_STAMP(1);
    // ---- edge features ----
    if (F == 1) {
        if (tid < ROWS) A[(size_t)tid * lda] = ef1;
    } else if (F > 0) {
        constexpr int GU = 8;
        const bool vecF = (F & 3) == 0 && ((size_t)efeat & 15) == 0;
        if (vecF) {
            const int F4 = F >> 2;
            const unsigned mF4 = fastdiv_magic((unsigned)F4);
            for (int f0 = tid; f0 < ROWS * F4; f0 += AGG_THREADS * GU) {
                f32x4 v[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF4), c = f - g * F4;
                    v[u] = (f < ROWS * F4 && g < rows) ? *reinterpret_cast<const f32x4 *>(efeat + (size_t)g_ei[g] * F + 4 * c)
                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF4), c = f - g * F4;
                    if (f < ROWS * F4) *reinterpret_cast<f32x4 *>(A + (size_t)g * lda + 4 * c) = v[u];
                }
            }
        } else {
            const unsigned mF = fastdiv_magic((unsigned)F);
            for (int f0 = tid; f0 < ROWS * F; f0 += AGG_THREADS * GU) {
                float v[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF), c = f - g * F;
                    v[u] = (f < ROWS * F && g < rows) ? efeat[(size_t)g_ei[g] * F + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int f = f0 + u * AGG_THREADS;
                    const int g = fastdiv(f, mF), c = f - g * F;
                    if (f < ROWS * F) A[(size_t)g * lda + c] = v[u];
                }
            }
        }
    }
    // ---- time encoding: column cc of rows [40 * (tid / 128), +40); lanes 100..127 write the zero padding ----
    // The common-range formula of time_cosf runs branch-free; the largest |argument| is tracked on the side and
    // the (rare: dt * w >= 4e6) columns are redone with the full routine afterwards.
    {
        const int g0 = (tid >> 7) * 40;
        const bool is_cos = cc < T;
        if (is_cos || F + cc < K1p) {
            float *dst = A + (size_t)g0 * lda + F + cc;
            float xmax = 0.f;
#pragma unroll 2
            for (int i = 0; i < 40; i += 4) {
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(g_dt + g0 + i);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float x = d4[u] * wc;
                    xmax = fmaxf(xmax, fabsf(x));
                    dst[(size_t)(i + u) * lda] = time_cosf_fast(x);          // cos(t*w + 0); wc = 0 in the padding: 1 -> fixed below
                }
            }
            if (!is_cos) {
                for (int i = 0; i < 40; ++i) dst[(size_t)i * lda] = 0.f;
            } else if (!(xmax < 4.0e6f)) {
                for (int i = 0; i < 40; ++i) dst[(size_t)i * lda] = time_cosf(g_dt[g0 + i] * wc);
            }
        }
    }
    AGG_STAMP(2);
    AGG_STAMP(3);
    __syncthreads();
    AGG_STAMP(4);

    // ---- fc1 on f32 MFMA ----
    {
        const float *bp = W1p + (size_t)col0 * K1p + 4 * g4;
        const int K4 = (F + T + 3) & ~3, nchunk = K4 >> 4, nrem = (K4 & 15) >> 2;
        if (wave < 3) d100_mfma<2>(A, lda, bp, K1p, nchunk, nrem, r16, g4, acc);
        else          d100_mfma<1>(A, lda, bp, K1p, nchunk, nrem, r16, g4, acc);
    }
    AGG_STAMP(5);
    __syncthreads();   // every wave is done reading the A tile: reuse it for the hidden rows

    // ---- bias + ReLU + weight, staged as Hs[g][col] in the A region ----
    float *Hs = A;
    {
        float *hp = Hs + (size_t)(g4 * 4) * LDH + col0;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wnn + a * 16 + g4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[a][0][j] + bias0;
                v = v > 0.f ? v : 0.f;
                hp[(size_t)(a * 16 + j) * LDH] = v * w4[j];
            }
        }
        if (wave < 3) {
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wnn + a * 16 + g4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v1 = acc[a][1][j] + bias1;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    hp[(size_t)(a * 16 + j) * LDH + 64] = v1 * w4[j];
                }
            }
        }
    }
    __syncthreads();
    AGG_STAMP(6);
    // ---- reduce over the k neighbours of each query row ----
    if (cc < D) {
        for (int q = tid >> 7; q < nq; q += 2) {
            const float *hq = Hs + (size_t)(q * KK) * LDH + cc;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KK; ++j) s += hq[(size_t)j * LDH];
            H[((size_t)m * N + q0 + q) * D + cc] = s;
        }
    }
    AGG_STAMP(7);
}

// ---------------------------------------------------------------------------
// k_fc1_agg_reg<KK>: the same computation for the shapes the headline configurations run (D = T = 100, F <= 4
// so that F + T <= 104, k in {20, 40}), built around the matrix pipe instead of around a workgroup's LDS tile:
//   * PERSISTENT waves, each one on its own: a wave owns 80 gathered rows (4 or 2 whole query rows) x all 7 N-tiles
//     of the hidden layer; no workgroup barrier anywhere, nothing but 1.3 KB of per-row scalars in LDS;
//   * fc1's [ef | time] weights stay in REGISTERS for the whole launch (7 N-tiles x 26 k-steps = 182 per lane, one
//     wave per SIMD: 512 registers) -- the MFMA's B operand never touches memory again;
//   * the A operand is never staged either: MFMA 16x16x4 wants A[row = lane & 15][k = 4 s + (lane >> 4)] from each
//     lane, which is cos(dt[row] * w_time[k]) -- the lane evaluates exactly that (one cosine per 7 MFMAs) while the
//     matrix pipe works on the previous step (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles);
//   * the projected rows P[nbr] (fc1 is linear: W_m memory[nbr], see k_fc1_agg) are added in the epilogue of each
//     16-row M-tile, their loads issued when the M-tile starts;
//   * bias + ReLU + w / sum(w) and the reduction over the k neighbours of a query row happen in registers: k is a
//     multiple of 4, so the 4 rows a lane holds of an M-tile belong to one query row; partial sums per (query
//     row, N-tile) are combined across the four 16-lane groups with two cross-lane adds at the end of the tile.
// Time encoding: every argument goes through time_cosf_rev (below), whose reduction is exact enough for any dt the
// stream produces: arguments beyond 4e6 (old neighbours: dt is seconds) are everyday data at stream scale.
// ---------------------------------------------------------------------------
#ifdef ZT_DIAG
__device__ unsigned long long g_regclk[4 * 1024];   // diagnostic builds (-DZT_DIAG, tools/build_diag.sh): per wave: shader cycles, wall start, wall end (100 MHz), tiles
#endif
constexpr int REG_NS = 26;       // MFMA k-steps: F + T <= 104
constexpr int REG_NB = 7;        // N-tiles of the hidden layer (D = 100 -> 112 columns)
constexpr int REG_FRAG_GROUPS = (REG_NB * REG_NS + 3) / 4;      // the weights in fragment order: groups of four values per lane

// fc1's [ef | time] block ([112][K2p], zero padded) in the order the lanes of k_fc1_agg_reg hold it: value q = 26 b + st
// of lane (r16, g4) is W[16 b + r16][4 st + g4]; image[q / 4][lane][q % 4]
__global__ void k_pack_reg_frag(const float *__restrict__ W1t, int K2p, float *__restrict__ image)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= REG_FRAG_GROUPS * 64 * 4) return;
    const int e = i & 3, lane = (i >> 2) & 63, G = i >> 8;
    const int q = 4 * G + e, b = q / REG_NS, st = q % REG_NS;
    image[i] = q < REG_NB * REG_NS ? W1t[(16 * b + (lane & 15)) * K2p + 4 * st + (lane >> 4)] : 0.f;
}

template <int KK, int DBG = 0>     // DBG != 0: only instantiated in diagnostic builds (-DZT_DIAG + ZT_AGG_DBG; WRONG results: 1 no cosine, 2 bare epilogue, 4 no projected rows)
__global__ __launch_bounds__(AGG_THREADS, 1) void k_fc1_agg_reg(
    const float *__restrict__ P, const float *__restrict__ efeat, const float *__restrict__ time_w, long long num_nodes,
    long long num_edges, int F, long long N, int M, const int *__restrict__ nbr, const int *__restrict__ eix,
    const float *__restrict__ dt, const float *__restrict__ w, const float *__restrict__ Wfrag,
    const float *__restrict__ b1, float *__restrict__ H, float *__restrict__ S, int *status,
    const int *gate_word, int gate_target, int *gate_latch)
{
    constexpr int D = 100, T = 100, DP = 112, ROWS = 80, RQ = ROWS / KK;
    static_assert(KK % 4 == 0 && ROWS % KK == 0, "groups of 4 rows must not straddle query rows");
    __shared__ int s_nb[AGG_WAVES][ROWS];
    __shared__ __attribute__((aligned(16))) float s_dt[AGG_WAVES][ROWS];
    __shared__ __attribute__((aligned(16))) float s_w[AGG_WAVES][ROWS];
    __shared__ float s_sum[AGG_WAVES][4];
    __shared__ __attribute__((aligned(16))) float s_ef[AGG_WAVES][ROWS * 4];      // edge-feature columns k < F <= 4 of every row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    int *nb_s = s_nb[wave];
    float *dt_s = s_dt[wave], *w_s = s_w[wave], *sum_s = s_sum[wave], *ef_s = s_ef[wave];
#ifdef ZT_DIAG
    const unsigned long long wall_in = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif

    // ---- once per launch: this lane's slice of the weights, the time frequencies and the biases ----
    // The 182 values a lane needs come from an image in FRAGMENT order (k_pack_reg_frag: [46 groups][64 lanes][4]): 46
    // coalesced 16-byte loads per lane, every wave of the launch reading the same 47 KB from L2.  (Round 3 staged the
    // [112][112] matrix through LDS and picked the values with 182 four-byte reads at a stride that puts sixteen lanes on
    // two banks: 10.6 us from kernel entry to the first tile -- 12 % of the kernel at C4's batch.)
    float Breg[REG_NB][REG_NS];
    {
        const f32x4 *Wf = reinterpret_cast<const f32x4 *>(Wfrag) + lane;
        static_assert(REG_FRAG_GROUPS % 2 == 0, "two batches");
        constexpr int HB = REG_FRAG_GROUPS / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // (all loads of a batch in flight before the first register write: the accumulator-register writes are volatile
            //  asm, and left alone the compiler issues load, wait, four writes, load, wait ... -- 46 L2 round trips in a row)
            f32x4 v[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) v[i] = Wf[(half * HB + i) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * (half * HB + i) + e;
                    // defined by an accumulator-register write: the value then LIVES in the accumulator half of the register
                    // file (an MFMA takes its B operand from either half); left to itself the allocator keeps all 182 in
                    // vector registers, runs out of them and spills
                    if (q < REG_NB * REG_NS) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Breg[q / REG_NS][q % REG_NS]) : "v"(v[i][e]));
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float freq[REG_NS];
#pragma unroll
    for (int st = 0; st < REG_NS; ++st) {
        const int kc = 4 * st + g4 - F;
        freq[st] = (kc >= 0 && kc < T) ? time_w[kc] : 0.f;
    }
    float bias[REG_NB];
#pragma unroll
    for (int b = 0; b < REG_NB; ++b) bias[b] = (16 * b + r16) < D ? b1[16 * b + r16] : 0.f;

#ifdef ZT_DIAG
    const unsigned long long clk0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull, wall0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
    const long long tpm = (N + RQ - 1) / RQ;                           // tiles per model
    const long long n_tiles = tpm * M;
    const long long stride = (long long)gridDim.x * AGG_WAVES;
    long long tile = (long long)blockIdx.x * AGG_WAVES + wave;

    // per-row scalars of a tile: lane l holds rows l and (l < 16) 64 + l; fetched one tile ahead
    int pf_nb[2] = {0, 0}, pf_ei[2] = {0, 0};
    float pf_dt[2] = {0.f, 0.f}, pf_w[2] = {0.f, 0.f};
    auto fetch = [&](long long t) {
        const long long m = t / tpm, q0 = (t - m * tpm) * RQ;
        const int rows = (int)((N - q0) < RQ ? (N - q0) : RQ) * KK;
        const size_t mb = ((size_t)m * N + q0) * KK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 64 + lane;
            const bool in = row < rows && (h == 0 || lane < ROWS - 64);
            pf_nb[h] = in ? nbr[mb + row] : 0;
            pf_ei[h] = in ? eix[mb + row] : 0;
            pf_dt[h] = in ? dt[mb + row] : 0.f;
            pf_w[h] = in ? w[mb + row] : 0.f;
        }
    };
    // ... and, once those edge ids are there, the rows' edge features (second stage, half a tile later)
    float pf_ef[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto fetch_ef = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ei = pf_ei[h];
            const bool ok = ei >= 0 && ei < num_edges && (h == 0 || lane < ROWS - 64);
#pragma unroll
            for (int c = 0; c < 4; ++c) pf_ef[h][c] = (ok && c < F) ? efeat[(size_t)ei * F + c] : 0.f;
        }
    };
    // the T-PPR rows of this batch may still be on their way (a launch group released batch by batch: common.hpp,
    // member_gate): the weights above were requested first, the wait is behind them
    member_gate_enter(gate_word, gate_target, status, gate_latch);
    if (tile < n_tiles) { fetch(tile); fetch_ef(); }

#pragma nounroll
    for (; tile < n_tiles; tile += stride) {
        const long long m = tile / tpm, q0 = (tile - m * tpm) * RQ;
        const int nq = (int)((N - q0) < RQ ? (N - q0) : RQ);
        // ---- this tile's scalars: registers -> LDS (the previous tile is done with the buffer) ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 64 + lane;
            if (h == 0 || lane < ROWS - 64) {
                int nb = pf_nb[h], ei = pf_ei[h];
                if (nb < 0 || nb >= num_nodes || ei < 0 || ei >= num_edges) { atomicExch(status, ZT_ERR_RANGE); nb = 0; ei = 0; }
                nb_s[row] = nb; dt_s[row] = pf_dt[h]; w_s[row] = pf_w[h];
                *reinterpret_cast<f32x4 *>(ef_s + 4 * row) = f32x4{pf_ef[h][0], pf_ef[h][1], pf_ef[h][2], pf_ef[h][3]};
            }
        }
        wave_sync();
        if (lane < RQ) {                                               // sum(w) in entry order, like torch.sum(dim=1)
            // (KK / 4 sixteen-byte reads, all in flight, then the adds in entry order: twenty dependent four-byte reads cost
            //  ~1 500 cycles per tile with nothing beside them)
            f32x4 wv[KK / 4];
#pragma unroll
            for (int j = 0; j < KK / 4; ++j) wv[j] = *reinterpret_cast<const f32x4 *>(w_s + lane * KK + 4 * j);
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < KK; ++j) sm += wv[j / 4][j % 4];
            sum_s[lane] = sm;
            if (lane < nq) S[(size_t)m * N + q0 + lane] = (sm == 0.f) ? 0.f : 1.f;
        }
        wave_sync();
        // w / sum(w), 0 where the sum is 0 (:267-270): once per row here, not once per use in the M-tiles
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 64 + lane;
            if (h == 0 || lane < ROWS - 64) {
                const float qs = sum_s[row / KK];
                w_s[row] = (qs == 0.f) ? 0.f : pf_w[h] / qs;
            }
        }
        wave_sync();
        const bool has_next = tile + stride < n_tiles;
        if (has_next) fetch(tile + stride);                             // in flight while this tile computes

        float Hloc[RQ][REG_NB];
#pragma unroll
        for (int q = 0; q < RQ; ++q)
#pragma unroll
            for (int b = 0; b < REG_NB; ++b) Hloc[q][b] = 0.f;

        // Software pipeline over the five M-tiles (16 rows each): while the matrix pipe runs M-tile a, the vector
        // unit evaluates the NEXT k-step's cosine and, during the first seven steps, the epilogue of M-tile a - 1 (one
        // N-tile per step) from a copy of its accumulators taken at the boundary; the projected rows of M-tile a are
        // requested at step 7, once that epilogue is through with the previous ones (one register set), and used
        // twenty steps later.  A sixth round drains the last epilogue.  sched_barrier between steps keeps the compiler
        // from hoisting five M-tiles' worth of loads and cosines to the top.  Register budget: accumulator half 182
        // weights + 28 accumulators; vector half ~180.
        f32x4 acc[REG_NB];
        float accv[REG_NB][4], Pv[2][REG_NB][4], wn_cur[4], wn_prv[4];
        // the projected rows of M-tile x into register set x & 1: requested TWO M-tiles ahead of their epilogue (a wave
        // has its SIMD to itself: nothing else hides a 2-4 us gather from a multi-gigabyte table)
        auto load_P = [&](int x) {
            const int4 nb4 = *reinterpret_cast<const int4 *>(nb_s + x * 16 + 4 * g4);         // epilogue rows 4 g4 + j
            const float *p0 = P + (size_t)nb4.x * DP + r16, *p1 = P + (size_t)nb4.y * DP + r16,
                        *p2 = P + (size_t)nb4.z * DP + r16, *p3 = P + (size_t)nb4.w * DP + r16;
#pragma unroll
            for (int b = 0; b < REG_NB; ++b) {
                Pv[x & 1][b][0] = p0[16 * b]; Pv[x & 1][b][1] = p1[16 * b]; Pv[x & 1][b][2] = p2[16 * b]; Pv[x & 1][b][3] = p3[16 * b];
            }
        };
        if (!(DBG & 4)) { load_P(0); load_P(1); }
#pragma unroll
        for (int a = 0; a <= 5; ++a) {
            float dta = 0.f, A_next[2] = {0.f, 0.f};
            if (a > 0) {
#pragma unroll
                for (int b = 0; b < REG_NB; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accv[b][j] = acc[b][j];
#pragma unroll
                for (int j = 0; j < 4; ++j) wn_prv[j] = wn_cur[j];
            }
            if (a < 5) {
                // ---- M-tile a: rows 16 a .. 16 a + 15 ----
                dta = dt_s[a * 16 + r16];                              // A operand: row r16 of the M-tile
                const float ef = ef_s[4 * (a * 16 + r16) + (g4 & 3)];  // k = g4 < F: an edge-feature column
                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(w_s + a * 16 + 4 * g4);   // already w / sum(w)
#pragma unroll
                for (int j = 0; j < 4; ++j) wn_cur[j] = w4[j];
#pragma unroll
                for (int b = 0; b < REG_NB; ++b) acc[b] = f32x4{bias[b], bias[b], bias[b], bias[b]};     // C operand of the first MFMA
                A_next[0] = (DBG & 1) ? dta * freq[0] : time_cosf_rev(dta * freq[0]);
                A_next[1] = (DBG & 1) ? dta * freq[1] : time_cosf_rev(dta * freq[1]);
                if (g4 < F) A_next[0] = ef;
                if (a == 2 && has_next) fetch_ef();                    // the next tile's edge ids have arrived by now
            }
            __builtin_amdgcn_sched_barrier(0);
            // query rows M-tile a - 1 touches (constants once unrolled)
            const int ap = a > 0 ? a - 1 : 0;
            const int q_lo = (16 * ap) / KK, q_hi = (16 * ap + 15) / KK;
            // Two k-steps at a time: 14 MFMAs, the cosines of the NEXT two steps (two independent chains: the f32 MFMA
            // keeps vector instructions off the SIMD while it runs, so what counts is that they issue back to back
            // once they do run) and, in the first rounds, two slices of the previous M-tile's epilogue.
#pragma unroll
            for (int st0 = 0; st0 < REG_NS; st0 += 2) {
                if (a == 5 && st0 >= REG_NB) break;
                const float A_cur[2] = {A_next[0], A_next[1]};
                if (a < 5) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (st0 + 2 + u < REG_NS)
                            A_next[u] = (DBG & 1) ? dta * freq[st0 + 2 + u] : time_cosf_rev(dta * freq[st0 + 2 + u]);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int b = 0; b < REG_NB; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_cur[u], Breg[b][st0 + u], acc[b], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int st = st0 + u;
                    if (!(a > 0 && st < REG_NB)) continue;
                    // epilogue of M-tile a - 1, N-tile st: + projected row (the bias went in as the MFMAs' C operand),
                    // ReLU, x w / sum(w); the lane's 4 rows belong to ONE query row
                    const int b = st;
                    float part = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = accv[b][j] + ((DBG & 2) ? 0.f : Pv[ap & 1][b][j]);
                        v = v > 0.f ? v : 0.f;
                        part += (DBG & 2) ? v : v * wn_prv[j];
                    }
                    if (DBG & 2) {
                        Hloc[q_lo][b] += part;
                    } else {
                        // The four 16-lane groups hold rows 4 g .. 4 g + 3 of the M-tile.  Every lane gets all four partial
                        // sums (two half-wave swaps + two row swaps, no LDS) and adds them to their query rows in ROW order:
                        // a query row's k / 4 groups are then always summed first to last, wherever in a tile the row falls
                        // -- a row's value must not depend on which shard or tile it is computed in.
                        const int pi = __float_as_int(part);
                        const auto h = __builtin_amdgcn_permlane32_swap(pi, pi, false, false);       // [0]: groups 0,1,0,1   [1]: 2,3,2,3
                        const auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);  // [0]: group 0 everywhere, [1]: group 1
                        const auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);  // [0]: group 2, [1]: group 3
                        const float pg[4] = {__int_as_float(lo[0]), __int_as_float(lo[1]), __int_as_float(hi[0]), __int_as_float(hi[1])};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int q = (16 * ap + 4 * g) / KK;                   // constant once unrolled
                            if (q < RQ) Hloc[q][b] += pg[g];
                        }
                    }
                    // (an empty side-effecting statement: it pins this arithmetic to THIS round.  Nothing else orders pure
                    //  vector work against the scheduling barriers, and instruction selection otherwise sinks all five
                    //  epilogues to the end of the tile, where their inputs -- 5 x 60 registers -- are still alive)
                    asm volatile("" : "+v"(Hloc[q_lo][b]));
                    if (q_hi != q_lo && q_hi < RQ) asm volatile("" : "+v"(Hloc[q_hi][b]));
                }
                if (a >= 1 && a <= 3 && st0 == REG_NB + 1 && !(DBG & 4)) load_P(a + 1);   // its register set has just been released
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- every lane holds every sum: group g writes query row g ----
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            if (g4 == (q & 3) && q < nq) {
                float *hrow = H + ((size_t)m * N + q0 + q) * D;
#pragma unroll
                for (int b = 0; b < REG_NB; ++b)
                    if (16 * b + r16 < D) hrow[16 * b + r16] = Hloc[q][b];
            }
        }
        wave_sync();                                                   // all reads of the scalar buffer are done
    }
#ifdef ZT_DIAG
    if (DBG && lane == 0 && blockIdx.x * AGG_WAVES + wave < 1024) {
        unsigned long long *o = g_regclk + 4 * (blockIdx.x * AGG_WAVES + wave);
        o[0] = __builtin_amdgcn_s_memtime() - clk0; o[1] = wall0; o[2] = __builtin_amdgcn_s_memrealtime();
        o[3] = wall_in;                                                         // (100 MHz) at kernel entry: the weight fill is wall0 - this
    }
#endif
}


// ---------------------------------------------------------------------------
// k_embed_out3 (round 5): the same three layers, PERSISTENT, for batches beyond k_embed_out2's reach.  k_embed_out's time
// at 3 000-12 000 rows was rounds of dependent weight-fragment fetches from L2 per 32-row workgroup (35-40 us for 1 GFLOP
// at C5's batch).  Here the weights live in LDS for the launch, in the order the MFMAs consume them (fragment (N-tile b,
// k-chunk c) = 64 lanes x 4 floats, one conflict-free ds_read_b128 per four MFMAs), one workgroup per CU of the stream,
// and every WAVE strides over (16 rows x ALL seven N-tiles) units by itself: the A operand comes straight from memory
// into the MFMA lanes (float4 at columns 16 c + 4 g, as in k_embed_out2), the next unit's rows are requested before the
// current unit's MFMAs, a row is read once.  Workgroups [0, n_src) take the source path (fc1s and fc2s resident: 98 KB, the
// hidden rows through 7 KB of the wave's own LDS), the others fc2 of the M models (49 KB).  Same accumulation order as
// the other two kernels (k-chunks first to last), so the three agree to the bit.  D = 100 (Dp = 112).
// ---------------------------------------------------------------------------
constexpr int EO3_NT = 7, EO3_KC = 7, EO3_DP = 112, EO3_LDY = 116;
constexpr int EO3_FRAG = EO3_NT * EO3_KC * 256;      // floats of one weight matrix in fragment order

__device__ __forceinline__ void eo3_fill(float *Wl, const float *__restrict__ Wp, int tid)
{
    constexpr int PER = (EO3_NT * EO3_KC * 64 + 255) / 256;             // fragments-lanes per thread: all in flight, then stored
    f32x4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int idx = tid + q * 256, frag = idx >> 6, ln = idx & 63;
        const int b = frag / EO3_KC, c = frag - b * EO3_KC;
        v[q] = idx < EO3_NT * EO3_KC * 64
                   ? *reinterpret_cast<const f32x4 *>(Wp + (size_t)(16 * b + (ln & 15)) * EO3_DP + 16 * c + 4 * (ln >> 4))
                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int idx = tid + q * 256;
        if (idx < EO3_NT * EO3_KC * 64) reinterpret_cast<f32x4 *>(Wl)[idx] = v[q];
    }
}

template <int HG>
__global__ __launch_bounds__(AGG_THREADS, 1) void k_embed_out3(
    const float *__restrict__ memory, long long num_nodes, const int *__restrict__ nodes, long long N, int D, int M,
    const float *__restrict__ H, const float *__restrict__ S, const float *__restrict__ fc2_p,
    const float *__restrict__ fc2_b, const float *__restrict__ fc1s_p, const float *__restrict__ fc1s_b,
    const float *__restrict__ fc2s_p, const float *__restrict__ fc2s_b, float *__restrict__ out, int *status, int n_src)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *W0 = reinterpret_cast<float *>(smem);                         // fc2 (neighbour role) / fc1s (source role)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int OW = D * (M + 1);
    const long long tiles = (N + 15) / 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    bool cin[EO3_KC];                                                    // this lane's float4 of chunk c lies inside the row
#pragma unroll
    for (int c = 0; c < EO3_KC; ++c) cin[c] = 16 * c + 4 * g4 < D;
    const f32x4 *Wf0 = reinterpret_cast<const f32x4 *>(W0) + lane;

    if ((int)blockIdx.x < n_src) {
        // ---------------- source path: memory[nodes] -> fc1s -> relu -> fc2s ----------------
        float *W1 = W0 + EO3_FRAG;                                       // fc2s
        float *Y = W1 + EO3_FRAG + wave * 16 * EO3_LDY;                  // this wave's hidden rows
        eo3_fill(W0, fc1s_p, tid);
        eo3_fill(W1, fc2s_p, tid);
        float b1v[EO3_NT], b2v[EO3_NT];
#pragma unroll
        for (int b = 0; b < EO3_NT; ++b) {
            b1v[b] = 16 * b + r16 < D ? fc1s_b[16 * b + r16] : 0.f;
            b2v[b] = 16 * b + r16 < D ? fc2s_b[16 * b + r16] : 0.f;
        }
        const f32x4 *Wf1 = reinterpret_cast<const f32x4 *>(W1) + lane;
        const long long stride = (long long)n_src * AGG_WAVES;
        long long t = (long long)blockIdx.x * AGG_WAVES + wave;
        // ids two units ahead, rows one unit ahead
        auto node_of = [&](long long tt) -> int { return (tt < tiles && tt * 16 + r16 < N) ? nodes[tt * 16 + r16] : -1; };
        auto rows_of = [&](int nd, f32x4 (&a)[EO3_KC]) {
            const bool ok = nd >= 0 && nd < num_nodes;
#pragma unroll
            for (int c = 0; c < EO3_KC; ++c)
                a[c] = (ok && cin[c]) ? *reinterpret_cast<const f32x4 *>(memory + (size_t)nd * D + 16 * c + 4 * g4) : zero4;
        };
        int nd_cur = node_of(t), nd_nxt = node_of(t + stride);
        f32x4 a[EO3_KC], an[EO3_KC];
        if (t < tiles) rows_of(nd_cur, a);
        __syncthreads();                                                 // both weight images are in LDS
        for (; t < tiles; t += stride) {
            const long long r0 = t * 16;
            if (r0 + r16 < N && (nd_cur < 0 || nd_cur >= num_nodes)) atomicExch(status, ZT_ERR_RANGE);
            rows_of(nd_nxt, an);                                         // the next unit's rows, under this unit's MFMAs
            nd_cur = nd_nxt;
            nd_nxt = node_of(t + 2 * stride);
            f32x4 acc1[EO3_NT];
#pragma unroll
            for (int b = 0; b < EO3_NT; ++b) acc1[b] = zero4;
#pragma unroll
            for (int c = 0; c < EO3_KC; ++c) {
                f32x4 w[EO3_NT];
#pragma unroll
                for (int b = 0; b < EO3_NT; ++b) w[b] = Wf0[(b * EO3_KC + c) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)                                  // consecutive MFMAs on different accumulators
#pragma unroll
                    for (int b = 0; b < EO3_NT; ++b) acc1[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], w[b][j], acc1[b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < EO3_NT; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc1[b][j] + b1v[b];
                    Y[(4 * g4 + j) * EO3_LDY + 16 * b + r16] = (16 * b + r16 < D && v > 0.f) ? v : 0.f;
                }
            wave_sync();
            f32x4 acc[EO3_NT];
#pragma unroll
            for (int b = 0; b < EO3_NT; ++b) acc[b] = zero4;
#pragma unroll
            for (int c = 0; c < EO3_KC; ++c) {
                const f32x4 y = *reinterpret_cast<const f32x4 *>(Y + r16 * EO3_LDY + 16 * c + 4 * g4);
                f32x4 w[EO3_NT];
#pragma unroll
                for (int b = 0; b < EO3_NT; ++b) w[b] = Wf1[(b * EO3_KC + c) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < EO3_NT; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[j], w[b][j], acc[b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < EO3_NT; ++b) {
                const int col = 16 * b + r16;
                if (col < D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (r0 + 4 * g4 + j < N) out[(size_t)(r0 + 4 * g4 + j) * OW + col] = acc[b][j] + b2v[b];
                }
            }
            wave_sync();                                                 // Y is free for the next unit
#pragma unroll
            for (int c = 0; c < EO3_KC; ++c) a[c] = an[c];
        }
        return;
    }
    // ---------------- neighbour paths: fc2 on the reduced rows of model m ----------------
    eo3_fill(W0, fc2_p, tid);
    float b2v[EO3_NT];
#pragma unroll
    for (int b = 0; b < EO3_NT; ++b) b2v[b] = 16 * b + r16 < D ? fc2_b[16 * b + r16] : 0.f;
    const long long units = tiles * M;
    const long long stride = (long long)((int)gridDim.x - n_src) * AGG_WAVES;
    long long u = (long long)((int)blockIdx.x - n_src) * AGG_WAVES + wave;
    // a unit's A operand: a query row's HG partial-sum groups (k_fc1_agg_wide), added first to last.  HG = 1: the next
    // unit's rows are requested under this unit's MFMAs; HG > 1 (F = 172: at most a few units per wave at those batch
    // sizes): groups [q0, q0 + GQ) at a time, at the top of the unit
    constexpr int GQ = HG == 1 ? 1 : 5;
    auto rows_of = [&](long long uu, int q0, f32x4 (&g)[GQ][EO3_KC]) {
        const long long m = uu / tiles, r = (uu - m * tiles) * 16 + r16;
        const bool rin = uu < units && r < N;
        const float *hp = H + (((size_t)m * N + (rin ? r : 0)) * HG + q0) * D + 4 * g4;
#pragma unroll
        for (int q = 0; q < GQ; ++q)
#pragma unroll
            for (int c = 0; c < EO3_KC; ++c) g[q][c] = (rin && cin[c]) ? *reinterpret_cast<const f32x4 *>(hp + (size_t)q * D + 16 * c) : zero4;
    };
    f32x4 gn[GQ][EO3_KC];
    if (HG == 1) rows_of(u, 0, gn);
    __syncthreads();                                                     // the weight image is in LDS
    for (; u < units; u += stride) {
        const long long m = u / tiles, r0 = (u - m * tiles) * 16;
        f32x4 a[EO3_KC];
        if (HG == 1) {
#pragma unroll
            for (int c = 0; c < EO3_KC; ++c) a[c] = gn[0][c];
        } else {
#pragma nounroll
            for (int q0 = 0; q0 < HG; q0 += GQ) {
                rows_of(u, q0, gn);
#pragma unroll
                for (int c = 0; c < EO3_KC; ++c) {
                    if (q0 == 0) a[c] = gn[0][c]; else a[c] += gn[0][c];
#pragma unroll
                    for (int q = 1; q < GQ; ++q) a[c] += gn[q][c];
                }
            }
        }
        float sv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sv[j] = r0 + 4 * g4 + j < N ? S[(size_t)m * N + r0 + 4 * g4 + j] : 0.f;
        if (HG == 1) rows_of(u + stride, 0, gn);                         // the next unit's rows, under this unit's MFMAs
        f32x4 acc[EO3_NT];
#pragma unroll
        for (int b = 0; b < EO3_NT; ++b) acc[b] = zero4;
#pragma unroll
        for (int c = 0; c < EO3_KC; ++c) {
            f32x4 w[EO3_NT];
#pragma unroll
            for (int b = 0; b < EO3_NT; ++b) w[b] = Wf0[(b * EO3_KC + c) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < EO3_NT; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], w[b][j], acc[b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < EO3_NT; ++b) {
            const int col = 16 * b + r16;
            if (col < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0 + 4 * g4 + j < N) out[(size_t)(r0 + 4 * g4 + j) * OW + (size_t)D * (m + 1) + col] = acc[b][j] + b2v[b] * sv[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Projected memory table: P[v][0..Dp) = W_m memory[v] (W_m = the memory columns of fc1, zero padded to
// [Dp][Dp]; columns >= D of P are zero).  rows == nullptr: every node; else the n rows listed (ids < 0 or
// beyond *count are skipped): the rows a batch's GRU update / exchange has just rewritten.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(AGG_THREADS) void k_project_rows(const float *__restrict__ memory, long long num_nodes,
                                                              int D, const float *__restrict__ Wm_p,
                                                              const int *__restrict__ rows, const int *__restrict__ count,
                                                              long long n_max, float *__restrict__ P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Dp = (D + 15) / 16 * 16, ldx = Dp + 4, NT = Dp / 16;
    float *X = reinterpret_cast<float *>(smem);          // [32][ldx]
    int *rid = reinterpret_cast<int *>(X + OUT_ROWS * ldx);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const long long r0 = (long long)blockIdx.x * OUT_ROWS;
    long long n = n_max;
    if (count != nullptr) { const long long c = *count; n = c < n ? c : n; }
    if (r0 >= n) return;
    if (tid < OUT_ROWS) {
        long long v = -1;
        if (r0 + tid < n) v = rows != nullptr ? (long long)rows[r0 + tid] : r0 + tid;
        rid[tid] = (v >= 0 && v < num_nodes) ? (int)v : -1;
    }
    __syncthreads();
    const unsigned mL = fastdiv_magic((unsigned)ldx);
    for (int f = tid; f < OUT_ROWS * ldx; f += AGG_THREADS) {
        const int g = fastdiv(f, mL), c = f - g * ldx;
        X[f] = (rid[g] >= 0 && c < D) ? memory[(size_t)rid[g] * D + c] : 0.f;
    }
    __syncthreads();
    f32x4 acc[OUT_MT][NTW];
    small_gemm(X, ldx, Wm_p, Dp, NT, wave, lane, acc);
#pragma unroll
    for (int b = 0; b < NTW; ++b) {
        const int col = (wave + b * AGG_WAVES) * 16 + r16;
        if (col >= Dp) continue;
#pragma unroll
        for (int a = 0; a < OUT_MT; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = a * 16 + g4 * 4 + j;
                if (rid[g] >= 0) P[(size_t)rid[g] * Dp + col] = acc[a][b][j];
            }
    }
}

struct EmbedPlan {
    int Dp, K1p, lda, mt, rq;
    size_t lds;
    // table path (k_fc1_agg<true>): the tile holds only [ef | cos]
    int K2p, lda2, mt2, rq2;
    size_t lds2;
    size_t off_w1p, off_H, off_S, off_fc2t, off_fc1st, off_fc2st, off_w1t, off_wm, off_wl, off_wfrag, total;
    int hg;                                     // partial-sum groups per query row in H (k_fc1_agg_wide: k / 4; else 1)
    bool big;                                   // k > 80: the MAX_MT_BIG instantiation of the generic kernel (mt = 0: that path's tile does not fit)
};

// tile shape for a contraction over Kp columns: as many whole query rows as fit MAX_MT tiles / the LDS budget
bool tile_shape(int Kp, int Dp, int k, int T, int *lda_out, int *mt_out, int *rq_out, size_t *lds_out, int max_mt = MAX_MT)
{
    const int lda = Kp + 4;
    int mt = max_mt;
    while (mt > 1 && ((size_t)mt * 16 * lda * 4 + (size_t)mt * 16 * 16) > (size_t)LDS_BUDGET) --mt;
    int rq = (mt * 16) / k;
    if (rq < 1) {
        // one query row must fit: grow to ceil(k/16) tiles if the budget allows
        mt = (k + 15) / 16;
        rq = 1;
        if (mt > max_mt || ((size_t)mt * 16 * lda * 4 + (size_t)mt * 16 * 16) > (size_t)LDS_BUDGET) return false;
    }
    mt = (rq * k + 15) / 16;
    // the hidden staging [rows_p][Dp+1] reuses the A region: it must fit
    if ((size_t)mt * 16 * (Dp + 1) * 4 > (size_t)mt * 16 * lda * 4) return false;
    *lda_out = lda; *mt_out = mt; *rq_out = rq;
    *lds_out = (size_t)mt * 16 * lda * 4 + (size_t)mt * 16 * 16 + (size_t)T * 4;   // A tile + wn, nbr, eix, dt per row + time_w
    return true;
}

bool make_plan(int64_t N, int D, int F, int T, int M, int k, EmbedPlan &p)
{
    const int K1 = D + F + T;
    p.Dp = round_up(D, 16);
    p.K1p = round_up(K1, 16);
    p.big = k > MAX_MT * 16;
    const int max_mt = p.big ? MAX_MT_BIG : MAX_MT;
    const bool full_ok = tile_shape(p.K1p, p.Dp, k, T, &p.lda, &p.mt, &p.rq, &p.lds, max_mt);
    if (!full_ok) {
        if (!p.big) return false;
        p.mt = 0; p.rq = 1; p.lda = p.K1p + 4; p.lds = 0;       // (wide k: only the table path's narrower tile may fit)
    }
    p.K2p = round_up(F + T, 16);
    p.mt2 = 0;                                  // table path unavailable (e.g. F + T < D: the staging would not fit)
    if (!tile_shape(p.K2p, p.Dp, k, T, &p.lda2, &p.mt2, &p.rq2, &p.lds2, max_mt)) p.mt2 = 0;
    if (p.mt == 0 && p.mt2 == 0) return false;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    // padded weights first: their offsets depend on (D, F, T) only, so a workspace prepared once serves every N
    p.off_w1p = take((size_t)p.Dp * p.K1p * 4);
    p.off_fc2t = take((size_t)p.Dp * p.Dp * 4);
    p.off_fc1st = take((size_t)p.Dp * p.Dp * 4);
    p.off_fc2st = take((size_t)p.Dp * p.Dp * 4);
    p.off_w1t = take((size_t)p.Dp * p.K2p * 4);
    p.off_wm = take((size_t)p.Dp * p.Dp * 4);
    p.hg = fc1_agg_wide_supported(D, F, T, k) ? k / 4 : 1;
    p.off_wl = take(p.hg > 1 ? fc1_agg_wide_weight_bytes() : 0);
    p.off_wfrag = take((size_t)REG_FRAG_GROUPS * 64 * 4 * 4);      // k_fc1_agg_reg's weights in fragment order (K2p = 112 shapes)
    p.off_H = take((size_t)M * N * D * 4 * p.hg);
    p.off_S = take((size_t)M * N * 4);
    p.total = o;
    return true;
}

}  // namespace

#ifdef ZT_AGG_STAMP
extern "C" int zt_debug_agg(unsigned long long *host, int reset)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_agg), sizeof(unsigned long long) * 8));
    if (reset) { unsigned long long z[8] = {0}; ZT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_agg), z, sizeof(z))); }
    return ZT_OK;
}
#endif

#ifdef ZT_DIAG
extern "C" int zt_debug_regclk(unsigned long long *host)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_regclk), sizeof(unsigned long long) * 4 * 1024));
    return ZT_OK;
}
#endif

extern "C" int64_t zt_embed_workspace_bytes(int64_t N, int32_t D, int32_t F, int32_t T, int32_t M, int32_t k)
{
    EmbedPlan p;
    if (N < 0 || D <= 0 || F < 0 || T < 0 || M <= 0 || k <= 0) return -1;
    if (!make_plan(N > 0 ? N : 1, D, F, T, M, k, p)) return -1;
    return (int64_t)p.total;
}

// pads every weight matrix into the workspace (once per weight change: zt_embed's weights_ready = 0)
static void embed_prepare(const zt_embed_weights *wt, int D, int F, int T, const EmbedPlan &p, char *ws, hipStream_t s)
{
    const int K1 = D + F + T;
    auto pad = [&](const float *W, int ld, int c0, int cols, size_t off, int cols_p) {
        k_pad_matrix<<<(p.Dp * cols_p + 255) / 256, 256, 0, s>>>(W, ld, c0, D, cols, reinterpret_cast<float *>(ws + off), p.Dp, cols_p);
    };
    pad(wt->fc1_w, K1, 0, K1, p.off_w1p, p.K1p);
    pad(wt->fc2_w, D, 0, D, p.off_fc2t, p.Dp);
    pad(wt->fc1s_w, D, 0, D, p.off_fc1st, p.Dp);
    pad(wt->fc2s_w, D, 0, D, p.off_fc2st, p.Dp);
    pad(wt->fc1_w, K1, D, F + T, p.off_w1t, p.K2p);       // the [ef | time] columns of fc1 (table path)
    pad(wt->fc1_w, K1, 0, D, p.off_wm, p.Dp);             // W_m: the memory columns of fc1 (k_project_rows)
    if (p.hg > 1) fc1_agg_wide_pack(wt->fc1_w, wt->time_w, wt->fc1_b, reinterpret_cast<float *>(ws + p.off_wl), s);   // LDS image (aggregate_wide.hip)
    if (p.K2p == 112 && p.Dp == 112)
        k_pack_reg_frag<<<(REG_FRAG_GROUPS * 64 * 4 + 255) / 256, 256, 0, s>>>(reinterpret_cast<const float *>(ws + p.off_w1t), p.K2p,
                                                                          reinterpret_cast<float *>(ws + p.off_wfrag));
}

const float *zt::embed_wm_ptr(void *embed_ws, int64_t N, int32_t D, int32_t F, int32_t T, int32_t M, int32_t k)
{
    EmbedPlan p;
    if (!embed_ws || D > 128 || !make_plan(N > 0 ? N : 1, D, F, T, M, k, p)) return nullptr;
    return reinterpret_cast<const float *>(reinterpret_cast<char *>(embed_ws) + p.off_wm);
}

extern "C" int64_t zt_project_table_bytes(int64_t num_nodes, int32_t D)
{
    if (num_nodes <= 0 || D <= 0) return -1;
    return (int64_t)num_nodes * round_up(D, 16) * 4;
}

extern "C" int zt_project_memory(const float *memory_dev, int64_t num_nodes, int32_t D, int32_t F, int32_t T,
                                 const zt_embed_weights *wt, int32_t weights_ready, const int32_t *rows_dev,
                                 const int32_t *count_dev, int64_t max_rows, float *table_dev, void *workspace_dev,
                                 int64_t ws_N, int32_t ws_M, int32_t ws_k, void *stream)
{
    if (!memory_dev || !wt || !table_dev || !workspace_dev || num_nodes <= 0 || D <= 0 || F < 0 || T < 0 || max_rows < 0) {
        set_error("zt_project_memory: bad argument");
        return ZT_ERR_ARG;
    }
    EmbedPlan p;
    if (D > 128 || !make_plan(ws_N > 0 ? ws_N : 1, D, F, T, ws_M, ws_k, p)) {
        set_error("zt_project_memory: unsupported shape");
        return ZT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace_dev);
    if (!weights_ready) embed_prepare(wt, D, F, T, p, ws, s);
    const long long n = rows_dev ? max_rows : num_nodes;
    if (n == 0) return ZT_OK;
    const size_t lds = (size_t)OUT_ROWS * (p.Dp + 4) * 4 + OUT_ROWS * 4;
    k_project_rows<<<(unsigned)((n + OUT_ROWS - 1) / OUT_ROWS), AGG_THREADS, lds, s>>>(
        memory_dev, num_nodes, D, reinterpret_cast<const float *>(ws + p.off_wm), rows_dev, count_dev, n, table_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

// zt_embed; mid_wait (pipeline.hip): an event the stream is told to wait for BETWEEN the aggregation and the output layer
static int embed_impl(const float *memory_dev, const float *efeat_dev, int64_t num_nodes, int64_t num_edges,
                      int32_t D, int32_t F, int32_t T, const int32_t *nodes_dev, int64_t N, int32_t M, int32_t k,
                      const int32_t *nbr_dev, const int32_t *eix_dev, const float *dt_dev, const float *w_dev,
                      const zt_embed_weights *wt, float *out_dev, void *workspace_dev, int32_t *status_dev,
                      const float *proj_table_dev, int32_t weights_ready, void *stream, hipEvent_t mid_wait,
                      zt::embed_out_deferred *defer = nullptr, const zt::member_gate *gate = nullptr)
{
    if (defer) defer->valid = false;
    if (!memory_dev || !efeat_dev || !wt || !status_dev || N < 0 || D <= 0 || F < 0 || T < 0 || M <= 0 || k <= 0) {
        set_error("zt_embed: bad argument");
        return ZT_ERR_ARG;
    }
    if (N == 0) return ZT_OK;
    if (!nodes_dev || !nbr_dev || !eix_dev || !dt_dev || !w_dev || !out_dev || !workspace_dev) {
        set_error("zt_embed: NULL buffer");
        return ZT_ERR_ARG;
    }
    EmbedPlan p;
    if (D > 16 * NTW * AGG_WAVES || D > 128 || !make_plan(N, D, F, T, M, k, p)) {
        set_error("zt_embed: D=%d F=%d T=%d k=%d outside the supported shapes (D<=128, one query row of k "
                  "neighbours must fit the %d KB LDS tile)", D, F, T, k, LDS_BUDGET / 1024);
        return ZT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace_dev);
    float *W1p = reinterpret_cast<float *>(ws + p.off_w1p);
    float *W1t = reinterpret_cast<float *>(ws + p.off_w1t);
    const float *Wfrag = reinterpret_cast<const float *>(ws + p.off_wfrag);
    float *H = reinterpret_cast<float *>(ws + p.off_H);
    float *S = reinterpret_cast<float *>(ws + p.off_S);
    float *fc2t = reinterpret_cast<float *>(ws + p.off_fc2t);
    float *fc1st = reinterpret_cast<float *>(ws + p.off_fc1st);
    float *fc2st = reinterpret_cast<float *>(ws + p.off_fc2st);
    if (!weights_ready) {
        ZT_PROF_BEGIN(s, P_EMBED_PREP);
        embed_prepare(wt, D, F, T, p, ws, s);
        ZT_PROF_END(s, P_EMBED_PREP);
    }
    const bool tab = proj_table_dev != nullptr && p.mt2 > 0;
    const size_t lds = tab ? p.lds2 : p.lds;
    static size_t attr_lds[2] = {0, 0};
    if (!tab && p.mt == 0) {
        set_error("zt_embed: k=%d needs the projected table (a query row's [memory | ef | time] tile does not fit %d KB of LDS)", k, LDS_BUDGET / 1024);
        return ZT_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024 && lds > attr_lds[tab ? 1 : 0]) {
        const void *fn = tab ? reinterpret_cast<const void *>(k_fc1_agg<true>) : reinterpret_cast<const void *>(k_fc1_agg<false>);
        ZT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds[tab ? 1 : 0] = lds;
    }
    static size_t attr_big[2] = {0, 0};
    if (p.big && lds > 48 * 1024 && lds > attr_big[tab ? 1 : 0]) {
        const void *fn = tab ? reinterpret_cast<const void *>(k_fc1_agg<true, MAX_MT_BIG>) : reinterpret_cast<const void *>(k_fc1_agg<false, MAX_MT_BIG>);
        ZT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_big[tab ? 1 : 0] = lds;
    }
    // the reference's default widths take the specialised kernels (zt_set_kernel_choice(ZT_CHOICE_AGGREGATE, ZT_AGG_GENERIC): the generic one)
    const bool generic = zt::kernel_choice(ZT_CHOICE_AGGREGATE) == ZT_AGG_GENERIC;
    const bool d100 = tab && D == 100 && T == 100 && p.mt2 == 5 && (k == 10 || k == 20 || k == 40) && !generic;
    static size_t attr_fast[3] = {0, 0, 0};
    if (d100 && lds > 48 * 1024) {
        const int ki = k == 10 ? 0 : (k == 20 ? 1 : 2);
        if (lds > attr_fast[ki]) {
            const void *fn = k == 10 ? reinterpret_cast<const void *>(k_fc1_agg_d100<10>)
                                     : (k == 20 ? reinterpret_cast<const void *>(k_fc1_agg_d100<20>)
                                                : reinterpret_cast<const void *>(k_fc1_agg_d100<40>));
            ZT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_fast[ki] = lds;
        }
    }
    // register-resident kernel: D = T = 100, F + T <= 104, k in {20, 40}
    const bool regk = tab && D == 100 && T == 100 && F <= 4 && (k == 20 || k == 40) && p.K2p == 112 && !generic;
    // wide edge features (F = 172), D = T = 100, k in {20, 40}: weights resident in LDS (aggregate_wide.hip)
    const bool widek = tab && p.hg > 1 && !generic;
    // (a wait for this batch's T-PPR rows: inside the two persistent kernels, a one-wave kernel in front of the others)
    if (gate != nullptr && gate->word != nullptr && !widek && !regk) {
        const int rc = zt::member_gate_launch(*gate, status_dev, s);
        if (rc != ZT_OK) return rc;
    }
    ZT_PROF_BEGIN(s, P_FC1_AGG);
    if (widek) {
        const int rc = fc1_agg_wide_launch(proj_table_dev, efeat_dev, wt->time_w, num_nodes, num_edges, N, M, k, nbr_dev, eix_dev,
                                           dt_dev, w_dev, reinterpret_cast<const float *>(ws + p.off_wl), wt->fc1_b, H, S,
                                           status_dev, stream_cu_count(s), s, gate);
        if (rc != ZT_OK) return rc;
    } else if (regk) {
        const long long rq = 80 / k, tiles = ((N + rq - 1) / rq) * M;
        long long wgs = (tiles + AGG_WAVES - 1) / AGG_WAVES;
        const int cus = stream_cu_count(s);
        if (wgs > cus) wgs = cus;                                      // persistent: one workgroup (4 waves) per CU
        const size_t reg_lds = 0;
        const int *gw = gate ? gate->word : nullptr;
        const int gt = gate ? gate->target : 0;
        int *gl = gate ? gate->latch : nullptr;
#ifdef ZT_DIAG
        static const int dbg = getenv("ZT_AGG_DBG") ? atoi(getenv("ZT_AGG_DBG")) : 0;       // diagnostic (wrong results)
#define ZT_REG_DBG(DB) k_fc1_agg_reg<20, DB><<<(unsigned)wgs, AGG_THREADS, reg_lds, s>>>(proj_table_dev, efeat_dev, wt->time_w, num_nodes, \
            num_edges, F, N, M, nbr_dev, eix_dev, dt_dev, w_dev, Wfrag, wt->fc1_b, H, S, status_dev, gw, gt, gl)
        if (k == 20 && dbg == 1) ZT_REG_DBG(1);
        else if (k == 20 && dbg == 2) ZT_REG_DBG(2);
        else if (k == 20 && dbg == 6) ZT_REG_DBG(6);
        else if (k == 20 && dbg == 7) ZT_REG_DBG(7);
        else if (k == 20 && dbg == 8) ZT_REG_DBG(8);       // (all parts, with the per-wave clocks)
#undef ZT_REG_DBG
        else
#endif
        if (k == 20)
            k_fc1_agg_reg<20><<<(unsigned)wgs, AGG_THREADS, reg_lds, s>>>(proj_table_dev, efeat_dev, wt->time_w, num_nodes, num_edges, F, N,
                                                                 M, nbr_dev, eix_dev, dt_dev, w_dev, Wfrag, wt->fc1_b, H, S, status_dev,
                                                                 gw, gt, gl);
        else
            k_fc1_agg_reg<40><<<(unsigned)wgs, AGG_THREADS, reg_lds, s>>>(proj_table_dev, efeat_dev, wt->time_w, num_nodes, num_edges, F, N,
                                                                 M, nbr_dev, eix_dev, dt_dev, w_dev, Wfrag, wt->fc1_b, H, S, status_dev,
                                                                 gw, gt, gl);
    } else if (d100) {
        dim3 grid((unsigned)((N + p.rq2 - 1) / p.rq2), (unsigned)M);
#define ZT_D100(KK) k_fc1_agg_d100<KK><<<grid, AGG_THREADS, lds, s>>>(proj_table_dev, efeat_dev, wt->time_w, num_nodes,     \
            num_edges, F, N, p.lda2, nbr_dev, eix_dev, dt_dev, w_dev, W1t, p.K2p, wt->fc1_b, H, S, status_dev)
        if (k == 10) ZT_D100(10); else if (k == 20) ZT_D100(20); else ZT_D100(40);
#undef ZT_D100
    } else if (tab) {
        dim3 grid((unsigned)((N + p.rq2 - 1) / p.rq2), (unsigned)M);
        if (p.big)
            k_fc1_agg<true, MAX_MT_BIG><<<grid, AGG_THREADS, lds, s>>>(proj_table_dev, efeat_dev, wt->time_w, num_nodes, num_edges, p.Dp, F,
                                                                   T, N, k, p.rq2, p.mt2, p.lda2, nbr_dev, eix_dev, dt_dev, w_dev, W1t,
                                                                   p.K2p, wt->fc1_b, H, S, status_dev, D, nullptr, nullptr);
        else
            k_fc1_agg<true><<<grid, AGG_THREADS, lds, s>>>(proj_table_dev, efeat_dev, wt->time_w, num_nodes, num_edges, p.Dp, F,
                                                           T, N, k, p.rq2, p.mt2, p.lda2, nbr_dev, eix_dev, dt_dev, w_dev, W1t,
                                                           p.K2p, wt->fc1_b, H, S, status_dev, D, nullptr, nullptr);
    } else {
        dim3 grid((unsigned)((N + p.rq - 1) / p.rq), (unsigned)M);
        if (p.big)
            k_fc1_agg<false, MAX_MT_BIG><<<grid, AGG_THREADS, lds, s>>>(memory_dev, efeat_dev, wt->time_w, num_nodes, num_edges, D, F, T, N,
                                                                    k, p.rq, p.mt, p.lda, nbr_dev, eix_dev, dt_dev, w_dev, W1p, p.K1p,
                                                                    wt->fc1_b, H, S, status_dev, D, nullptr, nullptr);
        else
            k_fc1_agg<false><<<grid, AGG_THREADS, lds, s>>>(memory_dev, efeat_dev, wt->time_w, num_nodes, num_edges, D, F, T, N,
                                                            k, p.rq, p.mt, p.lda, nbr_dev, eix_dev, dt_dev, w_dev, W1p, p.K1p,
                                                            wt->fc1_b, H, S, status_dev, D, nullptr, nullptr);
    }
    ZT_PROF_END(s, P_FC1_AGG);
    if (mid_wait != nullptr) ZT_HIP(hipStreamWaitEvent(s, mid_wait, 0));
    const int hg = widek ? p.hg : 1;
    // small batches: the latency-organised kernel (one wave per tile, path and N-tile; 1 - 2 memory round trips); large ones:
    // the persistent one (weights resident in LDS, a wave per 16 rows x all N-tiles); other widths: the tiled one (weights
    // streamed from L2 per 32-row workgroup).  zt_set_kernel_choice(ZT_CHOICE_EMBED_OUT, ..) pins one; the three give the
    // same results (tests/test_embed_gpu.py).
    const int oc = zt::kernel_choice(ZT_CHOICE_EMBED_OUT);
    const bool hg_ok = hg == 1 || hg == 5 || hg == 10;
    const bool can2 = D % 4 == 0 && (p.Dp == 112 || p.Dp == 128) && hg_ok;
    const bool can3 = D % 4 == 0 && p.Dp == EO3_DP && hg_ok;
    const bool use2 = can2 && (oc == ZT_OUT_LATENCY || (oc == 0 && N <= 1024));
    // (alone on the chip, tools/exp/p23_kernels.py: 12 288 rows persist 26 us / tiled 32 / latency 53 -- 29 / 38 / 66 on 192 CUs;
    //  3 000 rows 17 / 18 / 23; 1 800 rows with partial-sum groups 19 / 17.5 / 22; 600 rows 16.5 / 17 / 12)
    //  In the pipeline the persistent workgroups (130 KB of LDS, a whole register file per wave) do not fit beside another
    //  stream's kernels: C4 (3 000 rows, no CU masks, k_pruned_topk running beside) 36 us against the tiled kernel's 20; C5
    //  (12 288 rows, the main stream's CUs to itself) 29 against 38.  So: from 8 192 rows on.
    const bool use3 = can3 && !use2 && (oc == ZT_OUT_PERSIST || (oc == 0 && N >= 8192));
    const bool held_back = defer != nullptr && !use3;                    // (launched by the caller, beside the GRU update)
    if (!held_back) ZT_PROF_BEGIN(s, P_EMBED_OUT);
    if (use2) {
        // one wave per (16 rows, path, N-tile), striding over the row tiles with its weights in registers
        const long long tiles = (N + 15) / 16;
        zt::embed_out_deferred od;
        od.valid = true; od.form = 2; od.gx = (int)(tiles < 256 ? tiles : 256);
        od.memory = memory_dev; od.num_nodes = num_nodes; od.nodes = nodes_dev; od.N = N; od.D = D; od.M = M; od.hg = hg;
        od.H = H; od.S = S; od.fc2_p = fc2t; od.fc2_b = wt->fc2_b; od.fc1s_p = fc1st; od.fc1s_b = wt->fc1s_b; od.fc2s_p = fc2st;
        od.fc2s_b = wt->fc2s_b; od.out = out_dev; od.status = status_dev; od.latch = nullptr;
        if (defer != nullptr) {
            int *keep = defer->latch;
            *defer = od;
            defer->latch = keep;
            ZT_LAUNCH_CHECK();
            return ZT_OK;
        }
        const int rc = zt::embed_out_launch(od, s);
        if (rc != ZT_OK) return rc;
    } else if (use3) {
        // one workgroup per CU of the stream; the source path's two layers are as many MFMAs per row as M models' fc2
        const long long tiles = (N + 15) / 16;
        long long wgs = zt::stream_cu_count(s);
        const long long want = (tiles * (M + 2) + AGG_WAVES - 1) / AGG_WAVES;       // (units of 196 MFMAs) / waves
        if (wgs > want) wgs = want;
        if (wgs < 2) wgs = 2;
        int n_src = (int)((2 * wgs + (M + 2) / 2) / (M + 2));
        if (n_src < 1) n_src = 1;
        if (n_src > wgs - 1) n_src = (int)wgs - 1;
        const size_t lds3 = (size_t)(2 * EO3_FRAG + AGG_WAVES * 16 * EO3_LDY) * 4;
        static bool attr3[3] = {false, false, false};
        const int hi = hg == 1 ? 0 : (hg == 5 ? 1 : 2);
        if (!attr3[hi]) {
            const void *fn = hg == 1 ? reinterpret_cast<const void *>(k_embed_out3<1>)
                                     : (hg == 5 ? reinterpret_cast<const void *>(k_embed_out3<5>) : reinterpret_cast<const void *>(k_embed_out3<10>));
            ZT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            attr3[hi] = true;
        }
#define ZT_EO3(HGV) k_embed_out3<HGV><<<(unsigned)wgs, AGG_THREADS, lds3, s>>>(memory_dev, num_nodes, nodes_dev, N, D, M, H, S, fc2t, \
            wt->fc2_b, fc1st, wt->fc1s_b, fc2st, wt->fc2s_b, out_dev, status_dev, n_src)
        if (hg == 1) ZT_EO3(1); else if (hg == 5) ZT_EO3(5); else ZT_EO3(10);
#undef ZT_EO3
    } else {
        if (hg != 1 && hg != 5 && hg != 10) { set_error("zt_embed: %d partial-sum groups per row", hg); return ZT_ERR_UNSUPPORTED; }
        zt::embed_out_deferred od;
        od.valid = true; od.form = 1; od.gx = 0;
        od.memory = memory_dev; od.num_nodes = num_nodes; od.nodes = nodes_dev; od.N = N; od.D = D; od.M = M; od.hg = hg;
        od.H = H; od.S = S; od.fc2_p = fc2t; od.fc2_b = wt->fc2_b; od.fc1s_p = fc1st; od.fc1s_b = wt->fc1s_b; od.fc2s_p = fc2st;
        od.fc2s_b = wt->fc2s_b; od.out = out_dev; od.status = status_dev; od.latch = nullptr;
        if (defer != nullptr) {
            // the caller launches them beside the GRU update (gru_update_ex) -- or by embed_out_launch
            int *keep = defer->latch;
            *defer = od;
            defer->latch = keep;
            ZT_LAUNCH_CHECK();
            return ZT_OK;
        }
        const int rc = zt::embed_out_launch(od, s);
        if (rc != ZT_OK) return rc;
    }
    ZT_PROF_END(s, P_EMBED_OUT);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_embed(const float *memory_dev, const float *efeat_dev, int64_t num_nodes, int64_t num_edges,
                        int32_t D, int32_t F, int32_t T, const int32_t *nodes_dev, int64_t N, int32_t M, int32_t k,
                        const int32_t *nbr_dev, const int32_t *eix_dev, const float *dt_dev, const float *w_dev,
                        const zt_embed_weights *wt, float *out_dev, void *workspace_dev, int32_t *status_dev,
                        const float *proj_table_dev, int32_t weights_ready, void *stream)
{
    return embed_impl(memory_dev, efeat_dev, num_nodes, num_edges, D, F, T, nodes_dev, N, M, k, nbr_dev, eix_dev, dt_dev, w_dev,
                      wt, out_dev, workspace_dev, status_dev, proj_table_dev, weights_ready, stream, nullptr);
}

int zt::embed_out_launch(const zt::embed_out_deferred &d, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int Dp = round_up(d.D, 16);
    const size_t lds2 = (size_t)2 * OUT_ROWS * (Dp + 4) * 4 + OUT_ROWS * 4;
    EmbedOutArgs E;
    E.memory = d.memory; E.num_nodes = d.num_nodes; E.nodes = d.nodes; E.N = d.N; E.D = d.D; E.M = d.M; E.H = d.H; E.S = d.S;
    E.fc2_p = d.fc2_p; E.fc2_b = d.fc2_b; E.fc1s_p = d.fc1s_p; E.fc1s_b = d.fc1s_b; E.fc2s_p = d.fc2s_p; E.fc2s_b = d.fc2s_b;
    E.out = d.out; E.status = d.status;
    if (d.form == 2) {
        const dim3 grid2((unsigned)d.gx, (unsigned)(d.M + 1), (unsigned)(Dp / 16));
#define ZT_EO2(NTV, HGV) k_embed_out2<NTV, HGV><<<grid2, 64, 0, s>>>(E)
        if (Dp == 112) { if (d.hg == 1) ZT_EO2(7, 1); else if (d.hg == 5) ZT_EO2(7, 5); else ZT_EO2(7, 10); }
        else           { if (d.hg == 1) ZT_EO2(8, 1); else if (d.hg == 5) ZT_EO2(8, 5); else ZT_EO2(8, 10); }
#undef ZT_EO2
        ZT_LAUNCH_CHECK();
        return ZT_OK;
    }
    const dim3 grid((unsigned)((d.N + OUT_ROWS - 1) / OUT_ROWS), (unsigned)(d.M + 1));
    if (d.hg == 1) k_embed_out<1><<<grid, AGG_THREADS, lds2, s>>>(E);
    else if (d.hg == 5) k_embed_out<5><<<grid, AGG_THREADS, lds2, s>>>(E);
    else if (d.hg == 10) k_embed_out<10><<<grid, AGG_THREADS, lds2, s>>>(E);
    else { set_error("zt_embed: %d partial-sum groups per row", d.hg); return ZT_ERR_UNSUPPORTED; }
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

int zt::embed_ex(const float *memory_dev, const float *efeat_dev, int64_t num_nodes, int64_t num_edges,
                 int32_t D, int32_t F, int32_t T, const int32_t *nodes_dev, int64_t N, int32_t M, int32_t k,
                 const int32_t *nbr_dev, const int32_t *eix_dev, const float *dt_dev, const float *w_dev,
                 const zt_embed_weights *wt, float *out_dev, void *workspace_dev, int32_t *status_dev,
                 const float *proj_table_dev, int32_t weights_ready, void *stream, hipEvent_t mid_wait,
                 zt::embed_out_deferred *defer, const zt::member_gate *gate)
{
    return embed_impl(memory_dev, efeat_dev, num_nodes, num_edges, D, F, T, nodes_dev, N, M, k, nbr_dev, eix_dev, dt_dev, w_dev,
                      wt, out_dev, workspace_dev, status_dev, proj_table_dev, weights_ready, stream, mid_wait, defer, gate);
}

// The wait for one batch's T-PPR rows as a kernel of its own: a wave that polls the batch's counter.  The kernel behind it on
// the stream starts with the acquire every kernel starts with and reads rows that were stored write-through.
namespace {
__global__ void k_member_gate(const int *word, int target, int *status, int *latch)
{
    if (threadIdx.x == 0) (void)member_gate_wait(word, target, status, latch);
}
}  // namespace

int zt::member_gate_launch(const zt::member_gate &g, int32_t *status_dev, void *stream)
{
    if (g.word == nullptr) return ZT_OK;
    // (The command processor could do the waiting itself -- hipStreamWaitValue32 works on plain device memory and on CU-masked
    //  streams, tools/exp/waitvalue_probe.hip, and the runtime implements it as a one-wave kernel of its own: measured, the step
    //  is the same to within the noise.  But that wait has no time limit, and a tool that serialises kernels -- rocprofv3's
    //  counter mode -- turns it into a hang where this kernel gives up after 4 s and reports.  Every wait of the library is
    //  bounded; so is this one.)
    k_member_gate<<<1, 64, 0, (hipStream_t)stream>>>(g.word, g.target, status_dev, g.latch);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

// Training forward of the neighbour half (backward: aggregate_bwd.hip):
//   H[m][n][:] = sum_k w_k/sum(w) relu(fc1([memory'[nbr] | ef | cos(dt w)])),  S[m][n] = (sum_k w != 0)
// where memory'[v] = overlay[row_map[v]] if row_map[v] >= 0 (rows of the lazily updated memory,
// modules/memory_updater.py:61-90) else memory[v].  fc2 and the source transform run on the [N, D] results
// outside (plain GEMMs).  workspace: zt_embed_workspace_bytes(N, D, F, T, M, k).
extern "C" int zt_agg_train_forward(const float *memory_dev, const float *overlay_dev, const int32_t *row_map_dev,
                                    const float *efeat_dev, int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F,
                                    int32_t T, int64_t N, int32_t M, int32_t k, const int32_t *nbr_dev,
                                    const int32_t *eix_dev, const float *dt_dev, const float *w_dev,
                                    const zt_embed_weights *wt, float *H_dev, float *S_dev, void *workspace_dev,
                                    int32_t *status_dev, float drop_p, uint64_t drop_seed, void *stream)
{
    if (!memory_dev || !efeat_dev || !wt || !status_dev || !H_dev || !S_dev || !workspace_dev || N < 0 || D <= 0 || F < 0 ||
        T < 0 || M <= 0 || k <= 0 || (row_map_dev != nullptr && !overlay_dev) || !(drop_p >= 0.f && drop_p < 1.f)) {
        set_error("zt_agg_train_forward: bad argument");
        return ZT_ERR_ARG;
    }
    if (N == 0) return ZT_OK;
    EmbedPlan p;
    if (D > 128 || !make_plan(N, D, F, T, M, k, p)) {
        set_error("zt_agg_train_forward: unsupported shape");
        return ZT_ERR_UNSUPPORTED;
    }
    if (p.big) {
        set_error("zt_agg_train_forward: k=%d: the training kernels hold a query row of at most %d neighbours", k, MAX_MT * 16);
        return ZT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace_dev);
    embed_prepare(wt, D, F, T, p, ws, s);          // the weights change every optimizer step
    static size_t attr_lds = 0;
    if (p.lds > 48 * 1024 && p.lds > attr_lds) {
        ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fc1_agg<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
        attr_lds = p.lds;
    }
    dim3 grid((unsigned)((N + p.rq - 1) / p.rq), (unsigned)M);
    k_fc1_agg<false><<<grid, AGG_THREADS, p.lds, s>>>(memory_dev, efeat_dev, wt->time_w, num_nodes, num_edges, D, F, T, N, k,
                                                    p.rq, p.mt, p.lda, nbr_dev, eix_dev, dt_dev, w_dev,
                                                    reinterpret_cast<const float *>(ws + p.off_w1p), p.K1p, wt->fc1_b, H_dev,
                                                    S_dev, status_dev, D, row_map_dev, overlay_dev, (unsigned)drop_seed,
                                                    (unsigned)(drop_seed >> 32), zt::drop_threshold(drop_p),
                                                    1.f / (1.f - drop_p));
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
