// Training path of the top-k neighbour aggregation: forward with a sparse "updated rows"
// overlay and the fused backward.
//
// Reference (modules/embedding_module.py:227-276, train=True): `memory` is the lazily updated copy of the whole
// memory table (get_updated_memory clones [N, D] every batch, modules/memory_updater.py:79) and autograd flows
//   loss -> embeddings -> fc2 -> sum_k w_k relu(fc1([memory'[nbr] | ef | cos])) -> fc1, memory'[ids] -> GRU weights.
// Here the lazily updated rows live in a compact OVERLAY [U, D] (row_map[v] = overlay row of node v, or -1: the
// stored memory row, which carries no gradient); nothing of size [N, D] or [N, k, 2D+F] is ever materialised.
//
//   zt_agg_train_forward : H[m][n][:] = sum_k w_k relu(fc1(x_k)), S[m][n] = (sum w != 0)     (k_fc1_agg<false> + overlay)
//   zt_agg_train_backward: given dH, recomputes the pre-activations tile by tile and accumulates
//        dW1 += dpre^T x,   db1 += sum dpre,   d_overlay[row_map[nbr]] += dpre W_m      with dpre = w_k dH 1[pre > 0]
//     on f32 MFMA: three products per tile (the recompute and the two gradients), the [D, K1] weight gradient
//     held in registers across the tiles of a persistent workgroup and added to HBM once at the end.
#include "common.hpp"

using namespace zt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BW_THREADS = 256;
constexpr int BW_WAVES = 4;
constexpr int BW_MT = 5;           // 16-row tiles of gathered rows per step (80 rows)
constexpr int BW_NTW = 2;          // forward N-tiles per wave (D <= 128)
constexpr int BW_MAX_OT = 24;      // dW1 output tiles per wave and launch (96 accumulator registers)

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct BwdArgs {
    const float *memory, *overlay;     // [num_nodes][D], [U][D]
    const int *row_map;                // [num_nodes]: overlay row or -1
    const float *efeat, *time_w;
    long long num_nodes, num_edges, N;
    int D, F, T, k, M, rq, lda, ldp, K1p, Dp;
    const int *nbr, *eix;
    const float *dt, *w;
    const float *W1p;                  // [Dp][K1p] zero padded fc1 weight
    const float *b1;
    const float *dH;                   // [M][N][D]
    float *dW1, *db1, *d_overlay;      // [D][K1], [D], [U][D]  (accumulated with atomics)
    long long n_tiles;                 // tiles of rq query rows per model
    int mt;                            // 16-row tiles per step (<= BW_MT)
    int kt_lo, kt_hi;                  // this launch accumulates dW1 column tiles [kt_lo, kt_hi)
    int first;                         // 1: this launch also accumulates db1 and d_overlay
    unsigned drop_lo, drop_hi, drop_thr;   // training dropout of the hidden layer (common.hpp: drop_scale); thr = 0: none
    float drop_inv;
};

__global__ __launch_bounds__(BW_THREADS) void k_fc1_agg_bwd(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = a.D, F = a.F, T = a.T, k = a.k, lda = a.lda, ldp = a.ldp, K1p = a.K1p, Dp = a.Dp;
    const int K1 = D + F + T;
    const int mt = a.mt, rows_p = mt * 16;
    float *A = reinterpret_cast<float *>(smem);                 // [rows_p][lda]   gathered rows
    float *P = A + (size_t)rows_p * lda;                        // [rows_p][ldp]   dpre
    float *wn = P + (size_t)rows_p * ldp;                       // [rows_p]
    int *g_src = reinterpret_cast<int *>(wn + rows_p);          // >= 0: memory row, < 0: -(overlay row) - 1
    int *g_ei = g_src + rows_p;
    float *g_dt = reinterpret_cast<float *>(g_ei + rows_p);
    float *tw = g_dt + rows_p;                                  // [T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int NT = Dp / 16, KT = a.kt_hi - a.kt_lo;
    const int n_ot = NT * KT;                                   // 16x16 tiles of dW1 handled by this launch
    f32x4 gw[BW_MAX_OT];                                        // this wave's dW1 tiles (tile t = wave + 4*q)
#pragma unroll
    for (int q = 0; q < BW_MAX_OT; ++q) gw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;                                             // db1[tid] (tid < Dp)
    for (int c = tid; c < T; c += BW_THREADS) tw[c] = a.time_w[c];

    const long long total = a.n_tiles * a.M;
    for (long long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int m = (int)(tile / a.n_tiles);
        const long long q0 = (tile % a.n_tiles) * a.rq;
        const int nq = (int)((a.N - q0) < a.rq ? (a.N - q0) : a.rq);
        const int rows = nq * k;
        const size_t mb = ((size_t)m * a.N + q0) * k;
        __syncthreads();                                        // previous tile fully consumed
        for (int g = tid; g < rows_p; g += BW_THREADS) {
            int src = 0, ei = 0;
            float d = 0.f, wv = 0.f;
            if (g < rows) {
                int nb = a.nbr[mb + g];
                ei = a.eix[mb + g]; d = a.dt[mb + g]; wv = a.w[mb + g];
                if (nb < 0 || nb >= a.num_nodes || ei < 0 || ei >= a.num_edges) { nb = 0; ei = 0; wv = 0.f; }
                const int ov = a.row_map ? a.row_map[nb] : -1;
                src = ov >= 0 ? -ov - 1 : nb;
            }
            g_src[g] = src; g_ei[g] = ei; g_dt[g] = d; wn[g] = wv;
        }
        __syncthreads();
        float my_sum = 0.f;
        if (tid < rows) {
            const int q = tid / k;
            for (int j = 0; j < k; ++j) my_sum += wn[q * k + j];
        }
        __syncthreads();
        if (tid < rows) wn[tid] = (my_sum == 0.f) ? 0.f : wn[tid] / my_sum;
        // ---- gather x = [memory' | ef | cos] into the tile ----
        for (int f = tid; f < rows_p * K1p; f += BW_THREADS) {
            const int g = f / K1p, c = f - g * K1p;
            float v = 0.f;
            if (g < rows && c < K1) {
                if (c < D) {
                    const int s = g_src[g];
                    v = s >= 0 ? a.memory[(size_t)s * D + c] : a.overlay[(size_t)(-s - 1) * D + c];
                } else if (c < D + F) {
                    v = a.efeat[(size_t)g_ei[g] * F + (c - D)];
                } else {
                    v = time_cosf(g_dt[g] * tw[c - D - F]);
                }
            }
            A[(size_t)g * lda + c] = v;
        }
        __syncthreads();
        // ---- recompute pre = x W1^T (f32 MFMA; wave owns N-tiles {wave, wave+4}) ----
        f32x4 acc[BW_MT][BW_NTW];
#pragma unroll
        for (int x = 0; x < BW_MT; ++x)
#pragma unroll
            for (int b = 0; b < BW_NTW; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < K1p / 16; ++kc) {
            f32x4 av[BW_MT], bv[BW_NTW];
#pragma unroll
            for (int x = 0; x < BW_MT; ++x)
                av[x] = x < mt ? *reinterpret_cast<const f32x4 *>(A + (size_t)(x * 16 + r16) * lda + 16 * kc + 4 * g4)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < BW_NTW; ++b) {
                const int nt = wave + b * BW_WAVES;
                bv[b] = nt < NT ? *reinterpret_cast<const f32x4 *>(a.W1p + (size_t)(nt * 16 + r16) * K1p + 16 * kc + 4 * g4)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < BW_MT; ++x)
#pragma unroll
                    for (int b = 0; b < BW_NTW; ++b)
                        if (x < mt) acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x][j], bv[b][j], acc[x][b], 0, 0, 0);
        }
        // ---- dpre = w_k dH 1[pre > 0] into P ----
#pragma unroll
        for (int b = 0; b < BW_NTW; ++b) {
            const int nt = wave + b * BW_WAVES;
            if (nt >= NT) continue;
            const int col = nt * 16 + r16;
            const float bias = col < D ? a.b1[col] : 0.f;
#pragma unroll
            for (int x = 0; x < BW_MT; ++x)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (x >= mt) continue;
                    const int g = x * 16 + g4 * 4 + j;
                    float dp = 0.f;
                    if (g < rows && col < D && acc[x][b][j] + bias > 0.f) {
                        dp = wn[g] * a.dH[((size_t)m * a.N + q0 + g / k) * D + col];
                        if (a.drop_thr != 0u)
                            dp *= drop_scale(a.drop_lo, a.drop_hi, a.drop_thr, a.drop_inv,
                                             (unsigned long long)(((size_t)m * a.N + q0) * k + g) * D + col);
                    }
                    P[(size_t)g * ldp + col] = dp;
                }
        }
        __syncthreads();
        // ---- db1 ----
        if (a.first && tid < Dp) {
            float s = 0.f;
            for (int g = 0; g < rows; ++g) s += P[(size_t)g * ldp + tid];
            gb += s;
        }
        // ---- dW1[i][c] += sum_g P[g][i] x[g][c]: tiles t = wave, wave+4, ...; tile t = (dt, kt) ----
#pragma unroll
        for (int q = 0; q < BW_MAX_OT; ++q) {
            const int t = wave + q * BW_WAVES;
            if (t >= n_ot) break;
            const int dtile = t / KT, ktile = a.kt_lo + t - dtile * KT;
            f32x4 c4 = gw[q];
            for (int s = 0; s < rows_p / 4; ++s) {
                const int g = 4 * s + g4;
                const float pa = P[(size_t)g * ldp + dtile * 16 + r16];        // A operand: [m = D column][k = g]
                const float xb = A[(size_t)g * lda + ktile * 16 + r16];        // B operand: [k = g][n = input column]
                c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, xb, c4, 0, 0, 0);
            }
            gw[q] = c4;
        }
        // ---- d_overlay[row][c] += sum_col P[g][col] W1[col][c]  (memory columns c < D only) ----
        if (a.first && a.row_map != nullptr) {
            f32x4 dx[BW_MT][BW_NTW];
#pragma unroll
            for (int x = 0; x < BW_MT; ++x)
#pragma unroll
                for (int b = 0; b < BW_NTW; ++b) dx[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < Dp / 4; ++s) {
                const int col = 4 * s + g4;
                float pa[BW_MT], wb[BW_NTW];
#pragma unroll
                for (int x = 0; x < BW_MT; ++x) pa[x] = x < mt ? P[(size_t)(x * 16 + r16) * ldp + col] : 0.f;   // [m = row g][k = col]
#pragma unroll
                for (int b = 0; b < BW_NTW; ++b) {
                    const int c = (wave + b * BW_WAVES) * 16 + r16;
                    wb[b] = c < D ? a.W1p[(size_t)col * K1p + c] : 0.f;                           // [k = col][n = c]
                }
#pragma unroll
                for (int x = 0; x < BW_MT; ++x)
#pragma unroll
                    for (int b = 0; b < BW_NTW; ++b)
                        if (x < mt) dx[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[x], wb[b], dx[x][b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < BW_NTW; ++b) {
                const int c = (wave + b * BW_WAVES) * 16 + r16;
                if (c >= D) continue;
#pragma unroll
                for (int x = 0; x < BW_MT; ++x)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int g = x * 16 + g4 * 4 + j;
                        if (x >= mt || g >= rows) continue;
                        const int s = g_src[g];
                        if (s < 0 && dx[x][b][j] != 0.f) atomicAdd(a.d_overlay + (size_t)(-s - 1) * D + c, dx[x][b][j]);
                    }
            }
        }
    }
    // ---- flush the accumulated weight gradient ----
#pragma unroll
    for (int q = 0; q < BW_MAX_OT; ++q) {
        const int t = wave + q * BW_WAVES;
        if (t >= n_ot) break;
        const int dtile = t / KT, ktile = a.kt_lo + t - dtile * KT;
        const int c = ktile * 16 + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = dtile * 16 + g4 * 4 + j;
            if (i < D && c < K1 && gw[q][j] != 0.f) atomicAdd(a.dW1 + (size_t)i * K1 + c, gw[q][j]);
        }
    }
    if (a.first && tid < D && gb != 0.f) atomicAdd(a.db1 + tid, gb);
}

// Zero-padded copy W[rows][cols] -> Wp[rows_p][cols_p]
__global__ void k_pad(const float *__restrict__ W, int rows, int cols, float *__restrict__ Wp, int rows_p, int cols_p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_p * cols_p) return;
    const int r = i / cols_p, c = i % cols_p;
    Wp[i] = (r < rows && c < cols) ? W[(size_t)r * cols + c] : 0.f;
}

}  // namespace

extern "C" int64_t zt_agg_backward_workspace_bytes(int32_t D, int32_t F, int32_t T)
{
    if (D <= 0 || F < 0 || T < 0) return -1;
    return (int64_t)round_up(D, 16) * round_up(D + F + T, 16) * 4;
}

extern "C" int zt_agg_train_backward(const float *memory_dev, const float *overlay_dev, const int32_t *row_map_dev,
                                     const float *efeat_dev, const float *time_w_dev, int64_t num_nodes, int64_t num_edges,
                                     int32_t D, int32_t F, int32_t T, int64_t N, int32_t M, int32_t k,
                                     const int32_t *nbr_dev, const int32_t *eix_dev, const float *dt_dev, const float *w_dev,
                                     const float *fc1_w_dev, const float *fc1_b_dev, const float *dH_dev, float *dW1_dev,
                                     float *db1_dev, float *d_overlay_dev, void *workspace_dev, float drop_p,
                                     uint64_t drop_seed, void *stream)
{
    if (!memory_dev || !efeat_dev || !time_w_dev || !nbr_dev || !eix_dev || !dt_dev || !w_dev || !fc1_w_dev || !fc1_b_dev ||
        !dH_dev || !dW1_dev || !db1_dev || !workspace_dev || N < 0 || D <= 0 || F < 0 || T < 0 || M <= 0 || k <= 0 ||
        (row_map_dev != nullptr && (!overlay_dev || !d_overlay_dev))) {
        set_error("zt_agg_train_backward: bad argument");
        return ZT_ERR_ARG;
    }
    if (N == 0) return ZT_OK;
    const int K1 = D + F + T, Dp = round_up(D, 16), K1p = round_up(K1, 16);
    // tile: as many whole query rows as fit BW_MT 16-row tiles and 150 KB of LDS
    int mt = BW_MT;
    auto lds_of = [&](int t) { return ((size_t)t * 16 * (K1p + 4 + Dp + 4) + (size_t)t * 16 * 4 + T) * 4; };
    while (mt > 1 && lds_of(mt) > 150 * 1024) --mt;
    int rq = (mt * 16) / k;
    if (rq < 1) { mt = (k + 15) / 16; rq = 1; }
    if (D > 128 || mt > BW_MT || lds_of(mt) > 150 * 1024) {
        set_error("zt_agg_train_backward: D=%d F=%d T=%d k=%d outside the supported shapes", D, F, T, k);
        return ZT_ERR_UNSUPPORTED;
    }
    mt = (rq * k + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    float *W1p = reinterpret_cast<float *>(workspace_dev);
    k_pad<<<(Dp * K1p + 255) / 256, 256, 0, s>>>(fc1_w_dev, D, K1, W1p, Dp, K1p);
    BwdArgs a;
    a.memory = memory_dev; a.overlay = overlay_dev; a.row_map = row_map_dev; a.efeat = efeat_dev; a.time_w = time_w_dev;
    a.drop_lo = (unsigned)drop_seed; a.drop_hi = (unsigned)(drop_seed >> 32);
    a.drop_thr = (drop_p > 0.f && drop_p < 1.f) ? zt::drop_threshold(drop_p) : 0u;
    a.drop_inv = a.drop_thr ? 1.f / (1.f - drop_p) : 1.f;
    a.num_nodes = num_nodes; a.num_edges = num_edges; a.N = N;
    a.D = D; a.F = F; a.T = T; a.k = k; a.M = M; a.rq = rq; a.lda = K1p + 4; a.ldp = Dp + 4; a.K1p = K1p; a.Dp = Dp;
    a.nbr = nbr_dev; a.eix = eix_dev; a.dt = dt_dev; a.w = w_dev; a.W1p = W1p; a.b1 = fc1_b_dev; a.dH = dH_dev;
    a.dW1 = dW1_dev; a.db1 = db1_dev; a.d_overlay = d_overlay_dev;
    a.n_tiles = (N + rq - 1) / rq;
    a.mt = mt;
    const size_t lds = lds_of(mt);
    static size_t attr_lds = 0;
    if (lds > 48 * 1024 && lds > attr_lds) {
        ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fc1_agg_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_lds = lds;
    }
    hipDeviceProp_t prop;
    int dev = 0;
    ZT_HIP(hipGetDevice(&dev));
    ZT_HIP(hipGetDeviceProperties(&prop, dev));
    long long grid = a.n_tiles * M;
    if (grid > prop.multiProcessorCount) grid = prop.multiProcessorCount;      // persistent: one weight-gradient flush per CU
    // the weight gradient of one launch lives in registers: BW_MAX_OT tiles per wave.  Wide inputs (F = 172)
    // take several launches over windows of input columns (the recompute is repeated, db1 / d_overlay are not).
    const int NT = Dp / 16, KT = K1p / 16;
    const int win = (BW_MAX_OT * BW_WAVES) / NT;
    for (int lo = 0; lo < KT; lo += win) {
        a.kt_lo = lo; a.kt_hi = lo + win < KT ? lo + win : KT; a.first = lo == 0 ? 1 : 0;
        k_fc1_agg_bwd<<<(unsigned)grid, BW_THREADS, lds, s>>>(a);
    }
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
