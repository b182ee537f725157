// TemporalAttentionLayer forward (reference model/temporal_attention.py:7-68).
//
// NOT on the reference's live path: the layer is only instantiated by
// GraphAttentionEmbedding, which train.py can never reach, so
// there are no reference call sites or outputs; parity is against a torch
// restatement (nn.MultiheadAttention + MergeLayer, zebra_amd/modules.py).
// Built because BASELINE.json's north_star names it.
//
//   q = [src | src_time]  (1 token, E = D+T);  k = v = [nbr | edge | nbr_time]  (n_nbr tokens, Ek = D+F+T)
//   nn.MultiheadAttention(E, heads, kdim = vdim = Ek) with key_padding_mask; rows whose neighbours are all
//   padding get slot 0 unmasked and their output zeroed; then MergeLayer([attn_out | src]).
//
// One workgroup per tile of RQ query rows.  The RQ*n_nbr key rows are staged in LDS once and go through the
// K and the V projection on exact-f32 MFMA (v_mfma_f32_16x16x4_f32) -- the batched QKV contraction is where
// the FLOPs are; scores, softmax, the weighted sum and the small output layers run on the same LDS-to-LDS
// building block.  Eval forward only (dropout = identity).
#include "common.hpp"

#include <algorithm>

using namespace zt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_THREADS = 256;
constexpr int ATT_WAVES = 4;
constexpr int ATT_MAX_MT = 5;

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

__global__ void k_pad_w(const float *__restrict__ W, int rows, int cols, int ld_src, int col0, float *__restrict__ Wp,
                        int rows_p, int cols_p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_p * cols_p) return;
    const int r = i / cols_p, c = i % cols_p;
    Wp[i] = (r < rows && c < cols) ? W[(size_t)r * ld_src + col0 + c] : 0.f;
}

// Y[g][col] = act((sum_i X[g][i] * Wp[col][i] + bias[col]) * scale), g < mt*16, col < NT*16 (zero for
// col >= n_out).  X, Y in LDS; Wp zero-padded [NT*16][Kp] in global (L2); wave w owns N-tiles w, w+4, ...
__device__ inline void lds_linear(const float *X, int ldx, int mt, const float *__restrict__ Wp, int Kp, int n_out,
                                  int NT, const float *__restrict__ bias, float scale, bool relu, float *Y, int ldy)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    for (int nt = wave; nt < NT; nt += ATT_WAVES) {
        f32x4 acc[ATT_MAX_MT];
#pragma unroll
        for (int a = 0; a < ATT_MAX_MT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *bp = Wp + (size_t)(nt * 16 + r16) * Kp + 4 * g4;
        for (int kc = 0; kc < Kp / 16; ++kc) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bp + 16 * kc);
#pragma unroll
            for (int a = 0; a < ATT_MAX_MT; ++a) {
                if (a < mt) {
                    const f32x4 av = *reinterpret_cast<const f32x4 *>(X + (size_t)(a * 16 + r16) * ldx + 16 * kc + 4 * g4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[a], 0, 0, 0);
                }
            }
        }
        const int col = nt * 16 + r16;
        const float b = (col < n_out && bias) ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < ATT_MAX_MT; ++a) {
            if (a < mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = (acc[a][j] + b) * scale;
                    if (relu) v = v > 0.f ? v : 0.f;
                    Y[(size_t)(a * 16 + g4 * 4 + j) * ldy + col] = col < n_out ? v : 0.f;
                }
            }
        }
    }
}

struct AttnDims {
    int D, F, T, E, Ek, heads, hd, hidden, out_dim, k;
    int Ep, Ekp, E2p, Hp, Op;       // padded widths: E, Ek, E+D, hidden, out_dim
    int rq, mt, lda, ldk, ldq;
};

__global__ __launch_bounds__(ATT_THREADS) void k_temporal_attention(
    AttnDims d, long long N, const float *__restrict__ src, const float *__restrict__ src_t,
    const float *__restrict__ nf, const float *__restrict__ ef, const float *__restrict__ ntf,
    const unsigned char *__restrict__ mask, const float *__restrict__ Wq, const float *__restrict__ Wk,
    const float *__restrict__ Wv, const float *__restrict__ in_b, const float *__restrict__ Wo,
    const float *__restrict__ out_b, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, float *__restrict__ out, float *__restrict__ attn_w)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int rows_p = d.mt * 16;
    float *A = reinterpret_cast<float *>(smem);                  // [rows_p][lda]  key rows
    float *KV = A + (size_t)rows_p * d.lda;                      // [rows_p][ldk]  projected K, then V
    const size_t kv_words = (size_t)rows_p * d.ldk > (size_t)16 * d.ldq ? (size_t)rows_p * d.ldk : (size_t)16 * d.ldq;
    float *Xq = KV + kv_words;                                   // [16][ldq]      query rows / attention output
    float *Qp = Xq + 16 * d.ldq;                                 // [16][ldq]      projected, scaled queries; later scratch
    float *Z = KV;                                               // [16][ldq]      [attn_out | src] for the merger (K/V are dead by then)
    float *sc = Qp + 16 * d.ldq;                                 // [rows_p][heads] scores -> softmax weights
    int *inv = reinterpret_cast<int *>(sc + rows_p * 4);         // [16] row has no real neighbour
    const long long q0 = (long long)blockIdx.x * d.rq;
    const int nq = (int)((N - q0) < d.rq ? (N - q0) : d.rq);
    const int nrow = nq * d.k;
    const int k = d.k, E = d.E;

    // ---- stage inputs ----
    if (tid < 16) {
        int all = 1;
        if (tid < nq) for (int j = 0; j < k; ++j) all &= mask[(q0 + tid) * k + j] ? 1 : 0;
        inv[tid] = (tid < nq) ? all : 1;
    }
    for (int f = tid; f < rows_p * d.lda; f += ATT_THREADS) {
        const int g = f / d.lda, c = f - g * d.lda;
        float v = 0.f;
        if (g < nrow) {
            const size_t row = (size_t)q0 * k + g;
            if (c < d.D) v = nf[row * d.D + c];
            else if (c < d.D + d.F) v = ef[row * d.F + (c - d.D)];
            else if (c < d.Ek) v = ntf[row * d.T + (c - d.D - d.F)];
        }
        A[f] = v;
    }
    for (int f = tid; f < 16 * d.ldq; f += ATT_THREADS) {
        const int q = f / d.ldq, c = f - q * d.ldq;
        float v = 0.f;
        if (q < nq) {
            if (c < d.D) v = src[(q0 + q) * d.D + c];
            else if (c < E) v = src_t[(q0 + q) * d.T + (c - d.D)];
        }
        Xq[f] = v;
    }
    __syncthreads();
    // ---- q = (Wq x + bq) / sqrt(head_dim);  K = Wk key + bk ----
    lds_linear(Xq, d.ldq, 1, Wq, d.Ep, E, d.Ep / 16, in_b, rsqrtf((float)d.hd), false, Qp, d.ldq);
    lds_linear(A, d.lda, d.mt, Wk, d.Ekp, E, d.Ep / 16, in_b + E, 1.f, false, KV, d.ldk);
    __syncthreads();
    // ---- scores, key-padding mask, softmax over the neighbours ----
    for (int f = tid; f < rows_p * d.heads; f += ATT_THREADS) {
        const int g = f / d.heads, h = f - g * d.heads;
        float s = -INFINITY;
        if (g < nrow) {
            const int q = g / k, j = g - q * k;
            const bool masked = mask[(q0 + q) * k + j] != 0 && !(inv[q] && j == 0);
            if (!masked) {
                s = 0.f;
                const float *qr = Qp + (size_t)q * d.ldq + h * d.hd, *kr = KV + (size_t)g * d.ldk + h * d.hd;
                for (int c = 0; c < d.hd; ++c) s += qr[c] * kr[c];
            }
        }
        sc[f] = s;
    }
    __syncthreads();
    for (int f = tid; f < nq * d.heads; f += ATT_THREADS) {
        const int q = f / d.heads, h = f - q * d.heads;
        float mx = -INFINITY;
        for (int j = 0; j < k; ++j) mx = fmaxf(mx, sc[(q * k + j) * d.heads + h]);
        float sum = 0.f;
        for (int j = 0; j < k; ++j) sum += expf(sc[(q * k + j) * d.heads + h] - mx);
        for (int j = 0; j < k; ++j) sc[(q * k + j) * d.heads + h] = expf(sc[(q * k + j) * d.heads + h] - mx) / sum;
    }
    __syncthreads();
    // attention weights averaged over heads, zero for rows without neighbours (:60-63)
    for (int f = tid; f < nrow; f += ATT_THREADS) {
        const int q = f / k;
        float a = 0.f;
        for (int h = 0; h < d.heads; ++h) a += sc[f * d.heads + h];
        attn_w[(size_t)q0 * k + f] = inv[q] ? 0.f : a / (float)d.heads;
    }
    // ---- V = Wv key + bv (overwrites K), weighted sum over the neighbours ----
    lds_linear(A, d.lda, d.mt, Wv, d.Ekp, E, d.Ep / 16, in_b + 2 * E, 1.f, false, KV, d.ldk);
    __syncthreads();
    for (int f = tid; f < 16 * d.ldq; f += ATT_THREADS) {
        const int q = f / d.ldq, c = f - q * d.ldq;
        float v = 0.f;
        if (q < nq && c < E) {
            const int h = c / d.hd;
            for (int j = 0; j < k; ++j) v += sc[(q * k + j) * d.heads + h] * KV[(size_t)(q * k + j) * d.ldk + c];
        }
        Xq[f] = v;
    }
    __syncthreads();
    // ---- out_proj, zero rows without neighbours, merger([attn_out | src]) ----
    lds_linear(Xq, d.ldq, 1, Wo, d.Ep, E, d.Ep / 16, out_b, 1.f, false, Qp, d.ldq);
    __syncthreads();
    for (int f = tid; f < 16 * d.ldq; f += ATT_THREADS) {
        const int q = f / d.ldq, c = f - q * d.ldq;
        float v = 0.f;
        if (q < nq) {
            if (c < E) v = inv[q] ? 0.f : Qp[f];
            else if (c < E + d.D) v = src[(q0 + q) * d.D + (c - E)];
        }
        Z[f] = v;
    }
    __syncthreads();
    lds_linear(Z, d.ldq, 1, W1, d.E2p, d.hidden, d.Hp / 16, b1, 1.f, true, Xq, d.ldq);
    __syncthreads();
    lds_linear(Xq, d.ldq, 1, W2, d.Hp, d.out_dim, d.Op / 16, b2, 1.f, false, Qp, d.ldq);
    __syncthreads();
    for (int f = tid; f < nq * d.out_dim; f += ATT_THREADS) {
        const int q = f / d.out_dim, c = f - q * d.out_dim;
        out[(size_t)(q0 + q) * d.out_dim + c] = Qp[(size_t)q * d.ldq + c];
    }
}

bool attn_plan(int D, int F, int T, int heads, int hidden, int out_dim, int k, AttnDims &d, size_t &lds, size_t off[8],
               size_t &ws)
{
    d.D = D; d.F = F; d.T = T; d.E = D + T; d.Ek = D + F + T; d.heads = heads; d.hidden = hidden; d.out_dim = out_dim;
    d.k = k;
    if (heads <= 0 || heads > 4 || d.E % heads) return false;
    d.hd = d.E / heads;
    d.Ep = round_up(d.E, 16); d.Ekp = round_up(d.Ek, 16); d.E2p = round_up(d.E + D, 16);
    d.Hp = round_up(hidden, 16); d.Op = round_up(out_dim, 16);
    d.lda = d.Ekp + 4; d.ldk = d.Ep + 4;
    d.ldq = std::max(std::max(d.E2p, d.Ep), std::max(d.Hp, d.Op)) + 4;
    int mt = ATT_MAX_MT;
    auto bytes = [&](int m) {
        const size_t kv = std::max((size_t)m * 16 * d.ldk, (size_t)16 * d.ldq);   // K/V tile, later the merger input
        return ((size_t)m * 16 * d.lda + kv) * 4 + (size_t)2 * 16 * d.ldq * 4 + (size_t)m * 16 * 4 * 4 + 64;
    };
    while (mt > 1 && bytes(mt) > 158 * 1024) --mt;
    int rq = (mt * 16) / k;
    if (rq < 1) {
        mt = (k + 15) / 16; rq = 1;
        if (mt > ATT_MAX_MT || bytes(mt) > 158 * 1024) return false;
    }
    if (rq > 16) rq = 16;
    mt = (rq * k + 15) / 16;
    d.rq = rq; d.mt = mt;
    lds = bytes(mt);
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += (b + 255) & ~(size_t)255; return r; };
    off[0] = take((size_t)d.Ep * d.Ep * 4);    // Wq
    off[1] = take((size_t)d.Ep * d.Ekp * 4);   // Wk
    off[2] = take((size_t)d.Ep * d.Ekp * 4);   // Wv
    off[3] = take((size_t)d.Ep * d.Ep * 4);    // Wo
    off[4] = take((size_t)d.Hp * d.E2p * 4);   // merger.fc1
    off[5] = take((size_t)d.Op * d.Hp * 4);    // merger.fc2
    ws = o;
    return true;
}

}  // namespace

extern "C" int64_t zt_attention_workspace_bytes(int32_t D, int32_t F, int32_t T, int32_t n_head, int32_t hidden,
                                                int32_t out_dim, int32_t k)
{
    AttnDims d;
    size_t lds, off[8], ws;
    if (D <= 0 || F < 0 || T <= 0 || k <= 0 || hidden <= 0 || out_dim <= 0) return -1;
    if (!attn_plan(D, F, T, n_head, hidden, out_dim, k, d, lds, off, ws)) return -1;
    return (int64_t)ws;
}

extern "C" int zt_temporal_attention(const float *src_dev, const float *src_time_dev, const float *nbr_feat_dev,
                                     const float *edge_feat_dev, const float *nbr_time_dev, const uint8_t *mask_dev,
                                     int64_t N, int32_t k, int32_t D, int32_t F, int32_t T, int32_t n_head,
                                     int32_t hidden, int32_t out_dim, const zt_attn_weights *w, float *out_dev,
                                     float *attn_w_dev, void *workspace_dev, void *stream)
{
    if (!w || N < 0 || k <= 0 || D <= 0 || F < 0 || T <= 0) { set_error("zt_temporal_attention: bad argument"); return ZT_ERR_ARG; }
    if (N == 0) return ZT_OK;
    if (!src_dev || !src_time_dev || !nbr_feat_dev || !nbr_time_dev || (F > 0 && !edge_feat_dev) || !mask_dev ||
        !out_dev || !attn_w_dev || !workspace_dev) {
        set_error("zt_temporal_attention: NULL buffer");
        return ZT_ERR_ARG;
    }
    AttnDims d;
    size_t lds, off[8], ws;
    if (!attn_plan(D, F, T, n_head, hidden, out_dim, k, d, lds, off, ws)) {
        set_error("zt_temporal_attention: unsupported shape (heads<=4, E %% heads == 0, k rows must fit the LDS tile)");
        return ZT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    char *wsb = reinterpret_cast<char *>(workspace_dev);
    float *Wq = reinterpret_cast<float *>(wsb + off[0]), *Wk = reinterpret_cast<float *>(wsb + off[1]);
    float *Wv = reinterpret_cast<float *>(wsb + off[2]), *Wo = reinterpret_cast<float *>(wsb + off[3]);
    float *W1 = reinterpret_cast<float *>(wsb + off[4]), *W2 = reinterpret_cast<float *>(wsb + off[5]);
    auto pad = [&](const float *W, int rows, int cols, float *Wp, int rp, int cp) {
        k_pad_w<<<(rp * cp + 255) / 256, 256, 0, s>>>(W, rows, cols, cols, 0, Wp, rp, cp);
    };
    pad(w->q_w, d.E, d.E, Wq, d.Ep, d.Ep);
    pad(w->k_w, d.E, d.Ek, Wk, d.Ep, d.Ekp);
    pad(w->v_w, d.E, d.Ek, Wv, d.Ep, d.Ekp);
    pad(w->out_w, d.E, d.E, Wo, d.Ep, d.Ep);
    pad(w->m1_w, hidden, d.E + D, W1, d.Hp, d.E2p);
    pad(w->m2_w, out_dim, hidden, W2, d.Op, d.Hp);
    static size_t attr_lds = 0;
    if (lds > 48 * 1024 && lds > attr_lds) {
        ZT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_temporal_attention),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const unsigned grid = (unsigned)((N + d.rq - 1) / d.rq);
    k_temporal_attention<<<grid, ATT_THREADS, lds, s>>>(d, N, src_dev, src_time_dev, nbr_feat_dev, edge_feat_dev,
                                                        nbr_time_dev, mask_dev, Wq, Wk, Wv, w->in_b, Wo, w->out_b, W1,
                                                        w->m1_b, W2, w->m2_b, out_dev, attn_w_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
