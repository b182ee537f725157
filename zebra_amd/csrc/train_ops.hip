// Training-side dense operators on exact-f32 MFMA: what the reference's autograd does with
// nn.GRUCell and nn.Linear on the COMPACT rows of a training step --
//   * the lazily updated memory of the selected neighbours, get_updated_memory (reference
//     modules/memory_updater.py:61-90, nn.GRUCell :95-98): forward over the U flagged rows with the gate activations
//     kept for the backward, backward to dW_ih, dW_hh, db_ih, db_hh (the memory and the stored messages are buffers of the
//     reference, not parameters: no gradient flows into them);
//   * fc2 and transform_source of GraphDiffusionEmbedding (modules/embedding_module.py:86-98,320-328) on [N, D]
//     matrices: Y = X W^T (+ b) forward, dX = dY W, dW = dY^T X, db = column sums backward.
// One small tiled GEMM kernel serves all of them (v_mfma_f32_16x16x4_f32: products and sums are float32 FMAs in k
// order, so results agree with a float32 BLAS to rounding); the GRU's gate arithmetic is element-wise around it.
// The neighbour aggregation itself (fc1 + ReLU + dropout + weighted k-reduction, forward and backward) is
// aggregate.hip / aggregate_bwd.hip.
#include "common.hpp"

using namespace zt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[M][N] = op(A)[M][K] op(B)[K][N] (+ C), row-major; op(A)(m, k) = TA ? A[k * lda + m] : A[m * lda + k], likewise B.
// 64 x 64 tile per workgroup (4 waves, 32 x 32 each as 2 x 2 MFMA tiles), K in steps of 64 through LDS.
// (Round 6: the K step was 16 with a load, a store and two barriers per step -- a weight gradient dW = dY^T X of a training
//  step runs on 4-40 workgroups with K = the batch's rows, i.e. a chain of ~40 dependent memory round trips per workgroup:
//  222 us for 6 MFLOP, 1.3 of the training step's 3.5 ms of GPU time.  Now 64 columns of K per step, all sixteen loads of a
//  thread in flight before the first LDS store: a quarter of the round trips.  Same products, same k order: same results.)
constexpr int GK = 64;
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_f32(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                  long long M, long long N, long long K, long long lda, long long ldb,
                                                  long long ldc, int accumulate)
{
    __shared__ float As[64][GK + 1];      // [m][k]
    __shared__ float Bs[GK][65];          // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long m0 = (long long)blockIdx.y * 64, n0 = (long long)blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int r16 = lane & 15, g4 = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PT = 64 * GK / 256;     // elements of each operand per thread and step
    for (long long k0 = 0; k0 < K; k0 += GK) {
        float va[PT], vb[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int idx = tid + t * 256;
            // consecutive threads walk the contiguous dimension of the operand as it lies in memory
            const int am = TA ? (idx & 63) : (idx / GK), ak = TA ? (idx >> 6) : (idx % GK);
            const long long gm = m0 + am, gk = k0 + ak;
            va[t] = (gm < M && gk < K) ? (TA ? A[gk * lda + gm] : A[gm * lda + gk]) : 0.f;
            const int bn = TB ? (idx / GK) : (idx & 63), bk = TB ? (idx % GK) : (idx >> 6);
            const long long gn = n0 + bn, gk2 = k0 + bk;
            vb[t] = (gn < N && gk2 < K) ? (TB ? B[gn * ldb + gk2] : B[gk2 * ldb + gn]) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int idx = tid + t * 256;
            const int am = TA ? (idx & 63) : (idx / GK), ak = TA ? (idx >> 6) : (idx % GK);
            As[am][ak] = va[t];
            const int bn = TB ? (idx / GK) : (idx & 63), bk = TB ? (idx % GK) : (idx >> 6);
            Bs[bk][bn] = vb[t];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wm + i * 16 + r16][kk + g4];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[kk + g4][wn + j * 16 + r16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = m0 + wm + i * 16 + g4 * 4 + r, col = n0 + wn + j * 16 + r16;
                if (row < M && col < N) {
                    float *c = C + row * ldc + col;
                    *c = accumulate ? *c + acc[i][j][r] : acc[i][j][r];
                }
            }
}

int gemm(const float *A, const float *B, float *C, long long M, long long N, long long K, long long lda, long long ldb,
         long long ldc, bool ta, bool tb, bool accumulate, hipStream_t s)
{
    if (M <= 0 || N <= 0) return ZT_OK;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    const int acc = accumulate ? 1 : 0;
    if (ta && tb) k_gemm_f32<true, true><<<grid, 256, 0, s>>>(A, B, C, M, N, K, lda, ldb, ldc, acc);
    else if (ta) k_gemm_f32<true, false><<<grid, 256, 0, s>>>(A, B, C, M, N, K, lda, ldb, ldc, acc);
    else if (tb) k_gemm_f32<false, true><<<grid, 256, 0, s>>>(A, B, C, M, N, K, lda, ldb, ldc, acc);
    else k_gemm_f32<false, false><<<grid, 256, 0, s>>>(A, B, C, M, N, K, lda, ldb, ldc, acc);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

// out[c] (+)= sum_r X[r][c]; one workgroup of sixteen waves per 64 columns, rows strided over the waves, eight loads of a
// wave in flight (round 6: four waves with one dependent load each took 110 us for 600 rows -- 150 round trips in a row)
constexpr int CS_WAVES = 16;
__global__ __launch_bounds__(64 * CS_WAVES) void k_colsum(const float *__restrict__ X, long long R, long long Cn, long long ldx,
                                                          float *__restrict__ out, int accumulate)
{
    __shared__ float part[CS_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < Cn) {
        // (one accumulator, rows in ascending order per wave: the sum's association is fixed -- wave w adds rows w, w + 16, ...)
        for (long long r0 = wave; r0 < R; r0 += (long long)CS_WAVES * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long r = r0 + (long long)u * CS_WAVES;
                v[u] = r < R ? X[r * ldx + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < Cn) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CS_WAVES; ++w) t += part[w][lane];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// rows ids[u] of a [*, W] table -> dense [U, W]
__global__ void k_gather_rows_f32(const float *__restrict__ T, const int *__restrict__ ids, long long U, int W,
                                  float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U * W) return;
    const long long u = i / W;
    out[i] = T[(long long)ids[u] * W + (i - u * W)];
}

// The batch's own rows of the lazily updated memory (get_updated_memory(...)[nodes], modules/memory_updater.py:61-90 +
// modules/embedding_module.py:320-322): out[i] = overlay[row_map[nodes[i]]] where the map names an overlay row, else
// memory[nodes[i]]; sel[i] = that overlay row or -1, kept for the backward (the map is shared scratch and reset right after).
__global__ void k_overlay_rows(const float *__restrict__ memory, const float *__restrict__ overlay, const int *__restrict__ row_map,
                               const int *__restrict__ nodes, long long n, int D, long long num_nodes, float *__restrict__ out,
                               int *__restrict__ sel)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const long long r = i / D;
    const int c = (int)(i - r * D);
    int v = nodes[r];
    if (v < 0 || v >= num_nodes) v = 0;                          // (ids are checked by the T-PPR update of the same batch)
    const int ov = row_map != nullptr ? row_map[v] : -1;
    out[i] = ov >= 0 ? overlay[(long long)ov * D + c] : memory[(long long)v * D + c];
    if (c == 0) sel[r] = ov;
}

// d_overlay[sel[i]] += d_out[i] for the rows that came from the overlay (d_overlay zeroed by the caller; a node may appear
// several times in a batch: atomic adds, as in k_fc1_agg_bwd)
__global__ void k_overlay_rows_bwd(const float *__restrict__ d_out, const int *__restrict__ sel, long long n, int D,
                                   float *__restrict__ d_overlay)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const long long r = i / D;
    const int ov = sel[r];
    if (ov >= 0) atomicAdd(d_overlay + (long long)ov * D + (i - r * D), d_out[i]);
}

// torch.nn.GRUCell's gate arithmetic: gi = W_ih x + b_ih, gh = W_hh h + b_hh (gate order r, z, n);
// r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h.
// saved[u] = [r | z | n | gh_n] for the backward.
__global__ void k_gru_gates_fwd(const float *__restrict__ gi, const float *__restrict__ gh, const float *__restrict__ b_ih,
                                const float *__restrict__ b_hh, const float *__restrict__ H, long long U, int D,
                                float *__restrict__ h_out, float *__restrict__ saved)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U * D) return;
    const long long u = i / D;
    const int c = (int)(i - u * D);
    const float *gir = gi + u * 3 * D, *ghr = gh + u * 3 * D;
    const float r = 1.f / (1.f + expf(-((gir[c] + b_ih[c]) + (ghr[c] + b_hh[c]))));
    const float z = 1.f / (1.f + expf(-((gir[D + c] + b_ih[D + c]) + (ghr[D + c] + b_hh[D + c]))));
    const float hn = ghr[2 * D + c] + b_hh[2 * D + c];
    const float n = tanhf((gir[2 * D + c] + b_ih[2 * D + c]) + r * hn);
    const float h = H[i];
    h_out[i] = (1.f - z) * n + z * h;
    float *sv = saved + u * 4 * D;
    sv[c] = r; sv[D + c] = z; sv[2 * D + c] = n; sv[3 * D + c] = hn;
}

// d_gi = [dr_pre | dz_pre | dn_pre], d_gh = [dr_pre | dz_pre | dn_pre r] from d_h' (see k_gru_gates_fwd)
__global__ void k_gru_gates_bwd(const float *__restrict__ dh, const float *__restrict__ saved, const float *__restrict__ H,
                                long long U, int D, float *__restrict__ dgi, float *__restrict__ dgh)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U * D) return;
    const long long u = i / D;
    const int c = (int)(i - u * D);
    const float *sv = saved + u * 4 * D;
    const float r = sv[c], z = sv[D + c], n = sv[2 * D + c], hn = sv[3 * D + c];
    const float g = dh[i], h = H[i];
    const float dn_pre = g * (1.f - z) * (1.f - n * n);
    const float dz_pre = g * (h - n) * z * (1.f - z);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    float *a = dgi + u * 3 * D, *b = dgh + u * 3 * D;
    a[c] = dr_pre; a[D + c] = dz_pre; a[2 * D + c] = dn_pre;
    b[c] = dr_pre; b[D + c] = dz_pre; b[2 * D + c] = dn_pre * r;
}

struct GruTrainPlan {
    size_t off_x, off_h, off_gi, off_gh, total;
};

void gru_train_plan(long long U, int D, int msg, GruTrainPlan &p)
{
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    const size_t u = (size_t)(U > 0 ? U : 1);
    p.off_x = take(u * msg * 4);
    p.off_h = take(u * D * 4);
    p.off_gi = take(u * 3 * D * 4);
    p.off_gh = take(u * 3 * D * 4);
    p.total = o;
}

}  // namespace

extern "C" int zt_gemm_f32(const float *A_dev, const float *B_dev, float *C_dev, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldc, int32_t trans_a, int32_t trans_b, int32_t accumulate, void *stream)
{
    if (M < 0 || N < 0 || K < 0 || (M > 0 && N > 0 && (!C_dev || (K > 0 && (!A_dev || !B_dev))))) {
        set_error("zt_gemm_f32: bad argument");
        return ZT_ERR_ARG;
    }
    return gemm(A_dev, B_dev, C_dev, M, N, K, lda, ldb, ldc, trans_a != 0, trans_b != 0, accumulate != 0, (hipStream_t)stream);
}

extern "C" int zt_colsum_f32(const float *X_dev, int64_t rows, int64_t cols, int64_t ldx, float *out_dev, int32_t accumulate,
                             void *stream)
{
    if (rows < 0 || cols < 0 || (cols > 0 && (!out_dev || (rows > 0 && !X_dev)))) { set_error("zt_colsum_f32: bad argument"); return ZT_ERR_ARG; }
    if (cols == 0) return ZT_OK;
    k_colsum<<<(unsigned)((cols + 63) / 64), 64 * CS_WAVES, 0, (hipStream_t)stream>>>(X_dev, rows, cols, ldx, out_dev, accumulate != 0);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_overlay_rows(const float *memory_dev, const float *overlay_dev, const int32_t *row_map_dev, const int32_t *nodes_dev,
                               int64_t n, int32_t D, int64_t num_nodes, float *out_dev, int32_t *sel_dev, void *stream)
{
    if (n < 0 || D <= 0 || num_nodes <= 0 || (n > 0 && (!memory_dev || !nodes_dev || !out_dev || !sel_dev)) || (row_map_dev && !overlay_dev)) {
        set_error("zt_overlay_rows: bad argument");
        return ZT_ERR_ARG;
    }
    if (n == 0) return ZT_OK;
    k_overlay_rows<<<(unsigned)((n * D + 255) / 256), 256, 0, (hipStream_t)stream>>>(memory_dev, overlay_dev, row_map_dev, nodes_dev, n, D,
                                                                                    num_nodes, out_dev, sel_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_overlay_rows_backward(const float *d_out_dev, const int32_t *sel_dev, int64_t n, int32_t D, float *d_overlay_dev,
                                        void *stream)
{
    if (n < 0 || D <= 0 || (n > 0 && (!d_out_dev || !sel_dev || !d_overlay_dev))) { set_error("zt_overlay_rows_backward: bad argument"); return ZT_ERR_ARG; }
    if (n == 0) return ZT_OK;
    k_overlay_rows_bwd<<<(unsigned)((n * D + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_out_dev, sel_dev, n, D, d_overlay_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int64_t zt_gru_train_workspace_bytes(int64_t U, int32_t D, int32_t msg_dim)
{
    if (U < 0 || D <= 0 || msg_dim <= 0) return -1;
    GruTrainPlan p;
    gru_train_plan(U, D, msg_dim, p);
    return (int64_t)p.total;
}

extern "C" int zt_gru_train_forward(const float *messages_dev, const float *memory_dev, const int32_t *ids_dev, int64_t U,
                                    int32_t D, int32_t msg_dim, const zt_gru_weights *wt, float *h_out_dev, float *saved_dev,
                                    void *workspace_dev, void *stream)
{
    if (!messages_dev || !memory_dev || !wt || U < 0 || D <= 0 || msg_dim <= 0 || (U > 0 && (!ids_dev || !h_out_dev || !saved_dev || !workspace_dev))) {
        set_error("zt_gru_train_forward: bad argument");
        return ZT_ERR_ARG;
    }
    if (U == 0) return ZT_OK;
    hipStream_t s = (hipStream_t)stream;
    GruTrainPlan p;
    gru_train_plan(U, D, msg_dim, p);
    char *ws = reinterpret_cast<char *>(workspace_dev);
    float *X = reinterpret_cast<float *>(ws + p.off_x), *H = reinterpret_cast<float *>(ws + p.off_h);
    float *gi = reinterpret_cast<float *>(ws + p.off_gi), *gh = reinterpret_cast<float *>(ws + p.off_gh);
    k_gather_rows_f32<<<(unsigned)((U * msg_dim + 255) / 256), 256, 0, s>>>(messages_dev, ids_dev, U, msg_dim, X);
    k_gather_rows_f32<<<(unsigned)((U * D + 255) / 256), 256, 0, s>>>(memory_dev, ids_dev, U, D, H);
    int rc = gemm(X, wt->w_ih, gi, U, 3 * D, msg_dim, msg_dim, msg_dim, 3 * D, false, true, false, s);      // gi = X W_ih^T
    if (rc != ZT_OK) return rc;
    rc = gemm(H, wt->w_hh, gh, U, 3 * D, D, D, D, 3 * D, false, true, false, s);                            // gh = H W_hh^T
    if (rc != ZT_OK) return rc;
    k_gru_gates_fwd<<<(unsigned)((U * D + 255) / 256), 256, 0, s>>>(gi, gh, wt->b_ih, wt->b_hh, H, U, D, h_out_dev, saved_dev);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_gru_train_backward(const float *d_h_dev, const float *messages_dev, const float *memory_dev,
                                     const int32_t *ids_dev, int64_t U, int32_t D, int32_t msg_dim, const float *saved_dev,
                                     float *d_w_ih_dev, float *d_w_hh_dev, float *d_b_ih_dev, float *d_b_hh_dev,
                                     void *workspace_dev, void *stream)
{
    if (!messages_dev || !memory_dev || U < 0 || D <= 0 || msg_dim <= 0 || !d_w_ih_dev || !d_w_hh_dev || !d_b_ih_dev || !d_b_hh_dev ||
        (U > 0 && (!d_h_dev || !ids_dev || !saved_dev || !workspace_dev))) {
        set_error("zt_gru_train_backward: bad argument");
        return ZT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (U == 0) {
        ZT_HIP(hipMemsetAsync(d_w_ih_dev, 0, (size_t)3 * D * msg_dim * 4, s));
        ZT_HIP(hipMemsetAsync(d_w_hh_dev, 0, (size_t)3 * D * D * 4, s));
        ZT_HIP(hipMemsetAsync(d_b_ih_dev, 0, (size_t)3 * D * 4, s));
        ZT_HIP(hipMemsetAsync(d_b_hh_dev, 0, (size_t)3 * D * 4, s));
        return ZT_OK;
    }
    GruTrainPlan p;
    gru_train_plan(U, D, msg_dim, p);
    char *ws = reinterpret_cast<char *>(workspace_dev);
    float *X = reinterpret_cast<float *>(ws + p.off_x), *H = reinterpret_cast<float *>(ws + p.off_h);
    float *dgi = reinterpret_cast<float *>(ws + p.off_gi), *dgh = reinterpret_cast<float *>(ws + p.off_gh);
    // X and H are the rows the FORWARD gathered: the tables change between forward and backward (a training step
    // updates the memory and stores new messages before loss.backward(), model/tgn_model.py:155-168), so the workspace
    // of the forward call must be handed back untouched
    (void)messages_dev; (void)memory_dev; (void)ids_dev;
    k_gru_gates_bwd<<<(unsigned)((U * D + 255) / 256), 256, 0, s>>>(d_h_dev, saved_dev, H, U, D, dgi, dgh);
    int rc = gemm(dgi, X, d_w_ih_dev, 3 * D, msg_dim, U, 3 * D, msg_dim, msg_dim, true, false, false, s);   // dW_ih = dgi^T X
    if (rc != ZT_OK) return rc;
    rc = gemm(dgh, H, d_w_hh_dev, 3 * D, D, U, 3 * D, D, D, true, false, false, s);                         // dW_hh = dgh^T H
    if (rc != ZT_OK) return rc;
    k_colsum<<<(unsigned)((3 * D + 63) / 64), 64 * CS_WAVES, 0, s>>>(dgi, U, 3 * D, 3 * D, d_b_ih_dev, 0);
    k_colsum<<<(unsigned)((3 * D + 63) / 64), 64 * CS_WAVES, 0, s>>>(dgh, U, 3 * D, 3 * D, d_b_hh_dev, 0);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}
