/*
 * zo_nn.c -- CPU restatement of the reference's gather + TimeEncode +
 * transform + weighted-sum aggregation, last-message store and GRU memory
 * update.  TEST INFRASTRUCTURE ONLY (see zebra_oracle.h).
 *
 * Floating point: float32 storage as in the reference; dot products are
 * accumulated in float32 in input order (torch uses a BLAS with a different
 * summation order, so agreement with the reference is to rounding, ~1e-6;
 * the parity tolerance for embeddings is 1e-4).
 */
#include "zebra_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* y[j] = b[j] + sum_i x[i] * Wt[i][j]   (Wt = W transposed, [in][out]) */
static void affine_t(const float *x, const float *Wt, const float *b, int n_in,
                     int n_out, float *y)
{
    for (int j = 0; j < n_out; ++j) y[j] = b ? b[j] : 0.f;
    for (int i = 0; i < n_in; ++i) {
        const float xi = x[i];
        const float *wr = Wt + (size_t)i * n_out;
        for (int j = 0; j < n_out; ++j) y[j] += xi * wr[j];
    }
}

static float *transpose(const float *W, int n_out, int n_in)
{
    float *t = (float *)malloc(sizeof(float) * (size_t)n_out * n_in);
    for (int o = 0; o < n_out; ++o)
        for (int i = 0; i < n_in; ++i) t[(size_t)i * n_out + o] = W[(size_t)o * n_in + i];
    return t;
}

int zo_embed(const float *memory, const float *efeat, const float *time_w,
             int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F,
             int32_t T, const int32_t *nodes, int64_t N, int32_t M, int32_t k,
             const int32_t *nbr, const int32_t *eix, const float *dt,
             const float *w, const float *fc1_w, const float *fc1_b,
             const float *fc2_w, const float *fc2_b, const float *fc1s_w,
             const float *fc1s_b, const float *fc2s_w, const float *fc2s_b,
             float *out, int32_t n_threads)
{
    const int K1 = D + F + T;
    for (int64_t i = 0; i < N; ++i)
        if (nodes[i] < 0 || nodes[i] >= num_nodes) return -1;
    for (int64_t i = 0; i < (int64_t)M * N * k; ++i)
        if (nbr[i] < 0 || nbr[i] >= num_nodes || eix[i] < 0 || eix[i] >= num_edges) return -1;
    float *fc1_t = transpose(fc1_w, D, K1), *fc2_t = transpose(fc2_w, D, D);
    float *fc1s_t = transpose(fc1s_w, D, D), *fc2s_t = transpose(fc2s_w, D, D);
    const int OW = D * (M + 1);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
    {
        float *x = (float *)malloc(sizeof(float) * K1);
        float *h = (float *)malloc(sizeof(float) * D);
        float *y = (float *)malloc(sizeof(float) * D);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t r = 0; r < N; ++r) {
            float *o = out + (size_t)r * OW;
            /* transform_source(memory[nodes]) (embedding_module.py:243-246,320-323) */
            affine_t(memory + (size_t)nodes[r] * D, fc1s_t, fc1s_b, D, D, h);
            for (int j = 0; j < D; ++j) h[j] = h[j] > 0.f ? h[j] : 0.f;
            affine_t(h, fc2s_t, fc2s_b, D, D, o);
            for (int m = 0; m < M; ++m) {            /* :250-276 */
                size_t base = ((size_t)m * N + (size_t)r) * k;
                float wsum = 0.f;                    /* torch.sum(weights, dim=1) */
                for (int j = 0; j < k; ++j) wsum += w[base + j];
                float *acc = o + (size_t)D * (m + 1);
                for (int j = 0; j < D; ++j) acc[j] = 0.f;
                for (int q = 0; q < k; ++q) {
                    /* [memory | edge | time] (:264) */
                    memcpy(x, memory + (size_t)nbr[base + q] * D, sizeof(float) * D);
                    memcpy(x + D, efeat + (size_t)eix[base + q] * F, sizeof(float) * F);
                    for (int j = 0; j < T; ++j)      /* cos(t * w + 0) (time_encoding.py:27) */
                        x[D + F + j] = cosf(dt[base + q] * time_w[j]);
                    affine_t(x, fc1_t, fc1_b, K1, D, h);
                    for (int j = 0; j < D; ++j) h[j] = h[j] > 0.f ? h[j] : 0.f;
                    affine_t(h, fc2_t, fc2_b, D, D, y);
                    /* weights/weights_sum, 0 where the sum is 0 (:268-270) */
                    float wn = (wsum == 0.f) ? 0.f : w[base + q] / wsum;
                    for (int j = 0; j < D; ++j) acc[j] += y[j] * wn;
                }
            }
        }
        free(x); free(h); free(y);
    }
    free(fc1_t); free(fc2_t); free(fc1s_t); free(fc2s_t);
    return 0;
}

int64_t zo_store_messages(const float *memory, const float *last_update,
                          const float *efeat, const float *time_w,
                          int64_t num_nodes, int64_t num_edges, int32_t D,
                          int32_t F, int32_t T, const int32_t *src,
                          const int32_t *dst, const double *ts,
                          const int64_t *eidx, int64_t B, float *messages,
                          float *msg_ts, uint8_t *flags, int32_t *last)
{
    const int msg = 2 * D + F + T;
    for (int64_t i = 0; i < B; ++i) {
        if (src[i] < 0 || src[i] >= num_nodes || dst[i] < 0 || dst[i] >= num_nodes) return -1;
        if (eidx[i] < 0 || eidx[i] >= num_edges) return -1;
    }
    /* last occurrence per node over [src|dst] (tgn_model.py:206-208).  All
     * messages are built from the memory as it is on entry; a node's message
     * row only depends on memory/last_update, which this function does not
     * write, so building in place is safe. */
    for (int64_t p = 0; p < 2 * B; ++p) {
        int32_t v = p < B ? src[p] : dst[p - B];
        last[v] = (int32_t)p;
    }
    int64_t uniq = 0;
    for (int64_t p = 0; p < 2 * B; ++p) {
        int32_t v = p < B ? src[p] : dst[p - B];
        if (last[v] != p) continue;
        ++uniq;
        int64_t i = p < B ? p : p - B;
        int32_t partner = p < B ? dst[i] : src[i];
        float tf = (float)ts[i];                     /* .float() (:213) */
        float delta = tf - last_update[v];           /* :221 */
        float *mrow = messages + (size_t)v * msg;
        memcpy(mrow, memory + (size_t)v * D, sizeof(float) * D);
        memcpy(mrow + D, memory + (size_t)partner * D, sizeof(float) * D);
        memcpy(mrow + 2 * D, efeat + (size_t)eidx[i] * F, sizeof(float) * F);
        for (int j = 0; j < T; ++j) mrow[2 * D + F + j] = cosf(delta * time_w[j]);
        msg_ts[v] = tf;
        flags[v] = 1;
    }
    for (int64_t p = 0; p < 2 * B; ++p) last[p < B ? src[p] : dst[p - B]] = -1;
    return uniq;
}

static inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

int64_t zo_gru_update(float *memory, float *last_update, const float *messages,
                      const float *msg_ts, uint8_t *flags, int64_t num_nodes,
                      int32_t D, int32_t msg_dim, const int32_t *ids,
                      int64_t n_ids, const float *w_ih, const float *w_hh,
                      const float *b_ih, const float *b_hh, int32_t n_threads)
{
    /* collect the rows to update */
    int64_t cnt = 0;
    int32_t *rows;
    if (ids == NULL) {
        rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)num_nodes);
        for (int64_t v = 0; v < num_nodes; ++v) if (flags[v]) rows[cnt++] = (int32_t)v;
    } else {
        rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ids > 0 ? n_ids : 1));
        for (int64_t i = 0; i < n_ids; ++i) {
            if (ids[i] < 0 || ids[i] >= num_nodes) { free(rows); return -1; }
        }
        /* duplicates in ids would update a row twice; callers pass unique ids
         * (np.unique at model/tgn_model.py:129).  De-duplicate defensively. */
        for (int64_t i = 0; i < n_ids; ++i) {
            if (flags[ids[i]] == 1) { rows[cnt++] = ids[i]; flags[ids[i]] = 2; }
        }
        for (int64_t i = 0; i < cnt; ++i) flags[rows[i]] = 1;
    }
    float *wih_t = transpose(w_ih, 3 * D, msg_dim), *whh_t = transpose(w_hh, 3 * D, D);
    float *newh = (float *)malloc(sizeof(float) * (size_t)(cnt > 0 ? cnt : 1) * D);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
    {
        float *gi = (float *)malloc(sizeof(float) * 3 * D);
        float *gh = (float *)malloc(sizeof(float) * 3 * D);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t q = 0; q < cnt; ++q) {
            int32_t v = rows[q];
            const float *x = messages + (size_t)v * msg_dim;
            const float *hp = memory + (size_t)v * D;
            affine_t(x, wih_t, b_ih, msg_dim, 3 * D, gi);
            affine_t(hp, whh_t, b_hh, D, 3 * D, gh);
            for (int j = 0; j < D; ++j) {            /* torch.nn.GRUCell */
                float r = sigmoidf_(gi[j] + gh[j]);
                float z = sigmoidf_(gi[D + j] + gh[D + j]);
                float n = tanhf(gi[2 * D + j] + r * gh[2 * D + j]);
                newh[(size_t)q * D + j] = (1.f - z) * n + z * hp[j];
            }
        }
        free(gi); free(gh);
    }
    for (int64_t q = 0; q < cnt; ++q) {              /* memory_updater.py:40-43 */
        int32_t v = rows[q];
        memcpy(memory + (size_t)v * D, newh + (size_t)q * D, sizeof(float) * D);
        last_update[v] = msg_ts[v];
    }
    if (ids == NULL) { for (int64_t q = 0; q < cnt; ++q) flags[rows[q]] = 0; }
    else { for (int64_t i = 0; i < n_ids; ++i) flags[ids[i]] = 0; }
    free(wih_t); free(whh_t); free(newh); free(rows);
    return cnt;
}

int zo_affinity(const float *x1, const float *x2, int64_t rows, int32_t H,
                const float *fc1_w, const float *fc1_b, const float *fc2_w,
                const float *fc2_b, float *out)
{
    float *fc1_t = transpose(fc1_w, H, 2 * H);
    float *x = (float *)malloc(sizeof(float) * 2 * H), *h = (float *)malloc(sizeof(float) * H);
    for (int64_t r = 0; r < rows; ++r) {
        memcpy(x, x1 + (size_t)r * H, sizeof(float) * H);
        memcpy(x + H, x2 + (size_t)r * H, sizeof(float) * H);
        affine_t(x, fc1_t, fc1_b, 2 * H, H, h);
        float s = fc2_b[0];
        for (int j = 0; j < H; ++j) s += (h[j] > 0.f ? h[j] : 0.f) * fc2_w[j];
        out[r] = 1.f / (1.f + expf(-s));
    }
    free(fc1_t); free(x); free(h);
    return 0;
}
