/*
 * zo_tppr.c -- CPU restatement of the reference's T-PPR engines.
 * TEST INFRASTRUCTURE ONLY (see zebra_oracle.h).  Build with
 * -ffp-contract=off: the reference rounds every float64 `*` and `+`
 * separately (utils/util.py:523-541).
 */
#include "zebra_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* numba arithmetic                                                          */
/* ======================================================================== */

static inline int lt_f(double a, double b) { return isnan(b) || a < b; }

/* numba/misc/quicksort.py make_quicksort_impl(is_argsort=True), restated. */
void zo_numba_argsort(const double *a, int32_t n, int32_t *r)
{
    int32_t stk_lo[128], stk_hi[128];
    int sp = 0;
    for (int32_t i = 0; i < n; ++i) r[i] = i;
    if (n < 2) return;
    stk_lo[0] = 0; stk_hi[0] = n - 1; sp = 1;
    while (sp > 0) {
        --sp;
        int32_t low = stk_lo[sp], high = stk_hi[sp];
        while (high - low >= 15) {
            int32_t mid = (low + high) >> 1, t;
            if (lt_f(a[r[mid]], a[r[low]])) { t = r[low]; r[low] = r[mid]; r[mid] = t; }
            if (lt_f(a[r[high]], a[r[mid]])) { t = r[high]; r[high] = r[mid]; r[mid] = t; }
            if (lt_f(a[r[mid]], a[r[low]])) { t = r[low]; r[low] = r[mid]; r[mid] = t; }
            double pivot = a[r[mid]];
            t = r[high]; r[high] = r[mid]; r[mid] = t;
            int32_t i = low, j = high - 1;
            for (;;) {
                while (i < high && lt_f(a[r[i]], pivot)) ++i;
                while (j >= low && lt_f(pivot, a[r[j]])) --j;
                if (i >= j) break;
                t = r[i]; r[i] = r[j]; r[j] = t;
                ++i; --j;
            }
            t = r[i]; r[i] = r[high]; r[high] = t;
            if (high - i > i - low) {
                if (high > i) { stk_lo[sp] = i + 1; stk_hi[sp] = high; ++sp; }
                high = i - 1;
            } else {
                if (i > low) { stk_lo[sp] = low; stk_hi[sp] = i - 1; ++sp; }
                low = i + 1;
            }
        }
        for (int32_t p = low + 1; p <= high; ++p) {
            int32_t kk = r[p];
            double v = a[kk];
            int32_t q = p;
            while (q > low && lt_f(v, a[r[q - 1]])) { r[q] = r[q - 1]; --q; }
            r[q] = kk;
        }
    }
}

/* numba/cpython/numbers.py int_power_impl, restated. */
double zo_numba_int_pow(double a, int64_t b)
{
    double r = 1.0;
    int invert = 0;
    int64_t e = b;
    if (b < 0) { invert = 1; e = -b; }
    if (e > 0x10000) return pow(a, (double)b);
    while (e != 0) {
        if (e & 1) r *= a;
        e >>= 1;
        a *= a;
    }
    return invert ? 1.0 / r : r;
}

/* ======================================================================== */
/* streaming T-PPR state                                                     */
/* ======================================================================== */

struct zo_tppr {
    int64_t N;
    int32_t k, M;
    double *alpha, *beta;
    int32_t *len;  /* [M][N]    */
    double *norm;  /* [M][N]    */
    int64_t *eidx; /* [M][N][k] */
    int64_t *node; /* [M][N][k] */
    double *ts;    /* [M][N][k] */
    double *w;     /* [M][N][k] */
};

zo_tppr *zo_tppr_create(int64_t num_nodes, int32_t k, int32_t n_tppr,
                        const double *alpha_list, const double *beta_list)
{
    if (num_nodes <= 0 || k <= 0 || n_tppr <= 0) return NULL;
    zo_tppr *h = (zo_tppr *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->N = num_nodes; h->k = k; h->M = n_tppr;
    size_t rows = (size_t)n_tppr * (size_t)num_nodes;
    h->alpha = (double *)malloc(sizeof(double) * n_tppr);
    h->beta = (double *)malloc(sizeof(double) * n_tppr);
    h->len = (int32_t *)calloc(rows, sizeof(int32_t));
    h->norm = (double *)calloc(rows, sizeof(double));
    h->eidx = (int64_t *)calloc(rows * k, sizeof(int64_t));
    h->node = (int64_t *)calloc(rows * k, sizeof(int64_t));
    h->ts = (double *)calloc(rows * k, sizeof(double));
    h->w = (double *)calloc(rows * k, sizeof(double));
    if (!h->alpha || !h->beta || !h->len || !h->norm || !h->eidx || !h->node ||
        !h->ts || !h->w) { zo_tppr_destroy(h); return NULL; }
    memcpy(h->alpha, alpha_list, sizeof(double) * n_tppr);
    memcpy(h->beta, beta_list, sizeof(double) * n_tppr);
    return h;
}

void zo_tppr_destroy(zo_tppr *h)
{
    if (!h) return;
    free(h->alpha); free(h->beta); free(h->len); free(h->norm);
    free(h->eidx); free(h->node); free(h->ts); free(h->w);
    free(h);
}

void zo_tppr_reset(zo_tppr *h)
{
    size_t rows = (size_t)h->M * (size_t)h->N;
    memset(h->len, 0, rows * sizeof(int32_t));
    memset(h->norm, 0, rows * sizeof(double));
    memset(h->eidx, 0, rows * h->k * sizeof(int64_t));
    memset(h->node, 0, rows * h->k * sizeof(int64_t));
    memset(h->ts, 0, rows * h->k * sizeof(double));
    memset(h->w, 0, rows * h->k * sizeof(double));
}

int zo_tppr_copy(zo_tppr *dst, const zo_tppr *src)
{
    if (dst->N != src->N || dst->k != src->k || dst->M != src->M) return -1;
    size_t rows = (size_t)src->M * (size_t)src->N;
    memcpy(dst->len, src->len, rows * sizeof(int32_t));
    memcpy(dst->norm, src->norm, rows * sizeof(double));
    memcpy(dst->eidx, src->eidx, rows * src->k * sizeof(int64_t));
    memcpy(dst->node, src->node, rows * src->k * sizeof(int64_t));
    memcpy(dst->ts, src->ts, rows * src->k * sizeof(double));
    memcpy(dst->w, src->w, rows * src->k * sizeof(double));
    return 0;
}

void zo_tppr_export(const zo_tppr *h, int32_t m, int32_t *len, double *norm,
                    int64_t *eidx, int64_t *node, double *ts, double *w)
{
    size_t off = (size_t)m * (size_t)h->N;
    memcpy(len, h->len + off, h->N * sizeof(int32_t));
    memcpy(norm, h->norm + off, h->N * sizeof(double));
    memcpy(eidx, h->eidx + off * h->k, h->N * h->k * sizeof(int64_t));
    memcpy(node, h->node + off * h->k, h->N * h->k * sizeof(int64_t));
    memcpy(ts, h->ts + off * h->k, h->N * h->k * sizeof(double));
    memcpy(w, h->w + off * h->k, h->N * h->k * sizeof(double));
}

/* the export for n chosen nodes (arrays [n], [n][k]): full-size graphs, where only the touched rows are compared */
int zo_tppr_export_rows(const zo_tppr *h, int32_t m, const int64_t *ids, int64_t n, int32_t *len, double *norm,
                        int64_t *eidx, int64_t *node, double *ts, double *w)
{
    if (m < 0 || m >= h->M) return -2;
    const size_t k = (size_t)h->k;
    for (int64_t q = 0; q < n; ++q) {
        if (ids[q] < 0 || ids[q] >= h->N) return -1;
        size_t row = (size_t)m * (size_t)h->N + (size_t)ids[q];
        len[q] = h->len[row];
        norm[q] = h->norm[row];
        memcpy(eidx + (size_t)q * k, h->eidx + row * k, k * sizeof(int64_t));
        memcpy(node + (size_t)q * k, h->node + row * k, k * sizeof(int64_t));
        memcpy(ts + (size_t)q * k, h->ts + row * k, k * sizeof(double));
        memcpy(w + (size_t)q * k, h->w + row * k, k * sizeof(double));
    }
    return 0;
}

/* inverse of the export for n chosen nodes (arrays [n], [n][k]); other rows stay.  Used to start the
 * timed CPU baseline from the same warm state as the GPU run (bench.py). */
int zo_tppr_import_rows(zo_tppr *h, int32_t m, const int64_t *ids, int64_t n, const int32_t *len,
                        const double *norm, const int64_t *eidx, const int64_t *node, const double *ts,
                        const double *w)
{
    if (m < 0 || m >= h->M) return -2;
    const size_t k = (size_t)h->k;
    for (int64_t q = 0; q < n; ++q) {
        if (ids[q] < 0 || ids[q] >= h->N || len[q] < 0 || len[q] > h->k) return -1;
        size_t row = (size_t)m * (size_t)h->N + (size_t)ids[q];
        h->len[row] = len[q];
        h->norm[row] = norm[q];
        memcpy(h->eidx + row * k, eidx + (size_t)q * k, k * sizeof(int64_t));
        memcpy(h->node + row * k, node + (size_t)q * k, k * sizeof(int64_t));
        memcpy(h->ts + row * k, ts + (size_t)q * k, k * sizeof(double));
        memcpy(h->w + row * k, w + (size_t)q * k, k * sizeof(double));
    }
    return 0;
}

/* extract_streaming_tppr (utils/util.py:447-469).  Empty dict: the reference
 * leaves the zero-initialised row alone; here the row is written as zeros. */
static void emit_row(const zo_tppr *h, int32_t m, int64_t v, double t_now,
                     int32_t *o_node, int32_t *o_eidx, float *o_dt, float *o_w)
{
    const int32_t k = h->k;
    size_t row = ((size_t)m * h->N + (size_t)v);
    int32_t len = h->len[row];
    if (len == 0) {
        for (int32_t j = 0; j < k; ++j) { o_node[j] = 0; o_eidx[j] = 0; o_dt[j] = 0.f; o_w[j] = 0.f; }
        return;
    }
    const int64_t *e = h->eidx + row * k, *nd = h->node + row * k;
    const double *ts = h->ts + row * k, *w = h->w + row * k;
    for (int32_t j = 0; j < k; ++j) {
        float tsf = 0.f;
        if (j < len) {
            o_node[j] = (int32_t)nd[j];
            o_eidx[j] = (int32_t)e[j];
            o_w[j] = (float)w[j];
            tsf = (float)ts[j];      /* tmp_timestamps is float32 (:452,462) */
        } else {
            o_node[j] = 0; o_eidx[j] = 0; o_w[j] = 0.f;
        }
        /* float64 scalar - float32 array -> float64, stored to a float32 row
         * (:465,468): padding slots get f32(t_now). */
        o_dt[j] = (float)(t_now - (double)tsf);
    }
}

typedef struct { int64_t e, nd; double ts, w; } cand_t;

/* One (s1, s2) pair of the update block (utils/util.py:509-564): builds the
 * new dictionary of s1 in `out` (<= k entries), returns its size. */
static int32_t merge_pair(const zo_tppr *h, int32_t m, int64_t s1, int64_t s2,
                          int64_t eidx, double ts, cand_t *t, int32_t *perm,
                          double *vals, cand_t *out)
{
    const int32_t k = h->k;
    const double alpha = h->alpha[m], beta = h->beta[m];
    size_t r1 = (size_t)m * h->N + (size_t)s1, r2 = (size_t)m * h->N + (size_t)s2;
    int32_t n = 0;
    double scale_s1 = 0.0, scale_s2;
    if (h->norm[r1] == 0) {                         /* :514-519 */
        scale_s2 = 1 - alpha;
    } else {                                        /* :520-527 */
        int32_t l1 = h->len[r1];
        double last_norm = h->norm[r1];
        double new_norm = last_norm * beta + beta;
        scale_s1 = last_norm / new_norm * beta;
        scale_s2 = beta / new_norm * (1 - alpha);
        for (int32_t j = 0; j < l1; ++j) {
            t[n].e = h->eidx[r1 * k + j]; t[n].nd = h->node[r1 * k + j];
            t[n].ts = h->ts[r1 * k + j];
            t[n].w = h->w[r1 * k + j] * scale_s1;
            ++n;
        }
    }
    if (h->norm[r2] != 0) {                         /* :532-538 */
        int32_t l2 = h->len[r2];
        for (int32_t j = 0; j < l2; ++j) {
            int64_t e = h->eidx[r2 * k + j], nd = h->node[r2 * k + j];
            double tsj = h->ts[r2 * k + j];
            double add = h->w[r2 * k + j] * scale_s2;
            int32_t f = -1;
            for (int32_t q = 0; q < n; ++q)
                if (t[q].e == e && t[q].nd == nd && t[q].ts == tsj) { f = q; break; }
            if (f >= 0) t[f].w += add;
            else { t[n].e = e; t[n].nd = nd; t[n].ts = tsj; t[n].w = add; ++n; }
        }
    }
    {                                               /* :531 / :540-541 */
        double v = (alpha != 0) ? scale_s2 * alpha : scale_s2;
        int32_t f = -1;
        for (int32_t q = 0; q < n; ++q)
            if (t[q].e == eidx && t[q].nd == s2 && t[q].ts == ts) { f = q; break; }
        if (f >= 0) t[f].w = v;
        else { t[n].e = eidx; t[n].nd = s2; t[n].ts = ts; t[n].w = v; ++n; }
    }
    if (n <= k) {                                   /* :549-551 */
        memcpy(out, t, sizeof(cand_t) * n);
        return n;
    }
    for (int32_t q = 0; q < n; ++q) vals[q] = t[q].w; /* :553-559 */
    zo_numba_argsort(vals, n, perm);
    for (int32_t q = 0; q < k; ++q) out[q] = t[perm[n - k + q]];
    return k;
}

static void write_row(zo_tppr *h, int32_t m, int64_t v, const cand_t *c, int32_t n)
{
    const int32_t k = h->k;
    size_t row = (size_t)m * h->N + (size_t)v;
    h->len[row] = n;
    for (int32_t j = 0; j < k; ++j) {
        if (j < n) {
            h->eidx[row * k + j] = c[j].e; h->node[row * k + j] = c[j].nd;
            h->ts[row * k + j] = c[j].ts; h->w[row * k + j] = c[j].w;
        } else {
            h->eidx[row * k + j] = 0; h->node[row * k + j] = 0;
            h->ts[row * k + j] = 0; h->w[row * k + j] = 0;
        }
    }
}

int zo_tppr_stream(zo_tppr *h, const int32_t *nodes, const double *ts,
                   const int64_t *eidx, int64_t B, int32_t n_roles,
                   int32_t emit, int32_t model, int32_t *out_nodes,
                   int32_t *out_eidx, float *out_dt, float *out_w)
{
    const int32_t k = h->k;
    if (n_roles != 2 && n_roles != 3) return -2;
    for (int64_t i = 0; i < (int64_t)n_roles * B; ++i)
        if (nodes[i] < 0 || nodes[i] >= h->N) return -1;
    int32_t cap = 2 * k + 1;
    cand_t *t = (cand_t *)malloc(sizeof(cand_t) * cap);
    cand_t *n1 = (cand_t *)malloc(sizeof(cand_t) * k);
    cand_t *n2 = (cand_t *)malloc(sizeof(cand_t) * k);
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * cap);
    double *vals = (double *)malloc(sizeof(double) * cap);
    const int64_t rows = (int64_t)n_roles * B;
    int32_t m_lo = model < 0 ? 0 : model, m_hi = model < 0 ? h->M : model + 1;
    for (int32_t m = m_lo; m < m_hi; ++m) {          /* :489 */
        const double beta = h->beta[m];
        /* outputs are indexed by emitted model (single_streaming_topk -> 0) */
        size_t obase = (size_t)(model < 0 ? m : 0) * (size_t)rows * k;
        for (int64_t i = 0; i < B; ++i) {            /* :495 */
            int64_t s = nodes[i], d = nodes[i + B];
            double tnow = ts[i];
            int64_t e = eidx[i];
            if (emit) {                              /* :504-506 */
                for (int32_t r = 0; r < n_roles; ++r) {
                    size_t o = obase + (size_t)(i + (int64_t)r * B) * k;
                    emit_row(h, m, nodes[i + (int64_t)r * B], tnow, out_nodes + o,
                             out_eidx + o, out_dt + o, out_w + o);
                }
            }
            /* both pairs are computed from the OLD state (:509-564) ... */
            int32_t c1 = merge_pair(h, m, s, d, e, tnow, t, perm, vals, n1);
            int32_t c2 = 0;
            if (s != d) c2 = merge_pair(h, m, d, s, e, tnow, t, perm, vals, n2);
            /* ... then written back (:567-574) */
            size_t rs = (size_t)m * h->N + (size_t)s, rd = (size_t)m * h->N + (size_t)d;
            write_row(h, m, s, n1, c1);
            h->norm[rs] = h->norm[rs] * beta + beta;
            if (s != d) {
                write_row(h, m, d, n2, c2);
                h->norm[rd] = h->norm[rd] * beta + beta;
            }
        }
    }
    free(t); free(n1); free(n2); free(perm); free(vals);
    return 0;
}

/* ======================================================================== */
/* static adjacency + pruning                                                */
/* ======================================================================== */

typedef struct { double ts; int32_t nbr, eid; int64_t seq; } adj_t;

static void merge_sort_adj(adj_t *a, adj_t *tmp, int64_t n)
{
    /* stable bottom-up merge sort by ts (Python's sorted(key=ts) is stable,
     * utils/util.py:103) */
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) tmp[o++] = (a[j].ts < a[i].ts) ? a[j++] : a[i++];
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, sizeof(adj_t) * n);
    }
}

int zo_csr_build(const int32_t *src, const int32_t *dst, const int64_t *eidx,
                 const double *ts, int64_t E, int64_t num_nodes,
                 int64_t *indptr, int32_t *nbr, int32_t *eid, double *ats)
{
    for (int64_t i = 0; i < E; ++i)
        if (src[i] < 0 || src[i] >= num_nodes || dst[i] < 0 || dst[i] >= num_nodes) return -1;
    memset(indptr, 0, sizeof(int64_t) * (num_nodes + 1));
    for (int64_t i = 0; i < E; ++i) { indptr[src[i] + 1]++; indptr[dst[i] + 1]++; }
    for (int64_t v = 0; v < num_nodes; ++v) indptr[v + 1] += indptr[v];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * num_nodes);
    adj_t *a = (adj_t *)malloc(sizeof(adj_t) * (2 * E + 1));
    adj_t *tmp = (adj_t *)malloc(sizeof(adj_t) * (2 * E + 1));
    memcpy(cur, indptr, sizeof(int64_t) * num_nodes);
    for (int64_t i = 0; i < E; ++i) {                /* :94-96 */
        adj_t x; x.ts = ts[i]; x.eid = (int32_t)eidx[i]; x.seq = i;
        x.nbr = dst[i]; a[cur[src[i]]++] = x;
        x.nbr = src[i]; a[cur[dst[i]]++] = x;
    }
    for (int64_t v = 0; v < num_nodes; ++v)
        merge_sort_adj(a + indptr[v], tmp, indptr[v + 1] - indptr[v]);
    for (int64_t p = 0; p < 2 * E; ++p) { nbr[p] = a[p].nbr; eid[p] = a[p].eid; ats[p] = a[p].ts; }
    free(cur); free(a); free(tmp);
    return 0;
}

int64_t zo_find_before(const int64_t *indptr, const double *ats, int32_t v, double t)
{
    /* np.searchsorted(ts, t) side='left' */
    int64_t lo = indptr[v], hi = indptr[v + 1], base = lo;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ats[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo - base;
}

typedef struct { int32_t node; double ts, w; } query_t;

int zo_pruned_topk(const int64_t *indptr, const int32_t *nbr,
                   const int32_t *eid, const double *ats, int64_t num_nodes,
                   const int32_t *q_nodes, const double *q_ts, int64_t nq,
                   int32_t width, int32_t depth, double alpha, double beta,
                   int32_t k, int32_t *out_nodes, int32_t *out_eidx,
                   float *out_dt, float *out_w)
{
    for (int64_t i = 0; i < nq; ++i)
        if (q_nodes[i] < 0 || q_nodes[i] >= num_nodes) return -1;
    /* capacity: sum_{d=1..depth} width^d states */
    int64_t cap = 0, lvl = 1;
    for (int32_t d = 0; d < depth; ++d) { lvl *= width; cap += lvl; if (cap > (1 << 24)) return -2; }
    if (cap < 1) cap = 1;
    cand_t *dict = (cand_t *)malloc(sizeof(cand_t) * cap);
    query_t *ql = (query_t *)malloc(sizeof(query_t) * (lvl + 1));
    query_t *nql = (query_t *)malloc(sizeof(query_t) * (lvl + 1));
    int64_t hcap = 16; while (hcap < 2 * cap) hcap *= 2;
    int32_t *ht = (int32_t *)malloc(sizeof(int32_t) * hcap);
    double *vals = (double *)malloc(sizeof(double) * cap);
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * cap);

    for (int64_t i = 0; i < nq; ++i) {               /* :187 */
        int64_t nd = 0, nq_cur = 1;
        for (int64_t q = 0; q < hcap; ++q) ht[q] = -1;
        ql[0].node = q_nodes[i]; ql[0].ts = q_ts[i]; ql[0].w = 1.0;
        for (int32_t dep = 0; dep < depth; ++dep) {  /* :197 */
            int64_t nn = 0;
            for (int64_t qi = 0; qi < nq_cur; ++qi) { /* :201 */
                int32_t qn = ql[qi].node;
                int64_t n_ngh = zo_find_before(indptr, ats, qn, ql[qi].ts);
                if (n_ngh == 0) continue;
                double norm = beta / (1 - beta) * (1 - zo_numba_int_pow(beta, n_ngh)); /* :208 */
                double weight = (alpha != 0 && dep == 0)
                                    ? ql[qi].w * (1 - alpha) * beta / norm * alpha
                                    : ql[qi].w * (1 - alpha) * beta / norm;        /* :209 */
                int64_t lim = width < n_ngh ? width : n_ngh;
                int64_t end = indptr[qn] + n_ngh;    /* most recent first (:212-218) */
                for (int64_t z = 0; z < lim; ++z) {
                    int64_t p = end - (z + 1);
                    int64_t ke = eid[p], kn = nbr[p];
                    double kt = ats[p];
                    /* dict[(edge_idx,node,timestamp)] += weight (:219-225) */
                    uint64_t hsh = (uint64_t)ke * 0x9E3779B97F4A7C15ull ^ ((uint64_t)kn * 0xC2B2AE3D27D4EB4Full);
                    int64_t slot = (int64_t)(hsh & (uint64_t)(hcap - 1));
                    for (;;) {
                        int32_t di = ht[slot];
                        if (di < 0) {
                            dict[nd].e = ke; dict[nd].nd = kn; dict[nd].ts = kt; dict[nd].w = weight;
                            ht[slot] = (int32_t)nd; ++nd;
                            break;
                        }
                        if (dict[di].e == ke && dict[di].nd == kn && dict[di].ts == kt) {
                            dict[di].w = dict[di].w + weight;
                            break;
                        }
                        slot = (slot + 1) & (hcap - 1);
                    }
                    nql[nn].node = (int32_t)kn; nql[nn].ts = kt; nql[nn].w = weight; ++nn; /* :228-229 */
                    weight = weight * beta;          /* :232 */
                }
            }
            if (nn == 0) break;                      /* :234-237 */
            query_t *sw = ql; ql = nql; nql = sw; nq_cur = nn;
        }
        if (nd == 0) continue;                       /* :240-242 */
        double tnow = q_ts[i];
        int32_t take = nd <= k ? (int32_t)nd : k;
        if (nd > k) {
            for (int64_t q = 0; q < nd; ++q) vals[q] = dict[q].w;
            zo_numba_argsort(vals, (int32_t)nd, perm);
        }
        size_t o = (size_t)i * k;
        for (int32_t j = 0; j < k; ++j) {
            float tsf = 0.f;
            if (j < take) {
                const cand_t *c = nd <= k ? &dict[j] : &dict[perm[nd - k + j]];
                out_nodes[o + j] = (int32_t)c->nd; out_eidx[o + j] = (int32_t)c->e;
                out_w[o + j] = (float)c->w; tsf = (float)c->ts;
            } else {
                out_nodes[o + j] = 0; out_eidx[o + j] = 0; out_w[o + j] = 0.f;
            }
            out_dt[o + j] = (float)(tnow - (double)tsf); /* :272 */
        }
    }
    free(dict); free(ql); free(nql); free(ht); free(vals); free(perm);
    return 0;
}
