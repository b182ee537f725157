/*
 * sanitize_driver.c -- exercises every entry point of the CPU oracle under
 * AddressSanitizer + UndefinedBehaviorSanitizer ("race
 * detection / sanitizers: -fsanitize=address CPU build").  TEST INFRASTRUCTURE
 * ONLY; built by `make sanitize` into oracle/_san/ and run by
 * tests/test_oracle_golden.py::test_oracle_under_sanitizers.  Inputs are
 * seeded streams with hubs, self-loops, repeated pairs, duplicate timestamps,
 * empty rows and out-of-range ids (which must be REJECTED, not read).
 * Exit code 0 and no sanitizer report = pass.
 */
#include "zebra_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rs = 88172645463325252ull;
static uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double urand(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }
static int pl(int n) { double u = urand(); int v = (int)(pow(u, 3.0) * n); return v >= n ? n - 1 : v; }   /* skewed */
static float frand(void) { return (float)(urand() * 2.0 - 1.0); }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "sanitize_driver: check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void)
{
    const int N = 97, E = 3000, k = 7, M = 2, B = 50, D = 12, F = 3, T = 8;
    const double alpha[2] = {0.1, 0.0}, beta[2] = {0.5, 0.9};
    int32_t *src = malloc(sizeof(int32_t) * E), *dst = malloc(sizeof(int32_t) * E), *neg = malloc(sizeof(int32_t) * E);
    double *ts = malloc(sizeof(double) * E);
    int64_t *eidx = malloc(sizeof(int64_t) * E);
    double t = 0.0;
    for (int i = 0; i < E; ++i) {
        src[i] = 1 + pl(N - 1); dst[i] = 1 + pl(N - 1); neg[i] = 1 + (int)(rnd() % (N - 1));
        if (i && (rnd() % 20) == 0) { src[i] = src[i - 1]; dst[i] = dst[i - 1]; }     /* repeated pair */
        if ((rnd() % 30) == 0) dst[i] = src[i];                                       /* self loop */
        if ((rnd() % 10) != 0) t += urand() * 60.0;                                   /* duplicate timestamps */
        ts[i] = t; eidx[i] = i + 1;
    }
    /* ---- numba arithmetic ---- */
    {
        double a[41];
        int32_t r[41];
        for (int n = 0; n <= 41; ++n) {
            for (int i = 0; i < n; ++i) a[i] = (double)(rnd() % 5) * 0.25;            /* heavy ties */
            zo_numba_argsort(a, n, r);
            for (int i = 1; i < n; ++i) CHECK(a[r[i - 1]] <= a[r[i]]);
        }
        CHECK(zo_numba_int_pow(0.5, 10) == 0.0009765625);
        CHECK(zo_numba_int_pow(2.0, -2) == 0.25);
        CHECK(zo_numba_int_pow(3.0, 0) == 1.0);
    }
    /* ---- streaming T-PPR: 3 roles, 2 roles, single model, update only, bad ids, copy, export / import ---- */
    zo_tppr *h = zo_tppr_create(N, k, M, alpha, beta), *h2 = zo_tppr_create(N, k, M, alpha, beta);
    CHECK(h && h2);
    int32_t *nodes = malloc(sizeof(int32_t) * 3 * B), *on = malloc(sizeof(int32_t) * M * 3 * B * k),
            *oe = malloc(sizeof(int32_t) * M * 3 * B * k);
    float *od = malloc(sizeof(float) * M * 3 * B * k), *ow = malloc(sizeof(float) * M * 3 * B * k);
    for (int s = 0; s + B <= E; s += B) {
        memcpy(nodes, src + s, sizeof(int32_t) * B);
        memcpy(nodes + B, dst + s, sizeof(int32_t) * B);
        memcpy(nodes + 2 * B, neg + s, sizeof(int32_t) * B);
        const int mode = (s / B) % 4;
        int rc;
        if (mode == 0) rc = zo_tppr_stream(h, nodes, ts + s, eidx + s, B, 3, 1, -1, on, oe, od, ow);
        else if (mode == 1) rc = zo_tppr_stream(h, nodes, ts + s, eidx + s, B, 2, 1, -1, on, oe, od, ow);
        else if (mode == 2) rc = zo_tppr_stream(h, nodes, ts + s, eidx + s, B, 3, 1, 1, on, oe, od, ow);
        else rc = zo_tppr_stream(h, nodes, ts + s, eidx + s, B, 2, 0, -1, NULL, NULL, NULL, NULL);
        CHECK(rc == 0);
    }
    nodes[3] = N + 5;                                       /* out of range: rejected */
    CHECK(zo_tppr_stream(h, nodes, ts, eidx, B, 3, 1, -1, on, oe, od, ow) == -1);
    nodes[3] = -1;
    CHECK(zo_tppr_stream(h, nodes, ts, eidx, B, 3, 1, -1, on, oe, od, ow) == -1);
    CHECK(zo_tppr_copy(h2, h) == 0);
    {
        int32_t *len = malloc(sizeof(int32_t) * N);
        double *norm = malloc(sizeof(double) * N), *ets = malloc(sizeof(double) * N * k), *ew = malloc(sizeof(double) * N * k);
        int64_t *ee = malloc(sizeof(int64_t) * N * k), *en = malloc(sizeof(int64_t) * N * k), *ids = malloc(sizeof(int64_t) * N);
        for (int m = 0; m < M; ++m) {
            zo_tppr_export(h, m, len, norm, ee, en, ets, ew);
            for (int v = 0; v < N; ++v) { CHECK(len[v] >= 0 && len[v] <= k); ids[v] = v; }
            zo_tppr_reset(h2);
            CHECK(zo_tppr_import_rows(h2, m, ids, N, len, norm, ee, en, ets, ew) == 0);
        }
        ids[0] = N;                                         /* rejected */
        CHECK(zo_tppr_import_rows(h2, 0, ids, 1, len, norm, ee, en, ets, ew) != 0);
        free(len); free(norm); free(ets); free(ew); free(ee); free(en); free(ids);
    }
    zo_tppr_destroy(h2);
    zo_tppr_destroy(h);
    /* ---- adjacency + pruning T-PPR (k below, at and above the candidate count; isolated node; bad id) ---- */
    int64_t *indptr = malloc(sizeof(int64_t) * (N + 1));
    int32_t *nbr = malloc(sizeof(int32_t) * 2 * E), *eid = malloc(sizeof(int32_t) * 2 * E);
    double *ats = malloc(sizeof(double) * 2 * E);
    CHECK(zo_csr_build(src, dst, eidx, ts, E, N, indptr, nbr, eid, ats) == 0);
    CHECK(zo_find_before(indptr, ats, 0, 1e30) == 0);       /* node 0 (padding) has no entries */
    {
        const int nq = 64, kk[3] = {3, 20, 40};
        int32_t q[64];
        double qt[64];
        for (int i = 0; i < nq; ++i) { q[i] = i < 2 ? 0 : 1 + (int)(rnd() % (N - 1)); qt[i] = i < 4 ? 0.0 : ts[rnd() % E] + 1e-3; }
        for (int c = 0; c < 3; ++c) {
            const int K = kk[c];
            int32_t *pn = calloc((size_t)nq * K, 4), *pe = calloc((size_t)nq * K, 4);
            float *pd = calloc((size_t)nq * K, 4), *pw = calloc((size_t)nq * K, 4);
            CHECK(zo_pruned_topk(indptr, nbr, eid, ats, N, q, qt, nq, 10, 2, 0.1, 0.5, K, pn, pe, pd, pw) == 0);
            CHECK(zo_pruned_topk(indptr, nbr, eid, ats, N, q, qt, nq, 3, 3, 0.0, 0.9, K, pn, pe, pd, pw) == 0);
            q[5] = N;
            CHECK(zo_pruned_topk(indptr, nbr, eid, ats, N, q, qt, nq, 10, 2, 0.1, 0.5, K, pn, pe, pd, pw) != 0);
            q[5] = 1;
            free(pn); free(pe); free(pd); free(pw);
        }
    }
    /* ---- aggregate, messages, GRU, scorer (1 and 3 threads) ---- */
    {
        const int msg = 2 * D + F + T, H = D * (M + 1), n = 3 * B;
        float *memory = malloc(sizeof(float) * N * D), *efeat = malloc(sizeof(float) * (E + 1) * F), *tw = malloc(sizeof(float) * T);
        float *lu = calloc(N, 4), *messages = calloc((size_t)N * msg, 4), *mts = calloc(N, 4);
        uint8_t *flags = calloc(N, 1);
        int32_t *scratch = malloc(sizeof(int32_t) * N);
        for (int i = 0; i < N * D; ++i) memory[i] = frand();
        for (int i = 0; i < (E + 1) * F; ++i) efeat[i] = i < F ? 0.f : frand();
        for (int i = 0; i < T; ++i) tw[i] = (float)pow(10.0, -9.0 * i / (T - 1));
        for (int i = 0; i < N; ++i) scratch[i] = -1;
#define W(name, cnt) float *name = malloc(sizeof(float) * (cnt)); for (int i_ = 0; i_ < (cnt); ++i_) name[i_] = 0.2f * frand()
        W(fc1_w, D * (D + F + T)); W(fc1_b, D); W(fc2_w, D * D); W(fc2_b, D); W(fc1s_w, D * D); W(fc1s_b, D); W(fc2s_w, D * D); W(fc2s_b, D);
        W(w_ih, 3 * D * msg); W(w_hh, 3 * D * D); W(b_ih, 3 * D); W(b_hh, 3 * D); W(a1w, H * 2 * H); W(a1b, H); W(a2w, H); W(a2b, 1);
#undef W
        int32_t *qn = malloc(sizeof(int32_t) * n), *gn = calloc((size_t)M * n * k, 4), *ge = calloc((size_t)M * n * k, 4);
        float *gd = calloc((size_t)M * n * k, 4), *gw = calloc((size_t)M * n * k, 4), *out = malloc(sizeof(float) * n * H);
        for (int i = 0; i < n; ++i) qn[i] = 1 + (int)(rnd() % (N - 1));
        for (int i = 0; i < M * n * k; ++i) {
            const int empty_row = ((i / k) % 7) == 0;       /* rows whose weights sum to zero */
            gn[i] = empty_row ? 0 : (int)(rnd() % N); ge[i] = empty_row ? 0 : (int)(rnd() % (E + 1));
            gd[i] = (float)(urand() * 3e8); gw[i] = empty_row ? 0.f : (float)urand();
        }
        for (int th = 1; th <= 3; th += 2)
            CHECK(zo_embed(memory, efeat, tw, N, E + 1, D, F, T, qn, n, M, k, gn, ge, gd, gw, fc1_w, fc1_b, fc2_w, fc2_b,
                           fc1s_w, fc1s_b, fc2s_w, fc2s_b, out, th) == 0);
        for (int i = 0; i < n * H; ++i) CHECK(out[i] == out[i]);
        gn[11] = N;                                         /* rejected */
        CHECK(zo_embed(memory, efeat, tw, N, E + 1, D, F, T, qn, n, M, k, gn, ge, gd, gw, fc1_w, fc1_b, fc2_w, fc2_b,
                       fc1s_w, fc1s_b, fc2s_w, fc2s_b, out, 1) != 0);
        for (int s = 0; s + B <= 10 * B; s += B) {
            CHECK(zo_store_messages(memory, lu, efeat, tw, N, E + 1, D, F, T, src + s, dst + s, ts + s, eidx + s, B, messages,
                                    mts, flags, scratch) > 0);
            for (int i = 0; i < N; ++i) CHECK(scratch[i] == -1);
            if ((s / B) % 2) {
                int32_t ids[2 * 50];
                memcpy(ids, src + s, sizeof(int32_t) * B); memcpy(ids + B, dst + s, sizeof(int32_t) * B);
                CHECK(zo_gru_update(memory, lu, messages, mts, flags, N, D, msg, ids, 2 * B, w_ih, w_hh, b_ih, b_hh, 1 + (s / B) % 3) >= 0);
            }
        }
        CHECK(zo_gru_update(memory, lu, messages, mts, flags, N, D, msg, NULL, 0, w_ih, w_hh, b_ih, b_hh, 2) >= 0);
        for (int i = 0; i < N; ++i) CHECK(flags[i] == 0);
        float *pr = malloc(sizeof(float) * B);
        CHECK(zo_affinity(out, out + (size_t)B * H, B, H, a1w, a1b, a2w, a2b, pr) == 0);
        for (int i = 0; i < B; ++i) CHECK(pr[i] >= 0.f && pr[i] <= 1.f);
        free(memory); free(efeat); free(tw); free(lu); free(messages); free(mts); free(flags); free(scratch);
        free(fc1_w); free(fc1_b); free(fc2_w); free(fc2_b); free(fc1s_w); free(fc1s_b); free(fc2s_w); free(fc2s_b);
        free(w_ih); free(w_hh); free(b_ih); free(b_hh); free(a1w); free(a1b); free(a2w); free(a2b);
        free(qn); free(gn); free(ge); free(gd); free(gw); free(out); free(pr);
    }
    free(indptr); free(nbr); free(eid); free(ats);
    free(nodes); free(on); free(oe); free(od); free(ow);
    free(src); free(dst); free(neg); free(ts); free(eidx);
    printf("sanitize_driver: ok\n");
    return 0;
}
