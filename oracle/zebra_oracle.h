/*
 * zebra_oracle.h -- CPU restatement of the reference's T-PPR + aggregate +
 * memory-update path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the timed
 * CPU baseline ("port").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under zebra_amd/ links, imports or
 * calls it; the product path fails loudly when the HIP library is missing.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout).  Parity of this restatement is pinned by the golden
 * vectors in tests/golden/ (generated from the reference source itself, see
 * tests/golden/gen_golden.py).
 */
#ifndef ZEBRA_ORACLE_H
#define ZEBRA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- numba arithmetic (third-party dependency of the reference) ---------- */

/* numba 0.54.1 np.argsort quicksort (numba/misc/quicksort.py) over n float64
 * values; writes the permutation into r[0..n).  Call sites in the reference:
 * utils/util.py:258,555,658,762,851. */
void zo_numba_argsort(const double *a, int32_t n, int32_t *r);

/* numba pow(float64, int64) (numba/cpython/numbers.py:207-243); call site
 * utils/util.py:208. */
double zo_numba_int_pow(double a, int64_t b);

/* ---- streaming T-PPR (utils/util.py:391-873) ----------------------------- */

typedef struct zo_tppr zo_tppr;

/* tppr_finder.__init__ + reset_tppr (utils/util.py:393-434). */
zo_tppr *zo_tppr_create(int64_t num_nodes, int32_t k, int32_t n_tppr,
                        const double *alpha_list, const double *beta_list);
void zo_tppr_destroy(zo_tppr *h);
/* reset_tppr (utils/util.py:419-434). */
void zo_tppr_reset(zo_tppr *h);
/* deep copy of the live state into / out of a second handle of equal shape. */
int zo_tppr_copy(zo_tppr *dst, const zo_tppr *src);

/* streaming_topk (utils/util.py:473-576) / streaming_topk_no_fake (:682-782)
 * / single_streaming_topk (:581-679) / compute_val_tppr update loop (:787-870).
 *   nodes  : int32 [n_roles*B]  = [src | dst | (neg)]
 *   ts     : float64 [>=B]      (only the first B are read, :499)
 *   eidx   : int64 [B]
 *   n_roles: 3 (with negatives) or 2 (no_fake)
 *   emit   : 0 = update only (compute_val_tppr), 1 = also fill outputs
 *   model  : -1 = every model, else only that model (single_streaming_topk)
 *   out_*  : [n_models_emitted][n_roles*B][k], caller-zeroed is NOT required:
 *            every row is written (all-zero row when the dict is empty).
 * returns 0, or -1 on an out-of-range node id. */
int zo_tppr_stream(zo_tppr *h, const int32_t *nodes, const double *ts,
                   const int64_t *eidx, int64_t B, int32_t n_roles,
                   int32_t emit, int32_t model, int32_t *out_nodes,
                   int32_t *out_eidx, float *out_dt, float *out_w);

/* state export (dict items in iteration order), model m:
 *   len int32[N], norm float64[N], eidx int64[N*k], node int64[N*k],
 *   ts float64[N*k], w float64[N*k] (slots >= len are zero). */
void zo_tppr_export(const zo_tppr *h, int32_t m, int32_t *len, double *norm,
                    int64_t *eidx, int64_t *node, double *ts, double *w);

/* the same arrays for the n nodes ids[0..n) only (full-size graphs: the touched rows). */
int zo_tppr_export_rows(const zo_tppr *h, int32_t m, const int64_t *ids, int64_t n, int32_t *len, double *norm,
                        int64_t *eidx, int64_t *node, double *ts, double *w);

/* the same arrays for the n nodes ids[0..n) only, written INTO the state (bench.py: warm start). */
int zo_tppr_import_rows(zo_tppr *h, int32_t m, const int64_t *ids, int64_t n, const int32_t *len,
                        const double *norm, const int64_t *eidx, const int64_t *node, const double *ts,
                        const double *w);

/* ---- static adjacency + pruning T-PPR (utils/util.py:90-276) ------------- */

/* get_neighbor_finder (utils/util.py:90-107): undirected adjacency, each
 * node's entries stably sorted by timestamp.  indptr int64[num_nodes+1];
 * nbr/eid int32[2E]; ats float64[2E]. */
int zo_csr_build(const int32_t *src, const int32_t *dst, const int64_t *eidx,
                 const double *ts, int64_t E, int64_t num_nodes,
                 int64_t *indptr, int32_t *nbr, int32_t *eid, double *ats);

/* NeighborFinder.find_before (utils/util.py:152-154): number of entries of
 * node v with timestamp strictly below t. */
int64_t zo_find_before(const int64_t *indptr, const double *ats, int32_t v,
                       double t);

/* NeighborFinder.get_pruned_topk (utils/util.py:185-276).  Rows whose
 * dictionary is empty are left untouched (:241-242), all others are fully
 * written.  returns 0, -1 on bad id. */
int zo_pruned_topk(const int64_t *indptr, const int32_t *nbr,
                   const int32_t *eid, const double *ats, int64_t num_nodes,
                   const int32_t *q_nodes, const double *q_ts, int64_t nq,
                   int32_t width, int32_t depth, double alpha, double beta,
                   int32_t k, int32_t *out_nodes, int32_t *out_eidx,
                   float *out_dt, float *out_w);

/* ---- gather + TimeEncode + transform + weighted sum ----------------------- */

/* GraphDiffusionEmbedding.compute_embedding_tppr_ensemble, eval mode
 * (modules/embedding_module.py:243-276) with transform / transform_source
 * (:320-328) and TimeEncode.forward (model/time_encoding.py:23-28).
 *   memory [num_nodes][D], efeat [num_edges][F], time_w [T]
 *   nodes int32[N]; per model m: nbr/eix int32[N][k], dt/w float32[N][k]
 *   (model-major, i.e. [M][N][k])
 *   fc1_w [D][D+F+T], fc1_b [D], fc2_w [D][D], fc2_b [D]   (torch Linear layout)
 *   fc1s_w [D][D], fc1s_b, fc2s_w [D][D], fc2s_b
 *   out [N][D*(M+1)]
 * n_threads: OpenMP threads over rows (1 = serial). */
int zo_embed(const float *memory, const float *efeat, const float *time_w,
             int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F,
             int32_t T, const int32_t *nodes, int64_t N, int32_t M, int32_t k,
             const int32_t *nbr, const int32_t *eix, const float *dt,
             const float *w, const float *fc1_w, const float *fc1_b,
             const float *fc2_w, const float *fc2_b, const float *fc1s_w,
             const float *fc1s_b, const float *fc2s_w, const float *fc2s_b,
             float *out, int32_t n_threads);

/* ---- memory: last-message store + GRU update ------------------------------ */

/* TGN.get_raw_messages + Memory.store_raw_messages
 * (model/tgn_model.py:204-226, modules/memory.py:27-30): over the 2B sequence
 * [src|dst] keep the LAST occurrence of each node; message =
 * [mem[node] | mem[partner] | efeat[eidx] | cos((f32(t) - last_update[node])*w)].
 * Overwrites messages[node], msg_ts[node]; sets flags[node]=1.
 * scratch: int32[num_nodes], all -1 on entry and on exit (last-occurrence
 * table; persistent so that a call costs O(B), not O(num_nodes)).
 * returns the number of unique nodes, or -1 on bad id. */
int64_t zo_store_messages(const float *memory, const float *last_update,
                          const float *efeat, const float *time_w,
                          int64_t num_nodes, int64_t num_edges, int32_t D,
                          int32_t F, int32_t T, const int32_t *src,
                          const int32_t *dst, const double *ts,
                          const int64_t *eidx, int64_t B, float *messages,
                          float *msg_ts, uint8_t *flags, int32_t *scratch);

/* SequenceMemoryUpdater.update_memory / update_memory_in_test with
 * nn.GRUCell (modules/memory_updater.py:29-57,95-98).
 *   ids == NULL: every flagged node (update_memory_in_test, flags cleared);
 *   ids != NULL: the flagged subset of ids[0..n_ids) (update_memory; flags are
 *                cleared for ALL ids, as TGN does at model/tgn_model.py:155-157,
 *                170-172 via clear_messages(unique_positives)).
 *   GRU weights in torch layout: w_ih [3D][msg], w_hh [3D][D], b_ih/b_hh [3D],
 *   gate order r,z,n.
 * returns the number of rows updated. */
int64_t zo_gru_update(float *memory, float *last_update, const float *messages,
                      const float *msg_ts, uint8_t *flags, int64_t num_nodes,
                      int32_t D, int32_t msg_dim, const int32_t *ids,
                      int64_t n_ids, const float *w_ih, const float *w_hh,
                      const float *b_ih, const float *b_hh, int32_t n_threads);

/* MergeLayer link scorer (utils/util.py:14-26, model/tgn_model.py:185-188):
 * out[i] = sigmoid(fc2(relu(fc1([x1[i] | x2[i]])))), H = row width of x1/x2. */
int zo_affinity(const float *x1, const float *x2, int64_t rows, int32_t H,
                const float *fc1_w, const float *fc1_b, const float *fc2_w,
                const float *fc2_b, float *out);

#ifdef __cplusplus
}
#endif
#endif
