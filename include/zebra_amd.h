/*
 * zebra_amd.h -- C-ABI of libzebra_amd.so, the MI355X (gfx950) implementation
 * of Zebra's T-PPR + top-k aggregate + memory-update hot path.
 *
 * The reference has no FFI: its boundary is a set of Python attribute calls
 * (SURVEY.md section 8b).  Each entry point below names the reference method
 * it replaces (paths relative to the reference checkout).  The Python classes
 * in zebra_amd/ (tppr_finder, NeighborFinder, GraphDiffusionEmbedding, Memory,
 * GRUMemoryUpdater, TGN) bind these with ctypes; INTEGRATION.md shows the stub
 * a reference maintainer would add.
 *
 * Conventions
 *   - every function returns an int status (ZT_OK == 0) and never throws;
 *   - "dev" pointers are device (HBM) addresses, "host" pointers host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it and is stream-ordered; the library is not
 *     thread-safe per handle (the reference holds the GIL throughout);
 *   - node ids and edge ids are validated on the device: an out-of-range id
 *     rejects the whole call BEFORE any state is modified and latches
 *     ZT_ERR_RANGE in the handle's status word (zt_tppr_status / the status
 *     out-parameter), because the reference's Numba code would read out of
 *     bounds there (SURVEY.md 8b "Errors").
 */
#ifndef ZEBRA_AMD_H
#define ZEBRA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZT_OK 0
#define ZT_ERR_ARG (-1)         /* bad argument (NULL, non-positive size, k > ZT_MAX_K ...) */
#define ZT_ERR_RANGE (-2)       /* node / edge id out of range */
#define ZT_ERR_HIP (-3)         /* a HIP runtime call failed (see zt_last_error) */
#define ZT_ERR_UNSUPPORTED (-4) /* shape outside what the kernels were built for */
#define ZT_ERR_TIMEOUT (-5)     /* an in-kernel dependency wait hit its spin bound */

#define ZT_MAX_K 63             /* top-k width of the tuned paths: one dictionary entry per lane of a wavefront (streaming update with
                                 * hub chains, pruning query) */
#define ZT_MAX_K_WIDE 255       /* streaming T-PPR accepts k up to here (the reference's --topk is unbounded, train.py:46): beyond
                                 * ZT_MAX_K a correct-first path takes over -- one wavefront per model applies the edges in order
                                 * (csrc/tppr_wide.hpp), same state layout, same results as the oracle bit for bit, not tuned.  zt_embed
                                 * takes such rows through the generic kernel's 16-tile instantiation (k <= 80 in any case; up to 255
                                 * over the projected table, where a query row's tile fits LDS); the training kernels k <= 80; the
                                 * pruning query any k up to here (its kept set strides over the lanes; the selection beyond
                                 * ZT_MAX_K is the generic replay of numba_sort.hpp) */

/* Human-readable description of the last failure on this thread. */
const char *zt_last_error(void);
/* Library / build information ("zebra_amd <version> gfx950"). */
const char *zt_version(void);

/* HIP streams restricted to the compute units [cu_lo, cu_hi) (CU masking):
 * the dependency-bound T-PPR kernel and the throughput-bound aggregation can
 * then run side by side without sharing CUs.  New work; the reference is
 * single-stream (train.py:145-146). */
int zt_stream_create_masked(void **stream_out, int32_t cu_lo, int32_t cu_hi);
int zt_stream_destroy(void *stream);

/* Kernel selection, for VALIDATION.  Several steps of the path have more than one kernel -- a latency-organised
 * one for small batches, a throughput-organised one for large ones, a generic one for shapes outside the reference's
 * default widths (train.py:33-36) -- and the library picks by shape.  zt_set_kernel_choice pins the pick process-wide
 * (value 0 = the library's choice again) so that tests can hold the kernels against each other, and against the
 * oracle, on the same inputs.  A pinned kernel that cannot take a shape falls back to the library's pick.  Not a
 * tuning interface: every kernel of a selector computes the same function. */
#define ZT_CHOICE_AGGREGATE 0   /* neighbour aggregation (zt_embed): ZT_AGG_GENERIC */
#define ZT_CHOICE_EMBED_OUT 1   /* output layers (zt_embed): ZT_OUT_TILED, ZT_OUT_LATENCY, ZT_OUT_PERSIST */
#define ZT_CHOICE_GRU 2         /* zt_gru_update: ZT_GRU_TILE, ZT_GRU_SPLIT */
#define ZT_CHOICE_MESSAGES 3    /* zt_store_messages: ZT_MSG_ONE, ZT_MSG_TWO (batch positions per wavefront) */
#define ZT_CHOICE_TPPR_CHAIN 4  /* hub chains of zt_tppr_stream: ZT_CHAIN_SINGLE (one position per critical section: the
                                 * library's pick).  ZT_CHAIN_PAIRED / _SPINE / _DUO were built, are bit-exact and were measured
                                 * SLOWER: they are compiled into variant builds only
                                 * (tools/build_variant.sh, sources under tools/exp/variants/); the product library answers
                                 * ZT_ERR_UNSUPPORTED */
#define ZT_CHOICE_TPPR_PREPASS 5 /* dependency prepass of a launch of more than 4 096 accesses: ZT_PREPASS_LAUNCHES (one kernel per step,
                                 * eleven launches: the library's pick); ZT_PREPASS_COOP (one cooperative kernel with grid barriers:
                                 * not faster) likewise in variant builds only */
#define ZT_CHOICE_GROUP_RELEASE 6 /* zt_pipeline_*: when the aggregation of a batch may start whose streaming T-PPR update shares a launch
                                 * with other batches.  ZT_RELEASE_MEMBER: as soon as that batch's rows are written (a counter per batch
                                 * inside the launch: the library's pick); ZT_RELEASE_LAUNCH: when the whole launch has ended (an event;
                                 * launch groups then taper towards the end of the batches in sight); ZT_RELEASE_LAUNCH_FULL: the
                                 * same without the taper (uniform launches for counter collection, which serialises kernels and so
                                 * cannot run a release by member: tools/profile_round.sh) */
#define ZT_CHOICE_COUNT 7
#define ZT_AGG_GENERIC 1
#define ZT_OUT_TILED 1
#define ZT_OUT_LATENCY 2
#define ZT_OUT_PERSIST 3
#define ZT_GRU_TILE 1
#define ZT_GRU_SPLIT 2
#define ZT_CHAIN_SINGLE 1
#define ZT_CHAIN_PAIRED 2
#define ZT_CHAIN_SPINE 3   /* one wave per chain runs every critical section with the hub's row in registers (tools/exp/variants/tppr_spine.hpp) */
#define ZT_CHAIN_DUO 4     /* spine mode with the weights' recurrence (network + lane shift) on a wave of its own, ahead of the spine */
#define ZT_PREPASS_LAUNCHES 1
#define ZT_PREPASS_COOP 2
#define ZT_RELEASE_MEMBER 1
#define ZT_RELEASE_LAUNCH 2
#define ZT_RELEASE_LAUNCH_FULL 3
#define ZT_MSG_ONE 1
#define ZT_MSG_TWO 2
int zt_set_kernel_choice(int32_t which, int32_t value);

/* Per-kernel timing with HIP events recorded on the launch stream (replaces
 * the reference's unsynchronised wall-clock accumulators t_tppr etc.,
 * modules/embedding_module.py:73,220-225).  Names: tppr_prepass, tppr_stream,
 * tppr_cleanup, pruned_topk, embed_prep, fc1_agg, embed_out, store_messages,
 * gru_update.  zt_profile_enable(n): 0 off, 1 every launch, n > 1 every n-th
 * launch of each kernel (sampling).  zt_profile_read synchronises the device. */
int zt_profile_enable(int on);
int zt_profile_reset(void);
int zt_profile_read(const char *name, int64_t *count, double *total_ms);

/* ------------------------------------------------------------------------ */
/* Streaming T-PPR  --  utils/util.py:391-873 (class tppr_finder)            */
/* ------------------------------------------------------------------------ */

typedef struct zt_tppr zt_tppr;

/* tppr_finder.__init__ (utils/util.py:393-400): allocates the device-resident
 * state [n_tppr][num_nodes][k] (key, timestamp, weight) + length + normaliser,
 * all zero.  alpha/beta are host arrays of n_tppr doubles. */
int zt_tppr_create(zt_tppr **out, int64_t num_nodes, int32_t k, int32_t n_tppr,
                   const double *alpha_host, const double *beta_host);
int zt_tppr_destroy(zt_tppr *h);
/* Processes whose T-PPR update kernels share this device's compute units (default 1: the device is this
 * process's own -- one rank per GPU, the supported multi-GPU layout).  n > 1 (several ranks rehearsing on
 * ONE GPU): every launch takes 1/n of the CUs of its stream, so that the grids of all sharers are resident
 * together, and runs without hub chains, whose workgroups wait for one another.  New work; the reference
 * is one process on one device (train.py:145-146). */
int zt_tppr_set_device_share(zt_tppr *h, int32_t n_processes);
/* Statistics of the hub chains since the last call (synchronises `stream`, clears the counters): out5[0] pairs of
 * consecutive chain positions claimed by one wavefront, [1] pairs applied in ONE critical section, [2] / [3] pairs left
 * to the single hop before / inside the section (a precondition failed), [4] positions taken singly.
 * ZT_CHAIN_SPINE / ZT_CHAIN_DUO: [1] = sections the spine ran, [3] = positions it left to their helpers.
 * All zero in the product library, whose one chain mode keeps no statistics on its hot path (variant builds:
 * tools/exp/variants/tppr_pair.hpp, tppr_spine.hpp; the reference applies the edges one by one, utils/util.py:495-574.) */
int zt_tppr_chain_stats(zt_tppr *h, int64_t *out5, void *stream);

/* tppr_finder.reset_tppr (utils/util.py:419-434). */
int zt_tppr_reset(zt_tppr *h, void *stream);

/* Deep snapshot: dst <- src (same shape).  Replaces backup_tppr /
 * restore_tppr / restore_val_tppr (utils/util.py:436-444), which alias the
 * live state in the reference; the Python shim chooses
 * between deep and alias semantics. */
int zt_tppr_copy(zt_tppr *dst, const zt_tppr *src, void *stream);

/* tppr_finder.streaming_topk (utils/util.py:473-576), streaming_topk_no_fake
 * (:682-782), single_streaming_topk (:581-679) and the update loop of
 * compute_val_tppr (:787-870).
 *   nodes_dev : int32 [n_roles*B] = [src | dst | (neg)]
 *   ts_dev    : float64 [B]   (the reference reads only timestamps[:B], :499)
 *   eidx_dev  : int64 [B]
 *   n_roles   : 3 (with negatives) or 2
 *   emit      : 1 = fill the four output arrays, 0 = update only
 *   model     : -1 = all models, else only that one
 *   out_*_dev : [n_emitted_models][n_roles*B][k] int32/int32/float32/float32;
 *               every row is written (zeros when the dictionary is empty).
 *   plan_token: 0, or the token zt_tppr_plan returned for exactly this call.
 * Edges are applied in batch order, each seeing the state left by the edges
 * before it, exactly as the reference's sequential loop.
 * Errors: a batch with an out-of-range id is NOT applied, its output rows read
 * as empty dictionaries (zeros), and ZT_ERR_RANGE is latched in the handle:
 * every later zt_tppr_stream / zt_tppr_plan call returns it (as soon as the
 * device's write is visible to the host; no synchronisation is added) until
 * zt_tppr_status has reported and cleared it.  ZT_ERR_TIMEOUT likewise. */
int zt_tppr_stream(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev,
                   const int64_t *eidx_dev, int64_t B, int32_t n_roles,
                   int32_t emit, int32_t model, int32_t *out_nodes_dev,
                   int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev,
                   uint64_t plan_token, void *stream);

/* Optional, no reference counterpart: run the dependency prepass of a coming
 * zt_tppr_stream call ahead of time on another stream.  The prepass (which
 * edges of the batch touch the same node, in which order) reads only the ids,
 * never the T-PPR state, so it can overlap the previous call's update kernel.
 * *token_out identifies the plan: the zt_tppr_stream call it was made for
 * passes it back (same nodes_dev, B, n_roles, model; ordering between the two
 * streams is handled inside).  A call with token 0, or with a token whose plan
 * has been dropped (zt_tppr_reset / copy / import, or two newer plans), runs
 * its own prepass.  Two plans can be outstanding.  *token_out = 0 and a no-op
 * for B == 0 or B > 16384 (multi-launch calls plan inline).
 * The k_stream grid is sized for the compute units of the stream that RUNS it
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs of that stream's mask);
 * if that stream turns out to offer fewer CUs than the plan assumed, the grid
 * shrinks and hub chains are dropped for that launch (in-order queue only). */
int zt_tppr_plan(zt_tppr *h, const int32_t *nodes_dev, const int64_t *eidx_dev,
                 int64_t B, int32_t n_roles, int32_t model, uint64_t *token_out,
                 void *stream);

/* Synchronises `stream` and returns the latched status (ZT_OK, ZT_ERR_RANGE
 * or ZT_ERR_TIMEOUT) of the launches so far, clearing it. */
int zt_tppr_status(zt_tppr *h, void *stream);

/* State of model m to / from host arrays, dictionary items in iteration
 * order (attrs PPR_list / norm_list of the reference, utils/util.py:383-386):
 *   len int32[N], norm float64[N], eidx int64[N*k], node int64[N*k],
 *   ts float64[N*k], w float64[N*k]; slots >= len are zero.  Synchronous. */
int zt_tppr_export(zt_tppr *h, int32_t m, int32_t *len_host, double *norm_host,
                   int64_t *eidx_host, int64_t *node_host, double *ts_host,
                   double *w_host);
/* The same for the n nodes ids_host[0..n) only (outputs [n], [n][k]): what a
 * checkpoint of a 10^7-node graph wants -- only the nodes the stream has touched. */
int zt_tppr_export_rows(zt_tppr *h, int32_t m, const int64_t *ids_host, int64_t n,
                        int32_t *len_host, double *norm_host, int64_t *eidx_host,
                        int64_t *node_host, double *ts_host, double *w_host);
int zt_tppr_import(zt_tppr *h, int32_t m, const int32_t *len_host,
                   const double *norm_host, const int64_t *eidx_host,
                   const int64_t *node_host, const double *ts_host,
                   const double *w_host);
/* The same for the n nodes ids_host[0..n) only (inputs [n], [n][k]); rows of
 * other nodes are left as they are.  With zt_tppr_export_rows: a checkpoint
 * that only holds the nodes the stream has touched. */
int zt_tppr_import_rows(zt_tppr *h, int32_t m, const int64_t *ids_host, int64_t n,
                        const int32_t *len_host, const double *norm_host,
                        const int64_t *eidx_host, const int64_t *node_host,
                        const double *ts_host, const double *w_host);

/* ------------------------------------------------------------------------ */
/* Static adjacency + pruning T-PPR -- utils/util.py:90-276                  */
/* ------------------------------------------------------------------------ */

typedef struct zt_csr zt_csr;

/* get_neighbor_finder (utils/util.py:90-107): host edge arrays -> device CSR
 * (indptr int64[N+1], nbr int32[2E], eid int32[2E], ts float64[2E]); each
 * node's entries stably sorted by timestamp. */
int zt_csr_build(zt_csr **out, const int32_t *src_host, const int32_t *dst_host,
                 const int64_t *eidx_host, const double *ts_host, int64_t E,
                 int64_t num_nodes);
/* NeighborFinder(node_to_neighbors, node_to_edge_idxs, node_to_edge_timestamps)
 * (utils/util.py:146-149): adopt an adjacency that is already grouped by node
 * and time-sorted (host CSR arrays). */
int zt_csr_from_sorted(zt_csr **out, const int64_t *indptr_host, const int32_t *nbr_host,
                       const int32_t *eid_host, const double *ts_host, int64_t num_nodes);
int zt_csr_destroy(zt_csr *c);
/* Sizes and a host copy of the adjacency (attrs node_to_neighbors, ... of the
 * reference object). */
int zt_csr_size(const zt_csr *c, int64_t *num_nodes, int64_t *num_entries);
int zt_csr_export(const zt_csr *c, int64_t *indptr_host, int32_t *nbr_host, int32_t *eid_host,
                  double *ts_host);
/* NeighborFinder.find_before (utils/util.py:152-154) for one node: number of
 * entries strictly before t, and copies of that prefix (host, optional). */
int zt_csr_find_before(const zt_csr *c, int32_t v, double t, int64_t *count,
                       int32_t *nbr_host, int32_t *eid_host, double *ts_host,
                       int64_t cap);

/* NeighborFinder.get_pruned_topk (utils/util.py:185-276): rows whose
 * dictionary is empty are left untouched, all others fully written (the
 * reference mutates caller-owned arrays in place). */
int zt_pruned_topk(const zt_csr *c, const int32_t *q_nodes_dev,
                   const double *q_ts_dev, int64_t nq, int32_t width,
                   int32_t depth, double alpha, double beta, int32_t k,
                   int32_t *out_nodes_dev, int32_t *out_eidx_dev,
                   float *out_dt_dev, float *out_w_dev, int32_t *status_dev,
                   void *stream);
/* GraphDiffusionEmbedding.pruning_topk (modules/embedding_module.py:280-297):
 * the same query for every (alpha, beta) model of the ensemble in ONE walk of
 * the adjacency -- which states a query reaches depends on (node, time) only,
 * the models differ in the weights they carry along.  alpha_host / beta_host:
 * n_models host doubles; outputs are [n_models][nq][k], model-major, and follow
 * the same "empty rows stay untouched" rule. */
int zt_pruned_topk_multi(const zt_csr *c, const int32_t *q_nodes_dev,
                         const double *q_ts_dev, int64_t nq, int32_t width,
                         int32_t depth, int32_t n_models, const double *alpha_host,
                         const double *beta_host, int32_t k, int32_t *out_nodes_dev,
                         int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev,
                         int32_t *status_dev, void *stream);

/* ------------------------------------------------------------------------ */
/* Gather + TimeEncode + transform + weighted sum                            */
/*   GraphDiffusionEmbedding.compute_embedding_tppr_ensemble, eval forward   */
/*   (modules/embedding_module.py:243-276, 320-328; model/time_encoding.py)  */
/* ------------------------------------------------------------------------ */

typedef struct {
    const float *fc1_w, *fc1_b;   /* [D][D+F+T], [D]   torch Linear layout */
    const float *fc2_w, *fc2_b;   /* [D][D], [D] */
    const float *fc1s_w, *fc1s_b; /* transform_source */
    const float *fc2s_w, *fc2s_b;
    const float *time_w;          /* TimeEncode frequencies [T] */
} zt_embed_weights;               /* all device pointers */

/*   memory_dev [num_nodes][D], efeat_dev [num_edges][F]
 *   nodes_dev int32[N]; nbr/eix int32 [M][N][k]; dt/w float32 [M][N][k]
 *   out_dev [N][D*(M+1)] = [transform_source(memory[nodes]) | model 0 | ...]
 *   workspace_dev: at least zt_embed_workspace_bytes(N, D, F, T, M, k) bytes
 *   (-1 = shape unsupported). */
int64_t zt_embed_workspace_bytes(int64_t N, int32_t D, int32_t F, int32_t T,
                                 int32_t M, int32_t k);
/*   weights_ready: 0 = pad the weight matrices into the workspace first (needed
 *   once per weight change); 1 = the workspace already holds them (same
 *   workspace, same shape, unchanged weights).
 *   proj_table_dev: NULL, or the projected memory table of zt_project_memory for
 *   memory_dev (see below): fc1's memory columns are then taken from it. */
int zt_embed(const float *memory_dev, const float *efeat_dev, int64_t num_nodes,
             int64_t num_edges, int32_t D, int32_t F, int32_t T,
             const int32_t *nodes_dev, int64_t N, int32_t M, int32_t k,
             const int32_t *nbr_dev, const int32_t *eix_dev, const float *dt_dev,
             const float *w_dev, const zt_embed_weights *weights, float *out_dev,
             void *workspace_dev, int32_t *status_dev, const float *proj_table_dev,
             int32_t weights_ready, void *stream);

/* Projected memory table (no reference counterpart; an algebraic rewrite of
 * modules/embedding_module.py:264-265).  fc1 is linear, so
 *   fc1([memory[nbr] | edge | time]) = W_m memory[nbr] + W_e edge + W_t time + b,
 * and W_m memory[v] depends on the NODE only: P[v] = W_m memory[v] is kept in a
 * table [num_nodes][round_up(D,16)] (zt_project_table_bytes) and refreshed for
 * the <= 2B rows a batch rewrites instead of being recomputed for the 3B*k*M
 * gathered neighbour rows.  rows_dev == NULL: every node (after the memory or
 * the weights changed wholesale); else the rows listed (ids < 0 and entries
 * beyond *count_dev, if given, are skipped).  workspace_dev / ws_N / ws_M / ws_k:
 * the zt_embed workspace (and the N, M, k it was sized for) that holds the
 * padded weights. */
int64_t zt_project_table_bytes(int64_t num_nodes, int32_t D);
int zt_project_memory(const float *memory_dev, int64_t num_nodes, int32_t D,
                      int32_t F, int32_t T, const zt_embed_weights *weights,
                      int32_t weights_ready, const int32_t *rows_dev,
                      const int32_t *count_dev, int64_t max_rows, float *table_dev,
                      void *workspace_dev, int64_t ws_N, int32_t ws_M, int32_t ws_k,
                      void *stream);

/* Training path of the same aggregation: modules/embedding_module.py:227-276 with
 * train=True.  The reference clones the whole memory table per batch (modules/memory_updater.py:79) and lets
 * autograd walk an [N,k,2D+F] tensor; here the lazily updated rows are a compact overlay
 * (row_map_dev[v] = overlay row of node v or -1; NULL = no overlay) and the backward is one fused kernel.
 *   forward : H_dev [M][N][D] = sum_k w_k/sum(w) relu(fc1([memory'[nbr] | ef | cos])), S_dev [M][N] = (sum w != 0)
 *   backward: from dH_dev [M][N][D] ACCUMULATES (+=) dW1_dev [D][D+F+T], db1_dev [D] and d_overlay_dev [U][D]
 *             (the gradient wrt the overlay rows; stored memory rows carry none)
 * fc2 and transform_source act on [N, D] matrices and stay plain GEMMs outside.
 * drop_p > 0: the reference's training dropout of the hidden layer (nn.Dropout(0.1) between fc1's ReLU and fc2,
 *   modules/embedding_module.py:89,323-326) inside the kernels: the keep-mask is a hash of (drop_seed, element),
 *   regenerated by the backward from the same seed -- pass the forward's values.
 * workspaces: zt_embed_workspace_bytes(N, ...) / zt_agg_backward_workspace_bytes(D, F, T). */
int zt_agg_train_forward(const float *memory_dev, const float *overlay_dev, const int32_t *row_map_dev,
                         const float *efeat_dev, int64_t num_nodes, int64_t num_edges, int32_t D,
                         int32_t F, int32_t T, int64_t N, int32_t M, int32_t k, const int32_t *nbr_dev,
                         const int32_t *eix_dev, const float *dt_dev, const float *w_dev,
                         const zt_embed_weights *weights, float *H_dev, float *S_dev,
                         void *workspace_dev, int32_t *status_dev, float drop_p, uint64_t drop_seed,
                         void *stream);
int64_t zt_agg_backward_workspace_bytes(int32_t D, int32_t F, int32_t T);
int zt_agg_train_backward(const float *memory_dev, const float *overlay_dev, const int32_t *row_map_dev,
                          const float *efeat_dev, const float *time_w_dev, int64_t num_nodes,
                          int64_t num_edges, int32_t D, int32_t F, int32_t T, int64_t N, int32_t M,
                          int32_t k, const int32_t *nbr_dev, const int32_t *eix_dev, const float *dt_dev,
                          const float *w_dev, const float *fc1_w_dev, const float *fc1_b_dev,
                          const float *dH_dev, float *dW1_dev, float *db1_dev, float *d_overlay_dev,
                          void *workspace_dev, float drop_p, uint64_t drop_seed, void *stream);

/* ------------------------------------------------------------------------ */
/* Memory: last-message store + GRU update                                   */
/* ------------------------------------------------------------------------ */

/* TGN.get_raw_messages + Memory.store_raw_messages (model/tgn_model.py:204-226,
 * modules/memory.py:27-30).  "Last message wins" over the 2B sequence
 * [src|dst]; overwrites messages[node], msg_ts[node], sets flags[node] = 1.
 *   scratch_dev: int32[num_nodes], all -1 on entry and restored to -1 on exit.
 *   uniq_ids_dev (optional): receives the unique endpoint ids, *n_uniq_dev
 *   their count (order unspecified). */
int zt_store_messages(const float *memory_dev, const float *last_update_dev,
                      const float *efeat_dev, const float *time_w_dev,
                      int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F,
                      int32_t T, const int32_t *src_dev, const int32_t *dst_dev,
                      const double *ts_dev, const int64_t *eidx_dev, int64_t B,
                      float *messages_dev, float *msg_ts_dev, uint8_t *flags_dev,
                      int32_t *scratch_dev, int32_t *uniq_ids_dev,
                      int32_t *n_uniq_dev, int32_t *status_dev, void *stream);

/* Same, but only the winners whose batch position lies in [pos_lo, pos_hi)
 * (positions 0..2B over [src|dst]) are built and stored; "last occurrence" is
 * still resolved over the whole batch.  Used when the endpoints of a batch
 * are sharded across GPUs: every position has at most one winner, so a shard
 * touches at most pos_hi - pos_lo rows. */
int zt_store_messages_range(const float *memory_dev, const float *last_update_dev,
                            const float *efeat_dev, const float *time_w_dev,
                            int64_t num_nodes, int64_t num_edges, int32_t D, int32_t F,
                            int32_t T, const int32_t *src_dev, const int32_t *dst_dev,
                            const double *ts_dev, const int64_t *eidx_dev, int64_t B,
                            int64_t pos_lo, int64_t pos_hi, float *messages_dev,
                            float *msg_ts_dev, uint8_t *flags_dev, int32_t *scratch_dev,
                            int32_t *uniq_ids_dev, int32_t *n_uniq_dev,
                            int32_t *status_dev, void *stream);

typedef struct {
    const float *w_ih, *w_hh; /* [3D][msg], [3D][D]  (torch GRUCell, gates r,z,n) */
    const float *b_ih, *b_hh; /* [3D] */
} zt_gru_weights;             /* device pointers */

/* SequenceMemoryUpdater.update_memory / update_memory_in_test with
 * nn.GRUCell (modules/memory_updater.py:29-57,95-98), followed by
 * Memory.clear_messages (modules/memory.py:59-60) for the same ids.
 *   ids_dev == NULL : every node (update_memory_in_test);
 *   ids_dev != NULL : ids_dev[0 .. *n_ids_dev) if n_ids_dev != NULL, else
 *                     ids_dev[0 .. n_ids); duplicates are allowed.
 *   Flagged ids get last_update = msg_ts and memory = GRU(messages, memory);
 *   flags of all ids are cleared.
 *   flags_dev must be 4-byte aligned and padded to a multiple of 4 bytes (flag
 *   bytes are cleared with 32-bit atomics).
 *   workspace_dev: zt_gru_workspace_bytes(max_rows, D, msg_dim) bytes, where
 *   max_rows = n_ids (or num_nodes when ids_dev == NULL).  On return its first
 *   int32 holds the number of rows updated and the int32 array at byte
 *   zt_gru_rows_offset(D, msg_dim) their ids (duplicates in ids_dev are updated
 *   once).  The packed weights sit between the two, so their place does not
 *   depend on max_rows.
 *   weights_ready: 0 = pack the GRU weights into the workspace first; 1 = the
 *   workspace already holds them (same workspace, same weights; max_rows may
 *   differ from call to call). */
int64_t zt_gru_workspace_bytes(int64_t max_rows, int32_t D, int32_t msg_dim);
int64_t zt_gru_rows_offset(int32_t D, int32_t msg_dim);
int zt_gru_update(float *memory_dev, float *last_update_dev,
                  const float *messages_dev, const float *msg_ts_dev,
                  uint8_t *flags_dev, int64_t num_nodes, int32_t D,
                  int32_t msg_dim, const int32_t *ids_dev, int64_t n_ids,
                  const int32_t *n_ids_dev, const zt_gru_weights *weights,
                  void *workspace_dev, int32_t weights_ready, void *stream);

/* ------------------------------------------------------------------------ */
/* Training-side dense operators on exact-f32 MFMA (csrc/train_ops.hip).       */
/*   What the reference's autograd does with nn.GRUCell (get_updated_memory,   */
/*   modules/memory_updater.py:61-90,95-98) and nn.Linear (fc2,                */
/*   transform_source: modules/embedding_module.py:86-98,320-328) on the       */
/*   compact rows of a training step.                                          */
/* ------------------------------------------------------------------------ */
/* C[M][N] = op(A)[M][K] op(B)[K][N] (+ C if accumulate), row-major float32;
 * op(A)(m, k) = trans_a ? A[k * lda + m] : A[m * lda + k], likewise B. */
int zt_gemm_f32(const float *A_dev, const float *B_dev, float *C_dev, int64_t M, int64_t N,
                int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t trans_a,
                int32_t trans_b, int32_t accumulate, void *stream);
/* out[c] (+)= sum over rows of X[r][c] (bias gradients). */
int zt_colsum_f32(const float *X_dev, int64_t rows, int64_t cols, int64_t ldx, float *out_dev,
                  int32_t accumulate, void *stream);
/* The batch's own rows of the lazily updated memory -- get_updated_memory(...)[nodes]
 * (modules/memory_updater.py:61-90, modules/embedding_module.py:320-322): out[i] =
 * overlay[row_map[nodes[i]]] where row_map (NULL: none) names an overlay row, else
 * memory[nodes[i]]; sel[i] = that overlay row or -1.  The backward adds d_out[i] to
 * d_overlay[sel[i]] (atomic: a node may appear several times; the caller zeroes d_overlay). */
int zt_overlay_rows(const float *memory_dev, const float *overlay_dev, const int32_t *row_map_dev,
                    const int32_t *nodes_dev, int64_t n, int32_t D, int64_t num_nodes,
                    float *out_dev, int32_t *sel_dev, void *stream);
int zt_overlay_rows_backward(const float *d_out_dev, const int32_t *sel_dev, int64_t n, int32_t D,
                             float *d_overlay_dev, void *stream);
/* h_out[u] = GRUCell(messages[ids[u]], memory[ids[u]]) for the U rows ids (the
 * lazily updated rows of get_updated_memory); saved [U][4 D] keeps r, z, n and
 * W_hn h + b_hn for the backward.  workspace: zt_gru_train_workspace_bytes. */
int64_t zt_gru_train_workspace_bytes(int64_t U, int32_t D, int32_t msg_dim);
int zt_gru_train_forward(const float *messages_dev, const float *memory_dev,
                         const int32_t *ids_dev, int64_t U, int32_t D, int32_t msg_dim,
                         const zt_gru_weights *weights, float *h_out_dev, float *saved_dev,
                         void *workspace_dev, void *stream);
/* Gradients of the four GRU parameters from d_h [U][D] (written, not
 * accumulated).  Messages and memory are buffers in the reference: they get
 * no gradient.  workspace_dev must be the FORWARD call's workspace, untouched:
 * it holds the gathered rows (the tables themselves change between forward
 * and backward, model/tgn_model.py:155-168). */
int zt_gru_train_backward(const float *d_h_dev, const float *messages_dev,
                          const float *memory_dev, const int32_t *ids_dev, int64_t U,
                          int32_t D, int32_t msg_dim, const float *saved_dev,
                          float *d_w_ih_dev, float *d_w_hh_dev, float *d_b_ih_dev,
                          float *d_b_hh_dev, void *workspace_dev, void *stream);

/* ------------------------------------------------------------------------ */
/* One batch as one call -- TGN.compute_temporal_embeddings, train=False      */
/*   (model/tgn_model.py:124-174) for device-resident batches.  The reference */
/*   issues these steps from Python on one stream (train.py:145-146); here     */
/*   they are enqueued from C++ on three streams: the T-PPR query of batch b+1 */
/*   (side stream, CU-masked to tppr_cus compute units when > 0) beside        */
/*   aggregate + messages + GRU of batch b (main stream, the remaining CUs),   */
/*   the dependency prepass of batch b+2 on a third.                           */
/* ------------------------------------------------------------------------ */
typedef struct {
    zt_tppr *tppr;                 /* streaming strategy: the T-PPR state; else NULL      */
    const zt_csr *csr;             /* pruning strategy: the adjacency; else NULL          */
    int32_t width, depth;          /* pruning: n_degree, n_layer (train.py:25,28)         */
    double alpha[16], beta[16];    /* pruning: per model                                  */
    float *memory, *last_update, *messages, *msg_ts;   /* Memory (modules/memory.py:19-25) */
    uint8_t *flags;                /* pending-message flags, padded to 4 bytes            */
    int32_t *scratch;              /* int32[num_nodes], all -1 (zt_store_messages)        */
    const float *efeat;
    int64_t num_nodes, num_edges;
    int32_t D, F, T, M, k;
    zt_embed_weights ew;
    zt_gru_weights gw;
    void *embed_ws;                /* zt_embed_workspace_bytes(3*max_B, ...)              */
    void *gru_ws;                  /* zt_gru_workspace_bytes(2*max_B, ...)                */
    float *proj_table;             /* zt_project_memory table kept current by the step, or NULL */
    int32_t *status;               /* latched ZT_ERR_RANGE of the aggregate / message kernels   */
    int64_t max_B;                 /* largest batch                                       */
} zt_pipeline_desc;                /* all pointers device memory */

typedef struct {
    const int32_t *src, *dst, *neg;   /* int32[B] each */
    const double *ts;                 /* float64[B]    */
    const int64_t *eidx;              /* int64[B]; also identifies the batch between calls */
    int64_t B;
} zt_batch;

typedef struct zt_pipeline zt_pipeline;
int zt_pipeline_create(zt_pipeline **out, const zt_pipeline_desc *desc, int32_t tppr_cus);
int zt_pipeline_destroy(zt_pipeline *p);
/* The stream the outputs are produced on (hipStream_t): consumers and any collective that follows a step
 * (the row exchange of a sharded run) are enqueued there. */
void *zt_pipeline_main_stream(zt_pipeline *p);
/* New table pointers / handles (Memory tensors replaced, neighbour finder swapped: desc != NULL) and / or
 * weights changed in place (weights_changed: the padded copies in the workspaces are remade). */
int zt_pipeline_update(zt_pipeline *p, const zt_pipeline_desc *desc, int32_t weights_changed);
/* One eval-mode batch.  next / plan (optional): the batches of the following two steps; `next` is QUERIED now
 * (streaming: its T-PPR update is applied), so it must be the batch of the next call.  Rows [row_lo, row_hi)
 * of [src|dst|neg] are embedded into out_emb_dev [row_hi-row_lo][D*(M+1)] and the last messages of the
 * endpoints at positions [pos_lo, pos_hi) of [src|dst] are stored and applied (a single GPU passes 0, 3B, 0,
 * 2B; shards of a multi-GPU run their slices and exchange the touched rows afterwards).  No host
 * synchronisation; errors of the device side are latched (zt_tppr_status, desc.status). */
int zt_pipeline_step(zt_pipeline *p, const zt_batch *cur, const zt_batch *next, const zt_batch *plan,
                     int64_t row_lo, int64_t row_hi, int64_t pos_lo, int64_t pos_hi, float *out_emb_dev);
/* The same step with a longer view of the stream: ahead[0 .. n_ahead) are the batches that follow `cur`, in order.
 * With zt_pipeline_set_group(p, g), g <= 4, the streaming T-PPR update of g consecutive batches runs as ONE launch
 * (edges applied in order across them, exactly as in separate calls; every batch's output rows form their own
 * block), which pays a launch's fixed costs once per group; the pipeline then queries the group after the current
 * one and plans the one after that, so 3 g + 1 batches in sight keep it full (a group takes at most n - 1 of the n followers in sight along -- the last three batches of a stream are queried one by one --; fewer: smaller groups).  Batches of
 * a group are equally long (the last may be shorter) and together at most 16384 edges.  zt_pipeline_step is this
 * call with ahead = {next, plan}.  The pruning strategy carries no state between batches: its group is 1. */
int zt_pipeline_set_group(zt_pipeline *p, int32_t group);
/* embedding_module.average_topk (modules/embedding_module.py:232-233): with a non-NULL device float, every step
 * over a whole batch also writes the mean over the 2B rows of [src | dst] of the sum of model 0's T-PPR weights
 * there (main stream).  NULL switches it off. */
int zt_pipeline_set_stats(zt_pipeline *p, float *avg_topk_dev);
/* Batches whose T-PPR query was launched ahead of their step and that no step has consumed yet.  While it is
 * non-zero the streaming T-PPR state is AHEAD of the last step: a caller must not query or update the state
 * through zt_tppr_stream (or leave the announced order) before those batches have been stepped. */
int zt_pipeline_outstanding(const zt_pipeline *p);
/* Returns ZT_ERR_TIMEOUT (once) when a kernel of an EARLIER step gave up a bounded in-kernel wait -- the gate between the
 * output layers and the GRU update in their shared launch waits at most 4 s for the source path's reads, then leaves the
 * memory rows of its tile untouched and reports to the status word of the descriptor and to a host-mapped latch that
 * this call looks at on entry: that step's memory update is incomplete. */
int zt_pipeline_step_ahead(zt_pipeline *p, const zt_batch *cur, const zt_batch *ahead, int32_t n_ahead,
                           int64_t row_lo, int64_t row_hi, int64_t pos_lo, int64_t pos_hi, float *out_emb_dev);
/* n consecutive whole-batch steps from one host call (the batch loop of evaluation/evaluation.py:19-45): step b sees
 * batches b + 1 .. b + look as `ahead`, never beyond the n given; its [3 B_b][D (M + 1)] embeddings go to
 * out_emb_dev + b * out_stride floats (out_stride 0: one buffer, overwritten every step). */
/* ---- the row exchange of a multi-GPU run inside the native step loop (no reference counterpart: the
 * reference is one process on one device, train.py:145-146).  One process per GPU; T-PPR state and tables replicated;
 * rank r embeds the rows shard_range(3B, r, world) of every batch, stores and applies the last messages of the winners at
 * positions shard_range(2B, r, world), and after the GRU update of the batch the rows every rank rewrote -- [id | memory
 * row | last_update], and [message row | message time] when with_messages -- are all-gathered with ONE fixed-size
 * collective (cap_rows >= ceil(2 max_B / world) rows per rank), scattered into the local tables, and the projected table
 * follows: all on the pipeline's main stream, enqueued by zt_pipeline_step_ahead / zt_pipeline_run themselves once
 * zt_pipeline_set_exchange has been called (row_lo .. pos_hi of the step call are then this rank's shard, as before).
 *   ZT_XCHG_RCCL: RCCL's ncclAllGather (xGMI), one rank per GPU; unique_id = the 128 bytes zt_exchange_unique_id wrote
 *                 on rank 0, distributed by the caller; zt_exchange_create is collective (ncclCommInitRank);
 *   ZT_XCHG_SHM : tests / rehearsals with several ranks on ONE GPU (RCCL refuses that layout): a POSIX shared-memory
 *                 segment named shm_name, two host synchronisations per step.
 * with_messages = 0 is enough for the eval protocol: a batch's messages are consumed by the GRU update of the step that
 * stored them (model/tgn_model.py:159-172); a caller that compares the messages table across ranks switches it on. */
#define ZT_XCHG_RCCL 1
#define ZT_XCHG_SHM 2
typedef struct {
    int32_t rank, world, transport, with_messages;
    const void *unique_id;         /* ZT_XCHG_RCCL: 128 bytes */
    const char *shm_name;          /* ZT_XCHG_SHM */
    int64_t cap_rows;
    float *memory, *last_update, *messages, *msg_ts;   /* the tables of zt_pipeline_desc */
    int32_t D, msg_dim;
} zt_exchange_desc;
typedef struct zt_exchange zt_exchange;
int zt_exchange_unique_id(void *id_out, int64_t bytes);
int zt_exchange_create(zt_exchange **out, const zt_exchange_desc *desc);
int zt_exchange_set_tables(zt_exchange *x, float *memory, float *last_update, float *messages, float *msg_ts);
int zt_exchange_destroy(zt_exchange *x);
/* x != NULL: every step of this pipeline ends with the exchange (and zt_pipeline_run shards every batch by x's rank and
 * world: out_emb_dev + b * out_stride then receives the rank's rows_hi - rows_lo embedding rows of step b); NULL: off.
 * The pipeline does not own x. */
int zt_pipeline_set_exchange(zt_pipeline *p, zt_exchange *x);
int zt_pipeline_run(zt_pipeline *p, const zt_batch *batches, int32_t n, int32_t look, float *out_emb_dev,
                    int64_t out_stride);

/* ------------------------------------------------------------------------ */
/* TemporalAttentionLayer.forward -- model/temporal_attention.py:7-68        */
/*   NOT on the reference's live path (never reachable from train.py); built */
/*   because the north star names it.  Eval forward (dropout = identity).    */
/* ------------------------------------------------------------------------ */
typedef struct {
    const float *q_w;            /* multi_head_target.q_proj_weight [E][E], E = D+T      */
    const float *k_w, *v_w;      /* k_proj_weight, v_proj_weight    [E][Ek], Ek = D+F+T   */
    const float *in_b;           /* in_proj_bias [3E] (q, k, v)                          */
    const float *out_w, *out_b;  /* out_proj [E][E], [E]                                 */
    const float *m1_w, *m1_b;    /* merger.fc1 [hidden][E+D], [hidden]                   */
    const float *m2_w, *m2_b;    /* merger.fc2 [out_dim][hidden], [out_dim]              */
} zt_attn_weights;               /* device pointers */

/*   src [N][D], src_time [N][T], nbr_feat [N][k][D], edge_feat [N][k][F],
 *   nbr_time [N][k][T], mask uint8 [N][k] (non-zero = padding)
 *   -> out [N][out_dim], attn_w [N][k] (head-averaged attention weights). */
int64_t zt_attention_workspace_bytes(int32_t D, int32_t F, int32_t T, int32_t n_head,
                                     int32_t hidden, int32_t out_dim, int32_t k);
int zt_temporal_attention(const float *src_dev, const float *src_time_dev,
                          const float *nbr_feat_dev, const float *edge_feat_dev,
                          const float *nbr_time_dev, const uint8_t *mask_dev, int64_t N,
                          int32_t k, int32_t D, int32_t F, int32_t T, int32_t n_head,
                          int32_t hidden, int32_t out_dim, const zt_attn_weights *weights,
                          float *out_dev, float *attn_w_dev, void *workspace_dev,
                          void *stream);

/* ------------------------------------------------------------------------ */
/* Link scoring and link-prediction metrics      */
/* ------------------------------------------------------------------------ */
typedef struct {
    const float *fc1_w, *fc1_b;  /* affinity_score.fc1 [H][2H], [H]   (MergeLayer, utils/util.py:14-26) */
    const float *fc2_w, *fc2_b;  /* affinity_score.fc2 [1][H], [1]                                     */
} zt_affinity_weights;           /* device pointers */

/* TGN.compute_edge_probabilities' scorer (model/tgn_model.py:185-188):
 *   emb_dev [3B][H] = the batch's embeddings [src | dst | neg], H = D * (n_tppr + 1) in {200, 300};
 *   prob_dev [2B]   = sigmoid(fc2(relu(fc1([src | dst])))) for the B positive pairs, then the same for (src, neg).
 * workspace_dev: zt_affinity_workspace_bytes(max_B, H) bytes (-1 = H unsupported), sized for ws_max_B >= B;
 * weights_ready as for zt_embed (0 also clears the workspace's tile counters: use it for a fresh workspace). */
int64_t zt_affinity_workspace_bytes(int64_t max_B, int32_t H);
int zt_affinity(const float *emb_dev, int64_t B, int32_t H, const zt_affinity_weights *weights,
                float *prob_dev, void *workspace_dev, int64_t ws_max_B, int32_t weights_ready,
                void *stream);
/* evaluation/evaluation.py:34-45 and train.py:218-227 without the host: out_dev[0..3) (float64) =
 * (average_precision_score, roc_auc_score, mean(pos >= neg)) of B positive and B negative scores, scikit-learn's
 * definitions (distinct thresholds, ties included); accumulate != 0 adds to out_dev.  2B <= 16384. */
int zt_link_metrics(const float *pos_dev, const float *neg_dev, int64_t B, double *out_dev,
                    int32_t accumulate, void *stream);
/* The scorer as the tail of the native step: with non-NULL weights every zt_pipeline_step_ahead over a WHOLE batch
 * (rows [0, 3B)) also scores that batch's 2B pairs behind its aggregation (main stream).  prob_dev: [2][2 * max_B]
 * floats, consecutive steps alternate between the halves.  workspace_dev as for
 * zt_affinity with ws_max_B = the pipeline's max_B.  NULL weights: off.
 * zt_pipeline_last_scores: makes `stream` wait for the last scoring and returns where its 2B probabilities are. */
int zt_pipeline_set_scoring(zt_pipeline *p, const zt_affinity_weights *weights, void *workspace_dev,
                            float *prob_dev);
int zt_pipeline_last_scores(zt_pipeline *p, void *stream, float **prob_out, int64_t *B_out);

/* ------------------------------------------------------------------------ */
/* One-node multi-GPU exchange of touched rows.  The reference  */
/* has no distributed code; this is the data-path step around the one RCCL     */
/* all-gather per batch.  A row travels as float32                             */
/*   [ id (int32 bits) | row of table 0 | row of table 1 | ... ],              */
/* tables being [num_rows][width] float32 arrays (memory, last_update,         */
/* messages, message timestamps).                                              */
/* ------------------------------------------------------------------------ */
typedef struct zt_row_tables {
    float *ptr[8];       /* device pointers */
    int32_t width[8];    /* floats per row */
    int32_t n;           /* tables in use, 1..8 */
} zt_row_tables;

/* out_dev [cap][1 + sum(width)]: slot r carries ids_dev[r] and its rows for
 * r < *n_valid_dev, id = -1 and zeros beyond (fixed-size payload). */
int zt_pack_rows(const zt_row_tables *tables, const int32_t *ids_dev,
                 const int32_t *n_valid_dev, int64_t cap, float *out_dev,
                 void *stream);
/* Every received row with id >= 0 overwrites that row of the local tables. */
int zt_scatter_rows(const zt_row_tables *tables, const float *recv_dev,
                    int64_t rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ZEBRA_AMD_H */
