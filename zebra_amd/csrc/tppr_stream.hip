// Streaming top-k T-PPR maintenance on MI355X (gfx950).
//
// Replaces tppr_finder.streaming_topk & friends (reference utils/util.py:391-873).
// Compile with -ffp-contract=off: the reference rounds every float64 multiply
// and add separately (utils/util.py:523-541) and results must be bit-exact.
//
// Design (DESIGN.md "P1"):
//   * State is device resident.  A node's dictionary (one per model) is a ROW
//     of 8-byte GRANULES {tag:32 | payload:32}: header (len, norm) and, per
//     entry, key = (edge_idx, node), timestamp, weight split into 32-bit
//     halves, stored as six entry-major arrays so that lane j owns entry j.
//     Every granule of a row carries the tag (launch epoch, writer ordinal) of
//     the edge that wrote it: a row is self-validating, and "the data is the
//     flag" (cdna_hip_programming.md Guideline 16, recipe R2): a dependent edge
//     re-reads the row until all tags match -- ONE memory round trip per hop of
//     a hub chain, no release fence, no acquire, no separate flag.
//   * The edges of a batch must be applied in order, each reading the rows the
//     previous edges wrote.  A prepass groups the batch's node accesses by node
//     (atomic count / reserve / fill) and gives every access (a) the ordinal of
//     the last earlier WRITER of its node = the tag to expect, and (b) if its
//     immediate predecessor on that node was only a reader (a negative
//     sample), that edge's index: writers must not overwrite a row such a
//     reader has not read yet, which is what the per-edge "reads done" flag is
//     for.
//   * The update kernel is persistent: one wavefront per (edge, model) task,
//     tasks dequeued in order from an atomic head (a task only ever waits on
//     tasks that are already resident -> no deadlock); it loads its rows
//     (polling tags where a predecessor exists, with a back-off proportional to
//     the number of chain hops still ahead), publishes "reads done", merges in
//     LDS, prunes with the exact numba argsort semantics, and stores the new
//     rows as write-through (sc1) granules, fire and forget.
//   * The kernel's time is the longest chain of edges through one node times
//     the time of one hop.  The most-touched nodes of a launch (hubs) get a
//     workgroup of their own whose waves take the hub's edges in order and pass
//     the hub's row through an LDS mailbox; a hub row goes to memory only when
//     its next accessor reads it from there (process_edge: hub_to_memory).
//   * The prepass reads only node / edge ids.  The handle keeps two sets of its
//     buffers, so zt_tppr_plan can run it for a later call on another stream
//     while k_stream still works on the previous set.
//   * A merge matches keys through a per-wave hash table in LDS (all pairs on
//     slot collisions); the top-k prune is rank counting in registers, a
//     quicksort replay on the ranks when ties decide, LDS / sequential replays
//     for more than 64 candidates (numba_sort.hpp).
#include "numba_sort.hpp"
#ifdef ZT_CHAIN_VARIANTS                 // variant builds only (tools/build_variant.sh): the chain modes that were measured slower
#include "tppr_pair.hpp"                 // (tools/exp/variants/: two positions per critical section; spine / duo via tppr_chain.hpp)
#else
#include "tppr_chain.hpp"
#endif

#include "tppr_wide.hpp"                 // ZT_MAX_K < k <= ZT_MAX_K_WIDE: one wavefront per model, edges in order

#include <cstdlib>
#include <vector>

static_assert(WAVES_PER_WG == zt::TPPR_WAVES_PER_WG && REG_K_MAX == zt::TPPR_REG_K_MAX, "tppr_state.hpp");

namespace {

#ifndef ZT_STREAM_BOUNDS
#define ZT_STREAM_BOUNDS (WAVE * WAVES_PER_WG)      // (tools/exp/bounds_exp.sh: 768 = three waves per SIMD, 168 VGPRs)
#endif
// (variant builds, -DZT_CHAIN_VARIANTS, instantiate MODE 1 .. 3 as well; the product library has MODE 0 alone)
// PAIRS: the instantiation whose chain waves may take two positions per critical section (tppr_pair.hpp).  A kernel of its
// own: the paired hop is a real call that needs all 256 registers a wave may have at two waves per SIMD, and a k_stream of
// 256 registers leaves the message kernels no room beside its workgroups on the T-PPR stream's CUs (round 5: with the call
// merely PRESENT in the one kernel, k_last_pos waited 270 us per launch for a free register file and the driver-timed C5
// step went from 0.38 to 0.54 ms).
// MODE 2 (spine, tppr_chain.hpp): one wave of a chain workgroup runs every critical section, the others prepare and finish.
template <int MODE>
__global__ __launch_bounds__(ZT_STREAM_BOUNDS) void k_stream(zt_tppr h, StreamArgs A)
{
    constexpr bool PAIRS = MODE == 1;
    (void)PAIRS;
    __shared__ WaveLds lds[WAVES_PER_WG];
    __shared__ Mail mail;
    __shared__ int ch_edge[CH_MAX], ch_partner[CH_MAX], ch_wop[CH_MAX], ch_pch[CH_MAX];   // chain workgroups: HopRec
    WaveLds &L = lds[threadIdx.x / WAVE];
    const int lane = lane_id();
    if (ld_agent(h.ctl + 2) == ZT_ERR_RANGE) {         // rejected by k_count: the state is not touched,
        if (A.emit) {                                  // the output rows read as empty dictionaries
            const int k = h.k;
            for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
                 q < (long long)A.n_models * A.n_roles * A.B * k; q += (long long)gridDim.x * blockDim.x) {
                const long long c = q % k, row = q / k;
                const long long i = row % A.B, role = (row / A.B) % A.n_roles, mo = row / ((long long)A.B * A.n_roles);
                const long long o = (mo * A.out_rows + role * A.role_stride + i) * k + c;
                A.out_nodes[o] = 0; A.out_eidx[o] = 0; A.out_dt[o] = 0.f; A.out_w[o] = 0.f;
            }
        }
        if (A.member_done != nullptr) {                // batches released one by one: all of them, once every workgroup's rows are out
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0 && atomicAdd(A.member_done + zt::TPPR_MAX_MEMBERS, 1) == (int)gridDim.x - 1)
                for (int g = 0; g < zt::TPPR_MAX_MEMBERS; ++g) st_agent(A.member_done + g, 0x7fffffff);
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);                     // chain hops must not queue behind throughput kernels
    if (threadIdx.x < MAIL_R) { mail.slot[threadIdx.x].seq_set = 0; mail.slot[threadIdx.x].seq_ord = 0; mail.slot[threadIdx.x].seq_free = 0; }
    if (threadIdx.x == 0) mail.head = 0;
    if (MODE >= 2 && threadIdx.x < PREP_R) { mail.prep[threadIdx.x].seq = 0; mail.prep[threadIdx.x].res = 0; mail.prep[threadIdx.x].a_seq = 0; }
    if (MODE >= 2 && threadIdx.x == 0) { mail.a_gen = 0; mail.a_restart = 0; }
#ifdef ZT_CRIT
    if (threadIdx.x == 0) mail.t_start = (long long)__builtin_readcyclecounter();
#endif
    for (int q = lane; q < HTAB; q += WAVE) L.htab[q] = -1;
    __syncthreads();
    const int n_models = A.n_models;

    // ---- chain workgroups: blocks [0, chains * models) each own one hub of one model ----
    const int n_chain_wg = A.use_chains ? h.ctl[4] * n_models : 0;
    if ((int)blockIdx.x < n_chain_wg) {
        const int c = blockIdx.x / n_models, mo = blockIdx.x % n_models;
        const long long hub = h.chain_node[c];
        int len = h.chain_len[c];
        len = len < CH_MAX ? len : CH_MAX;
        const int *edges = h.chain_edges + c * CH_MAX;
        // what every hop needs to know about its edge (HopRec), once, into LDS
        for (int q = threadIdx.x; q < len; q += blockDim.x) {
            const int e = edges[q];
            const long long u = A.nodes[e], v = A.nodes[A.role_stride + e];
            const int role_h = u == hub ? 0 : 1;
            ch_edge[q] = e;
            ch_partner[q] = u == v ? -1 : (int)(u == hub ? v : u);
            ch_wop[q] = h.wo[(1 - role_h) * A.B + e];
            ch_pch[q] = u == v ? -1 : h.hv[(1 - role_h) * A.B + e];
        }
        // version 0 of the hub's row = the row as the launch finds it in `rows` (nobody stores there before the chain's last
        // hop): copied by the last wave right away, so that nothing that reads it -- the hub's readers ahead of its first
        // edge, that edge's partner task, another hub's chain whose first edge is this hub's first too -- waits for a hop
        if (len > 0 && (int)(threadIdx.x / WAVE) == WAVES_PER_WG - 1) {
            const int m0 = A.m_lo + mo;
            Row r0;
            (void)load_row(h, m0, hub, lane, 0u, r0);
            store_row_at(hub_version(h, m0, c, 0), h.k, lane, r0.len, r0.key, r0.ts, r0.w, r0.norm, (A.epoch << ORD_BITS) | 1u);
            // the hub's norm at every position of the chain: norm <- norm * beta + beta per edge (utils/util.py:567-572), a
            // recurrence that does not look at the rows -- one lane runs it from the norm the launch finds (zt_tppr::hubscale)
            if (lane == 0) {
                const double beta = h.beta[m0];
                double pn = r0.norm;
                double *tab = hub_scale(h, m0, c, 0);
                for (int t = 0; t <= len; ++t) { st_agent(tab + 4 * t, pn); pn = pn * beta + beta; }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (len > 0) {
            // ... and the scale factors that follow from it, two float64 divisions per position, every position by a thread of
            // its own: scale_s1 = norm / norm' * beta, scale_s2 = beta / norm' * (1 - alpha) with norm' = norm * beta + beta (:519-522)
            const int m0 = A.m_lo + mo;
            const double alpha = h.alpha[m0], beta = h.beta[m0];
            for (int t = threadIdx.x; t <= len; t += blockDim.x) {
                double *e = hub_scale(h, m0, c, t);
                const double pn = ld_agent(e), nn = pn * beta + beta;
                e[1] = nn;
                e[2] = pn / nn * beta;
                e[3] = beta / nn * (1.0 - alpha);
            }
            __syncthreads();
        }
        // chain_waves (ZT_CHAIN_WAVES, default all eight) waves take hops: what a hop needs besides the hub's update -- the
        // partner's update, the emission -- runs elsewhere (process_chain_partner), but a hop's preparation and its
        // off-chain half (replay, order, stores) still add up to ~5 hop periods of one wave's time.
        if ((int)(threadIdx.x / WAVE) >= A.chain_waves) return;
#ifdef ZT_CHAIN_VARIANTS
        const bool spine_on = MODE >= 2 && h.k <= REG_K_MAX && len > 0;
        if (spine_on && threadIdx.x < WAVE) { chain_spine<MODE == 3>(h, lds, lane, &mail, len); return; }
        if (spine_on && MODE == 3 && threadIdx.x < 2 * WAVE) { chain_weights(h, lds, lane, &mail, len); return; }
#else
        constexpr bool spine_on = false;
#endif
        ChainHint hint;
        hint.norm_out = 0.0; hint.tpos = -1;
        // (Assigning hop t to wave t mod 8 statically -- so that the SIMD mate of the wave on the chain is the one four
        // hops away -- was measured: the hops then run strictly one after the other, 10x slower.  A wave claiming its next
        // hop right after publishing, to have the partner's row requested early, was measured too: claims then follow
        // the order of publication -- the same fixed rotation -- and a wave with a long off-chain half holds the chain up.)
        for (;;) {
            const int t = wave_claim_lds(&mail.head, 1);          // one lane's add, no branch (common.hpp)
            if (t >= len) break;
            if (t == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);   // (the first hop has no mailbox to wait for)
#define ZT_U(x) __builtin_amdgcn_readfirstlane(x)
#ifdef ZT_CHAIN_VARIANTS
            // ---- two positions in one critical section (tppr_pair.hpp): both edges have a partner other than the hub, not
            // ---- the same one, and nobody has claimed position t + 1 yet ----
            if (PAIRS && t >= 1 && t + 1 < len && h.k <= PAIR_K_MAX) {
                const int pa = ZT_U(ch_partner[t]), pb = ZT_U(ch_partner[t + 1]);
                if (pa >= 0 && pb >= 0 && pa != pb) {
                    int old = 0;
                    if (lane == 0) old = atomicCAS(&mail.head, t + 1, t + 2);
                    old = ZT_U(old);
                    if (old == t + 1) {
                        chain_stat(h.ctl, lane, ST_PAIR_CLAIM);
                        HopRec r1, r2;
                        r1.partner = pa; r1.wo_p = ZT_U(ch_wop[t]); r1.pchain = ZT_U(ch_pch[t]);
                        r2.partner = pb; r2.wo_p = ZT_U(ch_wop[t + 1]); r2.pchain = ZT_U(ch_pch[t + 1]);
                        const int ea = ZT_U(ch_edge[t]), eb = ZT_U(ch_edge[t + 1]);
                        const int e2 = t + 2 < len ? ZT_U(ch_edge[t + 2]) : -1, e0 = ZT_U(ch_edge[t - 1]);
                        PairCtx X;
                        X.rows = h.rows; X.hubver = h.hubver; X.cdone = h.cdone; X.ctl = h.ctl; X.tsv = A.tsv; X.eidx = A.eidx;
                        X.N = h.N; X.m = A.m_lo + mo; X.alpha = h.alpha[X.m]; X.beta = h.beta[X.m]; X.epoch = A.epoch; X.k = h.k; X.rg = h.rg;
                        const double nrm = chain_hop2(X, &L, lane, ea, eb, &mail, hub, e2, t, hint.norm_out, hint.tpos, c, r1, r2);
                        if (nrm != 0.0) { hint.norm_out = nrm; hint.tpos = t + 1; continue; }
                        // a precondition failed before anything was written: the two hops one after the other
                        __builtin_amdgcn_s_setprio(1);
                        if (!chain_hop(h, A, L, lane, ea, mo, &mail, hub, e0, eb, t, &hint, c, r1))
                            process_edge(h, A, L, lane, ea, mo, &mail, hub, e0, eb, t, &hint, c);
                        __builtin_amdgcn_s_setprio(1);
                        if (!chain_hop(h, A, L, lane, eb, mo, &mail, hub, ea, e2, t + 1, &hint, c, r2))
                            process_edge(h, A, L, lane, eb, mo, &mail, hub, ea, e2, t + 1, &hint, c);
                        continue;
                    }
                }
            }
#endif
            if (MODE != 0) chain_stat(h.ctl, lane, ST_SINGLE);      // (statistics of the alternative chain modes only: the default kernel's hot path
                                                                    //  carries no atomic to a shared word that nobody reads -- round-5 advisor)
            const int pe = t > 0 ? ZT_U(ch_edge[t - 1]) : -1, ne = t + 1 < len ? ZT_U(ch_edge[t + 1]) : -1, ce = ZT_U(ch_edge[t]);
            HopRec rec;
            rec.partner = ZT_U(ch_partner[t]); rec.wo_p = ZT_U(ch_wop[t]); rec.pchain = ZT_U(ch_pch[t]);
#undef ZT_U
#ifdef ZT_CHAIN_VARIANTS
            if (spine_on && (pe < 0 || rec.partner < 0)) spine_post_none(&mail, lane, t);      // (chain_hop does not take these)
#endif
            if (!chain_hop(h, A, L, lane, ce, mo, &mail, hub, pe, ne, t, &hint, c, rec, spine_on))
                process_edge(h, A, L, lane, ce, mo, &mail, hub, pe, ne, t, &hint, c);
        }
        return;                                           // chain workgroups take no general tasks (letting them join the
                                                          // general queue once their chain is done was measured: no difference)
    }

    // ---- general queue: every (edge, model) task not owned by a chain, in order ----
    const int total = A.B * n_models;
    // member_done: the batch of the task this wave has just finished, not counted yet.  Counted behind the NEXT dequeue -- its
    // returning atomic has waited for every store of the task anyway, the emission's write-through stores included -- so the
    // count costs the queue one wave-level add per task and no wait of its own.
    int pend = -1;
    for (;;) {
        // Dequeue with NO divergent branch in the program (an `if (lane == 0)` here gets jump-threaded with lane-0 code at the
        // end of the previous iteration and the structurizer then replays the body for the remaining lanes: seen in the ISA)
        // and no scan over the lanes either (what `lane == 0 ? 1 : 0` as the addend compiled to until round 6): common.hpp
        const int idx = wave_claim_global(h.ctl + 1, 1);
        if (pend >= 0) {
            wave_add_global(A.member_done + pend, 1);      // (the claim above has waited for every store of the task: vmcnt(0))
            pend = -1;
        }
#ifdef ZT_CRIT
        if (idx >= total) {                                // diagnostic: when the last general wave left, and who it was
            if (lane == 0) atomicMax((unsigned long long *)&g_crit[8191 * 16 + 0], (unsigned long long)__builtin_readcyclecounter());
            return;
        }
#else
        if (idx >= total) return;
#endif
        const int i = idx / n_models;
        if (A.member_done != nullptr) pend = A.sub_B > 0 ? i / A.sub_B : 0;
        if (A.use_chains && h.owner_of[i] >= 0) {           // its chain applies the hub's update, this wave the rest
            process_chain_partner(h, A, L, lane, i, idx % n_models);
            continue;
        }
        process_edge(h, A, L, lane, i, idx % n_models, nullptr, -1, -1, -1, 0);
    }
}

// tags -> 0 for every granule (run when the launch epoch wraps)
__global__ void k_retag(u64 *rows, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        rows[i] &= 0xffffffffull;
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------
extern "C" int zt_tppr_create(zt_tppr **out, int64_t num_nodes, int32_t k, int32_t n_tppr,
                              const double *alpha_host, const double *beta_host)
{
    if (!out || num_nodes <= 0 || k <= 0 || n_tppr <= 0 || !alpha_host || !beta_host) {
        set_error("zt_tppr_create: bad argument");
        return ZT_ERR_ARG;
    }
    if (k > ZT_MAX_K_WIDE || n_tppr > 16) {
        set_error("zt_tppr_create: k=%d (max %d) or n_tppr=%d (max 16) unsupported", k, ZT_MAX_K_WIDE, n_tppr);
        return ZT_ERR_UNSUPPORTED;
    }
    zt_tppr *h = new zt_tppr();
    memset(h, 0, sizeof(*h));
    h->N = num_nodes; h->k = k; h->M = n_tppr; h->rg = HDR + 6 * k;
    for (int m = 0; m < n_tppr; ++m) { h->alpha[m] = alpha_host[m]; h->beta[m] = beta_host[m]; }
    const size_t rows = (size_t)n_tppr * (size_t)num_nodes;
    ZT_HIP(hipMalloc(&h->rows, rows * h->rg * sizeof(u64)));
    ZT_HIP(hipMalloc(&h->done, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    ZT_HIP(hipMemset(h->done, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    ZT_HIP(hipMalloc(&h->cdone, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    ZT_HIP(hipMemset(h->cdone, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * n_tppr));
    h->hubver = nullptr;                              // hub-row versions: allocated by the first launch that runs hub chains
                                                      // (run_chunk) -- snapshot / backup handles, which only ever receive
                                                      // zt_tppr_copy, never pay for them (65 MB at k = 20, M = 2)
    for (int q = 0; q < 2; ++q) {
        zt_tppr::PlanSet &P = h->set[q];
        ZT_HIP(hipMalloc(&P.cnt, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMalloc(&P.off, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMalloc(&P.ctl, CTL_WORDS * sizeof(int)));
        ZT_HIP(hipMalloc(&P.slot, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.list, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.wo, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.pflag, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.nxt, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.chain_of, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMalloc(&P.hot_node, sizeof(int) * (MAX_HOT + MAX_BIG)));
        ZT_HIP(hipMalloc(&P.hot_cnt, sizeof(int) * MAX_HOT));
        ZT_HIP(hipMalloc(&P.chain_node, sizeof(int) * MAX_CHAINS));
        ZT_HIP(hipMalloc(&P.chain_len, sizeof(int) * MAX_CHAINS));
        ZT_HIP(hipMalloc(&P.chain_edges, sizeof(int) * MAX_CHAINS * CH_MAX));
        ZT_HIP(hipMalloc(&P.owner_of, sizeof(int) * MAX_CHUNK));
        ZT_HIP(hipMalloc(&P.hv, sizeof(int) * 3 * MAX_CHUNK));
        ZT_HIP(hipMemset(P.chain_of, 0xff, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMemset(P.chain_len, 0, sizeof(int) * MAX_CHAINS));
        ZT_HIP(hipMemset(P.cnt, 0, (size_t)num_nodes * sizeof(int)));
        ZT_HIP(hipMemset(P.ctl, 0, CTL_WORDS * sizeof(int)));
        ZT_HIP(hipEventCreateWithFlags(&P.planned, hipEventDisableTiming | zt::sync_event_flags()));
        ZT_HIP(hipEventCreateWithFlags(&P.consumed, hipEventDisableTiming | zt::sync_event_flags()));
        P.used = false;
        P.valid = false;
    }
    ZT_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->latch_host), sizeof(int), hipHostMallocMapped));
    *h->latch_host = 0;
    ZT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->latch_dev), h->latch_host, 0));
    for (int q = 0; q < 2; ++q)                        // note_timeout finds the latch through ctl[14..15]
        ZT_HIP(hipMemcpy(h->set[q].ctl + 14, &h->latch_dev, sizeof(int *), hipMemcpyHostToDevice));
    h->plan_serial = 0;
    h->next_set = 0;
    use_set(h, 0);
    hipDeviceProp_t prop;
    int dev = 0;
    ZT_HIP(hipGetDevice(&dev));
    ZT_HIP(hipGetDeviceProperties(&prop, dev));
    h->n_cu = prop.multiProcessorCount;
    // every workgroup of a k_stream grid that runs hub chains must be resident: ask the runtime how many
    // fit on a CU (LDS, registers) rather than estimating it
    int per_cu = 0;
    ZT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream<0>, WAVE * WAVES_PER_WG, 0));
    h->wg_per_cu = per_cu;
    h->epoch = 0;
    h->share = 1;
    int rc = zt_tppr_reset(h, nullptr);
    if (rc != ZT_OK) return rc;
    ZT_HIP(hipDeviceSynchronize());
    *out = h;
    return ZT_OK;
}

extern "C" int zt_tppr_destroy(zt_tppr *h)
{
    if (!h) return ZT_OK;
    (void)hipDeviceSynchronize();
    (void)hipFree(h->rows); (void)hipFree(h->done); (void)hipFree(h->cdone);
    if (h->hubver) (void)hipFree(h->hubver);
    if (h->hubscale) (void)hipFree(h->hubscale);
    if (h->latch_host) (void)hipHostFree(h->latch_host);
    for (int q = 0; q < 2; ++q) {
        zt_tppr::PlanSet &P = h->set[q];
        (void)hipFree(P.cnt); (void)hipFree(P.off); (void)hipFree(P.ctl); (void)hipFree(P.slot); (void)hipFree(P.list);
        (void)hipFree(P.wo); (void)hipFree(P.pflag); (void)hipFree(P.nxt); (void)hipFree(P.chain_of);
        (void)hipFree(P.hot_node); (void)hipFree(P.hot_cnt); (void)hipFree(P.chain_node); (void)hipFree(P.chain_len);
        (void)hipFree(P.chain_edges); (void)hipFree(P.owner_of); (void)hipFree(P.hv);
        (void)hipEventDestroy(P.planned); (void)hipEventDestroy(P.consumed);
    }
    delete h;
    return ZT_OK;
}

extern "C" int zt_tppr_set_device_share(zt_tppr *h, int32_t n_processes)
{
    if (!h || n_processes < 1) { set_error("zt_tppr_set_device_share: bad argument"); return ZT_ERR_ARG; }
    h->share = n_processes;
    h->set[0].valid = h->set[1].valid = false;       // plans made for the other layout are void
    return ZT_OK;
}

// hub-chain statistics since the last call (summed over both plan sets, then cleared): out[0] pairs of positions claimed by
// one wave, [1] pairs completed in ONE critical section, [2] pairs left to the single hop in preparation (norm not
// predictable, a key shared between the partners' rows, a slot collision ...), [3] ... inside the section (a key of the hub's
// row in a partner's, a tie that reaches across the first cut ...), [4] positions taken singly.
extern "C" int zt_tppr_chain_stats(zt_tppr *h, int64_t *out5, void *stream)
{
    if (!h || !out5) return ZT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    ZT_HIP(hipStreamSynchronize(s));
    for (int q = 0; q < 5; ++q) out5[q] = 0;
#ifdef ZT_PAIR_STAT
    {
        long long g[16] = {0};
        ZT_HIP(hipMemcpyFromSymbol(g, HIP_SYMBOL(g_pstat), sizeof(g)));
        long long z[16] = {0};
        ZT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pstat), z, sizeof(z)));
        if (g[4] > 0)
            fprintf(stderr, "[pair-stat] %lld pairs: preparation %.0f clocks, waiting for the turn %.0f, critical section %.0f, off-chain halves %.0f\n",
                    g[4], (double)g[0] / g[4], (double)g[2] / g[4], (double)g[1] / g[4], (double)g[3] / g[4]);
        if (g[4] > 0)
            fprintf(stderr, "[pair-stat]   section: row + checks %.0f, network %.0f, masks + tests + slots %.0f, ring slot %.0f\n",
                    (double)g[5] / g[4], (double)g[6] / g[4], (double)g[7] / g[4], (double)g[8] / g[4]);
    }
#endif
    for (int q = 0; q < 2; ++q) {
        int c[5] = {0, 0, 0, 0, 0};
        ZT_HIP(hipMemcpy(c, h->set[q].ctl + ST_PAIR_CLAIM, sizeof(c), hipMemcpyDeviceToHost));
        ZT_HIP(hipMemset(h->set[q].ctl + ST_PAIR_CLAIM, 0, sizeof(c)));
        for (int j = 0; j < 5; ++j) out5[j] += c[j];
    }
    return ZT_OK;
}

extern "C" int zt_tppr_reset(zt_tppr *h, void *stream)
{
    if (!h) return ZT_ERR_ARG;
    const size_t rows = (size_t)h->M * (size_t)h->N;
    ZT_HIP(hipMemsetAsync(h->rows, 0, rows * h->rg * sizeof(u64), (hipStream_t)stream));
    h->set[0].valid = h->set[1].valid = false;       // plans made for the old state's stream are void
    return ZT_OK;
}

extern "C" int zt_tppr_copy(zt_tppr *dst, const zt_tppr *src, void *stream)
{
    if (!dst || !src) return ZT_ERR_ARG;
    if (dst->N != src->N || dst->k != src->k || dst->M != src->M) {
        set_error("zt_tppr_copy: shape mismatch");
        return ZT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)src->M * (size_t)src->N * src->rg;
    ZT_HIP(hipMemcpyAsync(dst->rows, src->rows, n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    // tags are relative to the owner's launch epoch: strip them in the copy
    k_retag<<<2048, 256, 0, s>>>(dst->rows, (long long)n);
    ZT_LAUNCH_CHECK();
    dst->set[0].valid = dst->set[1].valid = false;
    return ZT_OK;
}


// k_stream over plan set q (planned on any stream), on stream s.
static int run_chunk(zt_tppr *h, int q, const int32_t *nodes, const double *ts, const long long *eidx,
                     long long role_stride, int B, int n_roles, int emit, int model, long long out_rows, int32_t *on,
                     int32_t *oe, float *od, float *ow, hipStream_t s, bool plan_ordered = false,
                     hipEvent_t *done_out = nullptr, int sub_B = 0, int *member_done = nullptr)
{
    zt_tppr::PlanSet &P = h->set[q];
    if (h->epoch >= EPOCH_MAX) {               // launch epoch about to wrap: forget all tags
        k_retag<<<2048, 256, 0, s>>>(h->rows, (long long)h->M * h->N * h->rg);
        ZT_HIP(hipMemsetAsync(h->done, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * h->M, s));   // flags of old epochs
        ZT_HIP(hipMemsetAsync(h->cdone, 0, sizeof(unsigned) * (size_t)MAX_CHUNK * h->M, s));
        if (h->hubver) ZT_HIP(hipMemsetAsync(h->hubver, 0, (size_t)h->M * MAX_CHAINS * (CH_MAX + 1) * h->rg * sizeof(u64), s));
        h->epoch = 0;
    }
    h->epoch += 1;
    if (!plan_ordered) ZT_HIP(hipStreamWaitEvent(s, P.planned, 0));      // (the caller has ordered s behind the plan already)
    use_set(h, q);
    h->run_cus = tppr_stream_cus(h, s);
    // The plan sized the grid for the CUs it expected.  If THIS stream offers fewer (a CU mask the plan did
    // not know about), shrink the grid to what is resident here; hub chains only run when their workgroups
    // plus a general queue fit, otherwise every edge goes through the in-order queue, which needs no
    // residency (a task only waits on tasks already dequeued by resident waves).
    int grid = P.grid, use_chains = 1;
    int resident = h->run_cus * (h->wg_per_cu > 0 ? h->wg_per_cu : 1) / (h->share > 1 ? h->share : 1);
    if (resident < 1) resident = 1;
    if (grid > resident) {
        grid = resident;
        use_chains = P.max_chains <= tppr_chains_for_grid(h, grid, model < 0 ? h->M : 1) ? 1 : 0;
    }
    if (use_chains && P.max_chains > 0 && h->hubver == nullptr) {
        // first launch with hub chains on this handle: [M][MAX_CHAINS][CH_MAX + 1][rg] granules, tags 0 (cleared on the
        // launch stream, in front of the kernel; once per handle)
        const size_t vb = (size_t)h->M * MAX_CHAINS * (CH_MAX + 1) * h->rg * sizeof(u64);
        ZT_HIP(hipMalloc(&h->hubver, vb));
        ZT_HIP(hipMemsetAsync(h->hubver, 0, vb, s));
        ZT_HIP(hipMalloc(&h->hubscale, (size_t)h->M * MAX_CHAINS * (CH_MAX + 1) * 4 * sizeof(double)));   // (written by every chain workgroup
                                                                                                          //  before it is read: no clear)
    }
    StreamArgs sa;
    sa.use_chains = use_chains;
    sa.nodes = nodes; sa.tsv = ts; sa.eidx = eidx; sa.role_stride = role_stride; sa.B = B; sa.n_roles = n_roles;
    sa.emit = emit; sa.m_lo = model < 0 ? 0 : model; sa.n_models = model < 0 ? h->M : 1; sa.out_rows = out_rows;
    sa.out_nodes = on; sa.out_eidx = oe; sa.out_dt = od; sa.out_w = ow; sa.epoch = h->epoch;
    sa.sub_B = sub_B;
    sa.member_done = (emit && sub_B > 0) ? member_done : nullptr;
    // two chain positions per critical section (tppr_pair.hpp): bit-exact, 79 % of C5's chain positions pair up -- and slower
    // (the section 4.0-4.4 k clocks against 2 x 2.3 k, its preparation 25 k per pair: DESIGN.md section 5): on request only
    const int chain_choice = zt::kernel_choice(ZT_CHOICE_TPPR_CHAIN);
    sa.pairs = chain_choice == ZT_CHAIN_PAIRED ? 1 : 0;
    sa.chain_waves = WAVES_PER_WG;             // (4 / 6 / 8 waves per chain: 1874 / 1557 / 1432 us per four-batch C5 launch, round 3)
#ifdef ZT_CRIT
    static const int crit_multi_env = getenv("ZT_CRIT_MULTI") ? atoi(getenv("ZT_CRIT_MULTI")) : 0;    // (diagnostic build only)
    sa.crit_multi = crit_multi_env;
#else
    sa.crit_multi = 0;
#endif
#ifdef ZT_WAITLOG
    {
        void *wl = nullptr;
        ZT_HIP(hipGetSymbolAddress(&wl, HIP_SYMBOL(g_wl)));
        ZT_HIP(hipMemsetAsync(wl, 0, sizeof(int) * 2 * MAX_CHUNK * 8, s));
        h->dbg_nodes = nodes; h->dbg_stride = role_stride; h->dbg_B = B; h->dbg_roles = n_roles; h->dbg_models = sa.n_models;
    }
#endif
    ZT_PROF_BEGIN(s, P_STREAM);
#ifdef ZT_CHAIN_VARIANTS
    if (sa.pairs) k_stream<1><<<grid, WAVE * WAVES_PER_WG, 0, s>>>(*h, sa);
    else if (chain_choice == ZT_CHAIN_SPINE) k_stream<2><<<grid, WAVE * WAVES_PER_WG, 0, s>>>(*h, sa);
    else if (chain_choice == ZT_CHAIN_DUO) k_stream<3><<<grid, WAVE * WAVES_PER_WG, 0, s>>>(*h, sa);
    else
#endif
    k_stream<0><<<grid, WAVE * WAVES_PER_WG, 0, s>>>(*h, sa);      // (zt_set_kernel_choice refuses the other modes in a build without them)
    ZT_PROF_END(s, P_STREAM);
    ZT_LAUNCH_CHECK();
    ZT_HIP(hipEventRecord(P.consumed, s));
    if (done_out) *done_out = P.consumed;
    P.used = true;
    P.valid = false;
    return ZT_OK;
}


// which values of ZT_CHOICE_TPPR_CHAIN this build of the library can run (runtime.hip: zt_set_kernel_choice)
bool zt::tppr_chain_mode_compiled(int mode)
{
#ifdef ZT_CHAIN_VARIANTS
    return mode >= 0 && mode <= ZT_CHAIN_DUO;
#else
    return mode == 0 || mode == ZT_CHAIN_SINGLE;
#endif
}

// zt_tppr_stream with two extras for callers inside the library (pipeline.hip): plan_ordered = `stream` already
// waits for the stream that made the plan (no second wait packet); *done_out = the event recorded behind the
// (last) update kernel, so that the caller need not record one of its own.  Every packet between two update
// kernels on the T-PPR stream costs ~5 us of the step.
int zt::tppr_stream_ex(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev, const int64_t *eidx_dev, int64_t B,
                       int32_t n_roles, int32_t emit, int32_t model, int32_t *out_nodes_dev, int32_t *out_eidx_dev,
                       float *out_dt_dev, float *out_w_dev, uint64_t plan_token, void *stream, bool plan_ordered,
                       hipEvent_t *done_out, int32_t sub_B, int32_t *member_done_dev)
{
    if (done_out) *done_out = nullptr;
    if (member_done_dev != nullptr && (sub_B <= 0 || (B + sub_B - 1) / sub_B > TPPR_MAX_MEMBERS)) {
        set_error("zt_tppr_stream: batches are released one by one in launches over 2 .. %d batches", TPPR_MAX_MEMBERS);
        return ZT_ERR_ARG;
    }
    if (sub_B < 0 || (sub_B > 0 && (B > MAX_CHUNK || sub_B > B))) {
        set_error("zt_tppr_stream: a launch over several batches must fit one chunk (%d edges)", MAX_CHUNK);
        return ZT_ERR_ARG;
    }
    if (!h || B < 0 || (n_roles != 2 && n_roles != 3) || model >= h->M) {
        set_error("zt_tppr_stream: bad argument");
        return ZT_ERR_ARG;
    }
    if (int st = latched(h, "zt_tppr_stream")) return st;
    if (B == 0) return ZT_OK;
    if (!nodes_dev || !ts_dev || !eidx_dev ||
        (emit && (!out_nodes_dev || !out_eidx_dev || !out_dt_dev || !out_w_dev))) {
        set_error("zt_tppr_stream: NULL buffer");
        return ZT_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const long long *e64 = reinterpret_cast<const long long *>(eidx_dev);
    if (h->k > ZT_MAX_K) {
        // dictionaries wider than a wavefront (tppr_wide.hpp): the ids are checked, then one wavefront per model applies the
        // whole call's edges in order -- no prepass, no plan sets, no launch groups (a caller's sub_B is a single batch here)
        if (sub_B > 0 && sub_B < B) { set_error("zt_tppr_stream: k=%d takes one batch per launch", h->k); return ZT_ERR_UNSUPPORTED; }
        if (B > 0x7fffffffll / 3) { set_error("zt_tppr_stream: batch too large"); return ZT_ERR_ARG; }
        use_set(h, 0);
        k_wide_begin<<<1, 64, 0, s>>>(h->ctl);
        k_wide_check<<<(unsigned)((B * n_roles + 255) / 256 < 1024 ? (B * n_roles + 255) / 256 : 1024), 256, 0, s>>>(
            nodes_dev, e64, B, (int)B, n_roles, h->N, h->ctl, h->latch_dev);
        StreamArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.nodes = nodes_dev; sa.tsv = ts_dev; sa.eidx = e64; sa.role_stride = B; sa.B = (int)B; sa.n_roles = n_roles;
        sa.emit = emit; sa.m_lo = model < 0 ? 0 : model; sa.n_models = model < 0 ? h->M : 1; sa.out_rows = (long long)n_roles * B;
        sa.out_nodes = out_nodes_dev; sa.out_eidx = out_eidx_dev; sa.out_dt = out_dt_dev; sa.out_w = out_w_dev;
        ZT_PROF_BEGIN(s, P_STREAM);
        k_stream_wide<<<sa.n_models, WAVE, 0, s>>>(*h, sa);
        ZT_PROF_END(s, P_STREAM);
        ZT_LAUNCH_CHECK();
        return ZT_OK;
    }
    // a prepass made ahead of time by zt_tppr_plan for exactly this call: the caller hands back the token
    // that plan returned (a recycled device address alone must never select a stale plan)
    if (plan_token != 0) {
        for (int q = 0; q < 2; ++q) {
            zt_tppr::PlanSet &P = h->set[q];
            if (P.valid && P.token == plan_token) {
                if (P.nodes != nodes_dev || P.B != (int)B || P.n_roles != n_roles || P.model != model) {
                    set_error("zt_tppr_stream: plan token %llu was made for another call", (unsigned long long)plan_token);
                    return ZT_ERR_ARG;
                }
                return run_chunk(h, q, nodes_dev, ts_dev, e64, B, (int)B, n_roles, emit, model, (long long)n_roles * B,
                                 out_nodes_dev, out_eidx_dev, out_dt_dev, out_w_dev, s, plan_ordered, done_out, sub_B, member_done_dev);
            }
        }
        // the plan is gone (reset / copy / import, or two newer plans): fall through to an inline prepass
    }
    h->run_cus = tppr_stream_cus(h, s);        // plan and run on the same stream here
    // launches of at most MAX_CHUNK edges: writer ordinals must fit the tag
    for (int64_t c0 = 0; c0 < B; c0 += MAX_CHUNK) {
        const int bc = (int)((B - c0) < MAX_CHUNK ? (B - c0) : MAX_CHUNK);
        const size_t oo = (size_t)c0 * h->k;
        const int q = h->next_set;
        h->next_set ^= 1;
        h->set[q].valid = false;
        int rc = tppr_plan_chunk(h, q, nodes_dev + c0, e64 + c0, B, bc, n_roles, model, s);
        if (rc != ZT_OK) return rc;
        rc = run_chunk(h, q, nodes_dev + c0, ts_dev + c0, e64 + c0, B, bc, n_roles, emit, model,
                       (long long)n_roles * B, emit ? out_nodes_dev + oo : nullptr, emit ? out_eidx_dev + oo : nullptr,
                       emit ? out_dt_dev + oo : nullptr, emit ? out_w_dev + oo : nullptr, s, true, done_out, sub_B,
                       member_done_dev);                                                            // planned on s itself
        if (rc != ZT_OK) return rc;
    }
    return ZT_OK;
}

extern "C" int zt_tppr_stream(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev, const int64_t *eidx_dev,
                              int64_t B, int32_t n_roles, int32_t emit, int32_t model, int32_t *out_nodes_dev,
                              int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev, uint64_t plan_token,
                              void *stream)
{
    return zt::tppr_stream_ex(h, nodes_dev, ts_dev, eidx_dev, B, n_roles, emit, model, out_nodes_dev, out_eidx_dev,
                              out_dt_dev, out_w_dev, plan_token, stream, false, nullptr, 0);
}

#ifdef ZT_CRIT
extern "C" int zt_debug_crit(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_crit), sizeof(long long) * n * 16));
    return ZT_OK;
}
#endif
#ifdef ZT_STAMP
extern "C" int zt_debug_stamps(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n * 4));
    return ZT_OK;
}
extern "C" int zt_debug_paths(int *host)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_paths), sizeof(int) * 8));
    return ZT_OK;
}
extern "C" int zt_debug_stamps3(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps3), sizeof(long long) * n * 8));
    return ZT_OK;
}
extern "C" int zt_debug_hopst(long long *host)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hopst), sizeof(long long) * 16 * 2048 * 8));
    return ZT_OK;
}
extern "C" int zt_debug_stamps2(long long *host, int n)
{
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps2), sizeof(long long) * n * 8));
    return ZT_OK;
}
#endif


extern "C" int zt_tppr_status(zt_tppr *h, void *stream)
{
    if (!h) return ZT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int c[CTL_WORDS] = {0};
    int st = 0;
    ZT_HIP(hipStreamSynchronize(s));
    const int latch = *reinterpret_cast<volatile int *>(h->latch_host);
    *reinterpret_cast<volatile int *>(h->latch_host) = 0;
    for (int q = 0; q < 2 && st == 0; ++q) {          // the status word of either plan set (details of a time-out)
        ZT_HIP(hipMemcpyAsync(c, h->set[q].ctl, sizeof(c), hipMemcpyDeviceToHost, s));
        ZT_HIP(hipStreamSynchronize(s));
        st = c[2];
        if (st != 0) {
            ZT_HIP(hipMemsetAsync(h->set[q].ctl + 2, 0, sizeof(int), s));
            ZT_HIP(hipMemsetAsync(h->set[q].ctl + 13, 0, sizeof(int), s));
            ZT_HIP(hipStreamSynchronize(s));
            use_set(h, q);
        }
    }
    if (st == 0 && latch != 0) {                      // the failing set has been re-planned since: the latch remembers
        st = latch;
        c[13] = 0;
    }
    if (st != 0) {
        if (st == ZT_ERR_RANGE) {
            set_error("node or edge id out of range");
        } else {
            // kind 1: reads-done flag, 2: row tag (node, expect, seen, model), 3: mailbox (edge, expect, seen, prev edge)
#ifdef ZT_WAITLOG
            char msg[8192];
#else
            char msg[480];
#endif
            int o = snprintf(msg, sizeof(msg), "dependency wait timed out; %d waits gave up:", c[13]);
            for (int q = 0; q < CTL_LOG && q < c[13] && o < (int)sizeof(msg) - 80; ++q) {
                const int *r = c + 16 + 8 * q;
                o += snprintf(msg + o, sizeof(msg) - o, " [kind %d: %d expect 0x%x seen 0x%x aux %d wg %d wave %d t %d]", r[0],
                              r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
            }
            set_error("%.470s", msg);
#ifdef ZT_WAITLOG
            fprintf(stderr, "[waitlog] %s\n", msg);
            {
                const int B = h->dbg_B, R = h->dbg_roles;
                std::vector<int> wl(2 * MAX_CHUNK * 8), nodes((size_t)R * B), wo(3 * MAX_CHUNK), own(MAX_CHUNK), cn(MAX_CHAINS),
                    cl(MAX_CHAINS), ce(MAX_CHAINS * CH_MAX);
                (void)hipMemcpyFromSymbol(wl.data(), HIP_SYMBOL(g_wl), wl.size() * 4);
                for (int r = 0; r < R; ++r)
                    (void)hipMemcpy(nodes.data() + (size_t)r * B, h->dbg_nodes + r * h->dbg_stride, (size_t)B * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(wo.data(), h->wo, wo.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(own.data(), h->owner_of, own.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(cn.data(), h->chain_node, cn.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(cl.data(), h->chain_len, cl.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(ce.data(), h->chain_edges, ce.size() * 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "[waitlog] B %d roles %d models %d ctl4(after reset) %d\n", B, R, h->dbg_models, c[4]);
                for (int q = 0; q < MAX_CHAINS; ++q) {
                    fprintf(stderr, "[waitlog] chain %d node %d len %d:", q, cn[q], cl[q]);
                    for (int t = 0; t < cl[q] && t < 12; ++t) fprintf(stderr, " %d", ce[q * CH_MAX + t]);
                    fprintf(stderr, "\n");
                }
                for (int mo = 0; mo < h->dbg_models && mo < 2; ++mo) {
                    int tmin = 0x7fffffff;
                    for (int i = 0; i < B; ++i) {
                        const int *w = wl.data() + (mo * MAX_CHUNK + i) * 8;
                        if (w[0] && w[2] < tmin) tmin = w[2];
                    }
                    int shown = 0;
                    for (int i = 0; i < B && shown < 40; ++i) {
                        const int *w = wl.data() + (mo * MAX_CHUNK + i) * 8;
                        const bool slow = w[0] == 9 && (w[4] - w[2]) > 20000;     // > 25 ms
                        if (w[0] == 9 && !slow && (w[6] & 255) == 0) continue;
                        ++shown;
                        fprintf(stderr, "[waitlog] m%d edge %d (u %d v %d wo %d %d owner %d): state %d wg %d wave %d start %d rows +%d end +%d prev %d fail 0x%x seen 0x%x\n",
                                mo, i, nodes[i], nodes[B + i], wo[i], wo[B + i], own[i], w[0], w[1] / WAVES_PER_WG, w[1] % WAVES_PER_WG, w[2] - tmin,
                                w[3] - w[2], w[4] - w[2], w[5], w[6], w[7]);
                    }
                }
            }
#endif
        }
    }
    return st;
}

