// One eval-mode batch of the hot path as ONE host call (zt_pipeline_step): T-PPR query / update, gather +
// aggregate, last-message store, GRU memory update and the refresh of the projected table are enqueued from
// C++ on three HIP streams, with no Python between the launches.
//
// Mirrors TGN.compute_temporal_embeddings with train=False (reference model/tgn_model.py:124-174) for
// device-resident batches; the reference issues the same steps from Python on one stream
// (train.py:145-146).  The streams:
//   side  the T-PPR query of the batches AFTER the current group (streaming: k_stream, CU-masked to a few
//         compute units because it is latency-bound; pruning: k_pruned_topk) -- the T-PPR state depends only
//         on the edge stream, never on the node memory, so it runs ahead;
//   plan  the dependency prepass of the group after that (reads only ids);
//   main  aggregate + messages + GRU of the current batch (the remaining compute units).
// Three slots (ids + the four T-PPR output arrays each) rotate through plan -> query -> consume.
//
// GROUPS.  A slot holds a GROUP of up to `group` consecutive batches (zt_pipeline_set_group; default 1) whose
// streaming T-PPR update runs as ONE launch: edges are applied in order across the batches exactly as in
// separate calls (the reference's loop has no state between calls other than the dictionaries), each batch's
// output rows form their own block (StreamArgs::sub_B), and the fixed cost of a launch -- the packets between
// two kernels on the T-PPR stream, the kernel's start and tail, its cold instruction cache -- is paid once per
// group.  The caller shows the batches that follow (zt_pipeline_step_ahead); with fewer than 3 * group of
// them in sight the groups simply come out smaller.  The pruning strategy has no state to carry: group = 1.
//
// RELEASE BY MEMBER (round 6).  The aggregation of a group's batch b needs batch b's T-PPR rows, not the end of the launch
// that wrote them: k_stream counts, per member of the group, the (edge, model) tasks whose rows are out (write-through
// stores; Slot::mdone), and the main stream passes a one-wave gate kernel (k_member_gate: bounded wait, ZT_ERR_TIMEOUT to the
// status word and the pipeline's latch) instead of waiting for the launch's event.  So the main stream follows the T-PPR
// stream at a distance of ONE batch whatever the group size: no tapering of the groups towards the end of the batches in
// sight, and a short timed region runs in the launch groups of a long one.  ZT_CHOICE_GROUP_RELEASE = ZT_RELEASE_LAUNCH
// brings the event (and the taper) back -- also what a tool that SERIALISES kernels needs (rocprofv3's counter mode: a kernel
// that waits for a kernel of another stream never sees it run; the gate then gives up after 4 s and reports).
#include "common.hpp"

#include <cstring>

using namespace zt;

#ifndef ZT_MAX_GROUP
#define ZT_MAX_GROUP 8
#endif
constexpr int MAX_GROUP = ZT_MAX_GROUP;
static_assert(MAX_GROUP <= zt::TPPR_MAX_MEMBERS, "a slot's member counters (common.hpp: TPPR_MEMBER_WORDS)");

struct zt_pipeline {
    zt_pipeline_desc d;
    hipStream_t side, main_s, plan_s;
    zt::embed_out_deferred out_gru;     // the output layers of the step at hand, held back for k_out_gru (memory_update.hip)
    hipStream_t msg_s;         // the message build of the current batch, beside its aggregation (reads the memory tables only)
    hipEvent_t step_begin, msgs_done;
    int group;                 // batches per T-PPR launch (streaming)
    bool masked;               // the T-PPR stream and the main stream own disjoint compute units (zt_pipeline_create: tppr_cus > 0)
    struct Slot {
        int32_t *nodes;        // [3 * cap]   src of every member | dst ... | neg ...   (role stride = Btot)
        int32_t *nodes_m;      // [3 * cap]   per member: [src | dst | neg] of that batch (what zt_embed / the GRU read)
        double *ts;            // [3 * cap]   streaming: ts of every member, concatenated; pruning: query time of every row
        int64_t *eidx;         // [cap]       concatenated
        int32_t *buf;          // 4 x [M][3 * cap][k]
        int32_t *on, *oe;      // the group's four output arrays inside buf ([member][M][3 * B][k] each)
        float *od, *ow;
        int32_t *mdone;        // [TPPR_MEMBER_WORDS] per member: the (edge, model) tasks whose output rows are written (k_stream)
        bool by_member;        // the launch counts there: the main stream passes a gate per member instead of waiting for `ready_ev`
        hipEvent_t ready;      // T-PPR outputs complete (side stream)
        hipEvent_t ready_ev;   // the event to wait on for that: `ready`, or the one the T-PPR update recorded itself
        hipEvent_t consumed;   // main stream is done with the slot
        hipEvent_t filled;     // ids copied in (and, streaming, the prepass made)
        const int64_t *key[MAX_GROUP];   // the batches it holds (eidx pointers); n = 0: free
        int64_t B[MAX_GROUP];
        int n, n_done;
        int64_t Btot;
        uint64_t token;        // zt_tppr_plan token, 0 = none
        int64_t q_lo, q_hi;    // pruning strategy: the rows of the (single) member the query was launched for
        bool launched, used;
        bool waited;           // the main stream has been told to wait for `ready_ev` of this launch (a later member need not again)
    } slot[3];
    int next_slot;
    int64_t cap;               // edges a slot can hold
    // compact copies of a row shard's T-PPR outputs (sharded streaming runs only)
    int32_t *sh_on, *sh_oe;
    float *sh_od, *sh_ow;
    bool embed_ready, gru_ready;
    hipEvent_t entry;          // main stream at the moment a group is staged: the batches' tensors are written by then
    bool entry_recorded;       // ... recorded in the step call under way (only the calls that stage a group need it)
    float *avg_topk;           // zt_pipeline_set_stats: mean row sum of model 0's weights over [src | dst] (or NULL)
    // zt_pipeline_set_scoring: the link scorer behind the aggregation of every whole-batch step (or off)
    // On the main stream: on a stream of its own (measured: C5 0.421 instead of 0.41 ms/step, C4 0.256 instead of 0.208) it
    // takes registers and matrix-pipe time from the persistent aggregation kernels, which own a CU each.
    zt_affinity_weights aff;
    void *aff_ws;
    float *prob;               // [2][2 * max_B]: steps alternate between the halves
    bool aff_on, aff_ready;
    hipEvent_t scored[2];
    int score_n;               // scorings enqueued so far (parity = which half / which event)
    int64_t score_B;           // batch size of the last one
    zt_exchange *xchg;         // zt_pipeline_set_exchange: the row exchange of a multi-GPU run at the end of every step (or NULL)
    // Failure latch in host-mapped memory (as zt_tppr's): a bounded in-kernel wait of the step's own kernels that gives up
    // (the gate of k_out_gru / k_out_gru2, memory_update.hip) writes ZT_ERR_TIMEOUT here at system scope, and the NEXT step
    // call fails with it -- no synchronisation, and a caller that never reads the status word still hears of it.
    int *latch_host, *latch_dev;
};

namespace {

struct GroupPtrs {
    const int32_t *src[MAX_GROUP], *dst[MAX_GROUP], *neg[MAX_GROUP];
    const double *ts[MAX_GROUP];
    const int64_t *eidx[MAX_GROUP];
    long long B[MAX_GROUP], off[MAX_GROUP];
    int n;
    long long Btot;
};

// ids of a group into a slot: role-major over the whole group (what the T-PPR launch reads), batch-major per
// member (what the aggregation reads), the concatenated edge ids and times; ts3 = 1: the query time of every row
__global__ void k_stage_group(GroupPtrs g, int32_t *__restrict__ nodes, int32_t *__restrict__ nodes_m,
                              double *__restrict__ ts, long long *__restrict__ eidx, int ts3, int32_t *__restrict__ mdone)
{
    if (blockIdx.x == 0 && (int)threadIdx.x < zt::TPPR_MEMBER_WORDS) mdone[threadIdx.x] = 0;     // (the slot's previous user is done: `consumed`)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.Btot) return;
    int m = 0;
    while (m + 1 < g.n && t >= g.off[m + 1]) ++m;
    const long long i = t - g.off[m], B = g.B[m];
    const int32_t s = g.src[m][i], d = g.dst[m][i], n = g.neg[m][i];
    nodes[t] = s; nodes[g.Btot + t] = d; nodes[2 * g.Btot + t] = n;
    int32_t *nm = nodes_m + 3 * g.off[m];
    nm[i] = s; nm[B + i] = d; nm[2 * B + i] = n;
    const double tt = g.ts[m][i];
    eidx[t] = g.eidx[m][i];
    if (ts3) { double *q = ts + 3 * g.off[m]; q[i] = tt; q[B + i] = tt; q[2 * B + i] = tt; }
    else ts[t] = tt;
}

// embedding_module.average_topk (reference modules/embedding_module.py:232-233): mean over the 2B positive rows of
// the sum of model 0's T-PPR weights; one workgroup, float64 accumulation
__global__ __launch_bounds__(256) void k_avg_topk(const float *__restrict__ w, long long rows, int k, float *out)
{
    __shared__ double part[256];
    double acc = 0.0;
    for (long long q = threadIdx.x; q < rows * k; q += 256) acc += (double)w[q];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = rows > 0 ? (float)(part[0] / (double)rows) : 0.f;
}

bool same_batch(const zt_batch *a, const int64_t *key, int64_t B) { return a->eidx == key && a->B == B; }

zt_pipeline::Slot *find_slot(zt_pipeline *p, const zt_batch *b, int *member)
{
    for (auto &s : p->slot)
        for (int j = 0; j < s.n; ++j)
            if (same_batch(b, s.key[j], s.B[j])) { *member = j; return &s; }
    return nullptr;
}

bool valid_batch(const zt_pipeline *p, const zt_batch *b)
{
    return b != nullptr && b->src && b->dst && b->neg && b->ts && b->eidx && b->B > 0 && b->B <= p->d.max_B;
}

// A new group from `first` and up to want - 1 of the batches that follow it (equal sizes, the last may be shorter;
// one T-PPR launch must cover it), staged on stream `st` once the slot's previous user is done.
// record = false: the caller records `filled` itself (behind the prepass it enqueues after the copy).
int make_group(zt_pipeline *p, const zt_batch *first, const zt_batch *more, int n_more, int want, hipStream_t st,
               zt_pipeline::Slot **out, bool record)
{
    // a slot whose group is used up, else the next in turn (its `consumed` event orders the reuse)
    int pick = -1;
    for (int q = 0; q < 3; ++q) {
        const int c = (p->next_slot + q) % 3;
        if (p->slot[c].n == 0) { pick = c; break; }
    }
    const bool streaming = p->d.tppr != nullptr;
    if (pick < 0) {
        // every slot holds batches that have not been consumed: the caller left the order it announced.  A group that
        // was only planned (or a pruning query, which carries no state) can be dropped; a streaming group whose update
        // has been LAUNCHED cannot -- its edges are in the T-PPR state already and would be applied a second time.
        for (int q = 0; q < 3; ++q) {
            const int c = (p->next_slot + q) % 3;
            if (!p->slot[c].launched || !streaming) { pick = c; break; }
        }
        if (pick < 0) {
            set_error("zt_pipeline_step: batches must be presented in the order they were announced (three groups whose "
                      "T-PPR update has been applied are still waiting to be consumed)");
            return ZT_ERR_ARG;
        }
    }
    p->next_slot = (pick + 1) % 3;
    zt_pipeline::Slot &s = p->slot[pick];
    GroupPtrs g;
    memset(&g, 0, sizeof(g));
    auto add = [&](const zt_batch *b) {
        g.src[g.n] = b->src; g.dst[g.n] = b->dst; g.neg[g.n] = b->neg; g.ts[g.n] = b->ts; g.eidx[g.n] = b->eidx;
        g.B[g.n] = b->B; g.off[g.n] = g.Btot; g.Btot += b->B; ++g.n;
    };
    add(first);
    if (want > MAX_GROUP) want = MAX_GROUP;
    // Leave followers in sight: the aggregation of a group's FIRST batch waits for the whole launch, and after the LAST
    // launch of a stream (or of a timed region) the main stream still has every batch of that launch to aggregate.  So
    // where the view ahead ends the groups taper -- with n_more followers in sight a group takes at most n_more - 1 of
    // them along: the last THREE batches of a region are queried one by one (round 4; two before: the driver's 20-step
    // run ended with a two-batch launch and two aggregations behind it) and the aggregation of each runs beside the update
    // of the next.  A full group needs want + 1 followers in sight (synth.pipeline_look).
    // (round 6: with the batches of a launch released one by one -- see the top of the file -- neither holds: no taper)
    const int release = zt::kernel_choice(ZT_CHOICE_GROUP_RELEASE);
    const bool by_member = streaming && release != ZT_RELEASE_LAUNCH && release != ZT_RELEASE_LAUNCH_FULL;
    if (!by_member && release != ZT_RELEASE_LAUNCH_FULL && n_more <= want) want = n_more >= 2 ? n_more - 1 : 1;
    for (int q = 0; streaming && q < n_more && g.n < want; ++q) {
        const zt_batch *b = more + q;
        // members are equally long, except that the last one may be shorter; everything fits one launch and the slot
        if (!valid_batch(p, b) || g.B[g.n - 1] != first->B || b->B > first->B || g.Btot + b->B > TPPR_MAX_LAUNCH ||
            g.Btot + b->B > p->cap)
            break;
        add(b);
    }
    if (s.used) ZT_HIP(hipStreamWaitEvent(st, s.consumed, 0));
    // (released by member, `consumed` says that every batch's rows were written and read -- not that the launch that wrote
    //  them has ended: its last waves may still be on their way out.  The slot's ids and counters are not touched before it has.)
    if (s.used && s.by_member && s.launched && hipEventQuery(s.ready_ev) != hipSuccess) {
        (void)hipGetLastError();
        ZT_HIP(hipStreamWaitEvent(st, s.ready_ev, 0));
    }
    // the batches' tensors may have been produced on the caller's stream just before this call (the main stream is
    // ordered behind it): the staging copy must not read them earlier
    // (`entry` was recorded when the step call began -- BEFORE the main stream was told to wait for the current group's
    //  T-PPR update: recorded here, behind that wait, it held the staging and the plan of the NEXT groups back until the
    //  current update was done; a quarter of a millisecond at the start of a timed region)
    if (st != p->main_s) {
        if (!p->entry_recorded) { ZT_HIP(hipEventRecord(p->entry, p->main_s)); p->entry_recorded = true; }
        ZT_HIP(hipStreamWaitEvent(st, p->entry, 0));
    }
    k_stage_group<<<(unsigned)((g.Btot + 255) / 256), 256, 0, st>>>(g, s.nodes, s.nodes_m, s.ts,
                                                                     reinterpret_cast<long long *>(s.eidx), streaming ? 0 : 1, s.mdone);
    ZT_LAUNCH_CHECK();
    if (record) ZT_HIP(hipEventRecord(s.filled, st));
    // the four output arrays of the group, back to back in the slot's buffer
    const size_t per = (size_t)p->d.M * 3 * g.Btot * p->d.k;
    s.on = s.buf; s.oe = s.buf + per;
    s.od = reinterpret_cast<float *>(s.buf + 2 * per); s.ow = reinterpret_cast<float *>(s.buf + 3 * per);
    for (int j = 0; j < g.n; ++j) { s.key[j] = g.eidx[j]; s.B[j] = g.B[j]; }
    s.n = g.n; s.n_done = 0; s.Btot = g.Btot; s.token = 0; s.launched = false; s.used = true; s.waited = false;
    s.by_member = by_member && g.n > 1;
    s.q_lo = 0; s.q_hi = 0;
    *out = &s;
    return ZT_OK;
}

// the T-PPR query of the slot's group on the side stream; rows [row_lo, row_hi) only for the pruning strategy
int launch_tppr(zt_pipeline *p, zt_pipeline::Slot &s, int64_t row_lo, int64_t row_hi)
{
    const zt_pipeline_desc &d = p->d;
    // (a group staged steps ago: no wait packet in front of the query -- ~5 us of command-processor time each)
    if (hipEventQuery(s.filled) != hipSuccess) { (void)hipGetLastError(); ZT_HIP(hipStreamWaitEvent(p->side, s.filled, 0)); }
    s.ready_ev = s.ready;
    if (d.tppr != nullptr) {
        // `filled` was recorded behind the prepass (when there is one): no second wait; the event the update kernel
        // records for the plan set doubles as this slot's `ready`
        hipEvent_t done = nullptr;
        int rc = zt::tppr_stream_ex(d.tppr, s.nodes, s.ts, s.eidx, s.Btot, 3, 1, -1, s.on, s.oe, s.od, s.ow, s.token, p->side,
                                    true, &done, s.n > 1 ? (int32_t)s.B[0] : 0, s.by_member ? s.mdone : nullptr);
        if (rc != ZT_OK) return rc;
        if (s.by_member) {
            // nobody waits for this launch on the main stream; the slot's NEXT user does (make_group), and for that the plan
            // set's event will not do -- the launch after next records it again: an event of the slot's own (one packet per
            // launch group on this stream)
            ZT_HIP(hipEventRecord(s.ready, p->side));
            s.launched = true; s.waited = false;
            return ZT_OK;
        }
        if (done != nullptr) { s.ready_ev = done; s.launched = true; s.waited = false; return ZT_OK; }
    } else {
        // rows whose dictionary is empty are left untouched by the query (utils/util.py:185): the kernel writes them as zeros
        // (a memset in front of every query was a packet on this stream -- ~6 us of every C4 step, which this stream bounds)
        const int64_t n = row_hi - row_lo;
        int rc = zt::pruned_topk_multi_fill(d.csr, s.nodes_m + row_lo, s.ts + row_lo, n, d.width, d.depth, d.M, d.alpha, d.beta, d.k,
                                            s.on, s.oe, s.od, s.ow, d.status, p->side);      // every model in one walk
        if (rc != ZT_OK) return rc;
    }
    ZT_HIP(hipEventRecord(s.ready, p->side));
    s.launched = true;
    s.waited = false;
    s.q_lo = row_lo; s.q_hi = row_hi;
    return ZT_OK;
}

}  // namespace

extern "C" int zt_pipeline_create(zt_pipeline **out, const zt_pipeline_desc *desc, int32_t tppr_cus)
{
    if (!out || !desc || (desc->tppr == nullptr) == (desc->csr == nullptr) || !desc->memory || !desc->last_update ||
        !desc->messages || !desc->msg_ts || !desc->flags || !desc->scratch || !desc->efeat || !desc->embed_ws ||
        !desc->gru_ws || !desc->status || desc->max_B <= 0 || desc->M <= 0 || desc->M > 16 || desc->k <= 0) {
        set_error("zt_pipeline_create: bad argument (exactly one of tppr / csr, all tables and workspaces)");
        return ZT_ERR_ARG;
    }
    zt_pipeline *p = new zt_pipeline();
    memset(p, 0, sizeof(*p));
    p->d = *desc;
    p->group = 1;
    if (tppr_cus > 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        ZT_HIP(hipGetDevice(&dev));
        ZT_HIP(hipGetDeviceProperties(&prop, dev));
        void *a = nullptr, *b = nullptr;
        int rc = zt_stream_create_masked(&a, 0, tppr_cus);
        if (rc == ZT_OK) rc = zt_stream_create_masked(&b, tppr_cus, prop.multiProcessorCount);
        // the message kernels are small (a few dozen registers, no LDS): they share the T-PPR stream's compute units, where
        // they fit beside k_stream's workgroups -- the aggregation kernel fills the register files of its own
        void *c = nullptr;
        if (rc == ZT_OK) rc = zt_stream_create_masked(&c, 0, tppr_cus);
        if (rc != ZT_OK) {                     // (round-3 advisor: the streams already created leaked here)
            if (a) (void)zt_stream_destroy(a);
            if (b) (void)zt_stream_destroy(b);
            if (c) (void)zt_stream_destroy(c);
            delete p;
            return rc;
        }
        p->side = (hipStream_t)a; p->main_s = (hipStream_t)b; p->msg_s = (hipStream_t)c;
        p->masked = true;
    } else {
        ZT_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
        ZT_HIP(hipStreamCreateWithFlags(&p->main_s, hipStreamNonBlocking));
        ZT_HIP(hipStreamCreateWithFlags(&p->msg_s, hipStreamNonBlocking));
    }
    if (desc->tppr != nullptr) zt::tppr_hint_cus(desc->tppr, p->side);      // the first plans are made for THIS stream's CUs
    ZT_HIP(hipHostMalloc(reinterpret_cast<void **>(&p->latch_host), sizeof(int), hipHostMallocMapped));
    *p->latch_host = 0;
    ZT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->latch_dev), p->latch_host, 0));
    p->out_gru.latch = p->latch_dev;
    ZT_HIP(hipEventCreateWithFlags(&p->step_begin, hipEventDisableTiming | zt::sync_event_flags()));
    ZT_HIP(hipEventCreateWithFlags(&p->msgs_done, hipEventDisableTiming | zt::sync_event_flags()));
    ZT_HIP(hipEventCreateWithFlags(&p->scored[0], hipEventDisableTiming | zt::sync_event_flags()));
    ZT_HIP(hipEventCreateWithFlags(&p->scored[1], hipEventDisableTiming | zt::sync_event_flags()));
    ZT_HIP(hipStreamCreateWithFlags(&p->plan_s, hipStreamNonBlocking));
    ZT_HIP(hipEventCreateWithFlags(&p->entry, hipEventDisableTiming | zt::sync_event_flags()));
    // a slot holds one batch of max_B edges, or a group of smaller ones up to what one T-PPR launch covers
    p->cap = desc->max_B > TPPR_MAX_LAUNCH ? desc->max_B : TPPR_MAX_LAUNCH;
    const size_t rows = (size_t)3 * p->cap, per = (size_t)desc->M * rows * desc->k;
    for (auto &s : p->slot) {
        ZT_HIP(hipMalloc(&s.nodes, rows * sizeof(int32_t)));
        ZT_HIP(hipMalloc(&s.nodes_m, rows * sizeof(int32_t)));
        ZT_HIP(hipMalloc(&s.ts, rows * sizeof(double)));
        ZT_HIP(hipMalloc(&s.eidx, (size_t)p->cap * sizeof(int64_t)));
        ZT_HIP(hipMalloc(&s.buf, 4 * per * 4));
        ZT_HIP(hipMalloc(&s.mdone, zt::TPPR_MEMBER_WORDS * sizeof(int32_t)));
        ZT_HIP(hipMemset(s.mdone, 0, zt::TPPR_MEMBER_WORDS * sizeof(int32_t)));
        ZT_HIP(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming | zt::sync_event_flags()));
        ZT_HIP(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming | zt::sync_event_flags()));
        ZT_HIP(hipEventCreateWithFlags(&s.filled, hipEventDisableTiming | zt::sync_event_flags()));
    }
    const size_t per1 = (size_t)desc->M * 3 * desc->max_B * desc->k;
    ZT_HIP(hipMalloc(&p->sh_on, per1 * 4)); ZT_HIP(hipMalloc(&p->sh_oe, per1 * 4));
    ZT_HIP(hipMalloc(&p->sh_od, per1 * 4)); ZT_HIP(hipMalloc(&p->sh_ow, per1 * 4));
    *out = p;
    return ZT_OK;
}

extern "C" int zt_pipeline_destroy(zt_pipeline *p)
{
    if (!p) return ZT_OK;
    (void)hipDeviceSynchronize();
    for (auto &s : p->slot) {
        (void)hipFree(s.nodes); (void)hipFree(s.nodes_m); (void)hipFree(s.ts); (void)hipFree(s.eidx); (void)hipFree(s.buf); (void)hipFree(s.mdone);
        (void)hipEventDestroy(s.ready); (void)hipEventDestroy(s.consumed); (void)hipEventDestroy(s.filled);
    }
    (void)hipFree(p->sh_on); (void)hipFree(p->sh_oe); (void)hipFree(p->sh_od); (void)hipFree(p->sh_ow);
    (void)hipStreamDestroy(p->side); (void)hipStreamDestroy(p->main_s); (void)hipStreamDestroy(p->plan_s);
    (void)hipStreamDestroy(p->msg_s);
    (void)hipEventDestroy(p->scored[0]); (void)hipEventDestroy(p->scored[1]);
    (void)hipEventDestroy(p->entry); (void)hipEventDestroy(p->step_begin); (void)hipEventDestroy(p->msgs_done);
    if (p->latch_host) (void)hipHostFree(p->latch_host);
    delete p;
    return ZT_OK;
}

extern "C" void *zt_pipeline_main_stream(zt_pipeline *p) { return p ? (void *)p->main_s : nullptr; }

extern "C" int zt_pipeline_outstanding(const zt_pipeline *p)
{
    // batches whose T-PPR query has been launched ahead and that have not been consumed by a step yet
    if (!p) return 0;
    int n = 0;
    for (const auto &s : p->slot)
        if (s.n > 0 && s.launched) n += s.n - s.n_done;
    return n;
}

extern "C" int zt_pipeline_set_stats(zt_pipeline *p, float *avg_topk_dev)
{
    if (!p) return ZT_ERR_ARG;
    p->avg_topk = avg_topk_dev;
    return ZT_OK;
}

extern "C" int zt_pipeline_set_scoring(zt_pipeline *p, const zt_affinity_weights *weights, void *workspace_dev, float *prob_dev)
{
    if (!p) return ZT_ERR_ARG;
    if (weights == nullptr) { p->aff_on = false; return ZT_OK; }
    if (!workspace_dev || !prob_dev || !weights->fc1_w || !weights->fc1_b || !weights->fc2_w || !weights->fc2_b) {
        set_error("zt_pipeline_set_scoring: NULL buffer");
        return ZT_ERR_ARG;
    }
    if (zt_affinity_workspace_bytes(p->d.max_B, p->d.D * (p->d.M + 1)) < 0) {
        set_error("zt_pipeline_set_scoring: hidden width %d unsupported", p->d.D * (p->d.M + 1));
        return ZT_ERR_UNSUPPORTED;
    }
    p->aff = *weights; p->aff_ws = workspace_dev; p->prob = prob_dev;
    p->aff_on = true; p->aff_ready = false;              // (every call = new weights or a new workspace: packed again)
    return ZT_OK;
}

extern "C" int zt_pipeline_last_scores(zt_pipeline *p, void *stream, float **prob_out, int64_t *B_out)
{
    if (!p || !prob_out) return ZT_ERR_ARG;
    if (!p->aff_on || p->score_n == 0) { set_error("zt_pipeline_last_scores: no step has been scored"); return ZT_ERR_ARG; }
    const int par = (p->score_n - 1) & 1;
    ZT_HIP(hipStreamWaitEvent((hipStream_t)stream, p->scored[par], 0));
    *prob_out = p->prob + (size_t)par * 2 * p->d.max_B;
    if (B_out) *B_out = p->score_B;
    return ZT_OK;
}

extern "C" int zt_pipeline_set_exchange(zt_pipeline *p, zt_exchange *x)
{
    if (!p) return ZT_ERR_ARG;
    p->xchg = x;
    return ZT_OK;
}

extern "C" int zt_pipeline_set_group(zt_pipeline *p, int32_t group)
{
    if (!p || group < 1 || group > MAX_GROUP) { set_error("zt_pipeline_set_group: 1 <= group <= %d", MAX_GROUP); return ZT_ERR_ARG; }
    p->group = p->d.k > ZT_MAX_K ? 1 : group;        // (dictionaries wider than a wavefront: one batch per T-PPR launch, tppr_wide.hpp)
    return ZT_OK;
}

extern "C" int zt_pipeline_update(zt_pipeline *p, const zt_pipeline_desc *desc, int32_t weights_changed)
{
    if (!p) return ZT_ERR_ARG;
    if (desc) {
        if ((desc->tppr == nullptr) == (desc->csr == nullptr) || desc->max_B > p->d.max_B || desc->M != p->d.M || desc->k != p->d.k) {
            set_error("zt_pipeline_update: the strategy's handle is missing or the shape changed");
            return ZT_ERR_ARG;
        }
        p->d = *desc;
        for (auto &s : p->slot) if (!s.launched) { s.n = 0; }      // plans against the old state are void
    }
    if (weights_changed) { p->embed_ready = false; p->gru_ready = false; }
    return ZT_OK;
}

extern "C" int zt_pipeline_step_ahead(zt_pipeline *p, const zt_batch *cur, const zt_batch *ahead, int32_t n_ahead,
                                      int64_t row_lo, int64_t row_hi, int64_t pos_lo, int64_t pos_hi, float *out_emb_dev)
{
    if (!p || !valid_batch(p, cur) || n_ahead < 0 || (n_ahead > 0 && !ahead) ||
        !out_emb_dev || row_lo < 0 || row_hi > 3 * cur->B || row_lo > row_hi || pos_lo < 0 || pos_hi > 2 * cur->B || pos_lo > pos_hi) {
        set_error("zt_pipeline_step: bad argument");
        return ZT_ERR_ARG;
    }
    if (const int st = *reinterpret_cast<volatile int *>(p->latch_host)) {
        // a kernel of an earlier step gave up a bounded wait: its results are void.  Reported once, then cleared.
        *reinterpret_cast<volatile int *>(p->latch_host) = 0;
        set_error("zt_pipeline_step: a kernel of an earlier step gave up a bounded wait (status %d: the gate between the output "
                  "layers and the GRU update, memory_update.hip); that step's memory update is incomplete", st);
        return st;
    }
    const zt_pipeline_desc &d = p->d;
    const int64_t B = cur->B, n_rows = row_hi - row_lo;
    const bool pruning = d.csr != nullptr;
    const bool whole = row_lo == 0 && row_hi == 3 * B;
    const int want = pruning ? 1 : p->group;
    // the shard of a LATER batch is the same fraction of its rows (callers shard every batch alike)
    auto shard_of = [&](int64_t Bn, int64_t *lo, int64_t *hi) {
        *lo = Bn == B ? row_lo : (row_lo * 3 * Bn) / (3 * B);
        *hi = Bn == B ? row_hi : (row_hi * 3 * Bn) / (3 * B);
    };
    int rc, j = 0;
    // ---- this batch's T-PPR query: made ahead by an earlier step, or now (with as many followers as allowed) ----
    zt_pipeline::Slot *s = find_slot(p, cur, &j);
    // `entry` = everything the caller enqueued before this call: what a group staged on another stream must wait for
    // (make_group).  Recorded up front -- BEFORE the main stream is told to wait for the T-PPR update -- by the calls
    // that stage groups as a rule: a batch nobody has seen, or the first member of its group (the group after next is
    // planned then).  The other members' steps skip it, and the wait for an update the main stream has already been
    // told to wait for: two packets less between the GRU of one batch and the aggregation of the next (~5 us of
    // command-processor time each; round 4).  A call that stages a group after all records it late (make_group).
    // (round 5: every step records it -- the message kernels wait for it too, where they used to wait for an event of their
    //  own recorded BEHIND the wait for the T-PPR update: one packet less per step on the main stream, and the message build
    //  no longer waits for an update it does not read)
    ZT_HIP(hipEventRecord(p->entry, p->main_s));
    p->entry_recorded = true;
    // (a batch nobody has queried ahead is queried ALONE: the aggregation waits for this launch, and a group would
    //  make it wait for the followers' updates as well; they form the next group, which runs beside this batch)
    if (s == nullptr) { rc = make_group(p, cur, ahead, n_ahead, 1, p->side, &s, true); if (rc != ZT_OK) return rc; j = 0; }
    // (pruning: a query made ahead covers the rows shard_of() derived for it; if this call asks for other rows --
    //  the roundings differ when 3B is not divisible by the world size and the batch sizes change -- query again:
    //  the strategy carries no state)
    if (pruning && s->launched && (s->q_lo != row_lo || s->q_hi != row_hi)) s->launched = false;
    if (!s->launched) { rc = launch_tppr(p, *s, row_lo, row_hi); if (rc != ZT_OK) return rc; }
    // this member's rows, not the launch: one count per (edge, model) task (k_stream's general queue).  Small batches, where
    // the aggregation is the first reader of the rows on the main stream: the wait rides inside the aggregation kernel
    // (embed_ex) -- their steps are bound by the main stream, the gate is open when the kernel arrives, and a kernel of its own
    // in front would be ~13 us of every step (C2 0.083 -> 0.080 ms/step).  Large batches: a one-wave kernel here (bounded wait,
    // ZT_ERR_TIMEOUT to the status word and the pipeline's latch: aggregate.hip, member_gate_launch).  Measured on
    // C5 (0.296 against 0.305 ms/step at 200 steps): while the main stream waits for the T-PPR update, a persistent aggregation
    // kernel that is already resident fills all of its compute units' registers, and the prepass kernels of the next launch
    // group -- on a stream without a CU mask -- are left with the T-PPR stream's compute units, beside the hub chains.
    zt::member_gate gate = {nullptr, 0, p->latch_dev};
    if (s->by_member) {
        // The counters are zeroed by the group's staging kernel, on another stream: the gate must not look at them before that
        // kernel has run -- what is left there from the slot's previous group IS a full count.  In practice the staging is two
        // launch groups ahead of this step; by construction it is this wait, once per group (`filled` is recorded behind the
        // staging and the plan; the T-PPR launch that counts waits for it too).
        if (!s->waited) {
            if (hipEventQuery(s->filled) != hipSuccess) { (void)hipGetLastError(); ZT_HIP(hipStreamWaitEvent(p->main_s, s->filled, 0)); }
            s->waited = true;
        }
        gate.word = s->mdone + j; gate.target = (int32_t)(s->B[j] * d.M);
        // (inside the kernel only where the T-PPR stream has compute units of its own: a persistent aggregation kernel that
        //  waits on EVERY compute unit of an unmasked device, registers full, would keep out the very launch it waits for)
        if (!(p->masked && whole && n_rows > 0 && n_rows <= 2048 && p->avg_topk == nullptr)) {
            rc = zt::member_gate_launch(gate, d.status, p->main_s);
            if (rc != ZT_OK) return rc;
            gate.word = nullptr;
        }
    } else if (!s->waited) { ZT_HIP(hipStreamWaitEvent(p->main_s, s->ready_ev, 0)); s->waited = true; }
    // ---- the group after this one is queried beside this group's aggregation; the one after that is planned ----
    int a = s->n - 1 - j;                     // ahead[0 .. a) are the rest of this group
    if (a < n_ahead && valid_batch(p, ahead + a)) {
        int jn = 0;
        zt_pipeline::Slot *n = find_slot(p, ahead + a, &jn);
        if (n == nullptr) {
            // not planned by an earlier step (the start of a stream): its prepass goes to the plan stream all the same,
            // beside the current batch's update, not behind it on the T-PPR stream
            rc = make_group(p, ahead + a, ahead + a + 1, n_ahead - a - 1, want, p->plan_s, &n, pruning);
            if (rc != ZT_OK) return rc;
            if (!pruning) {
                rc = zt_tppr_plan(d.tppr, n->nodes, n->eidx, n->Btot, 3, -1, &n->token, p->plan_s);
                if (rc != ZT_OK) { (void)hipEventRecord(n->filled, p->plan_s); return rc; }
                ZT_HIP(hipEventRecord(n->filled, p->plan_s));
            }
        }
        if (!n->launched) {
            int64_t nlo, nhi;
            shard_of(n->B[0], &nlo, &nhi);
            rc = launch_tppr(p, *n, nlo, nhi);
            if (rc != ZT_OK) return rc;
        }
        a += n->n - jn;                       // ahead[a] now follows that group
        if (a < n_ahead && valid_batch(p, ahead + a) && find_slot(p, ahead + a, &jn) == nullptr) {
            zt_pipeline::Slot *q;
            rc = make_group(p, ahead + a, ahead + a + 1, n_ahead - a - 1, want, p->plan_s, &q, pruning);   // streaming: `filled`
            if (rc != ZT_OK) return rc;                                                              // follows the prepass
            if (!pruning) {
                rc = zt_tppr_plan(d.tppr, q->nodes, q->eidx, q->Btot, 3, -1, &q->token, p->plan_s);
                if (rc != ZT_OK) { (void)hipEventRecord(q->filled, p->plan_s); return rc; }
                ZT_HIP(hipEventRecord(q->filled, p->plan_s));
            }
        }
    }
    // ---- P2: gather + aggregate for rows [row_lo, row_hi) of member j ----
    size_t mo = 0;                            // first element of member j's block in each output array
    int64_t off = 0;
    for (int q = 0; q < j; ++q) { mo += (size_t)d.M * 3 * s->B[q] * d.k; off += s->B[q]; }
    const int32_t *nodes_cur = s->nodes_m + 3 * off;
    const int32_t *on = s->on + mo, *oe = s->oe + mo;
    const float *od = s->od + mo, *ow = s->ow + mo;
    if (!pruning && !whole) {
        // streaming T-PPR emits all 3B rows of every model: bring this shard's rows together
        const size_t w = (size_t)n_rows * d.k * 4, pitch = (size_t)3 * B * d.k * 4, o2 = (size_t)row_lo * d.k;
        ZT_HIP(hipMemcpy2DAsync(p->sh_on, w, on + o2, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        ZT_HIP(hipMemcpy2DAsync(p->sh_oe, w, oe + o2, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        ZT_HIP(hipMemcpy2DAsync(p->sh_od, w, od + o2, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        ZT_HIP(hipMemcpy2DAsync(p->sh_ow, w, ow + o2, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        on = p->sh_on; oe = p->sh_oe; od = p->sh_od; ow = p->sh_ow;
    }
    if (p->avg_topk != nullptr && whole) {
        k_avg_topk<<<1, 256, 0, p->main_s>>>(ow, 2 * B, d.k, p->avg_topk);
        ZT_LAUNCH_CHECK();
    }
    // ---- P3 (first half), on a stream of its own: the last messages of the endpoints at positions [pos_lo, pos_hi).  They
    // read the memory tables as everything enqueued on the main stream so far leaves them (the previous batch's GRU, a
    // caller's row exchange) and nothing the aggregation writes; the GRU below waits for them.
    ZT_HIP(hipStreamWaitEvent(p->msg_s, p->entry, 0));
    // Every endpoint of the batch gets a message here and is updated by the GRU below: in this protocol (eval steps, entered
    // with no flag pending -- step_device flushes first) "flagged among the batch's endpoints after the store" IS the list of
    // winners the message kernel already makes.  So the kernel hands its list to the GRU directly (the GRU workspace's row
    // list and counter; k_last_pos zeroes the counter) and sets no flag: Memory.clear_messages for the same ids would clear
    // it again -- and the compaction kernel (k_select_flagged) leaves the step.
    bool cnt_zeroed = false;
    const int msg_dim = 2 * d.D + d.F + d.T;
    char *gws = reinterpret_cast<char *>(d.gru_ws);
    int32_t *gru_cnt = reinterpret_cast<int32_t *>(gws);
    int32_t *gru_rows = reinterpret_cast<int32_t *>(gws + zt_gru_rows_offset(d.D, msg_dim));
    rc = zt::store_messages_ex(d.memory, d.last_update, d.efeat, d.ew.time_w, d.num_nodes, d.num_edges, d.D, d.F, d.T,
                               cur->src, cur->dst, cur->ts, cur->eidx, B, pos_lo, pos_hi, d.messages, d.msg_ts, d.flags,
                               d.scratch, gru_rows, gru_cnt, d.status, gru_cnt, p->msg_s, &cnt_zeroed, false);
    if (rc != ZT_OK) return rc;
    if (!cnt_zeroed) ZT_HIP(hipMemsetAsync(gru_cnt, 0, sizeof(int), p->msg_s));       // (B == 0 cannot get here; belt and braces)
    ZT_HIP(hipEventRecord(p->msgs_done, p->msg_s));
    const float *wm_p = (d.proj_table != nullptr && p->embed_ready) ? zt::embed_wm_ptr(d.embed_ws, 3 * d.max_B, d.D, d.F, d.T, d.M, d.k)
                                                                     : nullptr;
    bool msgs_waited = false;
    if (n_rows > 0) {
        // (the wait for the message build sits between the aggregation and the output layer -- the messages are ready long
        //  before the aggregation ends -- so that the GRU follows the output layer without a packet in between)
        msgs_waited = true;
        // (the output layers are held back: gru_update_ex below launches them in ONE kernel with the GRU update where both take
        //  their tiled forms -- k_out_gru --, otherwise in front of it)
        rc = zt::embed_ex(d.memory, d.efeat, d.num_nodes, d.num_edges, d.D, d.F, d.T, nodes_cur + row_lo, n_rows, d.M, d.k, on, oe,
                          od, ow, &d.ew, out_emb_dev, d.embed_ws, d.status, d.proj_table, p->embed_ready ? 1 : 0, p->main_s,
                          msgs_waited ? p->msgs_done : nullptr, &p->out_gru, gate.word ? &gate : nullptr);
        if (rc != ZT_OK) return rc;
        p->embed_ready = true;
    }
    // ---- P3: the GRU update over the messages built beside the aggregation; the refresh of the projected rows rides
    // inside the GRU kernel once the padded W_m is in the embed workspace ----
    if (!msgs_waited) ZT_HIP(hipStreamWaitEvent(p->main_s, p->msgs_done, 0));
    rc = zt::gru_update_ex(d.memory, d.last_update, d.messages, d.msg_ts, d.flags, d.num_nodes, d.D, msg_dim, nodes_cur, 2 * B,
                           nullptr, &d.gw, d.gru_ws, p->gru_ready ? 1 : 0, wm_p, wm_p ? d.proj_table : nullptr, p->main_s, true, true,
                           n_rows > 0 ? &p->out_gru : nullptr);
    if (rc != ZT_OK) return rc;
    p->gru_ready = true;
    if (n_rows > 0 && p->aff_on && whole) {      // compute_edge_probabilities' scorer (model/tgn_model.py:185-188) on the rows just written
        const int par = p->score_n & 1;
        rc = zt_affinity(out_emb_dev, B, d.D * (d.M + 1), &p->aff, p->prob + (size_t)par * 2 * d.max_B, p->aff_ws, d.max_B,
                         p->aff_ready ? 1 : 0, p->main_s);
        if (rc != ZT_OK) return rc;
        ZT_HIP(hipEventRecord(p->scored[par], p->main_s));
        p->aff_ready = true;
        p->score_n++;
        p->score_B = B;
    }
    if (d.proj_table != nullptr && wm_p == nullptr) {
        char *gw = reinterpret_cast<char *>(d.gru_ws);
        rc = zt_project_memory(d.memory, d.num_nodes, d.D, d.F, d.T, &d.ew, 1,
                               reinterpret_cast<const int32_t *>(gw + zt_gru_rows_offset(d.D, msg_dim)),
                               reinterpret_cast<const int32_t *>(gw), 2 * B, d.proj_table, d.embed_ws, 3 * d.max_B, d.M, d.k,
                               p->main_s);
        if (rc != ZT_OK) return rc;
    }
    // ---- multi-GPU: the rows every rank's GRU update rewrote, all-gathered and scattered into the local tables; the
    // projected table follows the rows the other ranks wrote (this rank's own were refreshed by its GRU kernel) ----
    if (p->xchg != nullptr) {
        const int32_t *xids = nullptr;
        int64_t n_x = 0;
        rc = zt::exchange_step(p->xchg, gru_rows, gru_cnt, p->main_s, &xids, &n_x);
        if (rc != ZT_OK) return rc;
        if (d.proj_table != nullptr) {
            rc = zt_project_memory(d.memory, d.num_nodes, d.D, d.F, d.T, &d.ew, p->embed_ready ? 1 : 0, xids, nullptr, n_x, d.proj_table,
                                   d.embed_ws, 3 * d.max_B, d.M, d.k, p->main_s);
            if (rc != ZT_OK) return rc;
            p->embed_ready = true;                 // (a rank whose row shard was empty: the padded weights were made just now)
        }
    }
    // the member is used up; with the last one the slot is free again
    s->key[j] = nullptr;
    if (++s->n_done >= s->n) {
        ZT_HIP(hipEventRecord(s->consumed, p->main_s));
        s->n = 0;
    }
    return ZT_OK;
}

// n consecutive whole-batch steps from ONE host call: the per-batch loop of evaluation/evaluation.py:19-45 (and of a bulk
// replay) without a trip through the caller's language per batch.  Step b sees the batches b + 1 .. b + look as `ahead`
// (never beyond the n given: nothing of a later batch is enqueued).  out_emb_dev + b * out_stride receives step b's
// [3 B_b][D (M + 1)] embeddings (out_stride = 0: every step overwrites the same buffer -- a caller that only wants the
// scores of zt_pipeline_set_scoring or the final state).
extern "C" int zt_pipeline_run(zt_pipeline *p, const zt_batch *batches, int32_t n, int32_t look, float *out_emb_dev,
                               int64_t out_stride)
{
    if (!p || !batches || n < 0 || look < 0 || !out_emb_dev || out_stride < 0) { set_error("zt_pipeline_run: bad argument"); return ZT_ERR_ARG; }
    int rank = 0, world = 1;
    if (p->xchg != nullptr) zt::exchange_shape(p->xchg, &rank, &world);
    for (int32_t b = 0; b < n; ++b) {
        const int32_t na = (b + 1 + look <= n) ? look : (n - b - 1);
        // a multi-GPU run: this rank's contiguous shard of the batch's 3B rows and of its 2B positions (zebra_amd/distributed.py:
        // shard_range); one GPU: everything
        const int64_t B = batches[b].B;
        const int64_t r0 = (3 * B * rank) / world, r1 = (3 * B * (rank + 1)) / world;
        const int64_t p0 = (2 * B * rank) / world, p1 = (2 * B * (rank + 1)) / world;
        const int rc = zt_pipeline_step_ahead(p, &batches[b], na > 0 ? &batches[b + 1] : nullptr, na, r0, r1, p0, p1,
                                              out_emb_dev + (size_t)b * (size_t)out_stride);
        if (rc != ZT_OK) return rc;
    }
    return ZT_OK;
}

extern "C" int zt_pipeline_step(zt_pipeline *p, const zt_batch *cur, const zt_batch *next, const zt_batch *plan,
                                int64_t row_lo, int64_t row_hi, int64_t pos_lo, int64_t pos_hi, float *out_emb_dev)
{
    zt_batch ahead[2];
    int n = 0;
    if (next != nullptr) { ahead[n++] = *next; if (plan != nullptr) ahead[n++] = *plan; }
    return zt_pipeline_step_ahead(p, cur, ahead, n, row_lo, row_hi, pos_lo, pos_hi, out_emb_dev);
}
