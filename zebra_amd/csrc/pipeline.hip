// One eval-mode batch of the hot path as ONE host call (zt_pipeline_step): T-PPR query / update, gather +
// aggregate, last-message store, GRU memory update and the refresh of the projected table are enqueued from
// C++ on three HIP streams, with no Python between the launches.
//
// Mirrors TGN.compute_temporal_embeddings with train=False (reference model/tgn_model.py:124-174) for
// device-resident batches; the reference issues the same steps from Python on one stream
// (train.py:145-146).  The streams:
//   side  the T-PPR query of batch b+1 (streaming: k_stream, CU-masked to a few compute units because it is
//         latency-bound; pruning: k_pruned_topk) -- the T-PPR state depends only on the edge stream, never
//         on the node memory, so it runs one batch ahead;
//   plan  the dependency prepass of batch b+2 (reads only ids);
//   main  aggregate + messages + GRU of batch b (the remaining compute units).
// Three batch slots (ids + the four T-PPR output arrays each) rotate through plan -> query -> consume.
#include "common.hpp"

#include <cstring>

using namespace zt;

struct zt_pipeline {
    zt_pipeline_desc d;
    hipStream_t side, main_s, plan_s;
    bool own_streams;
    struct Slot {
        int32_t *nodes;        // [3 * max_B]  src | dst | neg
        double *ts3;           // [3 * max_B]  (pruning: the query time of every row)
        int32_t *buf;          // 4 x [M][3 * max_B][k]
        int32_t *on, *oe;      // the current batch's four [M][3B][k] arrays inside buf
        float *od, *ow;
        hipEvent_t ready;      // T-PPR outputs complete (side stream)
        hipEvent_t ready_ev;   // the event to wait on for that: `ready`, or the one the T-PPR update recorded itself
        hipEvent_t consumed;   // main stream is done with the slot
        hipEvent_t filled;     // ids copied in (plan / side stream)
        const int64_t *key;    // the batch it holds (eidx pointer), nullptr = free
        int64_t B;
        uint64_t token;        // zt_tppr_plan token, 0 = none
        bool launched, used, filled_on_plan;
    } slot[3];
    int next_slot;
    // compact copies of a row shard's T-PPR outputs (sharded streaming runs only)
    int32_t *sh_on, *sh_oe;
    float *sh_od, *sh_ow;
    bool embed_ready, gru_ready;
};

namespace {

zt_pipeline::Slot *find_slot(zt_pipeline *p, const zt_batch *b)
{
    for (auto &s : p->slot)
        if (s.key != nullptr && s.key == b->eidx && s.B == b->B) return &s;
    return nullptr;
}

// [src | dst | neg] of a batch into a slot (and, for the pruning strategy, the query time of every row)
__global__ void k_stage_batch(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, const int32_t *__restrict__ neg,
                              const double *__restrict__ ts, long long B, int32_t *__restrict__ nodes, double *__restrict__ ts3)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    nodes[i] = src[i]; nodes[B + i] = dst[i]; nodes[2 * B + i] = neg[i];
    if (ts3 != nullptr) { const double t = ts[i]; ts3[i] = t; ts3[B + i] = t; ts3[2 * B + i] = t; }
}

// the slot that batch `b` will live in: ids are copied on stream `st` once the slot's previous user is done.
// record = false: the caller records `filled` itself (after the prepass it enqueues behind the copy).
int fill_slot(zt_pipeline *p, const zt_batch *b, hipStream_t st, zt_pipeline::Slot **out, bool record = true)
{
    zt_pipeline::Slot &s = p->slot[p->next_slot];
    p->next_slot = (p->next_slot + 1) % 3;
    if (s.used) ZT_HIP(hipStreamWaitEvent(st, s.consumed, 0));
    k_stage_batch<<<(unsigned)((b->B + 255) / 256), 256, 0, st>>>(b->src, b->dst, b->neg, b->ts, b->B, s.nodes,
                                                                  p->d.csr != nullptr ? s.ts3 : nullptr);
    ZT_LAUNCH_CHECK();
    if (record) ZT_HIP(hipEventRecord(s.filled, st));
    // the four output arrays [M][3B][k] of this batch, back to back in the slot's buffer
    const size_t per = (size_t)p->d.M * 3 * b->B * p->d.k;
    s.on = s.buf; s.oe = s.buf + per;
    s.od = reinterpret_cast<float *>(s.buf + 2 * per); s.ow = reinterpret_cast<float *>(s.buf + 3 * per);
    s.key = b->eidx; s.B = b->B; s.token = 0; s.launched = false; s.used = true;
    *out = &s;
    return ZT_OK;
}

// the T-PPR query of the slot's batch on the side stream; rows [row_lo, row_hi) only for the pruning strategy
int launch_tppr(zt_pipeline *p, zt_pipeline::Slot &s, const zt_batch *b, int64_t row_lo, int64_t row_hi)
{
    const zt_pipeline_desc &d = p->d;
    ZT_HIP(hipStreamWaitEvent(p->side, s.filled, 0));
    s.ready_ev = s.ready;
    if (d.tppr != nullptr) {
        // `filled` was recorded behind the prepass (when there is one): no second wait; the event the update kernel
        // records for the plan set doubles as this slot's `ready`
        hipEvent_t done = nullptr;
        int rc = zt::tppr_stream_ex(d.tppr, s.nodes, b->ts, b->eidx, b->B, 3, 1, -1, s.on, s.oe, s.od, s.ow, s.token, p->side,
                                    true, &done);
        if (rc != ZT_OK) return rc;
        if (done != nullptr) { s.ready_ev = done; s.launched = true; return ZT_OK; }
    } else {
        // rows whose dictionary is empty are left untouched by the query (utils/util.py:185): start from zeros
        const int64_t n = row_hi - row_lo;
        const size_t per = (size_t)d.M * n * d.k;
        if (n == 3 * b->B) {
            ZT_HIP(hipMemsetAsync(s.on, 0, 4 * per * 4, p->side));          // the four arrays follow each other (fill_slot)
        } else {
            ZT_HIP(hipMemsetAsync(s.on, 0, per * 4, p->side));
            ZT_HIP(hipMemsetAsync(s.oe, 0, per * 4, p->side));
            ZT_HIP(hipMemsetAsync(s.od, 0, per * 4, p->side));
            ZT_HIP(hipMemsetAsync(s.ow, 0, per * 4, p->side));
        }
        for (int m = 0; m < d.M; ++m) {
            const size_t o = (size_t)m * n * d.k;
            int rc = zt_pruned_topk(d.csr, s.nodes + row_lo, s.ts3 + row_lo, n, d.width, d.depth, d.alpha[m], d.beta[m], d.k,
                                    s.on + o, s.oe + o, s.od + o, s.ow + o, d.status, p->side);
            if (rc != ZT_OK) return rc;
        }
    }
    ZT_HIP(hipEventRecord(s.ready, p->side));
    s.launched = true;
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
    if (tppr_cus > 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        ZT_HIP(hipGetDevice(&dev));
        ZT_HIP(hipGetDeviceProperties(&prop, dev));
        void *a = nullptr, *b = nullptr;
        int rc = zt_stream_create_masked(&a, 0, tppr_cus);
        if (rc == ZT_OK) rc = zt_stream_create_masked(&b, tppr_cus, prop.multiProcessorCount);
        if (rc != ZT_OK) { delete p; return rc; }
        p->side = (hipStream_t)a; p->main_s = (hipStream_t)b;
    } else {
        ZT_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
        ZT_HIP(hipStreamCreateWithFlags(&p->main_s, hipStreamNonBlocking));
    }
    ZT_HIP(hipStreamCreateWithFlags(&p->plan_s, hipStreamNonBlocking));
    p->own_streams = true;
    const size_t rows = (size_t)3 * desc->max_B, per = (size_t)desc->M * rows * desc->k;
    for (auto &s : p->slot) {
        ZT_HIP(hipMalloc(&s.nodes, rows * sizeof(int32_t)));
        ZT_HIP(hipMalloc(&s.ts3, rows * sizeof(double)));
        ZT_HIP(hipMalloc(&s.buf, 4 * per * 4));
        ZT_HIP(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming | zt::sync_event_flags()));
        ZT_HIP(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming | zt::sync_event_flags()));
        ZT_HIP(hipEventCreateWithFlags(&s.filled, hipEventDisableTiming | zt::sync_event_flags()));
    }
    ZT_HIP(hipMalloc(&p->sh_on, per * 4)); ZT_HIP(hipMalloc(&p->sh_oe, per * 4));
    ZT_HIP(hipMalloc(&p->sh_od, per * 4)); ZT_HIP(hipMalloc(&p->sh_ow, per * 4));
    *out = p;
    return ZT_OK;
}

extern "C" int zt_pipeline_destroy(zt_pipeline *p)
{
    if (!p) return ZT_OK;
    (void)hipDeviceSynchronize();
    for (auto &s : p->slot) {
        (void)hipFree(s.nodes); (void)hipFree(s.ts3); (void)hipFree(s.buf);
        (void)hipEventDestroy(s.ready); (void)hipEventDestroy(s.consumed); (void)hipEventDestroy(s.filled);
    }
    (void)hipFree(p->sh_on); (void)hipFree(p->sh_oe); (void)hipFree(p->sh_od); (void)hipFree(p->sh_ow);
    (void)hipStreamDestroy(p->side); (void)hipStreamDestroy(p->main_s); (void)hipStreamDestroy(p->plan_s);
    delete p;
    return ZT_OK;
}

extern "C" void *zt_pipeline_main_stream(zt_pipeline *p) { return p ? (void *)p->main_s : nullptr; }

extern "C" int zt_pipeline_update(zt_pipeline *p, const zt_pipeline_desc *desc, int32_t weights_changed)
{
    if (!p) return ZT_ERR_ARG;
    if (desc) {
        if ((desc->tppr == nullptr) == (desc->csr == nullptr) || desc->max_B > p->d.max_B || desc->M != p->d.M || desc->k != p->d.k) {
            set_error("zt_pipeline_update: the strategy's handle is missing or the shape changed");
            return ZT_ERR_ARG;
        }
        p->d = *desc;
        for (auto &s : p->slot) if (!s.launched) { s.key = nullptr; }      // plans against the old state are void
    }
    if (weights_changed) { p->embed_ready = false; p->gru_ready = false; }
    return ZT_OK;
}

extern "C" int zt_pipeline_step(zt_pipeline *p, const zt_batch *cur, const zt_batch *next, const zt_batch *plan,
                                int64_t row_lo, int64_t row_hi, int64_t pos_lo, int64_t pos_hi, float *out_emb_dev)
{
    if (!p || !cur || !cur->src || !cur->dst || !cur->neg || !cur->ts || !cur->eidx || cur->B <= 0 || cur->B > p->d.max_B ||
        !out_emb_dev || row_lo < 0 || row_hi > 3 * cur->B || row_lo > row_hi || pos_lo < 0 || pos_hi > 2 * cur->B || pos_lo > pos_hi) {
        set_error("zt_pipeline_step: bad argument");
        return ZT_ERR_ARG;
    }
    const zt_pipeline_desc &d = p->d;
    const int64_t B = cur->B, n_rows = row_hi - row_lo;
    const bool pruning = d.csr != nullptr;
    const bool whole = row_lo == 0 && row_hi == 3 * B;
    int rc;
    // ---- this batch's T-PPR query: made ahead by the previous step, or now ----
    zt_pipeline::Slot *s = find_slot(p, cur);
    if (s == nullptr) { rc = fill_slot(p, cur, p->side, &s); if (rc != ZT_OK) return rc; }
    if (!s->launched) { rc = launch_tppr(p, *s, cur, row_lo, row_hi); if (rc != ZT_OK) return rc; }
    ZT_HIP(hipStreamWaitEvent(p->main_s, s->ready_ev, 0));
    // ---- the next batch's query beside this batch's aggregation; the prepass of the one after it ----
    if (next != nullptr && next->B > 0 && next->B <= d.max_B) {
        zt_pipeline::Slot *n = find_slot(p, next);
        if (n == nullptr) { rc = fill_slot(p, next, p->side, &n); if (rc != ZT_OK) return rc; }
        if (!n->launched) {
            // the shard of the NEXT step is the same fraction of its rows (callers shard every batch alike)
            const int64_t nlo = next->B == B ? row_lo : (row_lo * 3 * next->B) / (3 * B);
            const int64_t nhi = next->B == B ? row_hi : (row_hi * 3 * next->B) / (3 * B);
            rc = launch_tppr(p, *n, next, nlo, nhi);
            if (rc != ZT_OK) return rc;
        }
    }
    if (plan != nullptr && plan->B > 0 && plan->B <= d.max_B && find_slot(p, plan) == nullptr) {
        zt_pipeline::Slot *q;
        rc = fill_slot(p, plan, p->plan_s, &q, pruning);          // streaming: `filled` is recorded behind the prepass
        if (rc != ZT_OK) return rc;
        if (!pruning) {
            rc = zt_tppr_plan(d.tppr, q->nodes, plan->eidx, plan->B, 3, -1, &q->token, p->plan_s);
            if (rc != ZT_OK) { (void)hipEventRecord(q->filled, p->plan_s); return rc; }
            ZT_HIP(hipEventRecord(q->filled, p->plan_s));
        }
    }
    // ---- P2: gather + aggregate for rows [row_lo, row_hi) ----
    const int32_t *on = s->on, *oe = s->oe;
    const float *od = s->od, *ow = s->ow;
    if (!pruning && !whole) {
        // streaming T-PPR emits all 3B rows of every model: bring this shard's rows together
        const size_t w = (size_t)n_rows * d.k * 4, pitch = (size_t)3 * B * d.k * 4, off = (size_t)row_lo * d.k;
        ZT_HIP(hipMemcpy2DAsync(p->sh_on, w, s->on + off, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        ZT_HIP(hipMemcpy2DAsync(p->sh_oe, w, s->oe + off, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        ZT_HIP(hipMemcpy2DAsync(p->sh_od, w, s->od + off, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        ZT_HIP(hipMemcpy2DAsync(p->sh_ow, w, s->ow + off, pitch, w, d.M, hipMemcpyDeviceToDevice, p->main_s));
        on = p->sh_on; oe = p->sh_oe; od = p->sh_od; ow = p->sh_ow;
    }
    if (n_rows > 0) {
        rc = zt_embed(d.memory, d.efeat, d.num_nodes, d.num_edges, d.D, d.F, d.T, s->nodes + row_lo, n_rows, d.M, d.k, on, oe,
                      od, ow, &d.ew, out_emb_dev, d.embed_ws, d.status, d.proj_table, p->embed_ready ? 1 : 0, p->main_s);
        if (rc != ZT_OK) return rc;
        p->embed_ready = true;
    }
    // ---- P3: last messages of the endpoints at positions [pos_lo, pos_hi), GRU update, projected rows ----
    rc = zt_store_messages_range(d.memory, d.last_update, d.efeat, d.ew.time_w, d.num_nodes, d.num_edges, d.D, d.F, d.T,
                                 cur->src, cur->dst, cur->ts, cur->eidx, B, pos_lo, pos_hi, d.messages, d.msg_ts, d.flags,
                                 d.scratch, nullptr, nullptr, d.status, p->main_s);
    if (rc != ZT_OK) return rc;
    const int msg_dim = 2 * d.D + d.F + d.T;
    rc = zt_gru_update(d.memory, d.last_update, d.messages, d.msg_ts, d.flags, d.num_nodes, d.D, msg_dim, s->nodes, 2 * B,
                       nullptr, &d.gw, d.gru_ws, p->gru_ready ? 1 : 0, p->main_s);
    if (rc != ZT_OK) return rc;
    p->gru_ready = true;
    if (d.proj_table != nullptr) {
        char *gw = reinterpret_cast<char *>(d.gru_ws);
        rc = zt_project_memory(d.memory, d.num_nodes, d.D, d.F, d.T, &d.ew, 1, reinterpret_cast<const int32_t *>(gw + 256),
                               reinterpret_cast<const int32_t *>(gw), 2 * B, d.proj_table, d.embed_ws, 3 * d.max_B, d.M, d.k,
                               p->main_s);
        if (rc != ZT_OK) return rc;
    }
    ZT_HIP(hipEventRecord(s->consumed, p->main_s));
    s->key = nullptr;
    return ZT_OK;
}
