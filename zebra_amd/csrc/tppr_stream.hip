// Streaming top-k T-PPR maintenance on MI355X (gfx950).
//
// Replaces tppr_finder.streaming_topk & friends (reference utils/util.py:391-873).
// Compile with -ffp-contract=off: the reference rounds every float64 multiply
// and add separately (utils/util.py:523-541) and results must be bit-exact.
//
// Design (DESIGN.md "P1"):
//   * state is device resident, SoA [model][node][k]: key = (edge_idx << 32 |
//     node) u64, timestamp f64, weight f64, plus len i32 and norm f64 per row;
//   * the edges of a batch must be applied in order, each reading the rows the
//     previous edges wrote.  A 4-kernel prepass groups the batch's 3B node
//     accesses by node (atomic count / reserve / fill) and gives every access
//     the index of the latest EARLIER edge touching the same node;
//   * the update kernel is persistent: one wavefront per (edge, model) task,
//     tasks dequeued in order from an atomic head (so a task only ever waits
//     on tasks that are already resident -> no deadlock), each wave spins on
//     the done-flags of its <= 3 predecessor edges, merges in LDS, prunes with
//     the exact numba argsort semantics and publishes its rows write-through
//     (sc1) followed by its own done-flag (Guideline 16 recipe R1).
#include "numba_sort.hpp"

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace zt {

// ---- error plumbing ----------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- kernel timing with HIP events ----------------------------------------------
bool g_prof_on = false;
namespace {
struct ProfRec { hipEvent_t a, b; int id; bool open; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
double g_prof_ms[P_COUNT];
long long g_prof_n[P_COUNT];
const char *const g_prof_names[P_COUNT] = {"tppr_prepass", "tppr_stream", "tppr_cleanup", "pruned_topk",
                                           "embed_prep", "fc1_agg", "embed_out", "store_messages", "gru_update"};
hipEvent_t prof_event()
{
    hipEvent_t e;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace
void prof_begin(hipStream_t s, int id)
{
    ProfRec r{prof_event(), prof_event(), id, true};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    g_prof_recs.push_back(r);
}
void prof_end(hipStream_t s, int id)
{
    for (size_t i = g_prof_recs.size(); i-- > 0;)
        if (g_prof_recs[i].id == id && g_prof_recs[i].open) {
            (void)hipEventRecord(g_prof_recs[i].b, s);
            g_prof_recs[i].open = false;
            return;
        }
}

}  // namespace zt

using namespace zt;

extern "C" int zt_profile_enable(int on)
{
    g_prof_on = on != 0;
    return ZT_OK;
}

extern "C" int zt_profile_reset(void)
{
    (void)hipDeviceSynchronize();
    for (auto &r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    g_prof_recs.clear();
    for (int i = 0; i < P_COUNT; ++i) { g_prof_ms[i] = 0; g_prof_n[i] = 0; }
    return ZT_OK;
}

extern "C" int zt_profile_read(const char *name, int64_t *count, double *total_ms)
{
    if (!name) return ZT_ERR_ARG;
    ZT_HIP(hipDeviceSynchronize());
    for (auto &r : g_prof_recs) {
        if (!r.open) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { g_prof_ms[r.id] += ms; g_prof_n[r.id] += 1; }
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    for (int i = 0; i < P_COUNT; ++i)
        if (strcmp(name, g_prof_names[i]) == 0) {
            if (count) *count = g_prof_n[i];
            if (total_ms) *total_ms = g_prof_ms[i];
            return ZT_OK;
        }
    set_error("zt_profile_read: unknown kernel name %s", name);
    return ZT_ERR_ARG;
}

extern "C" const char *zt_last_error(void) { return g_err; }
extern "C" const char *zt_version(void) { return "zebra_amd 0.1 gfx950"; }

// ---- handle --------------------------------------------------------------------
struct zt_tppr {
    int64_t N;
    int32_t k, M;
    double alpha[16], beta[16];
    // state (device)
    int *len;      // [M][N]
    double *norm;  // [M][N]
    u64 *key;      // [M][N][k]
    double *ts;    // [M][N][k]
    double *w;     // [M][N][k]
    // per-node prepass scratch (device)
    int *cnt;      // [N], zero between calls
    int *off;      // [N]
    // per-batch scratch (device), sized for cap_acc accesses
    int64_t cap_acc;
    int *slot;     // [cap_acc] position of the access inside its node group
    int *list;     // [cap_acc] accesses grouped by node
    int *prev;     // [cap_acc] latest earlier edge touching the same node, or -1
    unsigned *done;  // [M][cap_acc/2 .. ] one flag per (model, edge)
    int64_t cap_done;
    // control words (device): [0] cursor, [1] queue head, [2] status
    int *ctl;
    unsigned epoch;
    int n_cu;
};

namespace {

constexpr int CAP = 128;          // candidates per merge: 2k+1 <= 127
constexpr int WAVES_PER_WG = 4;
constexpr long long WAIT_TICKS = 400000000ll;   // 4 s of the 100 MHz wall clock: bound on any dependency wait

struct WaveLds {
    u64 key[CAP];
    double ts[CAP];
    double w[CAP];
    int sel[64];
    SortLds sort;
};

// ---------------------------------------------------------------- prepass ----
// K1: validate ids; count accesses per node; remember each access' slot.
__global__ void k_count(const int *__restrict__ nodes, const long long *__restrict__ eidx, long long B,
                        int n_roles, long long N, int *cnt, int *slot, int *ctl)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long A = B * n_roles;
    if (a >= A) return;
    const int x = nodes[a];
    bool ok = x >= 0 && x < N;
    if (a < B) {
        const long long e = eidx[a];
        ok = ok && e >= 0 && e <= 0x7fffffffll;
    }
    if (!ok) {
        atomicExch(&ctl[2], ZT_ERR_RANGE);
        slot[a] = -1;
        return;
    }
    slot[a] = atomicAdd(&cnt[x], 1);
}

// K2: the first access of each node reserves a contiguous range of `list`.
__global__ void k_reserve(const int *__restrict__ nodes, long long A, const int *cnt, int *off, const int *slot,
                          int *ctl)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    if (slot[a] == 0) {
        const int x = nodes[a];
        off[x] = atomicAdd(&ctl[0], cnt[x]);
    }
}

// K3: scatter accesses into their node's range.
__global__ void k_fill(const int *__restrict__ nodes, long long A, const int *off, const int *slot, int *list)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const int s = slot[a];
    if (s < 0) return;
    list[off[nodes[a]] + s] = (int)a;
}

// K4: prev[a] = largest edge index < edge(a) among the accesses of a's node.
__global__ void k_prev(const int *__restrict__ nodes, long long A, long long B, const int *cnt, const int *off,
                       const int *slot, const int *list, int *prev)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    if (slot[a] < 0) { prev[a] = -1; return; }
    const int x = nodes[a];
    const int o = off[x], c = cnt[x];
    const int me = (int)(a % B);
    int best = -1;
    for (int p = 0; p < c; ++p) {
        const int e = list[o + p] % (int)B;
        if (e < me && e > best) best = e;
    }
    prev[a] = best;
}

// K5: restore the per-node counters and the control words for the next call.
__global__ void k_cleanup(const int *__restrict__ nodes, long long A, const int *slot, int *cnt, int *ctl)
{
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0) { ctl[0] = 0; ctl[1] = 0; }
    if (a >= A) return;
    if (slot[a] == 0) cnt[nodes[a]] = 0;
}

// ------------------------------------------------------------- row access ----
struct Row {
    u64 key;
    double ts, w;   // this lane's entry (valid for lane < len)
    int len;
    double norm;
};

__device__ __forceinline__ Row load_row(const zt_tppr &h, int m, long long x, int lane)
{
    Row r;
    const long long row = (long long)m * h.N + x;
    r.len = ld_agent(h.len + row);
    r.norm = ld_agent(h.norm + row);
    r.key = 0; r.ts = 0.0; r.w = 0.0;
    if (lane < h.k) {
        const long long p = row * h.k + lane;
        r.key = ld_agent(h.key + p);
        r.ts = ld_agent(h.ts + p);
        r.w = ld_agent(h.w + p);
    }
    return r;
}

// extract_streaming_tppr (utils/util.py:447-469)
__device__ __forceinline__ void emit_row(const Row &r, int k, int lane, double tnow, int *on, int *oe, float *od,
                                         float *ow)
{
    if (lane >= k) return;
    if (r.len == 0) { on[lane] = 0; oe[lane] = 0; od[lane] = 0.f; ow[lane] = 0.f; return; }
    const bool a = lane < r.len;
    on[lane] = a ? (int)(unsigned)(r.key & 0xffffffffull) : 0;
    oe[lane] = a ? (int)(unsigned)(r.key >> 32) : 0;
    ow[lane] = a ? (float)r.w : 0.f;
    const float tsf = a ? (float)r.ts : 0.f;      // tmp_timestamps is float32
    od[lane] = (float)(tnow - (double)tsf);        // f64 - f32 -> f64 -> stored f32
}

// One (s1, s2) pair of the update block (utils/util.py:509-564).  Returns the
// new length of s1's dictionary; lane j < length holds entry j in (ok, ot, ow).
__device__ inline int merge_pair(WaveLds &L, int lane, int k, double alpha, double beta, const Row &r1,
                                 const Row &r2, u64 newkey, double newts, u64 &ok, double &ot, double &ow)
{
    int n1;
    double scale_s1 = 0.0, scale_s2;
    if (r1.norm == 0.0) {                       // :514-519
        n1 = 0;
        scale_s2 = 1.0 - alpha;
    } else {                                    // :520-527
        n1 = r1.len;
        const double new_norm = r1.norm * beta + beta;
        scale_s1 = r1.norm / new_norm * beta;
        scale_s2 = beta / new_norm * (1.0 - alpha);
    }
    if (lane < n1) { L.key[lane] = r1.key; L.ts[lane] = r1.ts; L.w[lane] = r1.w * scale_s1; }
    wave_sync();
    int n = n1;
    if (r2.norm != 0.0) {                       // :532-538
        const bool active = lane < r2.len;
        const double add = r2.w * scale_s2;
        int f = -1;
        for (int q = 0; q < n1; ++q) {
            const bool hit = (L.key[q] == r2.key) && (L.ts[q] == r2.ts);
            f = (hit && f < 0) ? q : f;
        }
        const bool unmatched = active && f < 0;
        const u64 um = __ballot(unmatched);
        if (active && f >= 0) L.w[f] = L.w[f] + add;
        if (unmatched) {
            const int pos = n1 + __popcll(um & lanemask_lt());
            L.key[pos] = r2.key; L.ts[pos] = r2.ts; L.w[pos] = add;
        }
        n = n1 + __popcll(um);
        wave_sync();
    }
    {                                           // :531 / :540-541
        const double v = (alpha != 0.0) ? scale_s2 * alpha : scale_s2;
        int f = -1;
        for (int c = lane; c < CAP; c += WAVE) {
            const bool hit = c < n && L.key[c] == newkey && L.ts[c] == newts;
            const u64 hm = __ballot(hit);
            if (hm != 0ull && f < 0) f = (c - lane) + __ffsll((long long)hm) - 1;
        }
        if (lane == 0) {
            if (f >= 0) L.w[f] = v;
            else { L.key[n] = newkey; L.ts[n] = newts; L.w[n] = v; }
        }
        if (f < 0) ++n;
        wave_sync();
    }
    if (n <= k) {                               // :549-551
        if (lane < n) { ok = L.key[lane]; ot = L.ts[lane]; ow = L.w[lane]; }
        wave_sync();
        return n;
    }
    topk_select_wave(L.w, n, k, L.sel, L.sort, L.sort.r, L.sort.stk);   // :553-559
    if (lane < k) {
        const int c = L.sel[lane];
        ok = L.key[c]; ot = L.ts[c]; ow = L.w[c];
    }
    wave_sync();
    return k;
}

__device__ __forceinline__ void store_row(const zt_tppr &h, int m, long long x, int lane, int n, u64 key, double ts,
                                          double w, double new_norm)
{
    const long long row = (long long)m * h.N + x;
    if (lane < n) {
        const long long p = row * h.k + lane;
        st_agent(h.key + p, key);
        st_agent(h.ts + p, ts);
        st_agent(h.w + p, w);
    }
    if (lane == 0) {
        st_agent(h.len + row, n);
        st_agent(h.norm + row, new_norm);
    }
}

// ------------------------------------------------------------ main kernel ----
__global__ __launch_bounds__(WAVE * WAVES_PER_WG) void k_stream(zt_tppr h, const int *__restrict__ nodes,
                                                                const double *__restrict__ tsv,
                                                                const long long *__restrict__ eidx, long long B,
                                                                int n_roles, int emit, int m_lo, int n_models,
                                                                int *out_nodes, int *out_eidx, float *out_dt,
                                                                float *out_w, unsigned epoch)
{
    __shared__ WaveLds lds[WAVES_PER_WG];
    WaveLds &L = lds[threadIdx.x / WAVE];
    const int lane = lane_id();
    const int k = h.k;
    if (ld_agent(h.ctl + 2) == ZT_ERR_RANGE) return;   // rejected by k_count: touch nothing
    const long long total = B * n_models;
    const long long rows = B * n_roles;
    for (;;) {
        // Dequeue with NO divergent branch: every lane issues the add (lane 0
        // adds 1, the rest 0; the compiler folds it into one wave-level atomic).
        // An `if (lane == 0)` here gets jump-threaded with the `if (lane == 0)`
        // publish at the end of the previous iteration, and the structurizer
        // then replays the body for the remaining lanes (seen in the ISA).
        int idx = atomicAdd(h.ctl + 1, lane == 0 ? 1 : 0);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= total) return;
        const long long i = idx / n_models;
        const int mo = idx % n_models;          // emitted-model index
        const int m = m_lo + mo;
        const double alpha = h.alpha[m], beta = h.beta[m];

        // ---- wait for the predecessors of this edge's nodes ----
        if (lane < n_roles) {
            const int p = h.prev[(long long)lane * B + i];
            if (p >= 0) {
                const unsigned *flag = h.done + (long long)m * h.cap_done + p;
                unsigned spins = 0;
                long long t0 = 0;
                while (ld_agent(flag) != epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023u) == 0) {           // bounded: give up after WAIT_TICKS
                        const long long now = (long long)wall_clock64();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > WAIT_TICKS) { atomicExch(h.ctl + 2, ZT_ERR_TIMEOUT); break; }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

        const long long u = nodes[i], v = nodes[B + i];
        const double tnow = tsv[i];
        const long long e = eidx[i];
        const Row ru = load_row(h, m, u, lane);
        const Row rv = load_row(h, m, v, lane);

        if (emit) {                             // utils/util.py:504-506
            const long long ob = (long long)mo * rows * k;
            emit_row(ru, k, lane, tnow, out_nodes + ob + i * k, out_eidx + ob + i * k, out_dt + ob + i * k,
                     out_w + ob + i * k);
            emit_row(rv, k, lane, tnow, out_nodes + ob + (B + i) * k, out_eidx + ob + (B + i) * k,
                     out_dt + ob + (B + i) * k, out_w + ob + (B + i) * k);
            if (n_roles == 3) {
                const long long g = nodes[2 * B + i];
                const Row rg = load_row(h, m, g, lane);
                emit_row(rg, k, lane, tnow, out_nodes + ob + (2 * B + i) * k, out_eidx + ob + (2 * B + i) * k,
                         out_dt + ob + (2 * B + i) * k, out_w + ob + (2 * B + i) * k);
            }
        }

        // ---- both directions from the OLD rows (utils/util.py:509-564) ----
        const u64 key_uv = ((u64)(unsigned)e << 32) | (u64)(unsigned)v;   // (edge_idx, s2=v, ts) into u
        const u64 key_vu = ((u64)(unsigned)e << 32) | (u64)(unsigned)u;
        u64 ak = 0, bk = 0;
        double at = 0, aw = 0, bt = 0, bw = 0;
        const int na = merge_pair(L, lane, k, alpha, beta, ru, rv, key_uv, tnow, ak, at, aw);
        int nb = 0;
        if (u != v) nb = merge_pair(L, lane, k, alpha, beta, rv, ru, key_vu, tnow, bk, bt, bw);

        // ---- write back (utils/util.py:567-574) and publish ----
        store_row(h, m, u, lane, na, ak, at, aw, ru.norm * beta + beta);
        if (u != v) store_row(h, m, v, lane, nb, bk, bt, bw, rv.norm * beta + beta);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // all lanes store the same word (one store instruction, one address):
        // no lane-0 branch at the loop tail, see the dequeue comment above.
        st_agent(h.done + (long long)m * h.cap_done + i, epoch);
    }
}

__global__ void k_fill_zero(u64 *p, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = 0;
}

int ensure_batch_capacity(zt_tppr *h, int64_t A, int64_t B)
{
    if (A > h->cap_acc) {
        int64_t cap = h->cap_acc ? h->cap_acc : 1024;
        while (cap < A) cap *= 2;
        if (h->slot) { ZT_HIP(hipFree(h->slot)); ZT_HIP(hipFree(h->list)); ZT_HIP(hipFree(h->prev)); }
        ZT_HIP(hipMalloc(&h->slot, sizeof(int) * cap));
        ZT_HIP(hipMalloc(&h->list, sizeof(int) * cap));
        ZT_HIP(hipMalloc(&h->prev, sizeof(int) * cap));
        h->cap_acc = cap;
    }
    if (B > h->cap_done) {
        int64_t cap = h->cap_done ? h->cap_done : 512;
        while (cap < B) cap *= 2;
        if (h->done) ZT_HIP(hipFree(h->done));
        ZT_HIP(hipMalloc(&h->done, sizeof(unsigned) * cap * h->M));
        ZT_HIP(hipMemset(h->done, 0, sizeof(unsigned) * cap * h->M));
        h->cap_done = cap;
    }
    return ZT_OK;
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
    if (k > ZT_MAX_K || n_tppr > 16) {
        set_error("zt_tppr_create: k=%d (max %d) or n_tppr=%d (max 16) unsupported", k, ZT_MAX_K, n_tppr);
        return ZT_ERR_UNSUPPORTED;
    }
    zt_tppr *h = new zt_tppr();
    memset(h, 0, sizeof(*h));
    h->N = num_nodes; h->k = k; h->M = n_tppr;
    for (int m = 0; m < n_tppr; ++m) { h->alpha[m] = alpha_host[m]; h->beta[m] = beta_host[m]; }
    const size_t rows = (size_t)n_tppr * (size_t)num_nodes;
    ZT_HIP(hipMalloc(&h->len, rows * sizeof(int)));
    ZT_HIP(hipMalloc(&h->norm, rows * sizeof(double)));
    ZT_HIP(hipMalloc(&h->key, rows * k * sizeof(u64)));
    ZT_HIP(hipMalloc(&h->ts, rows * k * sizeof(double)));
    ZT_HIP(hipMalloc(&h->w, rows * k * sizeof(double)));
    ZT_HIP(hipMalloc(&h->cnt, (size_t)num_nodes * sizeof(int)));
    ZT_HIP(hipMalloc(&h->off, (size_t)num_nodes * sizeof(int)));
    ZT_HIP(hipMalloc(&h->ctl, 16 * sizeof(int)));
    ZT_HIP(hipMemset(h->cnt, 0, (size_t)num_nodes * sizeof(int)));
    ZT_HIP(hipMemset(h->ctl, 0, 16 * sizeof(int)));
    hipDeviceProp_t prop;
    int dev = 0;
    ZT_HIP(hipGetDevice(&dev));
    ZT_HIP(hipGetDeviceProperties(&prop, dev));
    h->n_cu = prop.multiProcessorCount;
    h->epoch = 0;
    int rc = zt_tppr_reset(h, nullptr);
    if (rc != ZT_OK) return rc;
    ZT_HIP(hipDeviceSynchronize());
    *out = h;
    return ZT_OK;
}

extern "C" int zt_tppr_destroy(zt_tppr *h)
{
    if (!h) return ZT_OK;
    (void)hipFree(h->len); (void)hipFree(h->norm); (void)hipFree(h->key); (void)hipFree(h->ts); (void)hipFree(h->w);
    (void)hipFree(h->cnt); (void)hipFree(h->off); (void)hipFree(h->ctl);
    if (h->slot) { (void)hipFree(h->slot); (void)hipFree(h->list); (void)hipFree(h->prev); }
    if (h->done) (void)hipFree(h->done);
    delete h;
    return ZT_OK;
}

extern "C" int zt_tppr_reset(zt_tppr *h, void *stream)
{
    if (!h) return ZT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)h->M * (size_t)h->N;
    ZT_HIP(hipMemsetAsync(h->len, 0, rows * sizeof(int), s));
    ZT_HIP(hipMemsetAsync(h->norm, 0, rows * sizeof(double), s));
    ZT_HIP(hipMemsetAsync(h->key, 0, rows * h->k * sizeof(u64), s));
    ZT_HIP(hipMemsetAsync(h->ts, 0, rows * h->k * sizeof(double), s));
    ZT_HIP(hipMemsetAsync(h->w, 0, rows * h->k * sizeof(double), s));
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
    const size_t rows = (size_t)src->M * (size_t)src->N;
    ZT_HIP(hipMemcpyAsync(dst->len, src->len, rows * sizeof(int), hipMemcpyDeviceToDevice, s));
    ZT_HIP(hipMemcpyAsync(dst->norm, src->norm, rows * sizeof(double), hipMemcpyDeviceToDevice, s));
    ZT_HIP(hipMemcpyAsync(dst->key, src->key, rows * src->k * sizeof(u64), hipMemcpyDeviceToDevice, s));
    ZT_HIP(hipMemcpyAsync(dst->ts, src->ts, rows * src->k * sizeof(double), hipMemcpyDeviceToDevice, s));
    ZT_HIP(hipMemcpyAsync(dst->w, src->w, rows * src->k * sizeof(double), hipMemcpyDeviceToDevice, s));
    return ZT_OK;
}

extern "C" int zt_tppr_stream(zt_tppr *h, const int32_t *nodes_dev, const double *ts_dev, const int64_t *eidx_dev,
                              int64_t B, int32_t n_roles, int32_t emit, int32_t model, int32_t *out_nodes_dev,
                              int32_t *out_eidx_dev, float *out_dt_dev, float *out_w_dev, void *stream)
{
    if (!h || B < 0 || (n_roles != 2 && n_roles != 3) || model >= h->M) {
        set_error("zt_tppr_stream: bad argument");
        return ZT_ERR_ARG;
    }
    if (B == 0) return ZT_OK;
    if (!nodes_dev || !ts_dev || !eidx_dev ||
        (emit && (!out_nodes_dev || !out_eidx_dev || !out_dt_dev || !out_w_dev))) {
        set_error("zt_tppr_stream: NULL buffer");
        return ZT_ERR_ARG;
    }
    if (B > (1ll << 24)) { set_error("zt_tppr_stream: batch too large"); return ZT_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int64_t A = B * n_roles;
    int rc = ensure_batch_capacity(h, A, B);
    if (rc != ZT_OK) return rc;
    h->epoch += 1;
    if (h->epoch == 0) h->epoch = 1;
    const int tb = 256;
    const int gb = (int)((A + tb - 1) / tb);
    const long long *e64 = reinterpret_cast<const long long *>(eidx_dev);
    static const bool dbg = getenv("ZT_DEBUG_SYNC") != nullptr;
#define ZT_DBG(tag)                                                                          \
    do {                                                                                     \
        if (dbg) {                                                                           \
            hipError_t e__ = hipStreamSynchronize(s);                                        \
            fprintf(stderr, "[zt] %s: %s\n", tag, hipGetErrorString(e__));                   \
            fflush(stderr);                                                                  \
        }                                                                                    \
    } while (0)
    ZT_PROF_BEGIN(s, P_PREPASS);
    k_count<<<gb, tb, 0, s>>>(nodes_dev, e64, B, n_roles, h->N, h->cnt, h->slot, h->ctl);
    ZT_DBG("k_count");
    k_reserve<<<gb, tb, 0, s>>>(nodes_dev, A, h->cnt, h->off, h->slot, h->ctl);
    ZT_DBG("k_reserve");
    k_fill<<<gb, tb, 0, s>>>(nodes_dev, A, h->off, h->slot, h->list);
    ZT_DBG("k_fill");
    k_prev<<<gb, tb, 0, s>>>(nodes_dev, A, B, h->cnt, h->off, h->slot, h->list, h->prev);
    ZT_DBG("k_prev");
    ZT_PROF_END(s, P_PREPASS);
    const int m_lo = model < 0 ? 0 : model;
    const int n_models = model < 0 ? h->M : 1;
    const long long total = B * n_models;
    long long waves = total;
    const long long max_waves = (long long)h->n_cu * WAVES_PER_WG * 2;
    if (waves > max_waves) waves = max_waves;
    const int grid = (int)((waves + WAVES_PER_WG - 1) / WAVES_PER_WG);
    ZT_PROF_BEGIN(s, P_STREAM);
    k_stream<<<grid, WAVE * WAVES_PER_WG, 0, s>>>(*h, nodes_dev, ts_dev, e64, B, n_roles, emit, m_lo, n_models,
                                                  out_nodes_dev, out_eidx_dev, out_dt_dev, out_w_dev, h->epoch);
    ZT_PROF_END(s, P_STREAM);
    ZT_DBG("k_stream");
    ZT_PROF_BEGIN(s, P_CLEANUP);
    k_cleanup<<<gb, tb, 0, s>>>(nodes_dev, A, h->slot, h->cnt, h->ctl);
    ZT_PROF_END(s, P_CLEANUP);
    ZT_DBG("k_cleanup");
    ZT_LAUNCH_CHECK();
    return ZT_OK;
}

extern "C" int zt_tppr_status(zt_tppr *h, void *stream)
{
    if (!h) return ZT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int st = 0;
    ZT_HIP(hipMemcpyAsync(&st, h->ctl + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    ZT_HIP(hipStreamSynchronize(s));
    if (st != 0) {
        ZT_HIP(hipMemsetAsync(h->ctl + 2, 0, sizeof(int), s));
        ZT_HIP(hipStreamSynchronize(s));
        set_error(st == ZT_ERR_RANGE ? "node or edge id out of range" : "dependency wait timed out");
    }
    return st;
}

extern "C" int zt_tppr_export(zt_tppr *h, int32_t m, int32_t *len_host, double *norm_host, int64_t *eidx_host,
                              int64_t *node_host, double *ts_host, double *w_host)
{
    if (!h || m < 0 || m >= h->M) return ZT_ERR_ARG;
    ZT_HIP(hipDeviceSynchronize());
    const size_t N = (size_t)h->N, k = (size_t)h->k, off = (size_t)m * N;
    std::vector<u64> key(N * k);
    ZT_HIP(hipMemcpy(len_host, h->len + off, N * sizeof(int), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(norm_host, h->norm + off, N * sizeof(double), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(key.data(), h->key + off * k, N * k * sizeof(u64), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(ts_host, h->ts + off * k, N * k * sizeof(double), hipMemcpyDeviceToHost));
    ZT_HIP(hipMemcpy(w_host, h->w + off * k, N * k * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t v = 0; v < N; ++v)
        for (size_t j = 0; j < k; ++j) {
            const size_t p = v * k + j;
            if ((int)j < len_host[v]) {
                eidx_host[p] = (int64_t)(key[p] >> 32);
                node_host[p] = (int64_t)(key[p] & 0xffffffffull);
            } else {
                eidx_host[p] = 0; node_host[p] = 0; ts_host[p] = 0.0; w_host[p] = 0.0;
            }
        }
    return ZT_OK;
}

extern "C" int zt_tppr_import(zt_tppr *h, int32_t m, const int32_t *len_host, const double *norm_host,
                              const int64_t *eidx_host, const int64_t *node_host, const double *ts_host,
                              const double *w_host)
{
    if (!h || m < 0 || m >= h->M) return ZT_ERR_ARG;
    const size_t N = (size_t)h->N, k = (size_t)h->k, off = (size_t)m * N;
    std::vector<u64> key(N * k);
    for (size_t p = 0; p < N * k; ++p) {
        if (eidx_host[p] < 0 || eidx_host[p] > 0x7fffffffll || node_host[p] < 0 || node_host[p] >= h->N) {
            set_error("zt_tppr_import: id out of range");
            return ZT_ERR_RANGE;
        }
        key[p] = ((u64)eidx_host[p] << 32) | (u64)node_host[p];
    }
    for (size_t v = 0; v < N; ++v)
        if (len_host[v] < 0 || len_host[v] > (int)k) { set_error("zt_tppr_import: bad length"); return ZT_ERR_ARG; }
    ZT_HIP(hipDeviceSynchronize());
    ZT_HIP(hipMemcpy(h->len + off, len_host, N * sizeof(int), hipMemcpyHostToDevice));
    ZT_HIP(hipMemcpy(h->norm + off, norm_host, N * sizeof(double), hipMemcpyHostToDevice));
    ZT_HIP(hipMemcpy(h->key + off * k, key.data(), N * k * sizeof(u64), hipMemcpyHostToDevice));
    ZT_HIP(hipMemcpy(h->ts + off * k, ts_host, N * k * sizeof(double), hipMemcpyHostToDevice));
    ZT_HIP(hipMemcpy(h->w + off * k, w_host, N * k * sizeof(double), hipMemcpyHostToDevice));
    return ZT_OK;
}
